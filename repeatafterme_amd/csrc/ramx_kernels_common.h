// ramx_kernels_common.h -- constants, control blocks, the pack kernel and the band recurrence shared by every kernel (band_step, run_band)
// (device code of libramx; included by ramx_device.hip only -- one translation unit so that everything inlines)
#pragma once

#include <hip/hip_runtime.h>
#include <utility>

#include "ramx_internal.h"

#define NEG RAMX_NEG_IMPOSSIBLE
#define SENT RAMX_OOB_SENTINEL
#define NSHARD 32
#define PF 8            // state slots kept in flight per lane
#define MAX_SAMPLES 64

struct RamxCtl
{
  long long max_ext;   // max_extension_score
  int max_row;         // max_extension_score_row_idx
  int stopped;
  int rows_done;       // row_idx iterations executed so far
  int overflow;        // a column sum left the int32 range
  int besta;
  int pad;
};

struct KArgs
{
  const int4 *S_in;
  int4 *S_out;
  const unsigned *bases;
  const int2 *bounds;
  int2 *trim;
  const long long *sums_in;
  long long *sums_out;
  long long *sums_zero;
  const RamxCtl *ctl_in;
  RamxCtl *ctl_out;
  signed char *cons_out;
  int Np, Nx, W, r, go, ge, cap, minimp, when_to_stop, nshards_in;
  int tab[RAMX_NCLASS][4];   // tab[class][candidate] = matrix[candidate][class]
  // -outmat trace (DBG instantiation of the column kernel only): per cell which state holds the cell's score
  // (0 substitution, 1 deletion, 2 insertion: bnw_extend.c:1027-1044), per flank the row's best score and cell
  signed char *dbg_codes;    // [Np][2W+1]
  int2 *dbg_best;            // [Np]
  // -vvvv trace (DBG + CHAIN instantiations): per flank the four candidate rows r+1 as the reference's per-candidate
  // compute_nw_row calls see them (ram_extend.c:992-1062) -- best cell value, its band cell, and the gap state of the row's
  // first and last cell (the **OUT_OF_SEQ** tests of :1036-1038 and :1134-1136) -- and the same two gap states of row r
  int *dbg_cand;             // [Np][16]: best[4], cell[4], gap_first[4], gap_last[4]; NULL: not wanted
  int *dbg_band;             // -vvvvv: [Np][4][2W+1][2]: every cell (sub, gap) of the four candidate rows; NULL: not wanted
  int2 *dbg_gap;             // [Np]: gap state of row r's first / last cell
};

struct FamDesc { int tile0, ntiles, nx, id; };      // first 64-flank tile, tiles, flanks of the family; its index in the caller's arrays

// ------------------------------------------------------------------------------------------
// pack kernel: 1-byte library -> transposed, pre-oriented 4-bit windows
// ------------------------------------------------------------------------------------------
#ifndef RAMX_SECONDARY_TU
__global__ void ramx_pack_kernel(const signed char *__restrict__ lib, unsigned long long lib_len,
                                 const ramx_flank *__restrict__ fl, int Nx, int Np, int W, int k0,
                                 unsigned *__restrict__ bases, int2 *__restrict__ bounds)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = k0 + blockIdx.y;                 // word k0 .. of every flank's window (the host packs a long window in pieces)
  if (n >= Np) return;
  // (eight consecutive words per thread, so that a thread reads the 64-byte line it has fetched whole, were tried in round 4: seam 1's
  // preparation got slower, not faster -- the words of a line are fetched by consecutive blocks and meet in L2 as it is)
  unsigned word = 0x88888888u;   // class 8 = N everywhere
  if (n < Nx)
  {
    const ramx_flank f = fl[n];
    word = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
      const int t = 8 * k + i - W - 8;   // one leading pad word: nibble index t'' = t + W + 8
      unsigned c = 8;
      if (t >= f.t_lo && t <= f.t_hi)
      {
        const long long p = f.start + (long long)f.step * t;
        if (p >= 0 && (unsigned long long)p < lib_len)
        {
          const int b = lib[p];
          if (b >= 0 && b <= 7)   // A C G T a c g t; complement keeps the case (sequence.c:1141-1160)
            c = f.compl_ ? (unsigned)((b & 4) | (3 - (b & 3))) : (unsigned)b;
        }
      }
      word |= c << (4 * i);
    }
    if (k == 0) bounds[n] = make_int2(f.t_lo + W, f.t_hi + W);
  }
  else if (k == 0)
    bounds[n] = make_int2(1, 0);   // empty interval: every cell out of bounds
  bases[(size_t)k * Np + n] = word;
}

// ------------------------------------------------------------------------------------------
// the same windows built straight from the .2bit payload (include/ramx.h ramx_packed_library): four bases per
// byte, first base in the most significant bits, T C A G = 0 1 2 3 (kentsrc/twoBitNew.c:531-594, dnautil.h:23-27); runs of N
// (:597-613) as sorted (start, length) pairs in library coordinates.  The one-byte-per-base library never exists.
// ------------------------------------------------------------------------------------------
struct PkLib
{
  const unsigned char *bytes;
  const unsigned long long *win_start, *win_byte;   // [n_windows + 1]
  const unsigned char *win_phase;
  const unsigned long long *n_start;
  const unsigned *n_len;
  unsigned long long length;
  int n_windows, n_blocks;
};

__device__ __forceinline__ int pk_window_of(const PkLib &L, unsigned long long p)     // last window that starts at or before p
{
  int lo = 0, hi = L.n_windows;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.win_start[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

// window of every flank's first base (its other bases lie in the same window unless the caller's bounds say otherwise:
// ramx_pack2_kernel looks again for those)
// the trim records of a finished direction, written straight into the session's pinned host buffer: the first device-to-host DMA
// of a process costs about 6 ms of engine set-up on this stack, a store over the bus from a kernel that is already loaded none
__global__ void ramx_to_host_kernel(const int2 *__restrict__ src, int2 *__restrict__ dst, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void ramx_flank_window_kernel(const PkLib L, const ramx_flank *__restrict__ fl, int Nx, int *__restrict__ flank_win)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Nx) return;
  const ramx_flank f = fl[n];
  long long p = f.start + (long long)f.step * (f.t_lo > 0 ? f.t_lo : 0);
  if (p < 0) p = 0;
  if ((unsigned long long)p >= L.length) p = L.length ? (long long)L.length - 1 : 0;
  flank_win[n] = L.n_windows > 0 ? pk_window_of(L, (unsigned long long)p) : 0;
}

#define RAMX_PK_WORDS 32
__global__ void ramx_pack2_kernel(const PkLib L, const ramx_flank *__restrict__ fl, const int *__restrict__ flank_win, int Nx, int Np, int W, int k0, int KW,
                                  unsigned *__restrict__ bases, int2 *__restrict__ bounds)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Np) return;
  // RAMX_PK_WORDS consecutive words (8 bases each) of one flank per thread: 256 bases are ONE 64-byte line of the payload, so a
  // thread's loads stay in the line it has just brought in (one word per thread, as in ramx_pack_kernel, touches 64 different
  // lines per wavefront for 2-3 bytes each)
  ramx_flank f;
  int w = 0;
  if (n < Nx) { f = fl[n]; w = flank_win[n]; }
  for (int k = k0 + blockIdx.y * RAMX_PK_WORDS; k < k0 + (int)(blockIdx.y + 1) * RAMX_PK_WORDS && k < KW; k++)      // words k0 .. KW-1
  {
  unsigned word = 0x88888888u;   // class 8 = N everywhere
  if (n < Nx)
  {
    word = 0;
    // Fast path (round 4; nearly every word): the word's eight positions all lie inside the flank and inside its window -- their
    // sixteen payload bits sit in at most three consecutive bytes, fetched once, and the runs of N are looked up once per word
    // (the per-base path below: eight byte loads and eight binary searches; 4.4 ms for the first 2,048 columns of 100,000 flanks)
    const int t0 = 8 * k - W - 8;
    bool done = false;
    if (t0 >= f.t_lo && t0 + 7 <= f.t_hi)
    {
      const long long pa = f.start + (long long)f.step * t0, pb = f.start + (long long)f.step * (t0 + 7);
      const long long plo = pa < pb ? pa : pb, phi = pa < pb ? pb : pa;
      if (plo >= 0 && (unsigned long long)phi < L.length && phi - plo == 7)
      {
        if ((unsigned long long)plo < L.win_start[w] || (unsigned long long)plo >= L.win_start[w + 1]) w = pk_window_of(L, (unsigned long long)plo);
        if ((unsigned long long)phi < L.win_start[w + 1])
        {
          const unsigned long long q0 = (unsigned long long)plo - L.win_start[w] + L.win_phase[w];
          const unsigned char *bp = L.bytes + L.win_byte[w] + (q0 >> 2);        // (the payload is padded: three bytes can always be read)
          const unsigned v24 = ((unsigned)bp[0] << 16) | ((unsigned)bp[1] << 8) | (unsigned)bp[2];
          const unsigned v16 = (v24 >> (8 - 2 * (unsigned)(q0 & 3))) & 0xffffu;     // position plo in the top two bits .. phi in the lowest two
          unsigned nmask = 0;                                                     // bit j: position plo + j lies in a run of N
          if (L.n_blocks > 0)
          {
            int a = -1, z = L.n_blocks;                                           // last run that starts at or before phi
            while (z - a > 1) { const int mid = (a + z) >> 1; if (L.n_start[mid] <= (unsigned long long)phi) a = mid; else z = mid; }
            for (; a >= 0 && L.n_start[a] + L.n_len[a] > (unsigned long long)plo; a--)
            {
              const long long s0 = (long long)L.n_start[a] - plo, e0 = s0 + (long long)L.n_len[a];
              const int lo_ = s0 < 0 ? 0 : (int)s0, hi_ = e0 > 8 ? 8 : (int)e0;
              if (hi_ > lo_) nmask |= ((1u << hi_) - 1u) & ~((1u << lo_) - 1u);
            }
          }
#pragma unroll
          for (int i = 0; i < 8; i++)
          {
            const int j = f.step > 0 ? i : 7 - i;                                 // position plo + j is nibble i
            const unsigned v = (v16 >> (14 - 2 * j)) & 3u;
            unsigned c = (0x2013u >> (4 * v)) & 0xfu;                             // T C A G -> 3 1 0 2 (sequence.h:7-15)
            if ((nmask >> j) & 1u) c = 8;
            else if (f.compl_) c = 3 - c;
            word |= c << (4 * i);
          }
          done = true;
        }
      }
    }
    if (!done)
    {
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
      const int t = 8 * k + i - W - 8;   // one leading pad word: nibble index t'' = t + W + 8
      unsigned c = 8;
      if (t >= f.t_lo && t <= f.t_hi)
      {
        const long long p = f.start + (long long)f.step * t;
        if (p >= 0 && (unsigned long long)p < L.length)
        {
          const unsigned long long up = (unsigned long long)p;
          if (up < L.win_start[w] || up >= L.win_start[w + 1]) w = pk_window_of(L, up);
          const unsigned long long q = up - L.win_start[w] + L.win_phase[w];
          const unsigned v = (L.bytes[L.win_byte[w] + (q >> 2)] >> (6 - 2 * (unsigned)(q & 3))) & 3u;
          c = (0x2013u >> (4 * v)) & 0xfu;                        // T C A G -> 3 1 0 2 (sequence.h:7-15)
          if (L.n_blocks > 0)
          {
            int a = -1, z = L.n_blocks;                            // last run that starts at or before p
            while (z - a > 1) { const int mid = (a + z) >> 1; if (L.n_start[mid] <= up) a = mid; else z = mid; }
            if (a >= 0 && up < L.n_start[a] + L.n_len[a]) c = 8;
          }
          if (c < 4 && f.compl_) c = 3 - c;
        }
      }
      word |= c << (4 * i);
    }
    }
    if (k == 0) bounds[n] = make_int2(f.t_lo + W, f.t_hi + W);
  }
  else if (k == 0)
    bounds[n] = make_int2(1, 0);   // empty interval: every cell out of bounds
  bases[(size_t)k * Np + n] = word;
  }
}
#endif

// ------------------------------------------------------------------------------------------
// column kernel
// ------------------------------------------------------------------------------------------
//
// State kept per band cell (two int32, as in the reference's score[..][..][2]) is stored TRANSFORMED:
//     m  = max(sub, gap)                  -- all the next row's substitution term needs   (bnw_extend.c:950-956)
//     e  = max(sub + go, gap) + ge        -- all the next row's deletion term needs       (:892-905), and, read
//                                            from the current row's previous cell, the insertion term (:972-985)
// (sub, gap) -> (m, e) loses nothing the recurrence ever reads, and saves three VALU ops per cell per row.
//
// Candidate rows (row r+1 for A,C,G,T; only their best cell is needed, ram_extend.c:1005-1062).  With
// go <= 0 and ge <= 0 (every built-in scoring system) the insertion chain can never hold the row maximum:
//   emit_k = max(sub_k + go, gap_k) <= cell_k,  ins_{k+1} = emit_k + ge <= cell_k, and for a masked cell
//   emit = max(v + go, v) = v = cell; by induction  max_k cell_k = max_k [ inb_k ? max(sub_k, del_k) : v_k ].
// So CHAIN = false evaluates the four candidates with NO serial chain: sub_k = m_k + M[a][base] and the shared
// del_k = e_{k+1}.  CHAIN = true keeps the full recurrence for user-supplied positive penalties.

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Sum over the wave of a value in [0, 2^31) per lane, without touching LDS: two 32-bit DPP reductions (low and high
// 16 bits; 64 lanes x 2^16 fits), total read from lane 63.  Six VALU steps each instead of six ds_bpermute pairs.
__device__ __forceinline__ unsigned wave_sum_u32_dpp(unsigned v)
{
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   // row_ror:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   // row_ror:8  -> row totals everywhere
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum over the wave (same DPP steps), returned to every lane through the scalar unit
__device__ __forceinline__ int wave_max_i32_dpp(int v)
{
  const int lowest = -2147483647 - 1;
#define RAMX_MAX_STEP(ctrl, rows) do { const int o_ = __builtin_amdgcn_update_dpp(lowest, v, ctrl, rows, 0xf, false); v = v > o_ ? v : o_; } while (0)
  RAMX_MAX_STEP(0xB1, 0xf);     // quad_perm:[1,0,3,2]
  RAMX_MAX_STEP(0x4E, 0xf);     // quad_perm:[2,3,0,1]
  RAMX_MAX_STEP(0x124, 0xf);    // row_ror:4
  RAMX_MAX_STEP(0x128, 0xf);    // row_ror:8  -> row maxima everywhere
  RAMX_MAX_STEP(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
  RAMX_MAX_STEP(0x143, 0xc);    // row_bcast:31 into rows 2 and 3
#undef RAMX_MAX_STEP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ long long wave_sum_nonneg31(int v)
{
  const unsigned lo = wave_sum_u32_dpp((unsigned)v & 0xffffu), hi = wave_sum_u32_dpp((unsigned)v >> 16);
  return ((long long)hi << 16) + (long long)lo;
}

// The DP rows are written once per column and read once by the next launch.  Plain (cacheable) accesses are the
// measured choice: the 131 MB ping-pong working set of the N = 100,000 workload stays largely resident in the
// 256 MB Infinity Cache between launches; non-temporal accesses (-DRAMX_NT_LDST) were 25 % slower
// (32.5 vs 26.1 us per column, profiles/r01_notes.md).
__device__ __forceinline__ int4 ld_stream(const int4 *p)
{
#ifndef RAMX_NT_LDST
  return *p;
#else
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p));
  return make_int4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st_stream(int4 *p, int4 v)
{
#ifndef RAMX_NT_LDST
  *p = v;
#else
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
  __builtin_nontemporal_store(x, reinterpret_cast<v4i *>(p));
#endif
}

__device__ __forceinline__ int imax(int x, int y) { return x > y ? x : y; }
__device__ __forceinline__ int imax3(int x, int y, int z) { return imax(imax(x, y), z); }
__device__ __forceinline__ int imed3(int x, int lo, int hi)   // median of three = clamp(x, lo, hi) for lo <= hi
{
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(lo), "v"(hi));
  return d;
}

// Per-lane running values of the skewed pair {row r, candidate rows r+1}.
struct LaneDP
{
  int eC;        // e of row r cell j-1: the insertion term of cell j, and the deletion term of candidate cell j-2
  int mPrev;     // m of row r cell j-1: the substitution predecessor of candidate cell j-1
  int bestF, jbest;
  int eA[4];     // CHAIN only: e of the candidates' previous cell
  int bestA[4];
  // -vvvv trace only (DBG && CHAIN): where each candidate row's best cell is, gap states of the rows' end cells
  int jA[4], gA0[4], gAL[4], gF0, gFL;
  // -vvvvv trace only: where the candidate rows' cells go (ram_extend.c:1013-1024 prints them): [4][2W+1][2] = (sub, gap); NULL: not wanted
  int *band;
};

// Uniform (scalar) per-step quantities.
struct StepU
{
  int j;         // band cell of row r handled by this step (candidates handle cell j-1 of row r+1)
  int vF;        // OOB fill of row r   cell j    (bnw_extend.c:990-1002)
  int vC;        // OOB fill of row r+1 cell j-1, or the "no such cell" value at j == 0
  bool first;    // j == 0: there is no candidate cell -1
  int hi;        // CHAIN fast path: upper clamp of the candidates' gap state (INT_MAX; NEG - ge at step 0)
};

// LDS score table: row b (base class 0..8, row 9 = zeros for masked cells) holds
// {M[A][b], M[C][b], M[G][b], M[T][b], M[besta][b], 0, 0, 0}; 32 B rows.
#define TAB_ROWS 10
#define TAB_STRIDE 8

// Table values of one step, fetched from LDS ahead of use (the lookups depend only on the base stream and
// the bounds, never on the DP chain, so they are issued one slot early to hide the LDS latency).
struct StepT
{
  int sF;        // M[besta][base]: substitution score of row r cell j
  int4 s;        // M[A..T][base] (zeros when the candidates' cell is masked)
  bool inb;      // t' = j + r inside the flank
  bool inbC;     // inb && j > 0
};

template <bool OOB>
__device__ __forceinline__ StepT fetch_step(const int *s_tab, unsigned bc, bool inb, bool first)
{
  StepT t;
  t.inb = OOB ? inb : true;
  t.inbC = OOB ? (inb && !first) : true;
#ifdef RAMX_DBG_NOLDS   // timing ablation only
  t.sF = (int)bc - 3;
  t.s = make_int4((int)bc, (int)bc - 1, (int)bc - 2, 3 - (int)bc);
#else
  t.sF = s_tab[bc * TAB_STRIDE + 4];
  const unsigned bcC = (OOB && !t.inbC) ? 9u : bc;   // row 9 of the table is all zeros
  t.s = *reinterpret_cast<const int4 *>(s_tab + bcC * TAB_STRIDE);
#endif
  return t;
}

// One band step.  FIN: compute row r cell j from the previous row (Pm = m of cell j, PeNext = e of cell j+1);
// !FIN (virtual step j == B): only the candidates' last cell.  INIT: row "r" is the boundary row
// (ram_extend.c:909-946).  OOB = false is the fast path taken by a wave whose 64 flanks all cover the whole
// band of both rows: no bounds selects at all.
template <bool INIT, bool FIN, bool OOB, bool CHAIN, bool DBG = false>
__device__ __forceinline__ void band_step(const int go, const int ge, const int W, const StepU u, const StepT t,
                                          const int Pm, const int PeNext, LaneDP &L, int &outM, int &outE, signed char *code = nullptr)
{
  int eCn, m = 0;
  if (FIN)
  {
    int sub, gap;
    if (INIT)
    {
      const int o = u.j - W;
      sub = (o == 0) ? 0 : (go + (o < 0 ? -o : o) * ge);
      gap = sub;
    }
    else
    {
      sub = Pm + t.sF;                             // bnw_extend.c:950-956
      gap = imax(L.eC, PeNext);                    // ins (:972-985) vs del (:892-905), :1007-1010
      if (OOB)
      {
        sub = t.inb ? sub : u.vF;                  // :990-1002
        gap = t.inb ? gap : u.vF;
      }
    }
    m = imax(sub, gap);                            // :1015-1018
    if (DBG && CHAIN)
    {
      if (u.j == 0) L.gF0 = gap;
      if (u.j == 2 * W) L.gFL = gap;
    }
    if (DBG && !INIT && code != nullptr)
    {
      // which state the reference's path string reports for this cell (:1038-1043): the substitution if it holds the
      // cell's score, else the deletion if it does, else the insertion; out-of-bounds cells have all three equal
      const int del = (OOB && !t.inb) ? u.vF : PeNext;
      code[u.j] = (signed char)((sub == m) ? 0 : ((del == m) ? 1 : 2));
    }
    if (!INIT)
    {
      const bool better = m > L.bestF;             // :1020-1024 strict >: lowest offset wins ties
      L.bestF = imax(m, L.bestF);
      L.jbest = better ? u.j : L.jbest;
    }
    eCn = imax(sub + go, gap) + ge;
    outM = m;
    outE = eCn;
  }
  else
    eCn = NEG;                                     // cell B does not exist: candidates' del is exactly NEG
  // candidates: cell j-1 of row r+1 for all four bases, from S(r) just computed (never stored)
  const int sv[4] = { t.s.x, t.s.y, t.s.z, t.s.w };
  if (CHAIN)
  {
    int mSel, lo, hi;
    if (OOB)
    {
      mSel = t.inbC ? L.mPrev : u.vC;              // masked cell: sub = gap = vC
      lo = t.inbC ? eCn : u.vC;                    // del of the candidates (shared by the four)
      hi = t.inbC ? 2147483647 : u.vC;
    }
    else
    {
      mSel = L.mPrev;                              // at step 0 this is the very negative initial value
      lo = eCn;
      hi = u.hi;                                   // INT_MAX, or NEG - ge at step 0: median(NEG, lo, hi) = NEG - ge
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
      const int subA = mSel + sv[c];
      const int gapA = imed3(L.eA[c], lo, hi);     // in bounds: max(ins, del); masked: vC
      if (DBG && !u.first)
      {
        // candidate cell j-1 of row r+1: strict > keeps the lowest cell on ties (bnw_extend.c:1020-1024)
        if (imax(subA, gapA) > L.bestA[c]) L.jA[c] = u.j - 1;
        if (u.j - 1 == 0) L.gA0[c] = gapA;
        if (u.j - 1 == 2 * W) L.gAL[c] = gapA;
        if (L.band != nullptr)
        {
          int *o = L.band + ((size_t)c * (2 * W + 1) + (u.j - 1)) * 2;
          o[0] = subA; o[1] = gapA;
        }
      }
      L.bestA[c] = imax3(L.bestA[c], subA, gapA);
      L.eA[c] = imax(subA + go, gapA) + ge;
    }
  }
  else
  {
    int mSel, lo;
    if (OOB)
    {
      mSel = t.inbC ? L.mPrev : u.vC;
      lo = t.inbC ? eCn : u.vC;
    }
    else
    {
      mSel = L.mPrev;
      lo = u.first ? NEG : eCn;                    // there is no candidate cell -1 to take e[0] as its deletion
    }
#pragma unroll
    for (int c = 0; c < 4; c++) L.bestA[c] = imax3(L.bestA[c], mSel + sv[c], lo);
  }
  L.mPrev = m;
  L.eC = eCn;
}

// The whole band of one flank (one lane) for column r: streams S(r-1) in, S(r) out.
template <bool INIT, bool OOB, bool CHAIN, bool DBG = false>
__device__ __forceinline__ void run_band(const KArgs &a, const int r, const int *s_tab, const int4 *Sin, int4 *Sout,
                                         const unsigned *bp, const int jlo, const int jhi, LaneDP &D, int &high, int &pos,
                                         int4 (&buf)[PF], int4 (&far)[PF], unsigned w0, unsigned w1, unsigned w2, signed char *code = nullptr)
{
  const int W = a.W, B = 2 * W + 1, Q = W + 1, go = a.go, ge = a.ge;
  // OOB fill values (bnw_extend.c:990-1002): uniform per (row, cell)
  const int edgeF = (r < W) ? go + (r + 1) * ge : SENT;        // row r,   cells j < W
  const int edgeC = (r + 1 < W) ? go + (r + 2) * ge : SENT;    // row r+1, cells j-1 < W
  const int vFirst = NEG - ge - (go > 0 ? go : 0);             // CHAIN: leaves eA = NEG after the masked step 0
  const int ph4 = 4 * ((r + 8) & 7);
  const size_t wstride = (size_t)a.Np;
  auto make_u = [&](int j) {
    StepU u;
    u.j = j;
    u.first = (j == 0);
    u.hi = u.first ? NEG - ge : 2147483647;
    if (OOB)
    {
      u.vF = (j < W) ? edgeF : SENT;
      u.vC = u.first ? vFirst : ((j - 1 < W) ? edgeC : SENT);
    }
    else { u.vF = 0; u.vC = 0; }
    return u;
  };
  // table lookups of the two steps of slot q (bc0/bc1: base classes of steps 2q and 2q+1)
  auto fetch_slot = [&](int q, unsigned bc0, unsigned bc1, StepT &t0, StepT &t1) {
    const int j0 = 2 * q, j1 = 2 * q + 1;
    t0 = fetch_step<OOB>(s_tab, bc0, (j0 >= jlo) && (j0 <= jhi), j0 == 0);
    t1 = fetch_step<OOB>(s_tab, bc1, (j1 >= jlo) && (j1 <= jhi), false);
  };
  // One regular slot q = (m,e) of cells 2q and 2q+1 of row r; candidate cells 2q-1 and 2q of row r+1.
  auto regular_slot = [&](int q, int4 cur, int4 nxt, const StepT &t0, const StepT &t1) {
    int m0, e0, m1, e1;
    band_step<INIT, true, OOB, CHAIN, DBG>(go, ge, W, make_u(2 * q), t0, cur.x, cur.w, D, m0, e0, code);
    band_step<INIT, true, OOB, CHAIN, DBG>(go, ge, W, make_u(2 * q + 1), t1, cur.z, nxt.y, D, m1, e1, code);
#if defined(RAMX_DBG_NOMEM) || defined(RAMX_DBG_NOSTORE)
    if (m0 == 0x7fffffff) Sout[(size_t)q * 64] = make_int4(m0, e0, m1, e1);
#else
    st_stream(Sout + (size_t)q * 64, make_int4(m0, e0, m1, e1));
#endif
  };
  auto final_slot = [&](int4 cur, const StepT &t0, const StepT &t1) {
    int m0, e0, m1, e1;
    // cell B-1 has no deletion predecessor (bnw_extend.c:892); then the virtual step j = B
    band_step<INIT, true, OOB, CHAIN, DBG>(go, ge, W, make_u(2 * W), t0, cur.x, NEG, D, m0, e0, code);
    band_step<INIT, false, OOB, CHAIN, DBG>(go, ge, W, make_u(B), t1, 0, 0, D, m1, e1, code);
    if (!INIT)
    {
      high = cur.z; pos = cur.w;
      if (D.bestF > high) { high = D.bestF; pos = r + D.jbest - W; }   // ram_extend.c:1140-1150
    }
    st_stream(Sout + (size_t)W * 64, make_int4(m0, e0, high, pos));
  };
  auto nib = [](unsigned A0, unsigned A1, int k) { return ((k < 8 ? A0 : A1) >> (4 * (k & 7))) & 15u; };

  // ---- full groups: 8 slots = 16 steps of branch-free straight-line code -------------------
  const int G = W >> 3;
  int q0 = 0;
  unsigned A0 = __builtin_amdgcn_alignbit(w1, w0, ph4);
  unsigned A1 = __builtin_amdgcn_alignbit(w2, w1, ph4);
  StepT t0, t1;
  fetch_slot(0, nib(A0, A1, 0), nib(A0, A1, 1), t0, t1);
  for (int g = 0; g < G; g++, q0 += 8)
  {
    const unsigned *bq = bp + (size_t)(2 * g + 3) * wstride;
    const unsigned w3 = bq[0], w4 = bq[wstride];               // next group's words, consumed at the END of this group
#pragma unroll
    for (int i = 0; i < PF; i++)
    {
      const int q = q0 + i;
      int4 cur = make_int4(0, 0, 0, 0), nxt = cur;
      if (!INIT)
      {
        cur = buf[i];
        nxt = buf[(i + 1) % PF];
        buf[i] = far[i];
        const int qn = q + 2 * PF;
#ifdef RAMX_DBG_NOMEM   // timing ablation only
        far[i] = make_int4(cur.x + 1, cur.y - 1, cur.z + 2, cur.w - 2);
#else
        far[i] = ld_stream(Sin + (size_t)(qn < Q ? qn : Q - 1) * 64);
#endif
      }
      StepT n0, n1;                                            // lookups of the NEXT slot, issued before this one's math
      if (i + 1 < PF) fetch_slot(q + 1, nib(A0, A1, 2 * i + 2), nib(A0, A1, 2 * i + 3), n0, n1);
      else
      {
        A0 = __builtin_amdgcn_alignbit(w3, w2, ph4);
        A1 = __builtin_amdgcn_alignbit(w4, w3, ph4);
        fetch_slot(q + 1, nib(A0, A1, 0), nib(A0, A1, 1), n0, n1);
      }
      regular_slot(q, cur, nxt, t0, t1);
      t0 = n0; t1 = n1;
    }
    w0 = w2; w1 = w3; w2 = w4;
  }
  // ---- tail group: remaining regular slots (W % 8 of them) and the final slot ---------------
#pragma unroll
  for (int i = 0; i < PF; i++)
  {
    const int q = q0 + i;
    if (q <= W)
    {
      int4 cur = make_int4(0, 0, 0, 0), nxt = cur;
      if (!INIT) { cur = buf[i]; nxt = buf[(i + 1) % PF]; }
      if (q < W)
      {
        StepT n0, n1;
        fetch_slot(q + 1, nib(A0, A1, (2 * i + 2) & 15), nib(A0, A1, (2 * i + 3) & 15), n0, n1);
        regular_slot(q, cur, nxt, t0, t1);
        t0 = n0; t1 = n1;
      }
      else
        final_slot(cur, t0, t1);
    }
  }
}

