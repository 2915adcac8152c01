// ramx_device.hip -- seam 2 of include/ramx.h: the thin device API and the gfx950 kernels.
//
// Replaces, on the device, the reference's per-column work:
//   compute_nw_row            bnw_extend.c:750-1048   (banded affine row, 2 states per cell)
//   candidate vote + cap      ram_extend.c:973-1086
//   recompute-with-winner     ram_extend.c:1097-1168  (folded away: see "column step" below)
//   fit-preferred stop rule   ram_extend.c:1169-1223
//   boundary row              ram_extend.c:909-960
//
// Layout in HBM (Np = flanks padded to a multiple of 64, W = half band, B = 2W+1, Q = W+1):
//   bases  uint32 [KW][Np]        4-bit base classes, 8 per word, pre-oriented per flank so that
//                                 nibble (t' & 7) of word (t' >> 3) is the base aligned to band
//                                 cell (row r, offset o) with t' = o + r + W.  Transposed: the 64
//                                 lanes of a wave read 256 contiguous bytes.
//   state  int4   [Np/64][Q][64]  one DP row per flank: slot q holds cells 2q and 2q+1 as
//                                 (sub,gap,sub,gap); the last slot holds cell B-1 and (high,pos)
//                                 (overall_sequence_high_score[_pos], ram_extend.c:900-901).
//                                 Exactly 16*B+16 bytes per flank, read once and written once per
//                                 column: the algorithmic traffic of SURVEY.md section 8(d).
//   trim   int2   [Np]            trimmed_sequence_high_score[_pos] (ram_extend.c:902-903)
//   sums   int64  [3][32][4]      sharded per-candidate column sums (vote), rotated by column
//   ctl    RamxCtl[2]             stop-rule state, flip-flopped by column
//
// Column step K(r), one launch per consensus column, ONE LANE PER FLANK:
//   every wave:  fold the 32x4 vote shards of row r -> besta(r), new-max / stop decision
//   every lane:  stream its previous row S(r-1) (coalesced 16 B loads, 8 deep in flight),
//                compute row r against besta(r) -> S(r) (stored), track best cell -> high/pos,
//                and, skewed by one cell, the four candidate rows r+1 from S(r) (never stored) ->
//                capped contributions -> wave shuffle reduce -> LDS block reduce -> 4 int64 atomics
//                into one of 32 shards.
// The serial dependency through the insertion term (cells -W..+W in order) stays a plain serial
// loop inside the lane, so the values are the reference's bit for bit; parallelism comes from the
// flanks (N >= 64 k fills the chip).  No MFMA: integer max-plus recurrences, HBM-bound.

#include <hip/hip_runtime.h>
#include <utility>
#include <rccl/rccl.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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
};

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512];

extern "C" void ramx_set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *ramx_last_error(void) { return g_err; }

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      ramx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); \
      return RAMX_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// ------------------------------------------------------------------------------------------
// pack kernel: 1-byte library -> transposed, pre-oriented 4-bit windows
// ------------------------------------------------------------------------------------------
__global__ void ramx_pack_kernel(const signed char *__restrict__ lib, unsigned long long lib_len,
                                 const ramx_flank *__restrict__ fl, int Nx, int Np, int W,
                                 unsigned *__restrict__ bases, int2 *__restrict__ bounds)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (n >= Np) return;
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
template <bool INIT, bool FIN, bool OOB, bool CHAIN>
__device__ __forceinline__ void band_step(const int go, const int ge, const int W, const StepU u, const StepT t,
                                          const int Pm, const int PeNext, LaneDP &L, int &outM, int &outE)
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
template <bool INIT, bool OOB, bool CHAIN>
__device__ __forceinline__ void run_band(const KArgs &a, const int r, const int *s_tab, const int4 *Sin, int4 *Sout,
                                         const unsigned *bp, const int jlo, const int jhi, LaneDP &D, int &high, int &pos,
                                         int4 (&buf)[PF], int4 (&far)[PF], unsigned w0, unsigned w1, unsigned w2)
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
    band_step<INIT, true, OOB, CHAIN>(go, ge, W, make_u(2 * q), t0, cur.x, cur.w, D, m0, e0);
    band_step<INIT, true, OOB, CHAIN>(go, ge, W, make_u(2 * q + 1), t1, cur.z, nxt.y, D, m1, e1);
#if defined(RAMX_DBG_NOMEM) || defined(RAMX_DBG_NOSTORE)
    if (m0 == 0x7fffffff) Sout[(size_t)q * 64] = make_int4(m0, e0, m1, e1);
#else
    st_stream(Sout + (size_t)q * 64, make_int4(m0, e0, m1, e1));
#endif
  };
  auto final_slot = [&](int4 cur, const StepT &t0, const StepT &t1) {
    int m0, e0, m1, e1;
    // cell B-1 has no deletion predecessor (bnw_extend.c:892); then the virtual step j = B
    band_step<INIT, true, OOB, CHAIN>(go, ge, W, make_u(2 * W), t0, cur.x, NEG, D, m0, e0);
    band_step<INIT, false, OOB, CHAIN>(go, ge, W, make_u(B), t1, 0, 0, D, m1, e1);
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

template <bool INIT, bool CHAIN, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ramx_column_kernel(const KArgs a)
{
  constexpr int WPB = BLOCK / 64;
  __shared__ __attribute__((aligned(16))) int s_tab[TAB_ROWS * TAB_STRIDE];
  __shared__ long long s_red[WPB][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  // ---- loads that do not depend on the vote are issued first, so that the prologue below (a dependent
  // round trip to the control block and the vote shards) overlaps with them: two rings of PF slots
  // (`buf` = slots 0..7, `far` = 8..15; inside the band every load is issued 2*PF slots = 32 band steps ahead
  // of its use, ~16 KB in flight per wave), the flank bounds and the first three base words.
  const int W = a.W, B = 2 * W + 1, Q = W + 1, r = a.r;
  const int tile = blockIdx.x * WPB + wave;
  const bool live = tile < (a.Np >> 6);
  const int n = (live ? tile : 0) * 64 + lane;
  const int4 *Sin = a.S_in + (size_t)(live ? tile : 0) * Q * 64 + lane;
  int4 *Sout = a.S_out + (size_t)(live ? tile : 0) * Q * 64 + lane;
  // base stream: step j reads nibble t'' = j + r + 8 (the packed windows carry one leading pad word so that
  // r = -1 stays non-negative); per group of 16 steps the nibbles sit at ph .. ph+15 of three words
  const unsigned *bp = a.bases + (size_t)((r + 8) >> 3) * a.Np + n;
  int4 buf[PF], far[PF];
  if (!INIT)
  {
#pragma unroll
    for (int i = 0; i < PF; i++) buf[i] = ld_stream(Sin + (size_t)(i < Q ? i : Q - 1) * 64);
#pragma unroll
    for (int i = 0; i < PF; i++) far[i] = ld_stream(Sin + (size_t)(i + PF < Q ? i + PF : Q - 1) * 64);
  }
  const int2 bd = a.bounds[n];
  const unsigned w0 = bp[0], w1 = bp[(size_t)a.Np], w2 = bp[2 * (size_t)a.Np];

  // ---- vote for row r, stop rule (every wave, redundantly; block 0 publishes) -------------
  int besta = 0;
  bool new_max = false;
  if (!INIT)
  {
    long long v[4] = { 0, 0, 0, 0 };
    if (lane < a.nshards_in)
    {
      const long long *p = a.sums_in + lane * 4;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    }
    const RamxCtl c = *a.ctl_in;
    if (c.stopped)           // uniform: the host runs ahead of the device-side stop decision
    {
      if (blockIdx.x == 0 && threadIdx.x == 0) *a.ctl_out = c;   // keep both flip-flop slots stopped
      return;
    }
    long long curr = 0;      // ram_extend.c:973-974
    int ovf = c.overflow;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      v[k] = wave_sum_ll(v[k]);
      if (v[k] > 2147483647LL || v[k] < -2147483648LL) ovf = 1;
      if (v[k] > curr) { curr = v[k]; besta = k; }   // :1081-1085 strict >, ties -> lowest base
    }
    int dist = c.max_row - a.r;
    dist = dist < 0 ? -dist : dist;
    new_max = curr >= c.max_ext + (long long)dist * a.minimp;   // :1194-1196
    const int max_row = new_max ? a.r : c.max_row;
    const long long max_ext = new_max ? curr : c.max_ext;
    int d2 = a.r - max_row;
    d2 = d2 < 0 ? -d2 : d2;
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
      RamxCtl o;
      o.max_ext = max_ext; o.max_row = max_row; o.stopped = (d2 >= a.when_to_stop) ? 1 : 0;   // :1216
      o.rows_done = a.r + 1; o.overflow = ovf; o.besta = besta; o.pad = 0;
      *a.ctl_out = o;
      a.cons_out[a.r] = (signed char)besta;   // :1092-1095 (host scatters into master[])
    }
  }
  else if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = 0; o.max_row = -1; o.stopped = 0; o.rows_done = 0; o.overflow = 0; o.besta = 0; o.pad = 0;
    *a.ctl_out = o;
  }
  if (threadIdx.x < TAB_ROWS * TAB_STRIDE)
  {
    const int row = threadIdx.x / TAB_STRIDE, col = threadIdx.x % TAB_STRIDE;
    int v = 0;
    if (row < RAMX_NCLASS) v = (col < 4) ? a.tab[row][col] : (col == 4 ? a.tab[row][besta] : 0);
    s_tab[threadIdx.x] = v;
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < NSHARD * 4; i += BLOCK) a.sums_zero[i] = 0;
  __syncthreads();

  // ---- the band: one lane = one flank -----------------------------------------------------
  int contrib[4] = { 0, 0, 0, 0 };              // each in [0, 2^31)
  if (live)
  {
    const int jlo = bd.x - r, jhi = bd.y - r;        // cell j (row r) / j-1 (row r+1) is in bounds iff jlo <= j <= jhi
    LaneDP D;
    D.eC = NEG; D.mPrev = NEG - 1000000; D.bestF = NEG; D.jbest = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { D.eA[c] = NEG; D.bestA[c] = NEG; }
    int high = 0, pos = 0;
    // wave-uniform choice: do all 64 flanks cover every cell of both rows?  (steps 0..B)
    const bool all_in = !INIT && __all((jlo <= 0) && (jhi >= B));
    if (all_in) run_band<INIT, false, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
    else run_band<INIT, true, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
    if (INIT || new_max) a.trim[n] = make_int2(high, pos);   // ram_extend.c:1203-1207 (913-914 at init)
    if (n < a.Nx)
    {
      const int capv = high + a.cap;
#pragma unroll
      for (int c = 0; c < 4; c++)
      {
        const int b = D.bestA[c] < 0 ? 0 : D.bestA[c];         // ram_extend.c:1042
        contrib[c] = (b >= capv) ? b : capv;                   // :1052-1062
      }
    }
  }

  // ---- 64 lanes -> wave -> block -> one int64 atomic per candidate into this block's shard ----
  {
    long long tot[4];
#pragma unroll
    for (int c = 0; c < 4; c++) tot[c] = wave_sum_nonneg31(contrib[c]);
    if (lane == 0)
    {
#pragma unroll
      for (int c = 0; c < 4; c++) s_red[wave][c] = tot[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 4)
  {
    long long t = 0;
#pragma unroll
    for (int wv = 0; wv < WPB; wv++) t += s_red[wv][threadIdx.x];
    atomicAdd((unsigned long long *)(a.sums_out + (blockIdx.x % NSHARD) * 4 + threadIdx.x), (unsigned long long)t);
  }
}

// ------------------------------------------------------------------------------------------
// persistent kernel: the whole direction in ONE launch, DP rows resident in registers
// ------------------------------------------------------------------------------------------
//
// For W known at compile time and N <= (resident waves) x 64 the row of a flank (2 x (2W+1) int32) fits the
// lane's registers, so the row never travels: HBM sees only the base words (~12 words per flank per column) and
// the 32-byte vote.  All L columns run inside one cooperative launch; the dependent-launch boundary of the
// streaming kernel becomes a device-wide barrier that is fused with the vote:
//
//   column c, every block:   4 x int64 atomicAdd into shard (blockIdx % 32) of vote set (c+1) % 3.  Each add
//                            carries its own arrival ticket: value = partial_sum + 2^41 + 2^54, so bits 54..63 of
//                            a shard word count the blocks that have contributed and the low 54 bits hold
//                            sum + count * 2^41 (|partial| <= 512 lanes * 2^31 < 2^41: exact for any input).
//   column c+1, wave 0:      lanes 0..31 poll "their" shard's four words (relaxed agent-scope loads + s_sleep,
//                            bounded) until all four show every block of the shard, decode, shuffle-reduce and
//                            publish the vote through LDS.  One fabric round trip after the last arrival.
//   Set (c+2) % 3 is zeroed by block 0 during column c, before block 0's own adds (everybody finished reading it
//   before contributing to column c; nobody adds to it before block 0 itself has contributed to column c+1).
//
// Placement independent: only agent-scope atomics / atomic loads touch shared words, no assumption on which
// XCD a block runs; co-residency is checked by hipLaunchCooperativeKernel and every spin is bounded (a timeout
// raises `err` and every block leaves).  Multi-GPU runs keep the per-column launches (RCCL sits between them).

struct PShard { unsigned long long word[4]; unsigned long long pad[4]; };   // 64 B: one cache line per shard
#define PRK_BIAS (1ULL << 41)
#define PRK_TICKET (1ULL << 54)

// Multi-GPU: every rank owns one PeerBox in fine-grained device memory, mapped into all other ranks through
// hipIpc handles.  After a rank's own blocks have all contributed to a column, its block 0 stores the rank's four
// totals into slot [set][rank] of EVERY box (its own included) over xGMI; each word carries the column number in
// its top 16 bits, so a reader knows a word is current without any flag or fence; every block then polls the local
// box until all ranks' words of this column are there.  3 sets rotate exactly like the vote shards.
#define RAMX_MAX_RANKS 16
struct PeerBox { unsigned long long slot[3][RAMX_MAX_RANKS][4]; unsigned long long token[RAMX_MAX_RANKS]; };
#define PEER_VBIAS (1LL << 46)
#define PEER_VMASK ((1ULL << 48) - 1)

struct PArgs
{
  int4 *S;                      // row state in HBM: read at start (boundary row from K(-1)), written back at the end
  const unsigned *bases;
  const int2 *bounds;
  int2 *trim;
  const long long *sums0;       // vote shards of row 0, produced by K(-1)
  PShard *vote;                 // [3][NSHARD]
  RamxCtl *ctl_out;
  signed char *cons_out;
  unsigned *err;                // != 0: a bounded spin gave up
  PeerBox *const *peers;        // [nranks] every rank's box as seen from this device (NULL on one GPU)
  PeerBox *box;                 // this rank's own box
  int rank, nranks;
  int Np, Nx, r0, L, go, ge, cap, minimp, when_to_stop, nblocks;
  int tab[RAMX_NCLASS][4];
  int pack_ok;                  // every reachable score fits 27 bits: the fast path may pack (score, cell) keys
  unsigned long long *dbg;      // -DRAMX_PRK_TIMING builds only: [block][8] phase sums in 10 ns ticks
};

#define PRK_SPIN_LIMIT (1u << 22)
#ifdef RAMX_PRK_TIMING
#define PRK_TICK(k) do { const unsigned long long t_ = wall_clock64(); tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define PRK_TICK(k) do { } while (0)
#endif

// Row state of the persistent kernel: m[B] in registers; e is kept as the 16-bit difference d = e - m in LDS.
// With go <= 0:  m + go + ge <= e <= m + ge  (e = max(sub+go, gap) + ge, m = max(sub, gap)), so d lies in
// [go + ge, ge] and int16 is exact whenever go + ge >= -32768 (checked on the host).  Each lane owns one dword per
// cell pair (layout [j/2][thread] dwords, halves by parity of j): conflict-free ds_read_i16 / ds_write_b16.
#ifndef PRK_OOB_GROUP
#define PRK_OOB_GROUP 2
#endif
#ifndef PRK_FAST_GROUP
#define PRK_FAST_GROUP 8
#endif
#ifndef PRK_FETCH_AHEAD
#define PRK_FETCH_AHEAD 2
#endif
__device__ __forceinline__ int vmax3(int x, int y, int z)   // forced v_max3_i32 (keeps the compiler from re-associating)
{
  int d;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}

template <class F, int... Js>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, Js...>)
{
  (f(std::integral_constant<int, Js>{}), ...);
}

// Score table of the in-bounds fast path, addressed straight from the packed base stream with ONE SDWA instruction:
// the eight nibbles of an aligned base word sit in four bytes; the LDS byte offset of the row of a class is
// (byte & 0xF0) for the high nibble and (byte << 4) for the low nibble of a word whose high nibbles have been cleared
// (one v_and per eight cells).  16 rows of 16 bytes: lanes reading the same class broadcast, different classes sit in
// different banks.  A row is {M[A][b] | M[C][b] | M[G][b] | M[T][b] as four int8, M[besta][b], -, -}: one ds_read_b64
// per cell; the candidates' scores are consumed by sign-extending SDWA adds.  Dword 1 is rewritten for every column
// (the winner changes), by wave 0 / before a block barrier.  Requires every score in [-128, 127] (checked on the
// host together with the key-packing bound).
struct FastTabs
{
  int row[16][4];
};

template <int BYTE>
__device__ __forceinline__ unsigned nib_lo_x16(unsigned A)   // ((A >> 8*BYTE) & 0xff) << 4
{
  unsigned d;
  if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(A));
  else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(A));
  else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(A));
  else asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(A));
  return d;
}
template <int BYTE>
__device__ __forceinline__ unsigned nib_hi_x16(unsigned A, unsigned mask_f0)   // (A >> 8*BYTE) & 0xf0
{
  unsigned d;
  if (BYTE == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "s"(mask_f0), "v"(A));
  else if (BYTE == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "s"(mask_f0), "v"(A));
  else if (BYTE == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "s"(mask_f0), "v"(A));
  else asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "s"(mask_f0), "v"(A));
  return d;
}

template <int BYTE>
__device__ __forceinline__ int add_sext_byte(int x, int packed)   // x + (int)(signed char)(packed >> 8*BYTE)
{
  int d;
  if (BYTE == 0) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(x), "v"(packed));
  else if (BYTE == 1) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(x), "v"(packed));
  else if (BYTE == 2) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(x), "v"(packed));
  else asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(x), "v"(packed));
  return d;
}

// candidate bytes: once per launch.
template <int BLOCK>
__device__ __forceinline__ void fast_tabs_init(FastTabs &ft, const int (&tab)[RAMX_NCLASS][4])
{
  if (threadIdx.x < 16)
  {
    const int cls = threadIdx.x;
    unsigned pk = 0;
    if (cls < RAMX_NCLASS)
      pk = ((unsigned)tab[cls][0] & 0xffu) | (((unsigned)tab[cls][1] & 0xffu) << 8) | (((unsigned)tab[cls][2] & 0xffu) << 16) |
           (((unsigned)tab[cls][3] & 0xffu) << 24);
    ft.row[cls][0] = (int)pk; ft.row[cls][1] = 0; ft.row[cls][2] = 0; ft.row[cls][3] = 0;
  }
}
// winner dword of the column whose winner is `besta` (threads 0..15 of the caller's group)
__device__ __forceinline__ void fast_tabs_winner(FastTabs &ft, const int *tab_besta /* old-format table of besta */, int i)
{
  if (i < 16) ft.row[i][1] = i < RAMX_NCLASS ? tab_besta[i * TAB_STRIDE + 4] : 0;
}

// In-bounds, chain-free band of the register-resident kernels (the steady state of a run): per cell
//   sub = Pm + sF;  m = max3(sub, eC, Pe);  e = max3(sub + go, eC, Pe) + ge          (5 VALU, chain of 2)
// The four candidates take two cells per v_max3; their shared deletion term max_k e_k is folded in at the end.
// The best cell of the row (value, lowest index on ties: bnw_extend.c:1020-1024) is tracked as a packed key
// (m << 4) | (15 - (j & 15)) per 16-cell group, two cells per v_max3; exact while |m| < 2^27 (checked on the host).
template <int W, int BLOCK>
__device__ __forceinline__ void prk_band_fast(const int go, const int ge, const FastTabs &ft, short *sD, const int r,
                                              const unsigned (&w)[(2 * W + 1 + 8) / 8 + 2], int (&M)[2 * W + 1], LaneDP &D)
{
  constexpr int B = 2 * W + 1, NG = (B + 15) / 16;
  const int ph4 = 4 * ((r + 8) & 7);
  const unsigned mask_f0 = 0xf0u;
  short *myD = sD + 2 * threadIdx.x;
  const char *tb = reinterpret_cast<const char *>(&ft.row[0][0]);
  int eC = NEG, mPrev = NEG, maxE = NEG, ePend = NEG, kPend = NEG;
  int bA[4] = { NEG, NEG, NEG, NEG }, pend[4] = { NEG, NEG, NEG, NEG };
  int kg[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) kg[g] = -2147483647 - 1;
  // compile-time cell index: the band is generated step by step (no reliance on the loop unroller, whose size limit
  // would otherwise leave the row in scratch memory).  Table rows and the previous row's e are fetched PD steps ahead
  // of their use (a lone wave per SIMD -- one family per workgroup -- has nobody to hide the LDS latency behind).
  constexpr int PD = PRK_FETCH_AHEAD;
  unsigned A = 0, Alo = 0;
  int2 rowQ[PD];                                    // {candidate bytes, M[besta][base]} of steps j .. j+PD-1
  int dQ[PD];                                       // e - m of the previous row's cells j+1 .. j+PD
  auto fetch_row = [&](auto jc) __attribute__((always_inline))
  {
    constexpr int jn = decltype(jc)::value;         // the step whose base is looked up
    if constexpr ((jn & 7) == 0 || jn == 0)
    {
      A = __builtin_amdgcn_alignbit(w[(jn >> 3) + 1], w[jn >> 3], ph4);
      Alo = A & 0x0f0f0f0fu;                        // low nibbles only: (byte << 4) is then the row offset of the class
    }
    constexpr int byte = (jn & 7) / 2;
    unsigned off;
    if constexpr ((jn & 1) == 0) off = nib_lo_x16<byte>(Alo);
    else off = nib_hi_x16<byte>(A, mask_f0);
    return *reinterpret_cast<const int2 *>(tb + off);
  };
  static_for([&](auto kc) __attribute__((always_inline))
  {
    constexpr int k = decltype(kc)::value;
    if constexpr (k <= B) rowQ[k] = fetch_row(std::integral_constant<int, (k <= B ? k : 0)>{});
    else rowQ[k] = make_int2(0, 0);
    dQ[k] = (k + 1 < B) ? (int)myD[((k + 1) >> 1) * (2 * BLOCK) + ((k + 1) & 1)] : 0;
  }, std::make_integer_sequence<int, PD>{});
  auto step = [&](auto jc) __attribute__((always_inline))
  {
    constexpr int j = decltype(jc)::value;
    if constexpr ((j & (PRK_FAST_GROUP - 1)) == 0)
    {
      // pin the accumulators to their group: nothing but data dependences orders pure arithmetic against
      // sched_barrier during instruction selection, and a sunk accumulation keeps every table row alive
      asm volatile("" ::"v"(bA[0]), "v"(bA[1]), "v"(bA[2]), "v"(bA[3]), "v"(maxE), "v"(kg[(j > 0 ? j - 1 : 0) >> 4]));
      __builtin_amdgcn_sched_barrier(0);
    }
    const int sv = rowQ[0].x, sF = rowQ[0].y;
    const int dCur = dQ[0];
#pragma unroll
    for (int k = 0; k + 1 < PD; k++) { rowQ[k] = rowQ[k + 1]; dQ[k] = dQ[k + 1]; }
    if constexpr (j + PD <= B) rowQ[PD - 1] = fetch_row(std::integral_constant<int, (j + PD <= B ? j + PD : 0)>{});
    if constexpr (j + PD + 1 < B) dQ[PD - 1] = (int)myD[((j + PD + 1) >> 1) * (2 * BLOCK) + ((j + PD + 1) & 1)];
    // candidates' cell j-1 of row r+1: substitution from m_{j-1} of row r (the base of that cell is this step's)
    if constexpr (j >= 1)
    {
      const int t4[4] = { add_sext_byte<0>(mPrev, sv), add_sext_byte<1>(mPrev, sv), add_sext_byte<2>(mPrev, sv), add_sext_byte<3>(mPrev, sv) };
      if constexpr ((j & 1) != 0)
      {
#pragma unroll
        for (int c = 0; c < 4; c++) pend[c] = t4[c];
      }
      else
      {
#pragma unroll
        for (int c = 0; c < 4; c++) bA[c] = imax3(bA[c], pend[c], t4[c]);
      }
    }
    if constexpr (j < B)
    {
      const int Pm = M[j];
      int Pe = NEG;
      if constexpr (j + 1 < B) Pe = M[j + 1] + dCur;
      const int sub = Pm + sF;                       // bnw_extend.c:950-956
      const int m = vmax3(sub, eC, Pe);              // max(sub, max(ins, del)), :1007-1018
      const int e = vmax3(sub + go, eC, Pe) + ge;
      M[j] = m;
      myD[(j >> 1) * (2 * BLOCK) + (j & 1)] = (short)(e - m);
      const int key = (int)(((unsigned)m << 4) | (unsigned)(15 - (j & 15)));
      if constexpr ((j & 1) == 0 && j + 1 < B) kPend = key;
      else if constexpr ((j & 1) != 0) kg[j >> 4] = imax3(kg[j >> 4], kPend, key);
      else kg[j >> 4] = imax(kg[j >> 4], key);
      // deletion term of candidate cell j-1 is e_j (cells 1..B-1)
      if constexpr (j >= 1)
      {
        if constexpr ((j & 1) != 0) ePend = e;
        else maxE = imax3(maxE, ePend, e);
      }
      mPrev = m;
      eC = e;
    }
  };
  static_for(step, std::make_integer_sequence<int, B + 1>{});
  // B is odd: the last candidate term (step B) is still pending; B-1 is even: every e has been folded
#pragma unroll
  for (int c = 0; c < 4; c++) D.bestA[c] = imax3(bA[c], (B & 1) ? pend[c] : NEG, maxE);
  // best cell: highest value, lowest group on ties (inside a group the key already prefers the lowest cell)
  int bestv = kg[NG - 1] >> 4, bkey = kg[NG - 1], bg = NG - 1;
#pragma unroll
  for (int g = NG - 2; g >= 0; g--)
  {
    const int v = kg[g] >> 4;
    const bool take = v >= bestv;
    bestv = take ? v : bestv;
    bkey = take ? kg[g] : bkey;
    bg = take ? g : bg;
  }
  D.bestF = bestv;
  D.jbest = 16 * bg + 15 - (bkey & 15);
}

template <int W, bool OOB, int BLOCK, bool INIT = false>
__device__ __forceinline__ void prk_band(const int go, const int ge, const int *s_tab, const FastTabs &ft, short *sD, const int r,
                                         const unsigned (&w)[(2 * W + 1 + 8) / 8 + 2], const int jlo, const int jhi,
                                         int (&M)[2 * W + 1], LaneDP &D)
{
  constexpr int B = 2 * W + 1;
  if (!OOB && !INIT)
  {
    prk_band_fast<W, BLOCK>(go, ge, ft, sD, r, w, M, D);
    return;
  }
  const int edgeF = (r < W) ? go + (r + 1) * ge : SENT;
  const int edgeC = (r + 1 < W) ? go + (r + 2) * ge : SENT;
  const int ph4 = 4 * ((r + 8) & 7);
  short *myD = sD + 2 * threadIdx.x;
#pragma unroll
  for (int j = 0; j <= B; j++)
  {
    // the row itself occupies B registers: keep the scheduler from hoisting every table lookup of the fully
    // unrolled band to the top.  This is the rarely taken masked path: small groups, lowest register pressure
    if ((j & (PRK_OOB_GROUP - 1)) == 0) __builtin_amdgcn_sched_barrier(0);
    const unsigned A = __builtin_amdgcn_alignbit(w[(j >> 3) + 1], w[j >> 3], ph4);
    const unsigned bc = (A >> (4 * (j & 7))) & 15u;
    StepU u;
    u.j = j; u.first = (j == 0); u.hi = 2147483647;
    u.vF = (j < W) ? edgeF : SENT;
    u.vC = u.first ? NEG : ((j - 1 < W) ? edgeC : SENT);
    const StepT t = fetch_step<OOB>(s_tab, bc, (j >= jlo) && (j <= jhi), j == 0);
    if (j < B)
    {
      const int Pm = M[j];
      int PeNext = NEG;
      if (j + 1 < B) PeNext = M[j + 1] + (int)myD[((j + 1) >> 1) * (2 * BLOCK) + ((j + 1) & 1)];   // previous row's e of cell j+1
      int m, e;
      band_step<INIT, true, OOB, false>(go, ge, W, u, t, Pm, PeNext, D, m, e);
      M[j] = m;
      myD[(j >> 1) * (2 * BLOCK) + (j & 1)] = (short)(e - m);
    }
    else
    {
      int dm, de;
      band_step<INIT, false, OOB, false>(go, ge, W, u, t, 0, 0, D, dm, de);
    }
  }
}

template <int W, int BLOCK>
__global__ __launch_bounds__(BLOCK, 2) void ramx_persistent_kernel(const PArgs a)
{
  constexpr int B = 2 * W + 1, Q = W + 1, NW = (B + 8) / 8 + 2, WPB = BLOCK / 64, RS = 2 * BLOCK;   // RS: shorts per cell-pair row of sD
  // one object, tables first: their LDS addresses must fit the 16-bit offset field of the ds_read that uses them
  struct Smem
  {
    FastTabs ft;
    int tab4[4][TAB_ROWS * TAB_STRIDE];                // one score table per winner base (masked path)
    long long red[WPB][4];
    long long vote[4];
    int fail, pad[3];
    short d[((B + 1) / 2) * RS];                       // d = e - m, [cell pair][thread][parity]
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
  FastTabs &s_ft = sm.ft;
  int (&s_tab4)[4][TAB_ROWS * TAB_STRIDE] = sm.tab4;
  long long (&s_red)[WPB][4] = sm.red;
  long long (&s_vote)[4] = sm.vote;
  int &s_fail = sm.fail;
  short *sD = sm.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * WPB + wave;
  const bool live = tile < (a.Np >> 6);
  const int n = (live ? tile : 0) * 64 + lane;
  int4 *S = a.S + (size_t)(live ? tile : 0) * Q * 64 + lane;
  short *myD = sD + 2 * threadIdx.x;

  // ---- row state -> registers (m) and LDS (e - m) -----------------------------------------
  int M[B];
  int high, pos, thigh = 0, tpos = 0;
  {
#pragma unroll
    for (int q = 0; q < W; q++)
    {
      const int4 v = S[(size_t)q * 64];
      M[2 * q] = v.x; M[2 * q + 1] = v.z;
      myD[q * RS] = (short)(v.y - v.x); myD[q * RS + 1] = (short)(v.w - v.z);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int4 v = S[(size_t)W * 64];
    M[B - 1] = v.x; myD[W * RS] = (short)(v.y - v.x); high = v.z; pos = v.w;
  }
  const int2 bd = a.bounds[n];
  const int shard = blockIdx.x % NSHARD;
  const int my_shard_blocks = (a.nblocks - (lane & (NSHARD - 1)) + NSHARD - 1) / NSHARD;   // wave 0: blocks arriving on shard `lane & 31`

  long long max_ext = 0;
  int max_row = -1, rows_done = 0, ovf = 0, stopped = 0, failed = 0;
  if (threadIdx.x == 0) s_fail = 0;
  for (int i = threadIdx.x; i < 4 * TAB_ROWS * TAB_STRIDE; i += BLOCK)
  {
    const int bt = i / (TAB_ROWS * TAB_STRIDE), e = i % (TAB_ROWS * TAB_STRIDE), row = e / TAB_STRIDE, col = e % TAB_STRIDE;
    int v = 0;
    if (row < RAMX_NCLASS) v = (col < 4) ? a.tab[row][col] : (col == 4 ? a.tab[row][bt] : 0);
    s_tab4[bt][e] = v;
  }
  fast_tabs_init<BLOCK>(s_ft, a.tab);
  __syncthreads();

#ifdef RAMX_PRK_TIMING
  unsigned long long tsum[6] = { 0, 0, 0, 0, 0, 0 }, tlast = wall_clock64();
#endif
  for (int r = 0; r < a.L; r++)
  {
    PRK_TICK(5);
    // ---- base words of this column (independent of the vote: issued before the wait) -------
    unsigned w[NW];
    {
      const unsigned *bp = a.bases + (size_t)((r + 8) >> 3) * a.Np + n;
#pragma unroll
      for (int k = 0; k < NW; k++) w[k] = bp[(size_t)k * a.Np];
    }
    // ---- vote of row r -----------------------------------------------------------------------
    if (wave == 0)
    {
      long long v[4] = { 0, 0, 0, 0 };
      if (r == 0)
      {
        if (lane < NSHARD) { const long long *p = a.sums0 + lane * 4; v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; }
      }
      else
      {
        // lane = shard + 32 * half polls words 2*half, 2*half+1 of "its" shard: one 16-byte load per lane and round
        // (a quarter of the requests of four 8-byte loads on 32 lanes; the poll competes with the adds it waits for)
        const int sidx = lane & (NSHARD - 1), half = lane >> 5;
        const unsigned long long *src = &a.vote[(size_t)(r % 3) * NSHARD + sidx].word[2 * half];
        unsigned spins = 0;
        bool done = my_shard_blocks <= 0 || (a.nranks > 1 && blockIdx.x != 0);   // multi-rank: only the exchanger needs the local total
        unsigned long long x0 = 0, x1 = 0;
        for (;;)
        {
          if (!done)
          {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u q;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(src) : "memory");
            x0 = ((unsigned long long)q.y << 32) | q.x;
            x1 = ((unsigned long long)q.w << 32) | q.z;
            done = (x0 >> 54) >= (unsigned long long)my_shard_blocks && (x1 >> 54) >= (unsigned long long)my_shard_blocks;
          }
          if (__all(done)) break;
          if (++spins > PRK_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
          {
            failed = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        long long y0 = 0, y1 = 0;
        if (my_shard_blocks > 0 && !failed && !(a.nranks > 1 && blockIdx.x != 0))
        {
          y0 = (long long)(x0 & (PRK_TICKET - 1)) - (long long)(x0 >> 54) * (long long)PRK_BIAS;
          y1 = (long long)(x1 & (PRK_TICKET - 1)) - (long long)(x1 >> 54) * (long long)PRK_BIAS;
        }
        // fold the 32 shards inside each half-wave; lanes 0 / 32 end up with words {0,1} / {2,3}
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) { y0 += __shfl_xor(y0, m, 64); y1 += __shfl_xor(y1, m, 64); }
        v[0] = __shfl(y0, 0, 64); v[1] = __shfl(y1, 0, 64); v[2] = __shfl(y0, 32, 64); v[3] = __shfl(y1, 32, 64);
      }
      if (r == 0)
      {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = wave_sum_ll(v[k]);
      }
      if (a.nranks > 1 && r > 0 && !failed)
      {
        // ---- cross-device step: v[] is this rank's total (identical in all lanes) -----------
        const unsigned long long tag = (unsigned long long)(r & 0xffff) << 48;
        if (blockIdx.x == 0 && lane < a.nranks)
        {
          PeerBox *pb = a.peers[lane];
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            if (v[k] >= PEER_VBIAS || v[k] <= -PEER_VBIAS) failed = 1;     // cannot be encoded: fail loudly
            __hip_atomic_store(&pb->slot[r % 3][a.rank][k], tag | ((unsigned long long)(v[k] + PEER_VBIAS) & PEER_VMASK),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        unsigned long long y[4] = { 0, 0, 0, 0 };
        bool got = lane >= a.nranks;
        unsigned spins = 0;
        for (;;)
        {
          if (!got)
          {
#pragma unroll
            for (int k = 0; k < 4; k++) y[k] = __hip_atomic_load(&a.box->slot[r % 3][lane][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            got = (y[0] >> 48) == (tag >> 48) && (y[1] >> 48) == (tag >> 48) && (y[2] >> 48) == (tag >> 48) && (y[3] >> 48) == (tag >> 48);
          }
          if (__all(got)) break;
          if (++spins > PRK_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
          {
            failed = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        failed = __any(failed) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
          v[k] = wave_sum_ll((lane < a.nranks && !failed) ? (long long)(y[k] & PEER_VMASK) - PEER_VBIAS : 0LL);
      }
      {
        // the winner's substitution column of the fast-path tables (same argmax rule as below, ram_extend.c:1064-1086)
        long long cw = 0;
        int bw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (v[k] > cw) { cw = v[k]; bw = k; }
        fast_tabs_winner(s_ft, s_tab4[bw], lane);
      }
      if (lane == 0)
      {
        s_vote[0] = v[0]; s_vote[1] = v[1]; s_vote[2] = v[2]; s_vote[3] = v[3];
        s_fail = failed;
        if (failed) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    PRK_TICK(0);                 // wave 0: vote seen (other waves: nothing)
    __syncthreads();
    PRK_TICK(1);                 // released by the block barrier
    if (__builtin_amdgcn_readfirstlane(s_fail)) { failed = 1; break; }
    long long curr = 0;
    int besta = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      // the vote is wave-uniform: move it to scalar registers so that the whole stop rule runs on the SALU and
      // none of its state (max_ext, max_row, ...) occupies vector registers next to the row
      const long long vv = s_vote[k];
      const long long vk = ((long long)__builtin_amdgcn_readfirstlane((int)(vv >> 32)) << 32) |
                           (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)vv);
      if (vk > 2147483647LL || vk < -2147483648LL) ovf = 1;
      if (vk > curr) { curr = vk; besta = k; }
    }
    int dist = max_row - r;
    dist = dist < 0 ? -dist : dist;
    const bool new_max = curr >= max_ext + (long long)dist * a.minimp;
    if (new_max) { max_row = r; max_ext = curr; }
    int d2 = r - max_row;
    d2 = d2 < 0 ? -d2 : d2;
    stopped = d2 >= a.when_to_stop;
    rows_done = r + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cons_out[r] = (signed char)besta;
    // block 0 clears the vote set of row r+2 (see the protocol above)
    if (blockIdx.x == 0 && threadIdx.x < NSHARD)
    {
      PShard *z = a.vote + (size_t)((r + 2) % 3) * NSHARD + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; k++) __hip_atomic_store(&z->word[k], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int *s_tab = s_tab4[besta];                // column 4 of table `besta` holds M[besta][class]

    // ---- the band, rows in registers ---------------------------------------------------------
    int contrib[4] = { 0, 0, 0, 0 };       // each in [0, 2^31): clamped at 0 below, capped from below by high + cap
    if (live)
    {
      const int jlo = bd.x - r, jhi = bd.y - r;
      LaneDP D;
      D.eC = NEG; D.mPrev = NEG - 1000000; D.bestF = NEG; D.jbest = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) { D.eA[c] = NEG; D.bestA[c] = NEG; }
      const bool all_in = a.pack_ok && __all((jlo <= 0) && (jhi >= B));
      if (all_in) prk_band<W, false, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
      else prk_band<W, true, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
      if (D.bestF > high) { high = D.bestF; pos = r + D.jbest - W; }   // ram_extend.c:1140-1150
      if (new_max) { thigh = high; tpos = pos; }                        // :1203-1207
      if (n < a.Nx)
      {
        const int capv = high + a.cap;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          const int b = D.bestA[c] < 0 ? 0 : D.bestA[c];
          contrib[c] = (b >= capv) ? b : capv;
        }
      }
    }
    PRK_TICK(2);                 // band done
    if (stopped || r == a.L - 1) break;     // the vote of row r+1 will not be consumed
    {
      long long tot[4];
#pragma unroll
      for (int c = 0; c < 4; c++) tot[c] = wave_sum_nonneg31(contrib[c]);
      if (lane == 0)
      {
#pragma unroll
        for (int c = 0; c < 4; c++) s_red[wave][c] = tot[c];
      }
    }
    __syncthreads();
    PRK_TICK(3);                 // wave reduction + block barrier
    if (blockIdx.x == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // block 0: its zeroing stores first
    if (threadIdx.x < 4)
    {
      long long t = 0;
#pragma unroll
      for (int wv = 0; wv < WPB; wv++) t += s_red[wv][threadIdx.x];
      PShard *sh = a.vote + (size_t)((r + 1) % 3) * NSHARD + shard;
      __hip_atomic_fetch_add(&sh->word[threadIdx.x], (unsigned long long)t + PRK_BIAS + PRK_TICKET, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    PRK_TICK(4);                 // contribution issued
  }
#ifdef RAMX_PRK_TIMING
  if (a.dbg != NULL && (threadIdx.x & 63) == 0)
  {
#pragma unroll
    for (int k = 0; k < 6; k++) a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + k] = tsum[k];
  }
#endif

  // ---- write back: rows (so that the device state can be inspected / resumed), trim, control ----
  if (live)
  {
#pragma unroll
    for (int q = 0; q < W; q++)
    {
      S[(size_t)q * 64] = make_int4(M[2 * q], M[2 * q] + (int)myD[q * RS], M[2 * q + 1], M[2 * q + 1] + (int)myD[q * RS + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    S[(size_t)W * 64] = make_int4(M[B - 1], M[B - 1] + (int)myD[W * RS], high, pos);
    a.trim[n] = make_int2(thigh, tpos);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = 0;
    o.pad = failed;
    *a.ctl_out = o;
  }
}

// ------------------------------------------------------------------------------------------
// batch mode: one BLOCK = one family
// ------------------------------------------------------------------------------------------
//
// Real inputs are hundreds of families of ~100 flanks each (util/extend-stk.pl runs one RAMExtend process per
// family).  A family of up to BLOCK flanks fits one workgroup, so its per-column vote is a block-local LDS
// reduction: no device-wide barrier, no atomics, no cooperative launch, any number of families per launch (blocks
// that are not resident simply wait their turn), each family stopping on its own fit-preferred rule.  Rows live in
// registers / LDS exactly as in the persistent kernel; the boundary row and the candidates of row 0 are produced
// in-kernel (column "-1").

struct FamDesc { int tile0, ntiles, nx, id; };      // first 64-flank tile, tiles, flanks of the family; its index in the caller's arrays

struct FArgs
{
  const unsigned *bases;
  const int2 *bounds;
  const FamDesc *fam;
  int2 *trim;                   // per flank
  RamxCtl *ctl_out;             // per family
  signed char *cons_out;        // [family][L]
  int Np, L, go, ge, cap, minimp, when_to_stop;
  int tab[RAMX_NCLASS][4];
  int pack_ok;
};

template <int W, int BLOCK>
__global__ __launch_bounds__(BLOCK, 2) void ramx_family_kernel(const FArgs a)
{
  constexpr int B = 2 * W + 1, NW = (B + 8) / 8 + 2, WPB = BLOCK / 64, RS = 2 * BLOCK;
  struct Smem     // tables first (16-bit ds offsets), see the persistent kernel
  {
    FastTabs ft;
    int tab4[4][TAB_ROWS * TAB_STRIDE];
    long long red[2][WPB][4];
    short d[((B + 1) / 2) * RS];
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
  FastTabs &s_ft = sm.ft;
  int (&s_tab4)[4][TAB_ROWS * TAB_STRIDE] = sm.tab4;
  long long (&s_red)[2][WPB][4] = sm.red;
  short *sD = sm.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const FamDesc fd = a.fam[blockIdx.x];
  const int vwave = wave;
  const bool live = vwave < fd.ntiles;
  const int n = (fd.tile0 + (live ? vwave : 0)) * 64 + lane;
  const bool active = live && (vwave * 64 + lane) < fd.nx;

  for (int i = threadIdx.x; i < 4 * TAB_ROWS * TAB_STRIDE; i += BLOCK)
  {
    const int bt = i / (TAB_ROWS * TAB_STRIDE), e = i % (TAB_ROWS * TAB_STRIDE), row = e / TAB_STRIDE, col = e % TAB_STRIDE;
    int v = 0;
    if (row < RAMX_NCLASS) v = (col < 4) ? a.tab[row][col] : (col == 4 ? a.tab[row][bt] : 0);
    s_tab4[bt][e] = v;
  }
  fast_tabs_init<BLOCK>(s_ft, a.tab);
  __syncthreads();

  int M[B];
#pragma unroll
  for (int j = 0; j < B; j++) M[j] = 0;
  int high = 0, pos = 0, thigh = 0, tpos = 0;
  const int2 bd = a.bounds[n];
  long long max_ext = 0;
  int max_row = -1, rows_done = 0, ovf = 0, stopped = 0;

  // base words of the lane's window: the window moves by one nibble per column, so the words are carried across
  // columns and ONE new word is loaded every eighth column (its first use is at the far end of the band)
  unsigned w[NW];
  {
    const unsigned *bp = a.bases + n;               // column -1 starts at word (r + 8) >> 3 = 0
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = bp[(size_t)k * a.Np];
  }
  for (int r = -1; r < a.L; r++)
  {
    if (r >= 0 && ((r + 8) & 7) == 0)
    {
#pragma unroll
      for (int k = 0; k + 1 < NW; k++) w[k] = w[k + 1];
      w[NW - 1] = a.bases[(size_t)(((r + 8) >> 3) + NW - 1) * a.Np + n];
    }
    int besta = 0;
    bool new_max = false;
    if (r >= 0)
    {
      // vote of row r: block-local (written at the end of the previous iteration, double buffered)
      long long curr = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        long long vv = 0;
#pragma unroll
        for (int wv = 0; wv < WPB; wv++) vv += s_red[r & 1][wv][k];
        // wave-uniform: keep the vote and the stop rule on the scalar unit (see the persistent kernel)
        const long long vk = ((long long)__builtin_amdgcn_readfirstlane((int)(vv >> 32)) << 32) |
                             (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)vv);
        if (vk > 2147483647LL || vk < -2147483648LL) ovf = 1;
        if (vk > curr) { curr = vk; besta = k; }
      }
      int dist = max_row - r;
      dist = dist < 0 ? -dist : dist;
      new_max = curr >= max_ext + (long long)dist * a.minimp;
      if (new_max) { max_row = r; max_ext = curr; }
      int d2 = r - max_row;
      d2 = d2 < 0 ? -d2 : d2;
      stopped = d2 >= a.when_to_stop;
      rows_done = r + 1;
      if (threadIdx.x == 0) a.cons_out[(size_t)fd.id * a.L + r] = (signed char)besta;
    }
    const int *s_tab = s_tab4[besta];
    if (r >= 0 && a.pack_ok)
    {
      // winner rows of the fast-path tables; everybody has left the previous column's band (barrier at its end)
      fast_tabs_winner(s_ft, s_tab, threadIdx.x);
      __syncthreads();
    }
    int contrib[4] = { 0, 0, 0, 0 };
    if (live)
    {
      const int jlo = bd.x - r, jhi = bd.y - r;
      LaneDP D;
      D.eC = NEG; D.mPrev = NEG - 1000000; D.bestF = NEG; D.jbest = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) { D.eA[c] = NEG; D.bestA[c] = NEG; }
      if (r < 0)
        prk_band<W, true, BLOCK, true>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
      else
      {
        const bool all_in = a.pack_ok && __all((jlo <= 0) && (jhi >= B));
        if (all_in) prk_band<W, false, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
        else prk_band<W, true, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
        if (D.bestF > high) { high = D.bestF; pos = r + D.jbest - W; }
        if (new_max) { thigh = high; tpos = pos; }
      }
      if (active)
      {
        const int capv = high + a.cap;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          const int b = D.bestA[c] < 0 ? 0 : D.bestA[c];
          contrib[c] = (b >= capv) ? b : capv;
        }
      }
    }
    if (stopped || r == a.L - 1) break;
    {
      long long tot[4];
#pragma unroll
      for (int c = 0; c < 4; c++) tot[c] = wave_sum_nonneg31(contrib[c]);
      if (lane == 0)
      {
#pragma unroll
        for (int c = 0; c < 4; c++) s_red[(r + 1) & 1][wave][c] = tot[c];
      }
    }
    __syncthreads();
  }
  if (live) a.trim[n] = make_int2(thigh, tpos);
  if (threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = 0; o.pad = 0;
    a.ctl_out[fd.id] = o;
  }
}

// Batch mode for everything the register-resident family kernel cannot take (any band width, positive penalties):
// the same one-workgroup-per-family loop with a block-local vote, but the rows stream through the family's slice of
// the in-place row buffer exactly as in ramx_column_kernel (run_band: runtime W, prefetch rings, CHAIN variant).  A
// lane only ever reads state it wrote itself, so no cross-lane visibility is needed between columns; a family's rows
// (W = 80, 100 flanks: 260 KB) stay in L2.
struct FSArgs
{
  KArgs k;                      // bases, bounds, trim, S_in == S_out, Np, W, go, ge, cap, minimp, when_to_stop, tab
  const FamDesc *fam;
  RamxCtl *ctl_out;             // per family
  signed char *cons_out;        // [family][L]
  int L;
};

template <bool CHAIN, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ramx_family_stream_kernel(const FSArgs fa)
{
  constexpr int WPB = BLOCK / 64;
  __shared__ __attribute__((aligned(16))) int s_tab[TAB_ROWS * TAB_STRIDE];
  __shared__ long long s_red[2][WPB][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const FamDesc fd = fa.fam[blockIdx.x];
  const bool live = wave < fd.ntiles;
  const int tile = fd.tile0 + (live ? wave : 0);
  const int n = tile * 64 + lane;
  const bool active = live && (wave * 64 + lane) < fd.nx;
  const KArgs &a = fa.k;
  const int W = a.W, B = 2 * W + 1, Q = W + 1;
  const int4 *Sin = a.S_in + (size_t)tile * Q * 64 + lane;
  int4 *Sout = a.S_out + (size_t)tile * Q * 64 + lane;
  const int2 bd = a.bounds[n];
  long long max_ext = 0;
  int max_row = -1, rows_done = 0, ovf = 0, stopped = 0;

  for (int r = -1; r < fa.L; r++)
  {
    const unsigned *bp = a.bases + (size_t)((r + 8) >> 3) * a.Np + n;
    int4 buf[PF], far[PF];
    if (r >= 0)
    {
#pragma unroll
      for (int i = 0; i < PF; i++) buf[i] = ld_stream(Sin + (size_t)(i < Q ? i : Q - 1) * 64);
#pragma unroll
      for (int i = 0; i < PF; i++) far[i] = ld_stream(Sin + (size_t)(i + PF < Q ? i + PF : Q - 1) * 64);
    }
    const unsigned w0 = bp[0], w1 = bp[(size_t)a.Np], w2 = bp[2 * (size_t)a.Np];
    int besta = 0;
    bool new_max = false;
    if (r >= 0)
    {
      long long curr = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        long long vk = 0;
#pragma unroll
        for (int wv = 0; wv < WPB; wv++) vk += s_red[r & 1][wv][k];
        if (vk > 2147483647LL || vk < -2147483648LL) ovf = 1;
        if (vk > curr) { curr = vk; besta = k; }
      }
      int dist = max_row - r;
      dist = dist < 0 ? -dist : dist;
      new_max = curr >= max_ext + (long long)dist * a.minimp;
      if (new_max) { max_row = r; max_ext = curr; }
      int d2 = r - max_row;
      d2 = d2 < 0 ? -d2 : d2;
      stopped = d2 >= a.when_to_stop;
      rows_done = r + 1;
      if (threadIdx.x == 0) fa.cons_out[(size_t)fd.id * fa.L + r] = (signed char)besta;
    }
    // the winner's score table; everybody has left the previous column's band (barrier at its end)
    for (int i = threadIdx.x; i < TAB_ROWS * TAB_STRIDE; i += BLOCK)      // BLOCK may be 64: fewer threads than entries
    {
      const int row = i / TAB_STRIDE, col = i % TAB_STRIDE;
      int v = 0;
      if (row < RAMX_NCLASS) v = (col < 4) ? a.tab[row][col] : (col == 4 ? a.tab[row][besta] : 0);
      s_tab[i] = v;
    }
    __syncthreads();
    int contrib[4] = { 0, 0, 0, 0 };
    if (live)
    {
      const int jlo = bd.x - r, jhi = bd.y - r;
      LaneDP D;
      D.eC = NEG; D.mPrev = NEG - 1000000; D.bestF = NEG; D.jbest = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) { D.eA[c] = NEG; D.bestA[c] = NEG; }
      int high = 0, pos = 0;
      if (r < 0)
        run_band<true, true, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
      else
      {
        const bool all_in = __all((jlo <= 0) && (jhi >= B));
        if (all_in) run_band<false, false, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
        else run_band<false, true, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
      }
      if (r < 0 || new_max) a.trim[n] = make_int2(high, pos);
      if (active)
      {
        const int capv = high + a.cap;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          const int b = D.bestA[c] < 0 ? 0 : D.bestA[c];
          contrib[c] = (b >= capv) ? b : capv;
        }
      }
    }
    if (stopped || r == fa.L - 1) break;
    {
      long long tot[4];
#pragma unroll
      for (int c = 0; c < 4; c++) tot[c] = wave_sum_nonneg31(contrib[c]);
      if (lane == 0)
      {
#pragma unroll
        for (int c = 0; c < 4; c++) s_red[(r + 1) & 1][wave][c] = tot[c];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = 0; o.pad = 0;
    fa.ctl_out[fd.id] = o;
  }
}

// ------------------------------------------------------------------------------------------
// host side of seam 2
// ------------------------------------------------------------------------------------------
struct ramx_dev
{
  int ordinal;
  hipStream_t stream;
  signed char *d_lib; unsigned long long lib_len; unsigned long long lib_cap;
  // per direction
  ramx_flank *d_flanks; unsigned *d_bases; int2 *d_bounds; int4 *d_state[2]; int2 *d_trim;
  long long *d_sums; RamxCtl *d_ctl; signed char *d_cons;
  PShard *d_vote; unsigned *d_err;   // persistent kernel: fused vote / barrier words
  int last_persistent;
  size_t cap_flanks, cap_bases, cap_state, cap_cons;
  RamxCtl *h_ctl;   // pinned, [2 checkpoints][2 slots]
  hipEvent_t ev_chk[2], ev_begin, ev_end, ev_s0[MAX_SAMPLES], ev_s1[MAX_SAMPLES];
  int Nx, Np, KW;
  ramx_params p; int tab[RAMX_NCLASS][4];
  int ready;
  RamxCtl final_ctl;
  // multi-GPU
  ncclComm_t comm; int rank, nranks;
  ramx_allreduce_cb cb; void *cb_user;
  // cross-device persistent path: this rank's box (fine-grained), every rank's box as mapped here, device copy of that table
  PeerBox *xbox; PeerBox *peer[RAMX_MAX_RANKS]; PeerBox **d_peer; int peer_ready;
  // host-memory variant of the boxes (POSIX shared memory registered with HIP): xbox/peer point into it
  void *hostbox_map; size_t hostbox_bytes; PeerBox *hostbox_host; int hostbox_registered;
  PeerBox *devbox;     // this rank's fine-grained device-memory box (exported over hipIpc)
  hipStream_t cls_stream[4]; hipEvent_t cls_ready, cls_done[4]; int cls_init;   // batch mode: one stream per workgroup shape
  int force_chain;   // RAMX_FORCE_CHAIN=1: always run the full candidate recurrence (test hook)
};

extern "C" int ramx_device_count(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { ramx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return RAMX_ERR_NO_DEVICE; }
  return n;
}

extern "C" int ramx_dev_create(int ordinal, ramx_dev **out)
{
  int n = ramx_device_count();
  if (n <= 0) { ramx_set_error("no HIP device visible (libramx has no CPU path)"); return RAMX_ERR_NO_DEVICE; }
  if (ordinal < 0 || ordinal >= n) { ramx_set_error("device ordinal %d out of range (%d devices)", ordinal, n); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(ordinal));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    ramx_set_error("device %d is %s; libramx carries gfx950 code objects only", ordinal, prop.gcnArchName);
    return RAMX_ERR_NO_DEVICE;
  }
  ramx_dev *d = (ramx_dev *)calloc(1, sizeof(ramx_dev));
  d->ordinal = ordinal;
  HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  HIPCHK(hipHostMalloc((void **)&d->h_ctl, 4 * sizeof(RamxCtl), hipHostMallocDefault));
  HIPCHK(hipMalloc((void **)&d->d_sums, (3 * NSHARD * 4 + 8) * sizeof(long long)));
  HIPCHK(hipMalloc((void **)&d->d_ctl, 2 * sizeof(RamxCtl)));
  HIPCHK(hipMalloc((void **)&d->d_vote, 3 * NSHARD * sizeof(PShard)));
  HIPCHK(hipMalloc((void **)&d->d_err, 64));

  for (int i = 0; i < 2; i++) HIPCHK(hipEventCreate(&d->ev_chk[i]));
  HIPCHK(hipEventCreate(&d->ev_begin));
  HIPCHK(hipEventCreate(&d->ev_end));
  for (int i = 0; i < MAX_SAMPLES; i++) { HIPCHK(hipEventCreate(&d->ev_s0[i])); HIPCHK(hipEventCreate(&d->ev_s1[i])); }
  const char *eb = getenv("RAMX_FORCE_CHAIN");
  d->force_chain = (eb && atoi(eb) != 0) ? 1 : 0;
  *out = d;
  return RAMX_OK;
}

extern "C" void ramx_dev_destroy(ramx_dev *d)
{
  if (!d) return;
  (void)hipSetDevice(d->ordinal);
  (void)hipStreamSynchronize(d->stream);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  (void)hipFree(d->d_lib); (void)hipFree(d->d_flanks); (void)hipFree(d->d_bases); (void)hipFree(d->d_bounds);
  (void)hipFree(d->d_state[0]); if (d->d_state[1] != d->d_state[0]) (void)hipFree(d->d_state[1]); (void)hipFree(d->d_trim); (void)hipFree(d->d_sums);
  (void)hipFree(d->d_ctl); (void)hipFree(d->d_cons); (void)hipFree(d->d_vote); (void)hipFree(d->d_err);
  (void)hipHostFree(d->h_ctl);
  if (d->hostbox_map)
  {
    if (d->hostbox_registered) (void)hipHostUnregister(d->hostbox_map);
    munmap(d->hostbox_map, d->hostbox_bytes);
  }
  if (d->cls_init)
  {
    for (int c = 0; c < 4; c++) { (void)hipStreamDestroy(d->cls_stream[c]); (void)hipEventDestroy(d->cls_done[c]); }
    (void)hipEventDestroy(d->cls_ready);
  }
  if (d->devbox) (void)hipFree(d->devbox);
  if (d->d_peer) (void)hipFree(d->d_peer);
  for (int i = 0; i < 2; i++) (void)hipEventDestroy(d->ev_chk[i]);
  (void)hipEventDestroy(d->ev_begin); (void)hipEventDestroy(d->ev_end);
  for (int i = 0; i < MAX_SAMPLES; i++) { (void)hipEventDestroy(d->ev_s0[i]); (void)hipEventDestroy(d->ev_s1[i]); }
  (void)hipStreamDestroy(d->stream);
  free(d);
}

extern "C" int ramx_dev_load_library(ramx_dev *d, const int8_t *sequence, uint64_t length)
{
  if (!d || (!sequence && length)) { ramx_set_error("ramx_dev_load_library: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (length > d->lib_cap)
  {
    if (d->d_lib) HIPCHK(hipFree(d->d_lib));
    d->d_lib = NULL;
    HIPCHK(hipMalloc((void **)&d->d_lib, length ? length : 1));
    d->lib_cap = length;
  }
  if (length) HIPCHK(hipMemcpyAsync(d->d_lib, sequence, length, hipMemcpyHostToDevice, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  d->lib_len = length;
  return RAMX_OK;
}

template <typename T>
static int ensure(T **p, size_t *cap, size_t need_bytes)
{
  if (need_bytes <= *cap && *p) return RAMX_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = NULL;
  HIPCHK(hipMalloc((void **)p, need_bytes ? need_bytes : 16));
  *cap = need_bytes;
  return RAMX_OK;
}

extern "C" int ramx_dev_begin_direction(ramx_dev *d, const ramx_flank *flanks, int32_t n_flanks, const ramx_params *p)
{
  if (!d || !p || n_flanks < 0 || (n_flanks && !flanks)) { ramx_set_error("ramx_dev_begin_direction: bad argument"); return RAMX_ERR_ARG; }
  if (p->bandwidth < 0 || p->L < 0) { ramx_set_error("bandwidth and L must be >= 0"); return RAMX_ERR_ARG; }
  if (!p->matrix) { ramx_set_error("scoring matrix missing"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int W = p->bandwidth, Q = W + 1;
  const int Nx = n_flanks;
  const int Np = ((Nx + 63) / 64) * 64 > 0 ? ((Nx + 63) / 64) * 64 : 64;
  // t'' = o + r + W + 8 runs over [7, L + 2W + 9]
  const int KW = (p->L + 2 * W + 2) / 8 + 8;   // + pad word in front, + lookahead words read by the kernel
  d->Nx = Nx; d->Np = Np; d->KW = KW; d->p = *p;
  // class table: tab[class][cand] = matrix[cand][code(class)], reference index order [cons][seq]
  for (int c = 0; c < RAMX_NCLASS; c++)
  {
    const int code = (c == 8) ? RAMX_SYM_N : c;
    for (int k = 0; k < 4; k++) d->tab[c][k] = p->matrix[k * 100 + code];
  }
  int rc;
  if ((rc = ensure(&d->d_flanks, &d->cap_flanks, (size_t)Np * sizeof(ramx_flank)))) return rc;
  if ((rc = ensure(&d->d_bases, &d->cap_bases, (size_t)KW * Np * sizeof(unsigned)))) return rc;
  if (d->d_bounds) { HIPCHK(hipFree(d->d_bounds)); d->d_bounds = NULL; }
  if (d->d_trim) { HIPCHK(hipFree(d->d_trim)); d->d_trim = NULL; }
  HIPCHK(hipMalloc((void **)&d->d_bounds, (size_t)Np * sizeof(int2)));
  HIPCHK(hipMalloc((void **)&d->d_trim, (size_t)Np * sizeof(int2)));
  const size_t state_bytes = (size_t)Np * Q * sizeof(int4);
  if (state_bytes > d->cap_state || !d->d_state[0])
  {
    // The row is updated IN PLACE: a lane only ever touches its own 16-byte column of the tile, reads slot q+16
    // before it writes slot q, and launches are serialised, so one buffer serves as S(r-1) and S(r).  Halving the
    // footprint (65.6 MB at N = 100,000) keeps the whole row set inside the Infinity Cache.
    const bool pingpong = getenv("RAMX_PINGPONG") != NULL;     // A/B switch kept for profiling
    if (d->d_state[0]) HIPCHK(hipFree(d->d_state[0]));
    if (d->d_state[1] && d->d_state[1] != d->d_state[0]) HIPCHK(hipFree(d->d_state[1]));
    d->d_state[0] = d->d_state[1] = NULL;
    HIPCHK(hipMalloc((void **)&d->d_state[0], state_bytes));
    if (pingpong) HIPCHK(hipMalloc((void **)&d->d_state[1], state_bytes));
    else d->d_state[1] = d->d_state[0];
    d->cap_state = state_bytes;
  }
  if ((rc = ensure(&d->d_cons, &d->cap_cons, (size_t)p->L + 16))) return rc;
  if (Nx) HIPCHK(hipMemcpyAsync(d->d_flanks, flanks, (size_t)Nx * sizeof(ramx_flank), hipMemcpyHostToDevice, d->stream));
  dim3 grid((Np + 255) / 256, KW);
  hipLaunchKernelGGL(ramx_pack_kernel, grid, dim3(256), 0, d->stream, d->d_lib, (unsigned long long)d->lib_len,
                     d->d_flanks, Nx, Np, W, d->d_bases, d->d_bounds);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
  HIPCHK(hipMemsetAsync(d->d_cons, 0, (size_t)p->L + 16, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  d->ready = 1;
  return RAMX_OK;
}

template <bool INIT>
static void launch_column(ramx_dev *d, const KArgs &a)
{
  const int tiles = d->Np / 64;
  const dim3 grid((tiles + 3) / 4), block(256);
  // the chain-free candidate evaluation is exact iff neither gap penalty is positive (see the kernel header)
  if (a.go <= 0 && a.ge <= 0 && !d->force_chain)
    hipLaunchKernelGGL((ramx_column_kernel<INIT, false, 256>), grid, block, 0, d->stream, a);
  else
    hipLaunchKernelGGL((ramx_column_kernel<INIT, true, 256>), grid, block, 0, d->stream, a);
}

// ---- host-side collective on 4 x int64 in device memory (RCCL, or the test hook) ---------------
static int host_allreduce_shards(ramx_dev *d, long long *dptr)   // dptr: NSHARD x 4 int64, summed over ranks in place
{
  if (d->cb)
  {
    long long h[NSHARD * 4], h4[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpyAsync(h, dptr, sizeof(h), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    for (int i = 0; i < NSHARD * 4; i++) h4[i & 3] += h[i];
    d->cb(h4, d->cb_user);
    memset(h, 0, sizeof(h));
    memcpy(h, h4, sizeof(h4));
    HIPCHK(hipMemcpyAsync(dptr, h, sizeof(h), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return RAMX_OK;
  }
  ncclResult_t nr = ncclAllReduce(dptr, dptr, NSHARD * 4, ncclInt64, ncclSum, d->comm, d->stream);
  if (nr != ncclSuccess) { ramx_set_error("ncclAllReduce: %s", ncclGetErrorString(nr)); return RAMX_ERR_COMM; }
  return RAMX_OK;
}

// one small integer agreed over all ranks (max): used to agree on a fallback
static int host_allreduce_flag(ramx_dev *d, int *flag)
{
  if (d->cb)
  {
    long long h4[4] = { *flag ? 1 : 0, 0, 0, 0 };
    d->cb(h4, d->cb_user);
    *flag = h4[0] != 0;
    return RAMX_OK;
  }
  long long *tmp = d->d_sums + 3 * NSHARD * 4;   // spare 4 words behind the three vote slots
  long long h = *flag ? 1 : 0;
  HIPCHK(hipMemcpyAsync(tmp, &h, sizeof(h), hipMemcpyHostToDevice, d->stream));
  ncclResult_t nr = ncclAllReduce(tmp, tmp, 1, ncclInt64, ncclMax, d->comm, d->stream);
  if (nr != ncclSuccess) { ramx_set_error("ncclAllReduce: %s", ncclGetErrorString(nr)); return RAMX_ERR_COMM; }
  HIPCHK(hipMemcpyAsync(&h, tmp, sizeof(h), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  *flag = h != 0;
  return RAMX_OK;
}

// ---- peer boxes ------------------------------------------------------------------------------------
extern "C" int ramx_dev_peer_export(ramx_dev *d, uint8_t handle[64])
{
  if (!d || !handle) { ramx_set_error("ramx_dev_peer_export: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (!d->devbox)
  {
    hipError_t e = hipExtMallocWithFlags((void **)&d->devbox, sizeof(PeerBox), hipDeviceMallocFinegrained);
    if (e != hipSuccess) { d->devbox = NULL; ramx_set_error("fine-grained allocation for the peer box failed: %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
    HIPCHK(hipMemset(d->devbox, 0, sizeof(PeerBox)));
  }
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, d->devbox));
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "ipc handle size");
  memset(handle, 0, 64);
  memcpy(handle, &h, sizeof(h));
  return RAMX_OK;
}

__global__ void ramx_peer_token_kernel(PeerBox *const *peers, int rank, int nranks, unsigned long long token)
{
  if (threadIdx.x < nranks)
    __hip_atomic_store(&peers[threadIdx.x]->token[rank], token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int ramx_dev_peer_import(ramx_dev *d, const uint8_t *handles, int rank, int nranks)
{
  if (!d || !handles || rank < 0 || rank >= nranks || nranks > RAMX_MAX_RANKS || !d->devbox)
  { ramx_set_error("ramx_dev_peer_import: bad argument (export first; at most %d ranks)", RAMX_MAX_RANKS); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->peer_ready = 0;
  d->rank = rank; d->nranks = nranks;
  d->xbox = d->devbox; d->hostbox_host = NULL;
  for (int q = 0; q < nranks; q++)
  {
    if (q == rank) { d->peer[q] = d->xbox; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + 64 * q, sizeof(h));
    void *ptr = NULL;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { ramx_set_error("hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e)); return RAMX_ERR_HIP; }
    d->peer[q] = (PeerBox *)ptr;
  }
  if (!d->d_peer) HIPCHK(hipMalloc((void **)&d->d_peer, RAMX_MAX_RANKS * sizeof(PeerBox *)));
  HIPCHK(hipMemcpy(d->d_peer, d->peer, nranks * sizeof(PeerBox *), hipMemcpyHostToDevice));
  return RAMX_OK;
}

/* Host-memory mailboxes: the same PeerBox protocol with every rank's box in ONE POSIX shared-memory segment that each
 * process registers with HIP (hipHostRegisterMapped).  Stores and polls cross PCIe instead of xGMI; no IPC handles,
 * no peer access between devices needed.  Used when the device-memory boxes cannot be mapped or fail their
 * self-test.  Every rank calls attach with the same name (ranks of one node); after a barrier rank 0 may unlink. */
extern "C" int ramx_dev_hostbox_attach(ramx_dev *d, const char *shm_name, int rank, int nranks)
{
  if (!d || !shm_name || rank < 0 || rank >= nranks || nranks > RAMX_MAX_RANKS)
  { ramx_set_error("ramx_dev_hostbox_attach: bad argument (at most %d ranks)", RAMX_MAX_RANKS); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->peer_ready = 0;
  const size_t bytes = (((size_t)nranks * sizeof(PeerBox)) + 4095) & ~(size_t)4095;
  int fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { ramx_set_error("shm_open(%s): %s", shm_name, strerror(errno)); return RAMX_ERR_COMM; }
  if (ftruncate(fd, (off_t)bytes) != 0) { ramx_set_error("ftruncate(%s): %s", shm_name, strerror(errno)); close(fd); return RAMX_ERR_COMM; }
  void *map = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) { ramx_set_error("mmap(%s): %s", shm_name, strerror(errno)); return RAMX_ERR_COMM; }
  hipError_t e = hipHostRegister(map, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) { munmap(map, bytes); ramx_set_error("hipHostRegister(shared boxes): %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
  void *dptr = NULL;
  e = hipHostGetDevicePointer(&dptr, map, 0);
  if (e != hipSuccess) { (void)hipHostUnregister(map); munmap(map, bytes); ramx_set_error("hipHostGetDevicePointer: %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
  if (d->hostbox_map) { if (d->hostbox_registered) (void)hipHostUnregister(d->hostbox_map); munmap(d->hostbox_map, d->hostbox_bytes); }
  d->hostbox_map = map; d->hostbox_bytes = bytes; d->hostbox_registered = 1;
  d->hostbox_host = (PeerBox *)map + rank;
  memset(d->hostbox_host, 0, sizeof(PeerBox));          // my own box; the others clear theirs
  d->rank = rank; d->nranks = nranks;
  // the device-memory box (if any) is no longer the exchange target
  d->xbox = (PeerBox *)dptr + rank;
  for (int q = 0; q < nranks; q++) d->peer[q] = (PeerBox *)dptr + q;
  if (!d->d_peer) HIPCHK(hipMalloc((void **)&d->d_peer, RAMX_MAX_RANKS * sizeof(PeerBox *)));
  HIPCHK(hipMemcpy(d->d_peer, d->peer, nranks * sizeof(PeerBox *), hipMemcpyHostToDevice));
  return RAMX_OK;
}

extern "C" int ramx_hostbox_unlink(const char *shm_name)
{
  if (!shm_name) return RAMX_ERR_ARG;
  return shm_unlink(shm_name) == 0 ? RAMX_OK : RAMX_ERR_COMM;
}

/* self-test of the mapped boxes: write a token into every rank's box, then (after the caller has synchronised the
 * ranks) check that every rank's token arrived here.  phase 0 = write, phase 1 = check (returns 1 if all there). */
extern "C" int ramx_dev_peer_selftest(ramx_dev *d, int phase, unsigned long long token)
{
  if (!d || !d->d_peer) { ramx_set_error("ramx_dev_peer_selftest: import first"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (phase == 0)
  {
    hipLaunchKernelGGL(ramx_peer_token_kernel, dim3(1), dim3(64), 0, d->stream, (PeerBox *const *)d->d_peer, d->rank, d->nranks, token);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->stream));
    return RAMX_OK;
  }
  PeerBox h;
  for (int tries = 0; tries < 200; tries++)
  {
    if (d->hostbox_host) memcpy(&h, (const void *)d->hostbox_host, sizeof(h));
    else HIPCHK(hipMemcpy(&h, d->xbox, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int q = 0; q < d->nranks; q++) ok &= (h.token[q] == token);
    if (ok) return 1;
    struct timespec ts = { 0, 1000000 };
    nanosleep(&ts, NULL);
  }
  return 0;
}

extern "C" int ramx_dev_peer_enable(ramx_dev *d, int on)
{
  if (!d) return RAMX_ERR_ARG;
  d->peer_ready = (on && d->d_peer && d->xbox) ? 1 : 0;
  return RAMX_OK;
}

// The fast band packs (score << 4 | cell) keys: exact while every reachable |score| < 2^27 (a row-r cell is at most
// (r + W + 2) steps of at most max(|matrix|, |go| + |ge|) away from 0), and reads matrix entries as int8.  Scoring
// systems outside these bounds take the general (masked-path) band for every wave.
static int fast_pack_ok(const int (&tab)[RAMX_NCLASS][4], int go, int ge, int L, int W)
{
  long long mx = (long long)(go < 0 ? -go : go) + (long long)(ge < 0 ? -ge : ge);
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
    {
      const long long v = tab[c][k] < 0 ? -(long long)tab[c][k] : (long long)tab[c][k];
      if (v > mx) mx = v;
    }
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
      if (tab[c][k] < -128 || tab[c][k] > 127) return 0;       // the fast tables hold the candidates' scores as int8
  return ((long long)L + 2LL * W + 4) * mx < (1LL << 27) ? 1 : 0;
}

// ---- persistent path --------------------------------------------------------------------------
// Block shape of the persistent launch: at most ONE barrier participant per CU.
//   <= 4 tiles per CU (N <= 65,536): 256-thread blocks, one wave per SIMD, up to 256 blocks;
//   otherwise 512-thread blocks (two waves per SIMD), up to 256 blocks = 131,072 flanks.
template <int W, int BLOCK>
static int prk_capacity_blocks(int *out)
{
  int per_cu = 0, dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ramx_persistent_kernel<W, BLOCK>, BLOCK, 0));
  if (per_cu > 1) per_cu = 1;          // one block per CU by design; never trust the API for more
  *out = per_cu * cus;
  return RAMX_OK;
}

template <int W, int BLOCK>
static int prk_launch(ramx_dev *d, PArgs &pa, int blocks)
{
  void *args[] = { (void *)&pa };
  HIPCHK(hipLaunchCooperativeKernel((const void *)ramx_persistent_kernel<W, BLOCK>, dim3(blocks), dim3(BLOCK), args, 0, d->stream));
  return RAMX_OK;
}

// block shape that keeps the whole flank set resident, or 0
template <int W>
static int prk_plan(int tiles, int *block, int *blocks)
{
  int cap = 0, rc;
  *block = 0; *blocks = 0;
  if ((rc = prk_capacity_blocks<W, 256>(&cap)) != RAMX_OK) return rc;
  if ((tiles + 3) / 4 <= cap) { *block = 256; *blocks = (tiles + 3) / 4; return RAMX_OK; }
  if ((rc = prk_capacity_blocks<W, 512>(&cap)) != RAMX_OK) return rc;
  if ((tiles + 7) / 8 <= cap) { *block = 512; *blocks = (tiles + 7) / 8; }
  return RAMX_OK;
}

// which band widths have a register-resident instantiation
static bool prk_has_width(int W) { return W == 14 || W == 20 || W == 40; }

static int prk_run(ramx_dev *d, const KArgs &a, int L, bool *used)
{
  *used = false;
  const int W = a.W;
  const bool multi = (d->comm != NULL && d->nranks > 1) || d->cb != NULL;
  const int tiles = d->Np / 64;
  int block = 0, blocks = 0, rc;
  bool can = !(getenv("RAMX_NO_PERSISTENT") != NULL || d->force_chain || a.go > 0 || a.ge > 0 || a.go + a.ge < -32768 ||
               !prk_has_width(W) || L <= 0);
  if (multi && (!d->peer_ready || d->nranks < 2 || L >= 65536 || getenv("RAMX_NO_PEER") != NULL)) can = false;
  if (can)
  {
    rc = (W == 14) ? prk_plan<14>(tiles, &block, &blocks) : (W == 20) ? prk_plan<20>(tiles, &block, &blocks) : prk_plan<40>(tiles, &block, &blocks);
    if (rc != RAMX_OK) return rc;
    if (block == 0) can = false;          // not co-resident: keep the streaming kernel
  }
  if (multi)
  {
    // every rank must take the same path: agree (max over ranks of "cannot")
    int cannot = can ? 0 : 1;
    if ((rc = host_allreduce_flag(d, &cannot)) != RAMX_OK) return rc;
    can = !cannot;
  }
  if (!can) return RAMX_OK;
  PArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.S = d->d_state[0]; pa.bases = d->d_bases; pa.bounds = d->d_bounds; pa.trim = d->d_trim;
  pa.sums0 = d->d_sums; pa.vote = d->d_vote; pa.ctl_out = d->d_ctl; pa.cons_out = d->d_cons; pa.err = d->d_err;
  pa.Np = d->Np; pa.Nx = d->Nx; pa.r0 = 0; pa.L = L; pa.go = a.go; pa.ge = a.ge; pa.cap = a.cap; pa.minimp = a.minimp;
  pa.when_to_stop = a.when_to_stop; pa.nblocks = blocks;
  pa.nranks = 1; pa.rank = 0; pa.peers = NULL; pa.box = NULL;
  if (multi)
  {
    pa.nranks = d->nranks; pa.rank = d->rank; pa.peers = (PeerBox *const *)d->d_peer; pa.box = d->xbox;
    // my box is cleared BEFORE the collective below, which no remote launch can get past without my taking part:
    // nobody writes a word of this run into it too early, and nothing of the last run survives
    if (d->hostbox_host)
    {
      HIPCHK(hipStreamSynchronize(d->stream));
      memset((void *)d->hostbox_host, 0, sizeof(PeerBox));
      __sync_synchronize();
    }
    else HIPCHK(hipMemsetAsync(d->xbox, 0, sizeof(PeerBox), d->stream));
    if ((rc = host_allreduce_shards(d, d->d_sums)) != RAMX_OK) return rc;     // vote of row 0 (from K(-1)) over ranks
  }
  memcpy(pa.tab, a.tab, sizeof(pa.tab));
  pa.pack_ok = getenv("RAMX_NO_FASTPACK") ? 0 : fast_pack_ok(pa.tab, a.go, a.ge, L, W);
  HIPCHK(hipMemsetAsync(d->d_vote, 0, 3 * NSHARD * sizeof(PShard), d->stream));
  HIPCHK(hipMemsetAsync(d->d_err, 0, 64, d->stream));
  *used = true;
#ifdef RAMX_PRK_TIMING
  const size_t nw = (size_t)blocks * (block / 64);
  HIPCHK(hipMalloc((void **)&pa.dbg, nw * 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(pa.dbg, 0, nw * 8 * sizeof(unsigned long long)));
#endif
  if (block == 256)
    rc = (W == 14) ? prk_launch<14, 256>(d, pa, blocks) : (W == 20) ? prk_launch<20, 256>(d, pa, blocks) : prk_launch<40, 256>(d, pa, blocks);
  else
    rc = (W == 14) ? prk_launch<14, 512>(d, pa, blocks) : (W == 20) ? prk_launch<20, 512>(d, pa, blocks) : prk_launch<40, 512>(d, pa, blocks);
#ifdef RAMX_PRK_TIMING
  if (rc == RAMX_OK)
  {
    // debug build: phase breakdown per column, in ns (wall_clock64 ticks are 10 ns)
    HIPCHK(hipStreamSynchronize(d->stream));
    unsigned long long *h = (unsigned long long *)malloc(nw * 8 * sizeof(unsigned long long));
    HIPCHK(hipMemcpy(h, pa.dbg, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static const char *nm[6] = { "wait vote", "block barrier 1", "band", "reduce+barrier 2", "issue atomics", "loop top" };
    const int wpb = block / 64;
    fprintf(stderr, "PRK_TIMING blocks %d x %d threads, L %d (ns per column)\n", blocks, block, L);
    for (int k = 0; k < 6; k++)
    {
      double s0 = 0, sO = 0, mx = 0, mn = 1e30;
      for (size_t i = 0; i < nw; i++)
      {
        const double v = 10.0 * (double)h[i * 8 + k] / L;
        if ((int)(i % wpb) == 0) s0 += v; else sO += v;
        if (v > mx) mx = v;
        if (v < mn) mn = v;
      }
      fprintf(stderr, "PRK_TIMING %-18s wave0 avg %8.1f  other waves avg %8.1f  min %8.1f  max %8.1f\n", nm[k], s0 / blocks,
              wpb > 1 ? sO / (double)(nw - blocks) : 0.0, mn, mx);
    }
    fprintf(stderr, "PRK_TIMING band by wave index:");
    for (int wv = 0; wv < wpb; wv++)
    {
      double sw = 0;
      for (int b = 0; b < blocks - 1; b++) sw += 10.0 * (double)h[((size_t)b * wpb + wv) * 8 + 2] / L;
      fprintf(stderr, " w%d %.0f", wv, sw / (blocks > 1 ? blocks - 1 : 1));
    }
    fprintf(stderr, "\n");
    free(h);
    (void)hipFree(pa.dbg);
  }
#endif
  return rc;
}

// ---- batch mode -------------------------------------------------------------------------------
template <int W, int BLOCK>
static int fam_launch(ramx_dev *d, const FArgs &fa, int F, hipStream_t st)
{
  (void)d;
  hipLaunchKernelGGL((ramx_family_kernel<W, BLOCK>), dim3(F), dim3(BLOCK), 0, st, fa);
  HIPCHK(hipGetLastError());
  return RAMX_OK;
}

extern "C" int ramx_dev_run_families(ramx_dev *d, const ramx_flank *flanks, int32_t n_padded, const int32_t *fam_first,
                                     const int32_t *fam_count, int32_t n_families, const ramx_params *p,
                                     ramx_run_info *infos, int8_t *cons, int32_t *trim_high, int32_t *trim_pos)
{
  if (!d || !p || !p->matrix || n_families < 0 || n_padded < 0 || (n_padded & 63) || (n_families && (!fam_first || !fam_count || !flanks)))
  { ramx_set_error("ramx_dev_run_families: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int W = p->bandwidth, L = p->L;
  if (W < 1 || L < 0) { ramx_set_error("ramx_dev_run_families: bad bandwidth / L"); return RAMX_ERR_ARG; }
  // register-resident family kernel where it applies, the streaming family kernel for everything else
  const bool resident = prk_has_width(W) && p->gapopen <= 0 && p->gapextn <= 0 && p->gapopen + p->gapextn >= -32768 &&
                        !d->force_chain && getenv("RAMX_NO_PERSISTENT") == NULL;
  int maxn = 0;
  for (int f = 0; f < n_families; f++)
  {
    if (fam_count[f] < 0 || (fam_first[f] & 63) || fam_first[f] + fam_count[f] > n_padded) { ramx_set_error("ramx_dev_run_families: bad family layout"); return RAMX_ERR_ARG; }
    if (fam_count[f] > maxn) maxn = fam_count[f];
  }
  if (maxn > 512) { ramx_set_error("batch mode: a family has more than 512 flanks"); return RAMX_ERR_UNSUPPORTED; }
  if (n_families == 0) return RAMX_OK;
  const int Np = n_padded > 0 ? n_padded : 64;
  const int KW = (L + 2 * W + 2) / 8 + 8;
  int rc;
  if ((rc = ensure(&d->d_flanks, &d->cap_flanks, (size_t)Np * sizeof(ramx_flank)))) return rc;
  if ((rc = ensure(&d->d_bases, &d->cap_bases, (size_t)KW * Np * sizeof(unsigned)))) return rc;
  if (d->d_bounds) { HIPCHK(hipFree(d->d_bounds)); d->d_bounds = NULL; }
  if (d->d_trim) { HIPCHK(hipFree(d->d_trim)); d->d_trim = NULL; }
  HIPCHK(hipMalloc((void **)&d->d_bounds, (size_t)Np * sizeof(int2)));
  HIPCHK(hipMalloc((void **)&d->d_trim, (size_t)Np * sizeof(int2)));
  if ((rc = ensure(&d->d_cons, &d->cap_cons, (size_t)n_families * (L > 0 ? L : 1) + 16))) return rc;
  // descriptors grouped by workgroup shape (64, 128, 256, 512 threads = 1, 2, 4, 8 tiles): one launch per non-empty
  // class, so a 100-flank family occupies two waves, not four
  FamDesc *hfd = (FamDesc *)malloc(sizeof(FamDesc) * n_families);
  int cls_first[5] = { 0, 0, 0, 0, 0 }, cls_count[4] = { 0, 0, 0, 0 };
  auto cls_of = [](int nx) { return nx <= 64 ? 0 : nx <= 128 ? 1 : nx <= 256 ? 2 : 3; };
  for (int f = 0; f < n_families; f++) cls_count[cls_of(fam_count[f])]++;
  for (int c = 0; c < 4; c++) cls_first[c + 1] = cls_first[c] + cls_count[c];
  {
    int fill[4] = { cls_first[0], cls_first[1], cls_first[2], cls_first[3] };
    for (int f = 0; f < n_families; f++)
    {
      FamDesc &x = hfd[fill[cls_of(fam_count[f])]++];
      x.tile0 = fam_first[f] / 64; x.ntiles = (fam_count[f] + 63) / 64; x.nx = fam_count[f]; x.id = f;
    }
  }
  FamDesc *dfd = NULL; RamxCtl *dctl = NULL;
  HIPCHK(hipMalloc((void **)&dfd, sizeof(FamDesc) * n_families));
  HIPCHK(hipMalloc((void **)&dctl, sizeof(RamxCtl) * n_families));
  HIPCHK(hipMemcpyAsync(dfd, hfd, sizeof(FamDesc) * n_families, hipMemcpyHostToDevice, d->stream));
  if (n_padded) HIPCHK(hipMemcpyAsync(d->d_flanks, flanks, (size_t)n_padded * sizeof(ramx_flank), hipMemcpyHostToDevice, d->stream));
  dim3 grid((Np + 255) / 256, KW);
  hipLaunchKernelGGL(ramx_pack_kernel, grid, dim3(256), 0, d->stream, d->d_lib, (unsigned long long)d->lib_len, d->d_flanks,
                     n_padded, Np, W, d->d_bases, d->d_bounds);
  HIPCHK(hipGetLastError());
  FArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.bases = d->d_bases; fa.bounds = d->d_bounds; fa.fam = dfd; fa.trim = d->d_trim; fa.ctl_out = dctl; fa.cons_out = d->d_cons;
  fa.Np = Np; fa.L = L; fa.go = p->gapopen; fa.ge = p->gapextn; fa.cap = p->cappenalty; fa.minimp = p->minimprovement;
  fa.when_to_stop = p->when_to_stop;
  for (int c = 0; c < RAMX_NCLASS; c++)
  {
    const int code = (c == 8) ? RAMX_SYM_N : c;
    for (int k = 0; k < 4; k++) fa.tab[c][k] = p->matrix[k * 100 + code];
  }
  fa.pack_ok = getenv("RAMX_NO_FASTPACK") ? 0 : fast_pack_ok(fa.tab, fa.go, fa.ge, L, W);
  HIPCHK(hipEventRecord(d->ev_begin, d->stream));
  // the shapes run side by side: one stream per class, forked from / joined into the library's stream
  if (!d->cls_init)
  {
    for (int c = 0; c < 4; c++) { HIPCHK(hipStreamCreateWithFlags(&d->cls_stream[c], hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&d->cls_done[c], hipEventDisableTiming)); }
    HIPCHK(hipEventCreateWithFlags(&d->cls_ready, hipEventDisableTiming));
    d->cls_init = 1;
  }
  HIPCHK(hipEventRecord(d->cls_ready, d->stream));
  for (int c = 0; c < 4; c++) if (cls_count[c] > 0) HIPCHK(hipStreamWaitEvent(d->cls_stream[c], d->cls_ready, 0));
  if (!resident)
  {
    // rows of every family in the (in-place) row buffer: 16 B x (W + 1) slots per flank
    const size_t state_bytes = (size_t)Np * (W + 1) * sizeof(int4);
    if (state_bytes > d->cap_state || !d->d_state[0])
    {
      if (d->d_state[0]) HIPCHK(hipFree(d->d_state[0]));
      if (d->d_state[1] && d->d_state[1] != d->d_state[0]) HIPCHK(hipFree(d->d_state[1]));
      d->d_state[0] = d->d_state[1] = NULL;
      HIPCHK(hipMalloc((void **)&d->d_state[0], state_bytes));
      d->d_state[1] = d->d_state[0];
      d->cap_state = state_bytes;
    }
    FSArgs fs;
    memset(&fs, 0, sizeof(fs));
    fs.k.bases = d->d_bases; fs.k.bounds = d->d_bounds; fs.k.trim = d->d_trim; fs.k.S_in = d->d_state[0]; fs.k.S_out = d->d_state[0];
    fs.k.Np = Np; fs.k.Nx = Np; fs.k.W = W; fs.k.go = p->gapopen; fs.k.ge = p->gapextn; fs.k.cap = p->cappenalty;
    fs.k.minimp = p->minimprovement; fs.k.when_to_stop = p->when_to_stop;
    memcpy(fs.k.tab, fa.tab, sizeof(fs.k.tab));
    fs.fam = dfd; fs.ctl_out = dctl; fs.cons_out = d->d_cons; fs.L = L;
    const bool chain = p->gapopen > 0 || p->gapextn > 0 || d->force_chain;
#define RAMX_FS_LAUNCH(CH, BL, C) do { if (cls_count[C] > 0) { fs.fam = dfd + cls_first[C]; \
      hipLaunchKernelGGL((ramx_family_stream_kernel<CH, BL>), dim3(cls_count[C]), dim3(BL), 0, d->cls_stream[C], fs); } } while (0)
    if (chain) { RAMX_FS_LAUNCH(true, 64, 0); RAMX_FS_LAUNCH(true, 128, 1); RAMX_FS_LAUNCH(true, 256, 2); RAMX_FS_LAUNCH(true, 512, 3); }
    else { RAMX_FS_LAUNCH(false, 64, 0); RAMX_FS_LAUNCH(false, 128, 1); RAMX_FS_LAUNCH(false, 256, 2); RAMX_FS_LAUNCH(false, 512, 3); }
#undef RAMX_FS_LAUNCH
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { free(hfd); ramx_set_error("family stream kernel launch: %s", hipGetErrorString(le)); return RAMX_ERR_HIP; }
    rc = RAMX_OK;
  }
  else
  {
    // the register-resident kernel gains nothing from smaller workgroups (measured: 11.5 vs 10.3 ms; rotating the live
    // waves over the SIMDs by arrival order on the CU did not help either): 256 threads up to 256 flanks, in class order
    cls_count[2] += cls_count[0] + cls_count[1]; cls_first[2] = 0; cls_count[0] = cls_count[1] = 0;
    rc = RAMX_OK;
    for (int c = 0; c < 4 && rc == RAMX_OK; c++)
    {
      if (cls_count[c] == 0) continue;
      fa.fam = dfd + cls_first[c];
      const int n = cls_count[c];
      hipStream_t st = d->cls_stream[c];
      if (c <= 2) rc = (W == 14) ? fam_launch<14, 256>(d, fa, n, st) : (W == 20) ? fam_launch<20, 256>(d, fa, n, st) : fam_launch<40, 256>(d, fa, n, st);
      else             rc = (W == 14) ? fam_launch<14, 512>(d, fa, n, st) : (W == 20) ? fam_launch<20, 512>(d, fa, n, st) : fam_launch<40, 512>(d, fa, n, st);
    }
  }
  if (rc != RAMX_OK) { free(hfd); return rc; }
  for (int c = 0; c < 4; c++)
    if (cls_count[c] > 0) { HIPCHK(hipEventRecord(d->cls_done[c], d->cls_stream[c])); HIPCHK(hipStreamWaitEvent(d->stream, d->cls_done[c], 0)); }
  HIPCHK(hipEventRecord(d->ev_end, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, d->ev_begin, d->ev_end));
  RamxCtl *hctl = (RamxCtl *)malloc(sizeof(RamxCtl) * n_families);
  HIPCHK(hipMemcpy(hctl, dctl, sizeof(RamxCtl) * n_families, hipMemcpyDeviceToHost));
  if (cons && L > 0) HIPCHK(hipMemcpy(cons, d->d_cons, (size_t)n_families * L, hipMemcpyDeviceToHost));
  if ((trim_high || trim_pos) && n_padded)
  {
    int2 *tmp = (int2 *)malloc((size_t)n_padded * sizeof(int2));
    HIPCHK(hipMemcpy(tmp, d->d_trim, (size_t)n_padded * sizeof(int2), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_padded; i++) { if (trim_high) trim_high[i] = tmp[i].x; if (trim_pos) trim_pos[i] = tmp[i].y; }
    free(tmp);
  }
  for (int f = 0; f < n_families && infos; f++)
  {
    memset(&infos[f], 0, sizeof(ramx_run_info));
    infos[f].ret = hctl[f].max_row + 1;
    infos[f].rows_executed = hctl[f].rows_done;
    infos[f].limit_warning = (hctl[f].stopped && hctl[f].rows_done - 1 == L - 1) ? 1 : 0;
    infos[f].overflow32 = hctl[f].overflow;
    infos[f].n_extendable = fam_count[f];
    infos[f].launches = 1;
    infos[f].loop_ms = ms;
    infos[f].persistent = resident ? 1 : 2;     /* 2: streaming family kernel */
  }
  free(hctl); free(hfd);
  (void)hipFree(dfd); (void)hipFree(dctl);
  d->ready = 0;       // the single-family buffers were reused: begin_direction must be called again before run_direction
  return RAMX_OK;
}

extern "C" int ramx_dev_run_direction(ramx_dev *d, ramx_run_info *info)
{
  if (!d || !d->ready) { ramx_set_error("ramx_dev_run_direction: begin_direction has not been called"); return RAMX_ERR_STATE; }
  HIPCHK(hipSetDevice(d->ordinal));
  const ramx_params &p = d->p;
  const int L = p.L;
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.bases = d->d_bases; a.bounds = d->d_bounds; a.trim = d->d_trim; a.cons_out = d->d_cons;
  a.Np = d->Np; a.Nx = d->Nx; a.W = p.bandwidth; a.go = p.gapopen; a.ge = p.gapextn; a.cap = p.cappenalty;
  a.minimp = p.minimprovement; a.when_to_stop = p.when_to_stop;
  memcpy(a.tab, d->tab, sizeof(a.tab));
  const bool multi = (d->comm != NULL && d->nranks > 1) || d->cb != NULL;

  auto slot = [&](int r) { return d->d_sums + (size_t)(((r % 3) + 3) % 3) * NSHARD * 4; };
  HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
  HIPCHK(hipEventRecord(d->ev_begin, d->stream));
  // K(-1): boundary row + candidates of row 0
  a.r = -1; a.S_in = d->d_state[0]; a.S_out = d->d_state[1]; a.ctl_in = d->d_ctl; a.ctl_out = d->d_ctl + 1;
  a.sums_in = slot(0); a.sums_out = slot(0); a.sums_zero = slot(1); a.nshards_in = NSHARD;
  launch_column<true>(d, a);
  HIPCHK(hipGetLastError());
  int launches = 0, nsamp = 0, pending = -1, chk = 0;
  bool persistent = false;
  {
    // the in-place row buffer holds S(-1) after K(-1); d_ctl[1] holds the initial control block, the persistent
    // kernel writes its final one to d_ctl[0]
    if (d->d_state[0] != d->d_state[1]) { ramx_set_error("persistent path needs the in-place row buffer"); }
    else { int prc = prk_run(d, a, L, &persistent); if (prc != RAMX_OK) return prc; }
  }
  if (persistent && multi)
  {
    // agree over all ranks whether the cross-device launch went through; if any rank gave up (bounded spin), every
    // rank repeats the direction with the per-column launches and the host collective
    HIPCHK(hipStreamSynchronize(d->stream));
    RamxCtl c0;
    HIPCHK(hipMemcpy(&c0, d->d_ctl, sizeof(c0), hipMemcpyDeviceToHost));
    int bad = c0.pad != 0;
    int frc = host_allreduce_flag(d, &bad);
    if (frc != RAMX_OK) return frc;
    if (bad)
    {
      fprintf(stderr, "ramx: cross-device persistent launch gave up on some rank; repeating the direction with per-column launches\n");
      persistent = false;
      HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
      a.r = -1; a.S_in = d->d_state[0]; a.S_out = d->d_state[1]; a.ctl_in = d->d_ctl; a.ctl_out = d->d_ctl + 1;
      a.sums_in = slot(0); a.sums_out = slot(0); a.sums_zero = slot(1); a.nshards_in = NSHARD;
      launch_column<true>(d, a);
      HIPCHK(hipMemsetAsync(d->d_ctl, 0, sizeof(RamxCtl), d->stream));
    }
  }
  d->last_persistent = persistent ? 1 : 0;
  const int CHUNK = 64;
  const int stride = L > MAX_SAMPLES * 4 ? L / MAX_SAMPLES : 4;
  bool stopped = false;
  memset(d->h_ctl, 0, 4 * sizeof(RamxCtl));
  for (int r = 0; r < L && !stopped && !persistent; r++)
  {
    if (multi)
    {
      // the vote shards of row r (32 x 4 x int64 = 1 KB) are summed across ranks in place; the column kernel then
      // folds them exactly as in the single-GPU case.  One collective per column, no extra kernel.
      int hrc = host_allreduce_shards(d, slot(r));
      if (hrc != RAMX_OK) return hrc;
    }
    a.sums_in = slot(r); a.nshards_in = NSHARD;
    a.r = r; a.S_in = d->d_state[(r + 1) & 1]; a.S_out = d->d_state[r & 1];
    a.ctl_in = d->d_ctl + ((r + 1) & 1); a.ctl_out = d->d_ctl + (r & 1);
    a.sums_out = slot(r + 1); a.sums_zero = slot(r + 2);
    const bool sample = (r % stride) == (stride / 2) && nsamp < MAX_SAMPLES;
    if (sample) HIPCHK(hipEventRecord(d->ev_s0[nsamp], d->stream));
    launch_column<false>(d, a);
    if (sample) { HIPCHK(hipEventRecord(d->ev_s1[nsamp], d->stream)); nsamp++; }
    launches++;
    if (((r + 1) % CHUNK) == 0 || r == L - 1)
    {
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(d->h_ctl + 2 * chk, d->d_ctl, 2 * sizeof(RamxCtl), hipMemcpyDeviceToHost, d->stream));
      HIPCHK(hipEventRecord(d->ev_chk[chk], d->stream));
      if (pending >= 0)
      {
        HIPCHK(hipEventSynchronize(d->ev_chk[pending]));
        const RamxCtl *h = d->h_ctl + 2 * pending;
        if (h[0].stopped || h[1].stopped) stopped = true;
      }
      pending = chk;
      chk ^= 1;
    }
  }
  HIPCHK(hipEventRecord(d->ev_end, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  RamxCtl h[2];
  HIPCHK(hipMemcpy(h, d->d_ctl, sizeof(h), hipMemcpyDeviceToHost));
  const RamxCtl &f = (L == 0) ? h[1] : ((h[0].rows_done > h[1].rows_done) ? h[0] : h[1]);
  d->final_ctl = f;
  if (persistent)
  {
    launches = 1;
    if (f.pad != 0) { ramx_set_error("persistent kernel: device-wide barrier timed out (bounded spin gave up)"); return RAMX_ERR_STATE; }
  }
  if (info)
  {
    info->ret = f.max_row + 1;
    info->rows_executed = f.rows_done;
    info->limit_warning = (f.stopped && f.rows_done - 1 == L - 1) ? 1 : 0;   // ram_extend.c:1225-1231
    info->overflow32 = f.overflow;
    info->n_extendable = d->Nx;
    info->launches = launches;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, d->ev_begin, d->ev_end));
    info->loop_ms = ms;
    double acc = 0; int cnt = 0;
    for (int i = 0; i < nsamp; i++)
    {
      float t = 0;
      if (hipEventElapsedTime(&t, d->ev_s0[i], d->ev_s1[i]) == hipSuccess) { acc += t; cnt++; }
    }
    info->kernel_ms_avg = cnt ? acc / cnt : 0.0;
    info->kernel_samples = cnt;
    info->persistent = persistent ? 1 : 0;
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_download(ramx_dev *d, int8_t *cons, int32_t cons_cap, int32_t *trim_high, int32_t *trim_pos)
{
  if (!d || !d->ready) { ramx_set_error("ramx_dev_download: nothing to download"); return RAMX_ERR_STATE; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int rows = d->final_ctl.rows_done;
  if (cons)
  {
    if (cons_cap < rows) { ramx_set_error("cons buffer too small (%d < %d)", cons_cap, rows); return RAMX_ERR_ARG; }
    if (rows) HIPCHK(hipMemcpy(cons, d->d_cons, (size_t)rows, hipMemcpyDeviceToHost));
  }
  if ((trim_high || trim_pos) && d->Nx)
  {
    int2 *tmp = (int2 *)malloc((size_t)d->Nx * sizeof(int2));
    HIPCHK(hipMemcpy(tmp, d->d_trim, (size_t)d->Nx * sizeof(int2), hipMemcpyDeviceToHost));
    for (int i = 0; i < d->Nx; i++)
    {
      if (trim_high) trim_high[i] = tmp[i].x;
      if (trim_pos) trim_pos[i] = tmp[i].y;
    }
    free(tmp);
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_peek_state(ramx_dev *d, int32_t flank, int32_t *cells, int32_t *high, int32_t *pos)
{
  if (!d || !d->ready || flank < 0 || flank >= d->Nx) { ramx_set_error("ramx_dev_peek_state: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int W = d->p.bandwidth, Q = W + 1, B = 2 * W + 1;
  const int rows = d->final_ctl.rows_done;
  const int4 *S = d->d_state[(rows - 1) & 1];   // K(r) wrote state[r & 1]; rows == 0 -> boundary row in slot 1
  const int tile = flank / 64, lane = flank % 64;
  for (int q = 0; q < Q; q++)
  {
    int4 v;
    HIPCHK(hipMemcpy(&v, S + ((size_t)tile * Q + q) * 64 + lane, sizeof(v), hipMemcpyDeviceToHost));
    cells[4 * q] = v.x; cells[4 * q + 1] = v.y;
    if (2 * q + 1 < B) { cells[4 * q + 2] = v.z; cells[4 * q + 3] = v.w; }
    else { if (high) *high = v.z; if (pos) *pos = v.w; }
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_set_allreduce_cb(ramx_dev *d, ramx_allreduce_cb cb, void *user)
{
  if (!d) { ramx_set_error("ramx_dev_set_allreduce_cb: bad argument"); return RAMX_ERR_ARG; }
  d->cb = cb;
  d->cb_user = user;
  return RAMX_OK;
}

extern "C" int ramx_comm_unique_id(uint8_t id[128])
{
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { ramx_set_error("ncclGetUniqueId: %s", ncclGetErrorString(r)); return RAMX_ERR_COMM; }
  memcpy(id, &u, 128);
  return RAMX_OK;
}

extern "C" int ramx_dev_comm_init(ramx_dev *d, const uint8_t id[128], int rank, int nranks)
{
  if (!d || rank < 0 || rank >= nranks) { ramx_set_error("ramx_dev_comm_init: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->rank = rank; d->nranks = nranks;
  if (nranks == 1) return RAMX_OK;
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclResult_t r = ncclCommInitRank(&d->comm, nranks, u, rank);
  if (r != ncclSuccess) { ramx_set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); d->comm = NULL; return RAMX_ERR_COMM; }
  return RAMX_OK;
}
