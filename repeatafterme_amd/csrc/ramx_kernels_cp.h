// ramx_kernels_cp.h -- the CELL-PARALLEL band: K lanes of a wavefront share one flanking sequence
// (device code of libramx; included by ramx_cp.hip only)
//
// compute_nw_row (bnw_extend.c:750-1048) walks the 2W+1 cells of a band row in order because the insertion term of
// cell j reads cell j-1 of the SAME row (:972-985).  In the transformed state (m, e) of ramx_kernels_common.h that
// dependency is   e_j = max(x_j, e_{j-1}) + ge,  x_j = max(sub_j + go, del_j)   -- a max-plus prefix scan
// (G[j] = max(c[j], G[j-1] + ge) with resets at out-of-bounds cells).  Here a row is split into
// K blocks of C = ceil((2W+1)/K) consecutive cells, one block per lane (K = 2, 4, 8 or 16 lanes: one DPP row holds
// 16/K flanks, a wavefront 64/K):
//
//   pass 1   every lane runs the serial recurrence over its own C cells with no carry-in (5 VALU per cell);
//   scan     the block totals A_l, tilted by l*C*ge, go through a DPP prefix-max inside the K-lane group
//            (row_shr / quad_perm, no LDS), giving every lane the exact e of the cell left of its block;
//   fix-up   m_i = max(m_i, carry + i*ge), e_i = max(e_i, carry + (i+1)*ge)   (3 VALU per cell);
//   reduce   best cell of the row (packed (score, 255 - cell) keys) and the four chain-free candidate maxima of
//            row r+1 (see ramx_kernels_common.h) by a DPP butterfly over the group.
//
// All of it is integer max/add on values that never overflow, so re-association is exact: the stored cells equal the
// serial reference's bit for bit (tests/test_gpu_cp.py compares the per-cell state with the oracle).  A column costs
// ~20 VALU per cell / K lanes instead of 16.5 per cell in one lane: a 100-flank family runs on 13 waves instead of 2.
//
// Cells j >= 2W+1 of the last blocks ("dead" cells) and whole dead lanes exist because K*C > 2W+1.  On the masked
// path they are ordinary out-of-bounds cells (sentinel fill).  On the in-bounds fast path they run free on very
// negative values; four selects per row keep every live value out of them (the insertion chain and the carry are cut
// at cell 2W+1) and the deletion term of cell 2W is forced to the reference's "impossible" value (bnw_extend.c:892).
#pragma once

#include "ramx_kernels_resident.h"
#include "ramx_cp_api.h"

#define CP_IDN (-1500000000)      // identity of the max scans: below every value a live cell can hold, far from INT_MIN
#define CP_IMIN (-2147483647 - 1)
#ifndef CP_G_GROUP
#define CP_G_GROUP 2          // masked path: cells per scheduling group
#endif
#ifndef CP_F_GROUP
#define CP_F_GROUP 4          // fast path: cells per scheduling group (bounds the live ranges of the per-cell temporaries)
#endif


#define CP_QP(a, b, c, d) "quad_perm:[" #a "," #b "," #c "," #d "]"

// dst = max(dst, dpp(dst)): lanes whose DPP source is invalid (row boundary, bound_ctrl:0) or masked keep dst.
// A DPP read needs two wait states after the VALU write of its source (s_nop 1 in front; steps on different
// registers are interleaved where several values are reduced so that only the first step pays it).
#define CP_MAX1(v, ctrl, masks) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " " masks : "+v"(v))
#define CP_FULL "row_mask:0xf bank_mask:0xf"

template <int K>
__device__ __forceinline__ int cp_scan_max(int v)   // inclusive prefix max over the K-lane group (identity: keep own)
{
  if (K == 16)
  {
    CP_MAX1(v, "row_shr:1", CP_FULL); CP_MAX1(v, "row_shr:2", CP_FULL); CP_MAX1(v, "row_shr:4", CP_FULL); CP_MAX1(v, "row_shr:8", CP_FULL);
  }
  else if (K == 2) CP_MAX1(v, CP_QP(0, 0, 2, 2), CP_FULL);
  else
  {
    CP_MAX1(v, CP_QP(0, 0, 1, 2), CP_FULL);         // within quads: shift by one (lane 0 of a quad takes itself)
    CP_MAX1(v, CP_QP(0, 1, 0, 1), CP_FULL);         // shift by two
    if (K == 8)
    {
      int t = v;
      asm("s_nop 1\n\tv_mov_b32_dpp %0, %0 " CP_QP(3, 3, 3, 3) " " CP_FULL : "+v"(t));     // quad totals
      asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(v) : "v"(t));   // quads 1, 3 take quad 0, 2
    }
  }
  return v;
}

// five values at once over the group (butterfly: every lane ends with the maximum)
template <int K>
__device__ __forceinline__ void cp_allmax5(int &a, int &b, int &c, int &d, int &e)
{
#define CP_STEP5(ctrl) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " " CP_FULL "\n\tv_max_i32_dpp %1, %1, %1 " ctrl " " CP_FULL \
                           "\n\tv_max_i32_dpp %2, %2, %2 " ctrl " " CP_FULL "\n\tv_max_i32_dpp %3, %3, %3 " ctrl " " CP_FULL  \
                           "\n\tv_max_i32_dpp %4, %4, %4 " ctrl " " CP_FULL : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
  if (K >= 2) CP_STEP5(CP_QP(1, 0, 3, 2));
  if (K >= 4) CP_STEP5(CP_QP(2, 3, 0, 1));
  if (K >= 8) CP_STEP5("row_half_mirror");
  if (K >= 16) CP_STEP5("row_mirror");
#undef CP_STEP5
}

// Sum over the flanks of a wave of four values that are replicated inside every K-lane group, each < 2^23 (the key
// bound of cp_supported), 64/K groups: fits 32 bits.  Returns wave-uniform totals.
template <int K>
__device__ __forceinline__ void cp_flank_sum4(unsigned (&v)[4])
{
#define CP_ADD4(ctrl, masks) asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 " ctrl " " masks "\n\tv_add_u32_dpp %1, %1, %1 " ctrl " " masks \
                                 "\n\tv_add_u32_dpp %2, %2, %2 " ctrl " " masks "\n\tv_add_u32_dpp %3, %3, %3 " ctrl " " masks    \
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
  if (K <= 2) CP_ADD4("row_ror:2", CP_FULL);
  if (K <= 4) CP_ADD4("row_ror:4", CP_FULL);
  if (K <= 8) CP_ADD4("row_ror:8", CP_FULL);        // every lane: total of its row of 16
  CP_ADD4("row_bcast:15", "row_mask:0xa bank_mask:0xf");
  CP_ADD4("row_bcast:31", "row_mask:0xc bank_mask:0xf");
#undef CP_ADD4
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (unsigned)__builtin_amdgcn_readlane((int)v[k], 63);
}

// byte BYTE of `word` (a class index times 4), as it is / times 4: LDS offsets of the 4-byte and the 16-byte table rows
template <int BYTE>
__device__ __forceinline__ unsigned cp_byte(unsigned word)
{
  unsigned d;
  if (BYTE == 0) asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(d) : "v"(word));
  else if (BYTE == 1) asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(d) : "v"(word));
  else if (BYTE == 2) asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(d) : "v"(word));
  else asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(d) : "v"(word));
  return d;
}
template <int BYTE>
__device__ __forceinline__ unsigned cp_byte_x4(unsigned word)
{
  unsigned d;
  if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(word));
  else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(word));
  else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(word));
  else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(word));
  return d;
}

// LDS words that several waves of a workgroup meet at (vote-wave mode) are read and written with explicit DS instructions: a
// `volatile` access through a generic pointer is compiled to a FLAT load / store with sc0 sc1 -- an order of magnitude
// slower, and its s_waitcnt vmcnt(0) also waits for every global load in flight (the base word fetched a row ahead).
__device__ __forceinline__ unsigned cp_lds_off(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ unsigned long long cp_lds_ld64(const void *p)
{
  unsigned long long v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cp_lds_off(p)) : "memory");
  return v;
}
__device__ __forceinline__ void cp_lds_st64(void *p, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" : : "v"(cp_lds_off(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void cp_lds_st32(void *p, int v) { asm volatile("ds_write_b32 %0, %1" : : "v"(cp_lds_off(p)), "v"(v) : "memory"); }

// Score tables in LDS.  win[b][class] = M[b][class]: the substitution score of the row being computed (winner b of the
// vote), one ds_read_b32 per cell at (4 * class + 64 * b).  cand[class] = {M[A][class], M[C][class], M[G][class],
// M[T][class]} as four int32, one ds_read_b128 per cell: the candidates' adds are then plain v_add_u32 on two vector
// registers, which issue at twice the rate of the SDWA byte adds a packed row needs (tools/microbench/valu_rate.hip).
// Rows of one table sit in different LDS banks, lanes reading the same class broadcast.
// 64-bit row sums and lane reads of the vote fold: ramx_kernels_resident.h
#define cp_row_sum_u64 prk_row_sum_u64
#define cp_readlane_u64 prk_readlane_u64

struct CpTabs
{
  int4 cand[16];
  int win[4][16];
};

template <int W, int K>
struct CpCfg
{
  static constexpr int B = 2 * W + 1;
  static constexpr int C = (B + K - 1) / K;          // cells per lane
  static constexpr int LB = (B - 1) / C;             // lane holding cell B-1
  static constexpr int IB = (B - 1) % C;             // its index there
  static constexpr int NA = (C + 1 + 7) / 8;         // aligned 8-nibble windows covering the C+1 lookups of a row
  static constexpr int NWL = (C + 7) / 8 + 1;        // base words spanned by nibbles s .. s+C
  // Workgroups have at most 512 threads (two waves per SIMD, 256 VGPRs each): measured on one CU, a family runs
  // fastest on 4-8 waves -- fewer, longer blocks per lane beat more lanes per flank once every SIMD has a wave or two,
  // because the scan / reduction / vote cost is per wave (tools/cp_cmp_k.sh: 100 flanks at W = 40: 2.58 us per column
  // with 4 lanes per flank, 2.95 with 8).  Blocks of up to 41 cells fit that register budget.
  static constexpr int MAXT = C <= 41 ? 512 : 0;
  // device-wide mode with a vote wave (RAMX_CP_SYNCW_MAXC): eight band waves and the vote wave while the saved row fits
  // the registers of nine waves (168 each); seven band waves and the vote wave when it is kept in LDS
  static constexpr int MAXT_DEV = C <= RAMX_CP_SYNCW_REGC ? 576 : MAXT;
};

// per-lane constants of the cell-parallel band
struct CpLane
{
  int l;                 // lane inside the group: owns cells l*C .. l*C + C-1
  int j0;                // l*C
  int lCge, lm1Cge;      // l*C*ge, (l-1)*C*ge: tilt of the block totals
  bool pL0, pLast;       // l == 0, l == K-1
  bool pFull, pPart, pDead;   // l < LB, l == LB, l > LB
  int keyfix;            // 255 - j0 - (C-1): turns a block-local key into a band-global one
  // per row (masked path): intervals as (negated lower end, span) for one unsigned compare per cell;
  // an empty interval is (-(1 << 20), 0)
  int nlo, span;         // cell i of this lane is in bounds iff (unsigned)(i + nlo) <= span (dead cells excluded)
  int nclo, cspan;       // candidate cell i (row r+1) is in bounds iff (unsigned)(i + nclo) <= cspan
  int iW;                // cells i < iW have band index j < W
  int iWr;               // iW in the first W rows, else a value no cell index reaches: edge fill applies iff i < iWr
};

// Selects of the masked path.  A comparison and the select that consumes it are ONE asm statement joined through VCC:
// written as C++ the (loop-invariant or early-computable) predicates of a whole block are hoisted by the scheduler into
// dozens of SGPR pairs, which spill into vector registers and from there into scratch.  volatile: otherwise the fills of
// pass 1 are kept alive (common subexpressions) for the fix-up pass, one more register per cell.
__device__ __forceinline__ int cp_sel_in(int ipn /* i + nlo */, int span, int inside, int outside)
{
  int d;
  asm volatile("v_cmp_ge_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %4, %3, vcc" : "=v"(d) : "v"(span), "v"(ipn), "v"(inside), "v"(outside) : "vcc");
  return d;
}
template <int I>
__device__ __forceinline__ int cp_fill(int iWr, int edge)      // (I < iWr) ? edge : SENT      (bnw_extend.c:990-1002)
{
  int d, sent = SENT;
  asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %4, %3, vcc" : "=v"(d) : "n"(I), "v"(iWr), "v"(edge), "v"(sent) : "vcc");
  return d;
}

// Row r from row r-1 (bnw_extend.c:750-1048), cells split over the group.  G: masked (general) path.
// sf(ic): the winner's substitution score of cell i (an LDS lookup).  The scores are fetched CP_SF_AHEAD cells ahead of
// their use: blocks of up to 21 cells fetch everything up front (lowest latency), blocks of 41 cells keep twelve in flight
// instead of a register per cell (device-wide W = 40, K = 2: 225 spilled registers -> 74).
#ifndef CP_SF_AHEAD
#define CP_SF_AHEAD 12
#endif
template <int W, int K, bool G, class SF>
__device__ __forceinline__ void cp_update(const CpLane &ln, const int go, const int ge, const int edgeFx,
                                          SF &&sf, int (&m)[CpCfg<W, K>::C], int (&e)[CpCfg<W, K>::C])
{
  typedef CpCfg<W, K> Cfg;
  constexpr int C = Cfg::C, IB = Cfg::IB;
  constexpr int QA = C <= 21 ? C : CP_SF_AHEAD;       // measured: a queue costs blocks of 21 cells 8 %, saves blocks of 41 cells 5 %
  int sq[QA];
  static_for([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; sq[i] = sf(ic); },
             std::make_integer_sequence<int, QA>{});
  if (G)
  {
    // The masked and the fast variant start with the same arithmetic on every cell (score lookup, substitution term);
    // left alone, the optimiser hoists those 2C values above the branch that selects the variant and keeps them alive
    // across it.  Opaque copies of the row make the two instruction streams different.
    static_for([&](auto ic) __attribute__((always_inline))
    {
      constexpr int i = decltype(ic)::value;
      int x = m[i];
      asm("" : "+v"(x));
      m[i] = x;
    }, std::make_integer_sequence<int, C>{});
  }
  // previous row's e of the cell right of this block: the deletion term of the block's last cell
  int peNext = e[0];
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 " CP_FULL : "=v"(peNext) : "v"(e[0]), "0"(NEG));   // lane 15 of a row keeps NEG
  if (K < 16) peNext = ln.pLast ? NEG : peNext;
  int eprev = ln.pL0 ? NEG : CP_IDN;                 // bnw_extend.c:802-804: no insertion into cell 0
  // compile-time cell indices (static_for, not the loop unroller: an array indexed by a loop variable is turned into
  // one wide vector register by the alloca promotion that runs before unrolling, and every branch then copies it whole)
  static_for([&](auto ic) __attribute__((always_inline))
  {
    constexpr int i = decltype(ic)::value;
    // masked path: keep the scheduler from hoisting every cell's fills and range tests to the top of the block
    if constexpr ((i % (G ? CP_G_GROUP : CP_F_GROUP)) == 0) __builtin_amdgcn_sched_barrier(0);
    const int sub = m[i] + sq[i % QA];                           // :950-956, M[besta][base of cell i]
    if constexpr (i + QA < C) sq[i % QA] = sf(std::integral_constant<int, (i + QA < C ? i + QA : 0)>{});
    int del = peNext;                                            // :892-905
    if constexpr (i + 1 < C) del = e[i + 1];
    if constexpr (i == IB) del = ln.pPart ? NEG : del;           // cell 2W has no deletion predecessor
    if constexpr (!G && i == IB + 1) eprev = ln.pPart ? CP_IDN : eprev;   // cut the chain into the dead cells
    const int mN = vmax3(sub, del, eprev);                       // :1007-1018
    int eN = vmax3(sub + go, del, eprev) + ge;
    if (G) eN = cp_sel_in(i + ln.nlo, ln.span, eN, cp_fill<i>(ln.iWr, edgeFx) + ge);   // an out-of-bounds cell restarts the chain
    m[i] = mN; e[i] = eN; eprev = eN;
  }, std::make_integer_sequence<int, C>{});
  // carry: e of the cell left of the block = max over the lanes to the left of (their total decayed by ge per cell)
  int t = eprev - ln.lCge;
  t = cp_scan_max<K>(t);
  int carry = CP_IDN;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 " CP_FULL : "=v"(carry) : "v"(t), "0"(CP_IDN));
  if (K < 16) carry = ln.pL0 ? CP_IDN : carry;
  carry += ln.lm1Cge;
  if (!G) carry = ln.pDead ? CP_IDN : carry;
  int c = carry;
  static_for([&](auto ic) __attribute__((always_inline))
  {
    constexpr int i = decltype(ic)::value;
    if constexpr ((i % (G ? CP_G_GROUP : CP_F_GROUP)) == 0) __builtin_amdgcn_sched_barrier(0);
    if constexpr (!G && i == IB + 1) c = ln.pPart ? CP_IDN : c;
    int mi = imax(m[i], c);
    c += ge;
    int ei = imax(e[i], c);
    if (G)
    {
      const int vF = cp_fill<i>(ln.iWr, edgeFx), ipn = i + ln.nlo;
      mi = cp_sel_in(ipn, ln.span, mi, vF);
      ei = cp_sel_in(ipn, ln.span, ei, vF + ge);
    }
    m[i] = mi; e[i] = ei;
  }, std::make_integer_sequence<int, C>{});
}

// Best cell of row r (value, lowest cell on ties: bnw_extend.c:1020-1024) and the best cells of the four candidate rows
// r+1 (chain-free rule of ramx_kernels_common.h), reduced over the group: every lane returns the same values.
// one value over the group (butterfly: every lane ends with the maximum)
template <int K>
__device__ __forceinline__ void cp_allmax1(int &a)
{
  if (K >= 2) CP_MAX1(a, CP_QP(1, 0, 3, 2), CP_FULL);
  if (K >= 4) CP_MAX1(a, CP_QP(2, 3, 0, 1), CP_FULL);
  if (K >= 8) CP_MAX1(a, "row_half_mirror", CP_FULL);
  if (K >= 16) CP_MAX1(a, "row_mirror", CP_FULL);
}

// LEAN (in-bounds fast path only): the best VALUE of row r and nothing else -- no candidate rows, no index of the best cell.
// Taken for a wave none of whose flanks can contribute more than its cap to the next vote or set a record in this row; the
// proof is the one of ramx_kernels_resident.h (prk_band_fast): with go, ge <= 0 the best cell of row r is at most
// P = max(0, largest matrix entry) above the best cell of row r-1 and every candidate row r+1 at most 2P above it.
template <int W, int K>
__device__ __forceinline__ void cp_reduce_lean(const CpLane &ln, const int (&m)[CpCfg<W, K>::C], int &bestF, int &jbest, int (&bestA)[4])
{
  typedef CpCfg<W, K> Cfg;
  constexpr int C = Cfg::C, IB = Cfg::IB;
  int kb = CP_IMIN, kp = CP_IMIN;
  static_for([&](auto ic) __attribute__((always_inline))
  {
    constexpr int i = decltype(ic)::value;
    int key = m[i];
    if constexpr (i > IB) key = ln.pFull ? key : CP_IMIN;         // dead cells of the lane that holds cell 2W
    if constexpr ((i & 1) == 0 && i + 1 < C) kp = key;
    else if constexpr ((i & 1) != 0) kb = imax3(kb, kp, key);
    else kb = imax(kb, key);
  }, std::make_integer_sequence<int, C>{});
  kb = ln.pDead ? CP_IMIN : kb;
  cp_allmax1<K>(kb);
  bestF = kb;
  jbest = 0;
  bestA[0] = bestA[1] = bestA[2] = bestA[3] = CP_IDN;
}

template <int W, int K, bool G>
__device__ __forceinline__ void cp_reduce(const CpLane &ln, const CpTabs &tabs, const unsigned (&AE)[CpCfg<W, K>::NA],
                                          const unsigned (&AO)[CpCfg<W, K>::NA], const int (&m)[CpCfg<W, K>::C],
                                          const int (&e)[CpCfg<W, K>::C], int &bestF, int &jbest, int (&bestA)[4])
{
  typedef CpCfg<W, K> Cfg;
  constexpr int C = Cfg::C, IB = Cfg::IB;
  int kb = CP_IMIN, kp = CP_IMIN, mE = CP_IDN, eP = CP_IDN;
  int bA[4] = { CP_IDN, CP_IDN, CP_IDN, CP_IDN }, pend[4] = { CP_IDN, CP_IDN, CP_IDN, CP_IDN };
  static_for([&](auto ic) __attribute__((always_inline))
  {
    constexpr int i = decltype(ic)::value;
    if constexpr ((i % (G ? CP_G_GROUP : CP_F_GROUP)) == 0) __builtin_amdgcn_sched_barrier(0);
    int key = (int)(((unsigned)m[i] << 8) | (unsigned)(C - 1 - i));
    if (G) key = cp_sel_in(i + ln.nlo, ln.span, key, CP_IMIN);
    else if constexpr (i > IB) key = ln.pFull ? key : CP_IMIN;
    int ms = m[i];
    if (G) ms = cp_sel_in(i + ln.nclo, ln.cspan, ms, SENT);
    // candidate cell i of row r+1 is aligned to the base of row r's cell i+1
    constexpr int in = i + 1;
    const unsigned off = cp_byte_x4<((in & 7) >> 1)>((in & 1) ? AO[in >> 3] : AE[in >> 3]);
    const int4 sc = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(&tabs.cand[0]) + off);
    const int t4[4] = { ms + sc.x, ms + sc.y, ms + sc.z, ms + sc.w };
    int ev = e[i];
    if constexpr (i == 0) ev = ln.pL0 ? CP_IDN : ev;             // e of cell 0 is nobody's deletion term
    if constexpr ((i & 1) == 0 && i + 1 < C)
    {
      kp = key; eP = ev;
#pragma unroll
      for (int a = 0; a < 4; a++) pend[a] = t4[a];
    }
    else if constexpr ((i & 1) != 0)
    {
      kb = imax3(kb, kp, key); mE = imax3(mE, eP, ev);
#pragma unroll
      for (int a = 0; a < 4; a++) bA[a] = imax3(bA[a], pend[a], t4[a]);
    }
    else
    {
      kb = imax(kb, key); mE = imax(mE, ev);
#pragma unroll
      for (int a = 0; a < 4; a++) bA[a] = imax(bA[a], t4[a]);
    }
  }, std::make_integer_sequence<int, C>{});
  if (!G) kb = ln.pDead ? CP_IMIN : kb;
  kb += ln.keyfix;                                               // low byte: 255 - j of the block's best cell
  int r0 = imax(bA[0], mE), r1 = imax(bA[1], mE), r2 = imax(bA[2], mE), r3 = imax(bA[3], mE);
  cp_allmax5<K>(r0, r1, r2, r3, kb);
  bestA[0] = r0; bestA[1] = r1; bestA[2] = r2; bestA[3] = r3;
  bestF = kb >> 8;
  jbest = 255 - (kb & 255);
}

// ------------------------------------------------------------------------------------------
// the kernel: K lanes per flank.
//   DEV = false   one workgroup = one family (batch mode, seam 1 for families up to one workgroup): block-local vote
//   DEV = true    one flank set spread over the grid (plain launch of at most one workgroup per CU; several sets per launch
//                 in batch mode): the per-column vote goes through the sharded ticket words of the persistent kernel
//                 (ramx_kernels_resident.h: every workgroup adds its four partial sums, tagged with an arrival ticket, into
//                 one of 32 shards; wave 0 of every workgroup polls the shards of the column it is about to start); with
//                 the flanks sharded over ranks the ranks' totals cross the devices through the mailboxes (cross_device).
//                 Co-residency is not guaranteed by the launch: every spin is bounded, a timeout raises the set's error
//                 word, every workgroup of the set leaves and the host repeats the work on another route.
// ------------------------------------------------------------------------------------------
#ifndef CP_SYNC_FIRST_SLEEP
#define CP_SYNC_FIRST_SLEEP 4   // s_sleep units (64 clocks) before the vote wave's first look at the tickets of a row
#endif
#ifdef RAMX_CP_TIMING
#define CP_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                        __builtin_amdgcn_sched_barrier(0); tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define CP_TICK(k) do { } while (0)
#endif

// phase probes of the vote-wave mode (-DRAMX_CP_TIMING): sums of shader clocks per phase, block 0, written to a.dbg[wave * 16 + k]
#ifdef RAMX_CP_TIMING
#define SP_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                        __builtin_amdgcn_sched_barrier(0); sp_t[k] += t_ - sp_last; sp_last = t_; } while (0)
#else
#define SP_TICK(k) do { } while (0)
#endif

#ifndef CP_MIN_WAVES_PER_SIMD
#define CP_MIN_WAVES_PER_SIMD 2      // 512 threads, 2 waves per SIMD: 256 VGPRs
#endif
template <int W, int K, bool DEV>
__global__ __launch_bounds__((DEV ? (CpCfg<W, K>::MAXT_DEV > 0 ? CpCfg<W, K>::MAXT_DEV : 64) : (CpCfg<W, K>::MAXT > 0 ? CpCfg<W, K>::MAXT : 64)), CP_MIN_WAVES_PER_SIMD)
void ramx_cp_kernel(const CPArgs a)
{
  typedef CpCfg<W, K> Cfg;
  constexpr int B = Cfg::B, C = Cfg::C, NA = Cfg::NA, NWL = Cfg::NWL, FPW = 64 / K;
  constexpr bool SYNCW = DEV && C <= RAMX_CP_SYNCW_MAXC;  // device-wide mode, blocks of up to 21 cells: may run with a dedicated vote wave and speculative columns
  const bool vw = SYNCW && (C <= RAMX_CP_SYNCW_REGC || a.vote_wave != 0);   // (blocks of 12..21 cells: the host chooses, ramx_cp_device_plan)
  static_assert(K == 2 || K == 4 || K == 8 || K == 16, "lanes per flank");
  static_assert(B <= 255, "cell index must fit the key's low byte");
  struct Smem
  {
    CpTabs tabs;                                     // first: the table rows are addressed with 16-bit immediate offsets
    unsigned long long vote[3][4];                   // DEV: [0..1] = this workgroup's partial sums (double buffered), [2] = the device-wide vote
    int fail, dec1, pad[2];                         // dec1: the vote wave's decision word for the band waves (SYNCW, two-barrier variant)
    // two-barrier variant, blocks of 12..21 cells: the row a band wave may have to restore (m then e, four values per 16-byte slot)
    int4 save1[(SYNCW && C > RAMX_CP_SYNCW_REGC) ? (2 * C + 3) / 4 : 1][(SYNCW && C > RAMX_CP_SYNCW_REGC) ? 448 : 1];
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
  // wave index through readfirstlane: `live` must be PROVABLY wave-uniform, or the band sits in a divergent region and
  // every column ends with one predicated copy per state register (phi of old and new row)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  FamDesc fd;
  CpDevDesc dd;
  dd.first = 0; dd.nx = 0; dd.b = 0; dd.nb = 1; dd.id = 0; dd.vs = 0;
  if (DEV)
  {
    // workgroup b of a set holds (band waves) * 64 / K consecutive flanks of it
    dd = a.dev[blockIdx.x];
    const int per = (blockDim.x - (vw ? 64 : 0)) / K;
    fd.tile0 = 0; fd.ntiles = 0; fd.id = dd.id;
    fd.nx = dd.nx - dd.b * per;                      // flanks (of this workgroup) that exist
    fd.nx = fd.nx < 0 ? 0 : (fd.nx > per ? per : fd.nx);
  }
  else fd = a.fam[blockIdx.x];
  const int wg = DEV ? dd.b : 0;                     // index of this workgroup inside its flank set
  PShard *const vote = a.vote + (size_t)dd.vs * RAMX_CP_NSETS * NSHARD;      // DEV only
  unsigned *const errw = a.err + (size_t)dd.vs * 16;
  // SYNCW: wave 0 holds no flank -- it runs the device-wide vote (sync_column below) while waves 1.. run the band
  const int f = vw ? (wave > 0 ? ((int)threadIdx.x - 64) / K : 0) : threadIdx.x / K;   // flank inside the family / workgroup
  const bool live = vw ? (wave > 0 && (wave - 1) * FPW < fd.nx) : (wave * FPW < fd.nx);  // wave-uniform: does this wave hold any flank?
  const bool active = live && f < fd.nx;
  const int n = DEV ? dd.first + wg * ((int)(blockDim.x - (vw ? 64 : 0)) / K) + (live ? f : 0) : fd.tile0 * 64 + (live ? f : 0);
  const int my_shard_blocks = DEV ? (dd.nb - (lane & (NSHARD - 1)) + NSHARD - 1) / NSHARD : 0;   // wave 0: blocks arriving on shard lane & 31
  int failed = 0;

  if (threadIdx.x < 16)
  {
    const int cls = threadIdx.x;
    int4 v = make_int4(0, 0, 0, 0);
    if (cls < RAMX_NCLASS) v = make_int4(a.tab[cls][0], a.tab[cls][1], a.tab[cls][2], a.tab[cls][3]);
    sm.tabs.cand[cls] = v;
    sm.tabs.win[0][cls] = v.x; sm.tabs.win[1][cls] = v.y; sm.tabs.win[2][cls] = v.z; sm.tabs.win[3][cls] = v.w;
  }
  if (threadIdx.x < 12) sm.vote[threadIdx.x >> 2][threadIdx.x & 3] = 0ULL;
  if (threadIdx.x == 0) sm.fail = 0;
  __syncthreads();

  long long max_ext = 0;
  int max_row = -1, rows_done = 0, ovf = 0, stopped = 0;
#ifdef RAMX_CP_TIMING
  unsigned long long sp_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, sp_last = 0;
#endif
  // thread k < 4 of the workgroup: word k of the workgroup's sums for row r+1, with the arrival ticket
  auto send_words = [&](int r, unsigned long long t) __attribute__((always_inline))
  {
    if (a.test_drop_row > 0 && r + 1 == a.test_drop_row && wg == dd.nb - 1 && (a.test_drop_id < 0 || a.test_drop_id == dd.id)) return;   // test hook: every workgroup then times out
    PShard *sh = vote + (size_t)((r + 1) & (RAMX_CP_NSETS - 1)) * NSHARD + (wg % NSHARD);
    __hip_atomic_fetch_add(&sh->word[threadIdx.x], t + PRK_BIAS + PRK_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the two ticket words lane (shard + 32 * half) watches for row r
  auto vote_src = [&](int r) __attribute__((always_inline)) -> const unsigned long long *
  {
    const int sidx = lane & (NSHARD - 1), half = lane >> 5;
    return &vote[(size_t)(r & (RAMX_CP_NSETS - 1)) * NSHARD + sidx].word[2 * half];
  };
  // wave 0, all lanes: add the other ranks' totals for row r to v (this rank's), exchanging through the mailboxes
  auto cross_device = [&](const int r, long long (&v)[4]) __attribute__((always_inline))
  {
      // ---- cross-device step (flanks sharded over ranks).  Workgroup 0 of every rank stores its rank's four totals,
      // tagged with the column number, into slot [r % 3][rank] of every OTHER rank's mailbox (system-scope stores over
      // xGMI, or PCIe for the host-memory boxes); every workgroup adds the other ranks' words to the local totals it has
      // just folded itself -- the local part never takes the detour through a mailbox.
      const unsigned long long tag = (unsigned long long)((r + PEER_TAG_OFFSET) & 0xffff) << 48;
      const bool other = lane < a.nranks && lane != a.rank;
      if (wg == 0 && other)
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
      unsigned long long yy[4] = { 0, 0, 0, 0 };
      bool got = !other;
      unsigned spins2 = 0;
      const PeerBox *pollbox = (a.mirror != NULL && wg != 0) ? a.mirror : a.box;
      for (;;)
      {
        if (!got)
        {
#pragma unroll
          for (int k = 0; k < 4; k++) yy[k] = __hip_atomic_load(&pollbox->slot[r % 3][lane][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          got = (yy[0] >> 48) == (tag >> 48) && (yy[1] >> 48) == (tag >> 48) && (yy[2] >> 48) == (tag >> 48) && (yy[3] >> 48) == (tag >> 48);
          if (got && a.mirror != NULL && wg == 0)
          {
            // host-memory boxes: only workgroup 0 polls across PCIe; it passes every arriving word on to the local pollers
#pragma unroll
            for (int k = 0; k < 4; k++)
              __hip_atomic_store(&a.mirror->slot[r % 3][lane][k], yy[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (__all(got)) break;
        if (++spins2 > PRK_SPIN_LIMIT || ((spins2 & 1023u) == 0 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
        {
          failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      failed = __any(failed) ? 1 : 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] += wave_sum_ll((other && !failed) ? (long long)(yy[k] & PEER_VMASK) - PEER_VBIAS : 0LL);
  };
  // ---- a vote wave (SYNCW: device-wide mode, blocks of up to RAMX_CP_SYNCW_MAXC cells) ------------------------------------
  // The vote of row r needs two trips through the memory fabric; the serial chain of the reference
  // (ram_extend.c:1081-1085 feeding :970) is  decision(r) -> tickets(r+1) -> exchange -> decision(r+1).  Wave 0 holds no
  // flank: it is the VOTE WAVE and runs nothing but the chain; the band waves compute row r on the workgroup's own argmax
  // while the vote travels ("one speculative column" below).  (Round 3 also built a barrier-free variant that ran the band
  // waves up to four rows ahead through LDS rings; it measured slower -- the device-wide chain, not the band, is the floor:
  // profiles/r03_ab_spec5.log -- and was removed in round 4.)
  // test hook (RAMX_TEST_CP_WRONG_EVERY=n): the guess of every n-th row is replaced by another base, identically in the
  // band waves and in the vote wave, so that rollbacks happen at known rows
  auto perturb = [&](const int row, const int g) __attribute__((always_inline)) -> int
  {
    if (a.test_wrong_every > 0 && ((row + 1) % a.test_wrong_every) == 0) return (g + 1 + (row / a.test_wrong_every) % 3) & 3;
    return g;
  };

  int respec = 0;                                    // vote wave: epoch switches (mispredicted rows)


  CpLane ln;
  ln.l = threadIdx.x % K;
  ln.j0 = ln.l * C;
  ln.lCge = ln.l * C * a.ge;
  ln.lm1Cge = (ln.l - 1) * C * a.ge;
  ln.pL0 = ln.l == 0; ln.pLast = ln.l == K - 1;
  ln.pFull = ln.l < Cfg::LB; ln.pPart = ln.l == Cfg::LB; ln.pDead = ln.l > Cfg::LB;
  ln.keyfix = 255 - ln.j0 - (C - 1);
  ln.iW = W - ln.j0;
  ln.nlo = ln.span = ln.nclo = ln.cspan = ln.iWr = 0;

  // go and ge live in VECTOR registers on purpose: on gfx950 v_add_u32 issues at twice the rate with two VGPR operands
  // (tools/microbench/valu_rate.hip: 2 cycles per wave-instruction, 4 with an SGPR operand); the empty asm hides from
  // the compiler that the values are uniform
  int vgo = a.go, vge = a.ge;
  asm volatile("" : "+v"(vgo), "+v"(vge));
  int m[C], e[C];
  int high = 0, pos = 0, thigh = 0, tpos = 0;
  int prevBest = 0x3fffffff;           // best cell of the previous FINAL row (LEAN test in band()): unknown at first
  const int2 bd = a.bounds[n];

  // base words: the lane's window covers nibbles s .. s+C, s = j0 + r + 8; w[k] = word (s >> 3) + k, one word ahead
  unsigned w[NWL + 1];
  int s = ln.j0 + 7;                                 // r = -1
  {
    const int w0 = s >> 3;
    static_for([&](auto kc) __attribute__((always_inline))
    {
      constexpr int k = decltype(kc)::value;
      const int wi = (w0 + k < a.KW) ? w0 + k : a.KW - 1;
      w[k] = a.bases[(size_t)wi * a.Np + n];
    }, std::make_integer_sequence<int, NWL + 1>{});
  }
  unsigned wpre;                                     // the word the next slide brings in (see finish_column)
  {
    const int wn = (s >> 3) + 1 + NWL;
    wpre = a.bases[(size_t)(wn < a.KW ? wn : a.KW - 1) * a.Np + n];
  }
  // ---- pieces of a column --------------------------------------------------------------------
  // the lane's base window as LDS offsets: byte b of AE[k] / AO[k] = 4 * class of cell 8k + 2b / 8k + 2b + 1 (cells 0 .. C:
  // cell C is the candidates' base of the block's last cell).  They depend only on the base stream and are computed at
  // the end of the previous column, ahead of the vote exchange.
  auto window_words = [&](unsigned (&AE)[NA], unsigned (&AO)[NA]) __attribute__((always_inline))
  {
    const int ph4 = 4 * (s & 7);
    static_for([&](auto kc) __attribute__((always_inline))
    {
      constexpr int k = decltype(kc)::value;
      const unsigned A = __builtin_amdgcn_alignbit(w[k + 1], w[k], ph4);
      AE[k] = (A << 2) & 0x3c3c3c3cu;
      AO[k] = (A >> 2) & 0x3c3c3c3cu;
    }, std::make_integer_sequence<int, NA>{});
  };
  // substitution scores of row r's cells against the winner `besta`: win[besta][class]
  auto set_masks = [&](int r) __attribute__((always_inline))
  {
    const int jlo = bd.x - r, jhi = bd.y - r;        // cell j of row r is in bounds iff jlo <= j <= jhi
    const int ilo = jlo - ln.j0, ihi = (jhi < B - 1 ? jhi : B - 1) - ln.j0;
    const int iclo = ilo - 1;                        // candidate cell j' of row r+1 <-> cell j'+1 of row r
    const int ichi = (jhi - 1 < B - 1 ? jhi - 1 : B - 1) - ln.j0;
    ln.nlo = (ihi >= ilo) ? -ilo : -(1 << 20);
    ln.span = (ihi >= ilo) ? ihi - ilo : 0;
    ln.nclo = (ichi >= iclo) ? -iclo : -(1 << 20);
    ln.cspan = (ichi >= iclo) ? ichi - iclo : 0;
    ln.iWr = (r < W) ? ln.iW : -(1 << 20);
  };
  // clamp at 0, cap from below by high + CAPPENALTY (ram_extend.c:1042, 1052-1062); then slide the window by one base:
  // one new word every eighth column, loaded a whole word ahead of its first use
  // contributions of this flank to the vote of the next row (ram_extend.c:1062-1079), given the flank's best score so far
  auto contributions = [&](const int (&bestA)[4], const int high_now, unsigned (&contrib)[4]) __attribute__((always_inline))
  {
    const int capv = high_now + a.cap;
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
      const int b = bestA[c] < 0 ? 0 : bestA[c];
      contrib[c] = active ? (unsigned)((b >= capv) ? b : capv) : 0u;
    }
  };
  auto slide_window = [&]() __attribute__((always_inline))
  {
    // The window slides by one word when the lane's nibble pointer crosses a word (lanes of a flank do so in different
    // columns, so some lane slides in every column).  The incoming word was loaded a column ago (wpre): every lane loads,
    // in every column, the word its next slide will bring in, so nothing ever waits for a load it has just issued.
    s++;
    const bool slide = (s & 7) == 0;
    static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; w[k] = slide ? w[k + 1] : w[k]; },
               std::make_integer_sequence<int, NWL>{});
    w[NWL] = slide ? wpre : w[NWL];
    const int wn = (s >> 3) + 1 + NWL;
    wpre = a.bases[(size_t)(wn < a.KW ? wn : a.KW - 1) * a.Np + n];
  };
  auto finish_column = [&](const int (&bestA)[4], unsigned (&contrib)[4]) __attribute__((always_inline))
  {
    contributions(bestA, high, contrib);
    slide_window();
  };
#ifdef RAMX_CP_TIMING
  unsigned long long tsum[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_amdgcn_s_memtime();
#endif
  // Block-local vote (DEV = false): three sets rotate -- column r reads set r % 3, adds to set (r+1) % 3 and clears set
  // (r+2) % 3 (last read at the top of column r-1, next added to during column r+1; the barrier at the end of every
  // column separates the three uses).
  // Device-wide vote (DEV = true): the waves add into the workgroup's partial sums [(r+1) & 1]; after the barrier four
  // threads forward them, with the arrival ticket, to this workgroup's shard of device set (r+1) % 4 and clear the other
  // buffer; every wave reads the sums it has just helped to build as the workgroup's PREDICTION of the next vote.
  // Four device sets rotate.  Workgroup 0 clears the set of row r+3 in column r, once it has seen every ticket of row r
  // (so everybody has finished reading row r-1, the set's previous user).  The first adds to that set come from
  // workgroups that have seen workgroup 0's ticket for row r+2, which wave 0 of workgroup 0 sends at the end of its
  // column r+1 -- after its own wait_vote(r+1), which begins by draining the wave's outstanding stores.  The clear is
  // therefore complete a column before it has to be, and nobody stalls for it.
  // four wave-uniform values into four LDS words: lanes 0..3 add one word each (one ds_add_u64 for the wave; a single lane
  // adding four words gets four atomics, each wrapped in the compiler's same-address reduction sequence)
  auto add4_lds = [&](unsigned long long *dst, const unsigned long long v0, const unsigned long long v1, const unsigned long long v2,
                      const unsigned long long v3) __attribute__((always_inline))
  {
    if (lane < 4)
    {
      const unsigned long long mine = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
      atomicAdd(&dst[lane], mine);
    }
  };
  auto publish = [&](int r, unsigned (&contrib)[4]) __attribute__((always_inline))
  {
    if (live)
    {
      cp_flank_sum4<K>(contrib);
      add4_lds(sm.vote[DEV ? ((r + 1) & 1) : (r + 4) % 3], contrib[0], contrib[1], contrib[2], contrib[3]);
    }
    // LDS traffic only: the barrier must not wait for the global accesses in flight (the base word loaded for eight
    // columns ahead, the consensus byte) as __syncthreads() would
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (DEV)
    {
      if (threadIdx.x < 4)
      {
        const unsigned long long t = sm.vote[(r + 1) & 1][threadIdx.x];
        sm.vote[r & 1][threadIdx.x] = 0ULL;          // the OTHER buffer: read by everybody a column ago, added to again after the next barrier
        send_words(r, t);
      }
    }
  };
  // DEV: wave 0 waits until every workgroup's contribution to the vote of row r has arrived (bounded spin), folds the 32
  // shards and leaves the four sums in sm.vote[2]; returns after the workgroup barrier.  Same protocol and encoding as
  // ramx_persistent_kernel.
  // (blocks of more than RAMX_CP_SYNCW_MAXC cells: the simple order -- vote, then band; no vote wave)
  auto wait_vote = [&](int r) __attribute__((always_inline))
  {
    if (wave == 0)
    {
      const unsigned long long *src = vote_src(r);
      unsigned spins = 0;
      // workgroup 0: the clearing stores of the previous column (set of row r+2) are complete before this wave sends
      // its next ticket (see publish)
      if (wg == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CP_TICK(8);                // wave 0: drain
      unsigned long long x0 = 0, x1 = 0;
#ifdef CP_PROBE_NO_WAIT          // timing probe only (wrong results): whatever has arrived is the vote
      bool done = true;
#else
      bool done = my_shard_blocks <= 0;
#endif
      for (;;)
      {
        if (__all(done)) break;
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
        if (++spins > PRK_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
        {
          failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      CP_TICK(9);                // wave 0: early samples, polling
      // fold the shards: raw words first (sum + bias and ticket fields are both additive: at most 256 tickets, ten bits),
      // rows of 16 lanes with DPP butterflies, the four rows on the scalar unit; one decode per word at the end
      if (my_shard_blocks <= 0 || failed) { x0 = 0; x1 = 0; }
      x0 = cp_row_sum_u64(x0); x1 = cp_row_sum_u64(x1);
      long long v[4];                                              // this GPU's totals
      {
        const unsigned long long t0 = cp_readlane_u64(x0, 0) + cp_readlane_u64(x0, 16), t1 = cp_readlane_u64(x1, 0) + cp_readlane_u64(x1, 16);
        const unsigned long long t2 = cp_readlane_u64(x0, 32) + cp_readlane_u64(x0, 48), t3 = cp_readlane_u64(x1, 32) + cp_readlane_u64(x1, 48);
        v[0] = (long long)(t0 & (PRK_TICKET - 1)) - (long long)(t0 >> 54) * (long long)PRK_BIAS;
        v[1] = (long long)(t1 & (PRK_TICKET - 1)) - (long long)(t1 >> 54) * (long long)PRK_BIAS;
        v[2] = (long long)(t2 & (PRK_TICKET - 1)) - (long long)(t2 >> 54) * (long long)PRK_BIAS;
        v[3] = (long long)(t3 & (PRK_TICKET - 1)) - (long long)(t3 >> 54) * (long long)PRK_BIAS;
      }
      if (a.nranks > 1 && !failed) cross_device(r, v);
      CP_TICK(10);               // wave 0: fold (and the cross-device step)
      if (lane == 0)
      {
        sm.vote[2][0] = (unsigned long long)v[0]; sm.vote[2][1] = (unsigned long long)v[1];
        sm.vote[2][2] = (unsigned long long)v[2]; sm.vote[2][3] = (unsigned long long)v[3];
        sm.fail = failed;
        if (failed) __hip_atomic_store(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    CP_TICK(11);                 // LDS write, barrier (waves other than 0: the whole wait)
  };
  unsigned AE[NA], AO[NA];
  static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; AE[k] = 0; AO[k] = 0; },
             std::make_integer_sequence<int, NA>{});
  if (live) window_words(AE, AO);
  // ---- column -1: boundary row (ram_extend.c:909-946) and the candidates of row 0 -----------------
  if (a.L > 0)
  {
    unsigned contrib[4] = { 0, 0, 0, 0 };
    if (live)
    {
      int bestA[4], bestF, jb;
      set_masks(-1);
      static_for([&](auto ic) __attribute__((always_inline))
      {
        constexpr int i = decltype(ic)::value;
        const int j = ln.j0 + i, o = j - W;
        int v = (o == 0) ? 0 : a.go + (o < 0 ? -o : o) * a.ge;  // sub = gap = go + |o| * ge, 0 at the centre
        v = (j < B) ? v : SENT;                                    // dead cells: sentinel
        m[i] = v; e[i] = v + a.ge;                                 // e = max(sub + go, gap) + ge with go <= 0
      }, std::make_integer_sequence<int, C>{});
      cp_reduce<W, K, true>(ln, sm.tabs, AE, AO, m, e, bestF, jb, bestA);
      finish_column(bestA, contrib);
      window_words(AE, AO);
    }
    publish(-1, contrib);
  }

  // One column.  The fast and the masked variant are two separate loops below (a wave switches between them when its
  // flanks enter or leave the band's range): with both variants in ONE loop body the register allocator needed ~45
  // registers more than the larger of the two.  Returns true when the column loop ends.
  // the band of one row against winner b: row r from row r-1, best cell of row r, best cells of the candidate rows r+1
  auto band = [&](const int r, const int b, auto gc, int &bestF, int &jb, int (&bestA)[4]) __attribute__((always_inline))
  {
    constexpr bool G = decltype(gc)::value;
#ifdef CP_PROBE_NO_BAND          // timing probe only (wrong results): the column without its arithmetic
    bestF = r; jb = W;
    return;
#endif
    // the winner's score of cell i: table win[b][class of the cell's base], byte offset 64 * b + 4 * class
    const unsigned bb = 0x40404040u * (unsigned)b;
    const char *tw = reinterpret_cast<const char *>(&sm.tabs.win[0][0]);
    auto sf = [&](auto ic) __attribute__((always_inline)) -> int
    {
      constexpr int i = decltype(ic)::value;
      const unsigned off = cp_byte<((i & 7) >> 1)>(((i & 1) ? AO[i >> 3] : AE[i >> 3]) | bb);
      return *reinterpret_cast<const int *>(tw + off);
    };
    cp_update<W, K, G>(ln, vgo, vge, a.go + (r + 1) * a.ge /* edge fill, first W rows only (set_masks: iWr) */, sf, m, e);
    CP_TICK(2);                  // row update
    if constexpr (!G)
    {
      // LEAN: no flank of the wave can contribute more than its cap to the next vote or set a record in this row
      // (prevBest: best cell of row r-1; `high` before this row's record update: the cap only grows)
      const int capfloor = (high + a.cap) > 0 ? (high + a.cap) : 0;
      const bool lean = a.lean_p >= 0 && __all(!active || ((prevBest + 2 * a.lean_p <= capfloor) && (prevBest + a.lean_p <= high)));
      if (lean) cp_reduce_lean<W, K>(ln, m, bestF, jb, bestA);
      else cp_reduce<W, K, false>(ln, sm.tabs, AE, AO, m, e, bestF, jb, bestA);
    }
    else cp_reduce<W, K, G>(ln, sm.tabs, AE, AO, m, e, bestF, jb, bestA);
    CP_TICK(3);                  // reductions
  };
  // ---- one speculative column with a vote wave and two workgroup barriers (SYNCW) --------------------------------------------
  // The vote of row r needs two trips through the memory fabric.  Meanwhile the band waves compute row r against the
  // WORKGROUP'S OWN argmax of the candidate sums (guess) and carry the speculation through the flank records, the
  // contributions to row r+1 and their workgroup sums (barrier A).  Wave 0 holds no flank: during the band it waits for the
  // tickets of row r, folds the shards, applies the stop rule; after A it picks up the workgroup sums, and when the vote
  // agrees with the guess -- nearly always while the flanks still align -- sends them with the ticket for row r+1 at once.
  // One word (sm.dec1) tells the band waves the winner, the stop rule's verdict and the guess for row r+1 (barrier B).
  // When the guess was wrong the band waves restore the saved row r-1, run the band again with the true winner, correct
  // the workgroup sums by the difference (barrier C), and the sums are sent then.  Nothing computed from an unconfirmed
  // guess ever leaves the workgroup.
  // dec: bits 0-1 winner, 2 new maximum, 3 stop, 4 failed, 5 guess was wrong, 6-7 guess for the next row
  int guess_cur = 0;
  auto argmax4_lds = [&](const unsigned long long *ps) __attribute__((always_inline)) -> int
  {
    unsigned long long cbest = 0;
    int g = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const unsigned long long v = ps[k];
      const unsigned long long vk = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      if (vk > cbest) { cbest = vk; g = k; }
    }
    return g;
  };
  // wave 0, lanes 0..3 hold t: argmax with the vote's tie rule
  auto argmax4_lanes = [&](const unsigned long long t) __attribute__((always_inline)) -> int
  {
    unsigned long long cbest = 0;
    int g = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const unsigned long long vk = cp_readlane_u64(t, k);
      if (vk > cbest) { cbest = vk; g = k; }
    }
    return g;
  };
  auto sync_column = [&](const int r) __attribute__((always_inline)) -> bool
  {
    // ---- the vote of row r: every workgroup's contribution has arrived (bounded spin) ----
    const unsigned long long *src = vote_src(r);
    unsigned spins = 0;
    // workgroup 0: the clearing stores of the previous column (set of row r+2) are complete before this wave sends its
    // next ticket (see publish)
    if (wg == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long x0 = 0, x1 = 0;
    bool done = my_shard_blocks <= 0;
    __builtin_amdgcn_s_sleep(CP_SYNC_FIRST_SLEEP);   // the tickets were sent a moment ago: they need half a microsecond to land
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
      if (++spins > PRK_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
      {
        failed = 1;
        break;
      }
    }
    if (my_shard_blocks <= 0 || failed) { x0 = 0; x1 = 0; }
    x0 = cp_row_sum_u64(x0); x1 = cp_row_sum_u64(x1);
    long long v[4];                                              // this GPU's totals
    {
      const unsigned long long t0 = cp_readlane_u64(x0, 0) + cp_readlane_u64(x0, 16), t1 = cp_readlane_u64(x1, 0) + cp_readlane_u64(x1, 16);
      const unsigned long long t2 = cp_readlane_u64(x0, 32) + cp_readlane_u64(x0, 48), t3 = cp_readlane_u64(x1, 32) + cp_readlane_u64(x1, 48);
      v[0] = (long long)(t0 & (PRK_TICKET - 1)) - (long long)(t0 >> 54) * (long long)PRK_BIAS;
      v[1] = (long long)(t1 & (PRK_TICKET - 1)) - (long long)(t1 >> 54) * (long long)PRK_BIAS;
      v[2] = (long long)(t2 & (PRK_TICKET - 1)) - (long long)(t2 >> 54) * (long long)PRK_BIAS;
      v[3] = (long long)(t3 & (PRK_TICKET - 1)) - (long long)(t3 >> 54) * (long long)PRK_BIAS;
    }
    if (a.nranks > 1 && !failed) cross_device(r, v);
    // ---- winner and stop rule (ram_extend.c:1081-1085, 1194-1216) ----
    int besta = 0;
    unsigned chi = 0, clo = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const unsigned hi = (unsigned)((unsigned long long)v[k] >> 32), lo = (unsigned)(unsigned long long)v[k];
      if (hi != 0 || lo > 2147483647u) ovf = 1;                  // the reference's int accumulator would have wrapped
      if (hi > chi || (hi == chi && lo > clo)) { chi = hi; clo = lo; besta = k; }
    }
    const long long curr = (long long)(((unsigned long long)chi << 32) | clo);
    int dist = max_row - r;
    dist = dist < 0 ? -dist : dist;
    const bool new_max = curr >= max_ext + (long long)dist * a.minimp;
    if (new_max) { max_row = r; max_ext = curr; }
    int d2 = r - max_row;
    d2 = d2 < 0 ? -d2 : d2;
    stopped = d2 >= a.when_to_stop;
    rows_done = r + 1;
    if (wg == 0 && !failed)
    {
      if (lane == 0) a.cons_out[(size_t)dd.id * a.L + r] = (signed char)besta;
      if (lane < NSHARD)            // workgroup 0 clears the device set of row r+3 (see publish)
      {
        PShard *z = vote + (size_t)((r + 3) & (RAMX_CP_NSETS - 1)) * NSHARD + lane;
#pragma unroll
        for (int k = 0; k < 4; k++) __hip_atomic_store(&z->word[k], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const bool wrong = besta != guess_cur;
    if (wrong) respec++;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // A: the workgroup's sums for row r+1 are complete
    unsigned long long tsend = 0;
    if (lane < 4)
    {
      tsend = sm.vote[(r + 1) & 1][lane];
      sm.vote[r & 1][lane] = 0ULL;                   // row r's sums: the guess was taken from them a column ago; added to again after B
    }
    int gnext = perturb(r + 1, argmax4_lanes(tsend));
    if (!wrong && !failed && lane < 4) send_words(r, tsend);
    if (lane == 0)
    {
      sm.dec1 = besta | (new_max ? 4 : 0) | (stopped ? 8 : 0) | (failed ? 16 : 0) | (wrong ? 32 : 0) | (gnext << 6);
      if (failed) __hip_atomic_store(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // B: the decision
    if (failed) return true;
    if (wrong)
    {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // C: corrected sums complete
      unsigned long long t2 = 0;
      if (lane < 4) { t2 = sm.vote[(r + 1) & 1][lane]; send_words(r, t2); }
      gnext = perturb(r + 1, argmax4_lanes(t2));
    }
    guess_cur = gnext;
    return stopped || r == a.L - 1;
  };
  auto band_column = [&](const int r, auto gc) __attribute__((always_inline)) -> bool
  {
    constexpr bool G = decltype(gc)::value;
    CP_TICK(7);                  // barrier released .. loop top
    int bestA[4] = { 0, 0, 0, 0 }, bestF = 0, jb = 0;
    constexpr bool SAVE_LDS = C > RAMX_CP_SYNCW_REGC;
    int sm_[SAVE_LDS ? 1 : C], se_[SAVE_LDS ? 1 : C];
    int high1 = high, pos1 = pos;
    // the row before this column: in registers, or (longer blocks) in this thread's column of the LDS save area
    auto row_value = [&](auto qc) __attribute__((always_inline)) -> int &
    {
      constexpr int q = decltype(qc)::value;         // 0 .. C-1: m, C .. 2C-1: e
      if constexpr (q < C) return m[q]; else return e[q - C];
    };
    auto save_row = [&]() __attribute__((always_inline))
    {
      if constexpr (!SAVE_LDS)
        static_for([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; sm_[i] = m[i]; se_[i] = e[i]; },
                   std::make_integer_sequence<int, C>{});
      else
        static_for([&](auto gc) __attribute__((always_inline))
        {
          constexpr int g = decltype(gc)::value;
          int4 v = make_int4(0, 0, 0, 0);
          v.x = row_value(std::integral_constant<int, 4 * g>{});
          if constexpr (4 * g + 1 < 2 * C) v.y = row_value(std::integral_constant<int, (4 * g + 1 < 2 * C ? 4 * g + 1 : 0)>{});
          if constexpr (4 * g + 2 < 2 * C) v.z = row_value(std::integral_constant<int, (4 * g + 2 < 2 * C ? 4 * g + 2 : 0)>{});
          if constexpr (4 * g + 3 < 2 * C) v.w = row_value(std::integral_constant<int, (4 * g + 3 < 2 * C ? 4 * g + 3 : 0)>{});
          sm.save1[g][threadIdx.x - 64] = v;
        }, std::make_integer_sequence<int, (2 * C + 3) / 4>{});
    };
    auto restore_row = [&]() __attribute__((always_inline))
    {
      if constexpr (!SAVE_LDS)
        static_for([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; m[i] = sm_[i]; e[i] = se_[i]; },
                   std::make_integer_sequence<int, C>{});
      else
        static_for([&](auto gc) __attribute__((always_inline))
        {
          constexpr int g = decltype(gc)::value;
          const int4 v = sm.save1[g][threadIdx.x - 64];
          row_value(std::integral_constant<int, 4 * g>{}) = v.x;
          if constexpr (4 * g + 1 < 2 * C) row_value(std::integral_constant<int, (4 * g + 1 < 2 * C ? 4 * g + 1 : 0)>{}) = v.y;
          if constexpr (4 * g + 2 < 2 * C) row_value(std::integral_constant<int, (4 * g + 2 < 2 * C ? 4 * g + 2 : 0)>{}) = v.z;
          if constexpr (4 * g + 3 < 2 * C) row_value(std::integral_constant<int, (4 * g + 3 < 2 * C ? 4 * g + 3 : 0)>{}) = v.w;
        }, std::make_integer_sequence<int, (2 * C + 3) / 4>{});
    };
    unsigned contrib[4] = { 0, 0, 0, 0 };
    if (live)
    {
      if (G) set_masks(r);
      save_row();
      band(r, guess_cur, gc, bestF, jb, bestA);
      if (bestF > high1) { high1 = bestF; pos1 = r + jb - W; }   // ram_extend.c:1140-1150
      contributions(bestA, high1, contrib);
      cp_flank_sum4<K>(contrib);
      add4_lds(sm.vote[(r + 1) & 1], contrib[0], contrib[1], contrib[2], contrib[3]);
    }
    CP_TICK(1);                  // speculative band, records, workgroup sums
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // A
    asm volatile("s_barrier" ::: "memory");                               // B
    const int dec = __builtin_amdgcn_readfirstlane(sm.dec1);
    CP_TICK(0);                  // waiting for the decision
    if (dec & 16) { failed = 1; return true; }
    const int besta = dec & 3;
    const bool new_max = (dec & 4) != 0, wrong = (dec & 32) != 0;
    stopped = (dec >> 3) & 1;
    int gnext = (dec >> 6) & 3;
    if (live)
    {
      if (wrong)
      {
        restore_row();
        band(r, besta, gc, bestF, jb, bestA);
        high1 = high; pos1 = pos;
        if (bestF > high1) { high1 = bestF; pos1 = r + jb - W; }
        unsigned c2[4];
        contributions(bestA, high1, c2);
        cp_flank_sum4<K>(c2);
        // replace this wave's part of the workgroup sums (64-bit wrap-around: the total stays non-negative)
        add4_lds(sm.vote[(r + 1) & 1], (unsigned long long)c2[0] - contrib[0], (unsigned long long)c2[1] - contrib[1],
                 (unsigned long long)c2[2] - contrib[2], (unsigned long long)c2[3] - contrib[3]);
#ifdef RAMX_CP_TIMING
        tsum[6] += 1;            // mispredicted columns
#endif
      }
      high = high1; pos = pos1;
      prevBest = bestF;          // of the row as it stands now (speculative pass accepted, or recomputed)
      if (new_max) { thigh = high; tpos = pos; }                 // :1203-1207
      slide_window();
      window_words(AE, AO);      // next column's window
    }
    CP_TICK(4);                  // records, window slide, next window
    if (wrong)
    {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // C: corrected sums complete
      gnext = perturb(r + 1, argmax4_lds(sm.vote[(r + 1) & 1]));
    }
    guess_cur = gnext;
    CP_TICK(5);
    return stopped || r == a.L - 1;
  };
  auto column_spec = [&](const int r, auto gc) __attribute__((always_inline)) -> bool
  {
    if (wave == 0) return sync_column(r);
    return band_column(r, gc);
  };
  if (vw && a.L > 0) guess_cur = perturb(0, argmax4_lds(sm.vote[0]));     // row 0's workgroup sums (complete since the barrier of column -1)
  auto column = [&](const int r, auto gc) __attribute__((always_inline)) -> bool
  {
    constexpr bool G = decltype(gc)::value;
    if constexpr (SYNCW) { if (vw) return column_spec(r, gc); }
    CP_TICK(7);                  // barrier released .. loop top
    int bestA[4] = { 0, 0, 0, 0 }, bestF = 0, jb = 0;
    if (G && live) set_masks(r);
    if (DEV)
    {
      wait_vote(r);
      if (__builtin_amdgcn_readfirstlane(sm.fail)) { failed = 1; return true; }
    }
    // vote of row r: block-local (added during the previous column).  The sums are non-negative: they are compared
    // as (high, low) unsigned halves on the scalar unit (there is no 64-bit scalar compare; the compiler's choice for a
    // signed 64-bit compare is a chain of vector instructions)
    int besta = 0;
    unsigned chi = 0, clo = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const unsigned long long vv = sm.vote[DEV ? 2 : r % 3][k];
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(vv >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)vv);
      if (hi != 0 || lo > 2147483647u) ovf = 1;                  // the reference's int accumulator would have wrapped
      if (hi > chi || (hi == chi && lo > clo)) { chi = hi; clo = lo; besta = k; }   // ram_extend.c:1081-1085
    }
    const long long curr = (long long)(((unsigned long long)chi << 32) | clo);
    int dist = max_row - r;
    dist = dist < 0 ? -dist : dist;
    const bool new_max = curr >= max_ext + (long long)dist * a.minimp;   // :1194-1196
    if (new_max) { max_row = r; max_ext = curr; }
    int d2 = r - max_row;
    d2 = d2 < 0 ? -d2 : d2;
    stopped = d2 >= a.when_to_stop;                              // :1216
    rows_done = r + 1;
    if (!DEV)
    {
      if (threadIdx.x == 0) a.cons_out[(size_t)fd.id * a.L + r] = (signed char)besta;
      if (threadIdx.x < 4) sm.vote[(r + 2) % 3][threadIdx.x] = 0ULL;
    }
    else if (wg == 0)
    {
      if (threadIdx.x == 0) a.cons_out[(size_t)dd.id * a.L + r] = (signed char)besta;
      if (threadIdx.x < NSHARD)     // workgroup 0 clears the device set of row r+3 (see publish)
      {
        PShard *z = vote + (size_t)((r + 3) & (RAMX_CP_NSETS - 1)) * NSHARD + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; k++) __hip_atomic_store(&z->word[k], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned contrib[4] = { 0, 0, 0, 0 };
    CP_TICK(0);                  // vote read, stop rule
    if (live)
    {
      band(r, besta, gc, bestF, jb, bestA);
      prevBest = bestF;
      if (bestF > high) { high = bestF; pos = r + jb - W; }      // ram_extend.c:1140-1150
      if (new_max) { thigh = high; tpos = pos; }                 // :1203-1207
      finish_column(bestA, contrib);
      window_words(AE, AO);      // next column's window
    }
    CP_TICK(4);                  // records, contributions, window slide, next window
    if (stopped || r == a.L - 1) return true;
    publish(r, contrib);
    CP_TICK(5);                  // wave sum, LDS atomics, barrier
    return false;
  };
  // wave-uniform: every flank of the wave covers the whole band of rows r and r+1 (padding flanks: all-N stream, masked vote)
  auto is_fast = [&](const int r) __attribute__((always_inline)) -> bool
  {
#ifdef CP_PROBE_NO_G
    return true;
#elif defined(CP_PROBE_NO_F)
    return false;
#else
    return !live || __all(!active || ((bd.x - r <= 0) && (bd.y - r >= B)));
#endif
  };
#ifdef RAMX_CP_TIMING
  tlast = __builtin_amdgcn_s_memtime();
#endif
  for (int r = 0; r < a.L;)
  {
    bool end = false;
    if (is_fast(r))
    {
      do { end = column(r, std::false_type{}); r++; } while (!end && r < a.L && is_fast(r));
    }
    else
    {
      end = column(r, std::true_type{});
      r++;
    }
    if (end) break;
  }
#ifdef RAMX_CP_TIMING
  if (a.dbg != NULL && blockIdx.x == 0 && lane == 0 && !spec_mode)
  {
#pragma unroll
    for (int k = 0; k < 16; k++) a.dbg[wave * 16 + k] = tsum[k];
    if (wave == 0) a.dbg[16 * 16] = (unsigned long long)rows_done;
  }
#endif
  if (live && a.state_out != NULL && active)
  {
    static_for([&](auto ic) __attribute__((always_inline))
    {
      constexpr int i = decltype(ic)::value;
      if (ln.j0 + i < B) a.state_out[(size_t)n * B + ln.j0 + i] = make_int2(m[i], e[i]);
    }, std::make_integer_sequence<int, C>{});
  }
  if (DEV && live && a.S != NULL && n < a.Np)
  {
    // final rows in the layout of the lane-per-flank kernels (ramx_dev_peek_state): slot q of tile n / 64 holds cells
    // 2q, 2q+1 as (m, e, m, e); the last slot holds cell 2W and (high, pos)
    int *S = reinterpret_cast<int *>(a.S) + ((size_t)(n >> 6) * (W + 1) * 64 + (n & 63)) * 4;
    static_for([&](auto ic) __attribute__((always_inline))
    {
      constexpr int i = decltype(ic)::value;
      const int j = ln.j0 + i;
      if (j < B)
      {
        int *p = S + (size_t)(j >> 1) * 64 * 4 + (j & 1) * 2;
        p[0] = m[i]; p[1] = e[i];
      }
    }, std::make_integer_sequence<int, C>{});
    if (ln.pL0) { int *p = S + (size_t)W * 64 * 4 + 2; p[0] = high; p[1] = pos; }
  }
  if (live && ln.pL0 && (!DEV || n < a.Np)) a.trim[n] = make_int2(thigh, tpos);
  if (threadIdx.x == 0 && (!DEV || wg == 0))
  {
    RamxCtl o;
    o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = respec; o.pad = failed;   // (besta: columns computed twice after a wrong guess, vote-wave modes)
    a.ctl_out[fd.id] = o;
  }
}
