// ramx_kernels_packed.h -- the lane-per-flank persistent kernel on PACKED rows: two cells of a flank's row per register, int16
// relative to a per-flank 32-bit base (device code of libramx; included by ramx_packed.hip only).
//
// What is computed is the reference's column step (ram_extend.c:970-1223 around compute_nw_row, bnw_extend.c:750-1048) exactly
// as in ramx_kernels_resident.h -- one lane = one flank for the whole direction, the device-wide vote of ramx_kernels_vote.h --
// on another representation of the row:
//
//   R[k] = (m[2k], m[2k+1])        m = max(sub, gap) of a cell                                          k = 0 .. W
//   E[k] = (e[2k-1], e[2k])        e = max(sub + go, gap) + ge: the cell's value as a gap predecessor
//
// both as int16 relative to the lane's `pbase`, both in VECTOR REGISTERS (2 (W + 1) of them: 82 at W = 40 where the int32 row
// needed 81 registers plus 162 bytes of LDS per lane, and 162 at W = 80, which therefore runs two waves per SIMD as well).
// Why the values fit: an in-bounds cell of row r lies at most 3W (P + |min score| + |ge|) + |go| + W |ge| below the row's best
// cell (take the best path into the row's best cell, leave it 2W rows earlier, and walk to the cell with substitutions and ONE
// gap: every step of the detour is in bounds because sequence positions only grow along a path and the target is in bounds;
// rows below 3W are bounded absolutely), the best cell moves by at most max(P, |min score|) per row, and the base follows it
// every 16th row once it is more than `rebase` away.  The host admits a scoring system only if all of it fits int16 with the
// intermediates (ramx_pk_plan), and the entry check below refuses rows that do not (the direction then runs on the int32 rows).
//
// Per pair of cells (bnw_extend.c:892-1018 on the transformed state, see ramx_kernels_common.h):
//   sub  = R[k] + S                        two cells per instruction (v_pk_add_i16; S: score pair of the pair's two base classes,
//   sg   = sub + go                        one LDS read of a 256-entry table indexed by a byte of the phase-aligned base word)
//   t    = max3(sg.lo, E'[k+1].lo, C.lo)   cell 2k:   max(sub + go, deletion, insertion)            (E': previous row)
//   C.hi = t + ge                          e[2k]                                                      (C = (e[2k-1], e[2k]))
//   R[k] = max(sub, E'[k+1], C)            both cells
//   E[k] = C
//   t2   = max3(sg.hi, E'[k+1].hi, C.hi)   cell 2k+1
//   C'.lo = t2 + ge                        e[2k+1]: the next pair's chain register
// The insertion chain runs a cell at a time on register halves (v_max3_i16 / v_add_i16 with op_sel), everything else two cells
// per instruction.  FULL rows add the four candidate rows r+1 (chain-free, ramx_kernels_common.h: 4 packed adds + 4 packed maxima
// per pair on (m[2k-1], m[2k]), their common deletion term max e folded in at the end) and the best cell's index (keys
// (m << 16 | 31 - cell % 32), three groups of 32 cells at W = 40); LEAN rows (no lane can contribute more than its cap or set a
// record, ramx_kernels_resident.h) only the row and its best value.
//
// Far end of a flank.  Cells beyond the flank's last base (j > jhi) hold the sentinel in the reference (bnw_extend.c:990-1002).
// Here they need no masked variant of the band: base-word nibbles beyond the flank's end are replaced ONCE, when a word is
// loaded (every eighth column), by class 15, whose scores are -32768 in every table, and every addition saturates.  Then
//   * an out-of-bounds cell's substitution term is pinned far below every in-bounds value;
//   * in-bounds cells never read out-of-bounds ones (cell j of row r reads cells j, j+1 of row r-1 and cell j-1 of row r; the
//     in-bounds cells of a row are 0 .. jhi(r) and jhi(r-1) = jhi(r) + 1);
//   * what an out-of-bounds cell does take is the insertion chain out of cell jhi and deletions out of out-of-bounds cells of
//     the row before, and by induction m[j] <= m[jhi] + ge and e[j] <= e[jhi] + ge for j > jhi: such a cell never holds the
//     row's best (ties go to the lower cell), its e never raises the candidates' deletion term, and its candidate terms are
//     pinned by class 15 again.
// Lanes whose flank has run out completely (jhi < 0: best cell = sentinel) and lanes without a valid candidate cell (jhi < 1)
// are patched after the band, per lane.  The rows written back at the end carry the reference's fill values in those cells.
// What this kernel does NOT take is a row whose LOW cells are out of bounds (the first rows of a flank whose core is shorter
// than the band: their fill value go + (row+1) ge feeds in-bounds neighbours): the host starts it at the first row r0 from
// which every flank has jlo <= 0 and runs the rows before on the int32 kernel (ramx_pk_first_row).
//
// The vote.  One workgroup barrier per column (the vote wave publishes, everybody reads) and an LDS arrival counter instead of
// the second one: a wave adds its four sums to the workgroup's totals when it has them, the LAST arriver sends the ticket.  A
// LEAN wave without leaders has its sums BEFORE the band (its contribution is the sum of its caps, ram_extend.c:1052-1062), so
// an all-LEAN workgroup's ticket leaves at the start of the column and the exchange runs beside the band instead of behind it;
// a workgroup that holds a leader or a FULL wave sends when that wave is done -- the other seven do not wait for it.
#pragma once

#include "ramx_kernels_vote.h"
#include "ramx_pk_api.h"

template <int W>
struct PkCfg
{
  static constexpr int B = 2 * W + 1, NP = W + 1, NW = (B + 8) / 8 + 2, NG = (B + 31) / 32, Q = W + 1;
};

#define PKB_NEG2 ((int)0x80008000u)
#ifndef PKB_GROUP
#define PKB_GROUP 2          // pairs of cells between scheduling barriers
#endif
#ifndef PKB_PT
#define PKB_PT 2             // table rows are looked up this many pairs ahead of their use
#endif

__device__ __forceinline__ int pkb_add_sat(int a, int b) { return pk_i(__builtin_elementwise_add_sat(pk_v(a), pk_v(b))); }      // v_pk_add_i16 .. clamp
__device__ __forceinline__ int pkb_sub_sat(int a, int b) { return pk_i(__builtin_elementwise_sub_sat(pk_v(a), pk_v(b))); }      // v_pk_sub_i16 .. clamp
__device__ __forceinline__ int pkb_max(int a, int b) { return pk_i(__builtin_elementwise_max(pk_v(a), pk_v(b))); }
// dst.hi = sat(a.lo + b.lo), dst.lo kept
__device__ __forceinline__ void pkb_add_to_hi(int &dst, int a, int b) { asm("v_add_i16 %0, %1, %2 op_sel:[0,0,1] clamp" : "+v"(dst) : "v"(a), "v"(b)); }
// d.lo = sat(a.lo + b.lo) (d.hi: whatever the register held)
__device__ __forceinline__ int pkb_add_lo(int a, int b) { int d; asm("v_add_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ int pkb_hmax(int p) { const pk_s2 q = pk_v(p); return q.x > q.y ? (int)q.x : (int)q.y; }
__device__ __forceinline__ unsigned pkb_lds_off(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

// nibbles of base word `kb` (absolute index) beyond the flank's last base become class 15: nibble i of word kb is flank position
// t'' = 8 kb + i, out of bounds iff t'' > bdy + 8
__device__ __forceinline__ unsigned pkb_mask_word(unsigned wv, int kb, int bdy9x4 /* 4 (bdy + 9) */)
{
  int s = bdy9x4 - 32 * kb;                        // 4 x (number of in-bounds nibbles of this word), unclamped
  s = s < 0 ? 0 : s;
  const unsigned m = s >= 32 ? 0u : (0xffffffffu << s);
  return wv | m;
}

// Score tables, indexed by (winner << 8 | class pair byte): the band forms that index in the 16-bit halves of two registers per
// four cell pairs (v_perm_b32 of the phase-aligned base word with the winner), so ONE pair of band instantiations serves the four
// winners -- four instantiations under a switch meet in a block whose every row register is a PHI, and the allocator then keeps
// two rows in flight.
struct PkTabs
{
  int4 t4[4][256];        // FULL rows: the four candidates' score pairs (low class | high class << 16), ROTATED so that .x is the
                          // winner's: (w, w+1, w+2, w+3 mod 4)
  int tp[4][256];         // LEAN rows: the winner's score pair (entries 4 bytes apart: see ramx_kernels_resident.h)
};

template <int BLOCK>
__device__ __forceinline__ void pkb_tabs_init(PkTabs &pt, const int (&tab)[RAMX_NCLASS][4])
{
  for (int i = threadIdx.x; i < 256; i += BLOCK)
  {
    // entry (hi, lo) sits at index hi << 4 | (lo ^ ((hi & 3) << 2)): the LDS bank of a lookup is its index's low bits, and with
    // the plain index two lanes whose low classes agree and whose high classes differ meet in one bank (58 % of the LDS cycles
    // of the first version were bank conflicts); this way the sixteen pairs of A C G T sit in sixteen different banks.  The row
    // forms the same index from the base word: A ^ ((A >> 2) & 0x0c0c0c0c), three instructions per four pairs.
    const int hi = i >> 4, lo = (i & 15) ^ ((hi & 3) << 2);
    int q[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
      const int sl = lo < RAMX_NCLASS ? tab[lo][c] : (lo == 15 ? -32768 : 0);
      const int sh = hi < RAMX_NCLASS ? tab[hi][c] : (hi == 15 ? -32768 : 0);
      q[c] = (sl & 0xffff) | (sh << 16);
      pt.tp[c][i] = q[c];
    }
#pragma unroll
    for (int wv = 0; wv < 4; wv++) pt.t4[wv][i] = make_int4(q[wv], q[(wv + 1) & 3], q[(wv + 2) & 3], q[(wv + 3) & 3]);
  }
}

// (A hand-ordered version of this row -- every pair of cells one asm block in which no instruction follows its producer, the
// candidate terms of the pair before between the links of the insertion chain -- removed the 230 s_nop the compiler puts
// behind packed / op_sel producers (1,180 -> 1,050 instructions per FULL row) and was SLOWER: 7.65 against 7.25 us per column in
// the aligned phase, profiles/r04_notes.md.  With two waves per SIMD the wait states are the other wave's issue slots.)
// One row on the packed representation.  wsel: the winner (wave-uniform; part of the table index).  ph4: 4 x the nibble phase of
// step 0 in w[0].  FULL: kg[] = best cell keys per group of 32 cells, bA[i] = running maximum (packed) of candidate
// (wsel + i) mod 4, maxE = max e (packed); LEAN: best = the row's best value (packed halves).
template <int W, int BLOCK, bool FULL>
__device__ __forceinline__ void pkb_band(const int go2, const int ge2, const PkTabs &pt, const int wsel, const int ph4, const unsigned (&w)[PkCfg<W>::NW],
                                         int (&R)[PkCfg<W>::NP], int (&E)[PkCfg<W>::NP], int &best, int (&kg)[PkCfg<W>::NG], int (&bA)[4], int &maxE)
{
  constexpr int B = PkCfg<W>::B, NP = PkCfg<W>::NP, NG = PkCfg<W>::NG, PT = PKB_PT;
  const char *tb = FULL ? reinterpret_cast<const char *>(&pt.t4[0][0]) : reinterpret_cast<const char *>(&pt.tp[0][0]);
  const unsigned wrep = (unsigned)wsel * 0x01010101u;
  const int hmask = (int)0xffff0000u;
  int C = PKB_NEG2, Rprev = PKB_NEG2;
  best = PKB_NEG2; maxE = PKB_NEG2;
#pragma unroll
  for (int c = 0; c < 4; c++) bA[c] = PKB_NEG2;
#pragma unroll
  for (int g = 0; g < NG; g++) kg[g] = -2147483647 - 1;
  unsigned Alo = 0, Ahi = 0;          // (winner << 8 | class pair byte) of four cell pairs, one per 16-bit half
  int tQ[PT];
  int cQ[FULL ? PT : 1][4];
  auto lookup = [&](auto kc, auto qc) __attribute__((always_inline))
  {
    constexpr int k = decltype(kc)::value, qi = decltype(qc)::value;
    if constexpr ((k & 3) == 0)
    {
      unsigned A = __builtin_amdgcn_alignbit(w[(k >> 2) + 1], w[k >> 2], ph4);
      A ^= (A >> 2) & 0x0c0c0c0cu;                    // bank swizzle of the table index (pkb_tabs_init)
      Alo = __builtin_amdgcn_perm(wrep, A, 0x04010400u);
      Ahi = __builtin_amdgcn_perm(wrep, A, 0x04030402u);
    }
    const unsigned Aw = (k & 2) ? Ahi : Alo;
    if constexpr (FULL)
    {
      const int4 v4 = *reinterpret_cast<const int4 *>(tb + pk_word_shl<(k & 1), 4>(Aw));
      cQ[qi][0] = v4.x; cQ[qi][1] = v4.y; cQ[qi][2] = v4.z; cQ[qi][3] = v4.w;
      tQ[qi] = v4.x;
    }
    else tQ[qi] = *reinterpret_cast<const int *>(tb + pk_word_shl<(k & 1), 2>(Aw));
  };
  static_for([&](auto kc) __attribute__((always_inline)) { lookup(kc, kc); }, std::make_integer_sequence<int, PT>{});
  static_for([&](auto kc) __attribute__((always_inline))
  {
    constexpr int k = decltype(kc)::value;
    if constexpr ((k % PKB_GROUP) == 0)
    {
      // pin the accumulators to their group (a sunk accumulation keeps every table row alive, as in prk_band_fast)
      if constexpr (FULL) asm volatile("" ::"v"(C), "v"(bA[0]), "v"(bA[1]), "v"(bA[2]), "v"(bA[3]), "v"(maxE), "v"(kg[(k > 0 ? 2 * k - 1 : 0) >> 5]));
      else asm volatile("" ::"v"(best), "v"(C));
      __builtin_amdgcn_sched_barrier(0);
    }
    const int S = tQ[0];
    int sc[4] = { 0, 0, 0, 0 };
    if constexpr (FULL) { sc[0] = cQ[0][0]; sc[1] = cQ[0][1]; sc[2] = cQ[0][2]; sc[3] = cQ[0][3]; }
#pragma unroll
    for (int q = 0; q + 1 < PT; q++)
    {
      tQ[q] = tQ[q + 1];
      if constexpr (FULL) { cQ[q][0] = cQ[q + 1][0]; cQ[q][1] = cQ[q + 1][1]; cQ[q][2] = cQ[q + 1][2]; cQ[q][3] = cQ[q + 1][3]; }
    }
    if constexpr (k + PT < NP) lookup(std::integral_constant<int, (k + PT < NP ? k + PT : 0)>{}, std::integral_constant<int, PT - 1>{});
    const int PeP = k + 1 < NP ? E[k + 1 < NP ? k + 1 : 0] : PKB_NEG2;        // (e'[2k+1], e'[2k+2]): cell 2W has no deletion (bnw_extend.c:892)
    const int sub = pkb_add_sat(R[k], S);                       // bnw_extend.c:950-956, two cells
    const int sg = pkb_add_sat(sub, go2);
    const int t = pk_max3_lll(sg, PeP, C);                      // cell 2k: max(sub + go, del, ins), :1007-1018
    pkb_add_to_hi(C, t, ge2);                                   // C = (e[2k-1], e[2k])
    int m = pkb_max(pkb_max(sub, PeP), C);
    E[k] = C;
    if constexpr (FULL) { if constexpr (k > 0) maxE = pkb_max(maxE, C); }     // deletion terms of candidate cells 2k-2, 2k-1 (e[0] is nobody's)
    if constexpr (2 * k + 1 < B)
    {
      const int t2 = pk_max3_hhh(sg, PeP, C);                   // cell 2k+1
      C = pkb_add_lo(t2, ge2);                                  // the next pair's chain register: lo = e[2k+1]
    }
    else m = (m & 0xffff) | (int)0x80000000u;                   // there is no cell 2W+1
    R[k] = m;
    if constexpr (FULL)
    {
      // best cell of the row: keys (value << 16 | 31 - cell % 32), the lowest cell wins ties (bnw_extend.c:1020-1024)
      constexpr int g = (2 * k) >> 5;
      const int klo = (int)(((unsigned)m << 16) | (unsigned)(31 - ((2 * k) & 31)));
      if constexpr (2 * k + 1 < B)
      {
        const int khi = (m & hmask) | (31 - ((2 * k + 1) & 31));
        kg[g] = imax3(kg[g], klo, khi);
      }
      else kg[g] = imax(kg[g], klo);
      // candidate cells 2k-1 and 2k of row r+1: substitution from (m[2k-1], m[2k]) with the classes of steps 2k, 2k+1
      const int ms = (int)__builtin_amdgcn_alignbit((unsigned)m, (unsigned)Rprev, 16);      // k = 0: (-32768, m[0]): there is no cell -1
#pragma unroll
      for (int c = 0; c < 4; c++) bA[c] = pkb_max(bA[c], pkb_add_sat(ms, sc[c]));
      Rprev = m;
    }
    else best = pkb_max(best, m);
  }, std::make_integer_sequence<int, NP>{});
}

// Leaders (ramx_kernels_resident.h, prk_leader_rows): what a LEAN row skipped, for ONE flank of the wave, by all 64 lanes.  The
// leader's lane publishes its new row, its chain registers and its base words; lane i takes cells i, i + 64, ..: candidate cell k
// of row r+1 = m_k + M[a][class of step k+1], deletion terms e_k, five wave maxima, the lowest cell holding the row's best.
template <int W, int BLOCK>
__device__ __forceinline__ void pkb_leader_rows(const PkTabs &pt, int *scr /* [2 NP + NW] of this wave */, const int ph, const int leader,
                                                const int pbase, const int jhi, const int bestF,
                                                const unsigned (&w)[PkCfg<W>::NW], const int (&R)[PkCfg<W>::NP], const int (&E)[PkCfg<W>::NP],
                                                int (&bestA)[4], int &jbest)
{
  constexpr int B = PkCfg<W>::B, NW = PkCfg<W>::NW, NP = PkCfg<W>::NP;
  constexpr int NH = (B + 63) / 64;
  const int lane = threadIdx.x & 63;
  if (lane == leader)
  {
    static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; scr[k] = R[k]; scr[NP + k] = E[k]; },
               std::make_integer_sequence<int, NP>{});
    static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; scr[2 * NP + k] = (int)w[k]; },
               std::make_integer_sequence<int, NW>{});
  }
  __builtin_amdgcn_wave_barrier();          // LDS operations of a wave execute in order; this only pins the compiler's order
  const int bf = __builtin_amdgcn_readlane(bestF, leader);
  const int lbase = __builtin_amdgcn_readlane(pbase, leader);
  const int ljhi = __builtin_amdgcn_readlane(jhi, leader);
  int ta[4] = { NEG, NEG, NEG, NEG }, me = NEG;
  unsigned long long hit[NH];
#pragma unroll
  for (int h = 0; h < NH; h++)
  {
    const int k = lane + 64 * h;
    const bool ok = k < B && k <= ljhi;                 // cell k of row r in bounds
    const int kk = k < B ? k : 0;
    const int mw = scr[kk >> 1];
    const int m = lbase + ((kk & 1) ? (mw >> 16) : (int)(short)mw);
    const int g = kk + 1 + ph;                           // nibble of step k+1 in the base words
    const unsigned cls = ((unsigned)scr[2 * NP + (g >> 3)] >> (4 * (g & 7))) & 15u;
    if (ok && k + 1 <= ljhi)                             // candidate cell k of row r+1 in bounds (bounds move by one per row)
    {
#pragma unroll
      for (int c = 0; c < 4; c++) ta[c] = imax(ta[c], m + (int)(short)pt.tp[c][cls]);
    }
    if (ok && k >= 1)
    {
      const int cw = scr[NP + ((kk + 1) >> 1)];          // E[(k+1)/2] = (e[k] | .) for odd k, (. | e[k]) for even k
      me = imax(me, lbase + ((kk & 1) ? (int)(short)cw : (cw >> 16)));
    }
    hit[h] = __ballot(ok && m == bf);
  }
  const int mx = wave_max_i32_dpp(me);
  int best[4];
#pragma unroll
  for (int c = 0; c < 4; c++) best[c] = imax(wave_max_i32_dpp(ta[c]), mx);
  int jb = 0;                                            // lowest cell on ties (bnw_extend.c:1020-1024)
#pragma unroll
  for (int h = NH - 1; h >= 0; h--)
    if (hit[h]) jb = 64 * h + __builtin_ctzll(hit[h]);
  if (lane == leader)
  {
#pragma unroll
    for (int c = 0; c < 4; c++) bestA[c] = best[c];
    jbest = jb;
  }
}

// The vote wave's part of a column (ram_extend.c:1064-1086, 1194-1223), shared by the workgroups' vote waves and the exchanger:
// the four sums of row r -> winner, new maximum, stop; the stop rule's state lives in `st` (LDS: max_ext in two words, max_row,
// overflow).  Wave-uniform: runs on the scalar unit.  Returns winner | new maximum << 2 | stop << 3 | failure << 4.
template <class A>
__device__ __forceinline__ int pkb_stop_rule(const A &a, const int r, const long long (&v)[4], const int fl, int *st, const int lane)
{
  long long mx = ((long long)__builtin_amdgcn_readfirstlane(st[1]) << 32) | (long long)(unsigned)__builtin_amdgcn_readfirstlane(st[0]);
  int mr = __builtin_amdgcn_readfirstlane(st[2]), ov = __builtin_amdgcn_readfirstlane(st[3]);
  long long cw = 0;
  int bw = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const long long vk = ((long long)__builtin_amdgcn_readfirstlane((int)(v[k] >> 32)) << 32) |
                         (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v[k]);
    if (vk > 2147483647LL || vk < -2147483648LL) ov = 1;
    if (vk > cw) { cw = vk; bw = k; }
  }
  int dist = mr - r;
  dist = dist < 0 ? -dist : dist;
  const int nm = cw >= mx + (long long)dist * a.minimp ? 1 : 0;
  if (nm && !fl) { mr = r; mx = cw; }
  int d2 = r - mr;
  d2 = d2 < 0 ? -d2 : d2;
  const int stp = d2 >= a.when_to_stop ? 1 : 0;
  if (lane == 0) { st[0] = (int)mx; st[1] = (int)(mx >> 32); st[2] = mr; st[3] = ov; }
  return bw | (nm << 2) | (stp << 3) | (fl << 4);
}

#ifdef RAMX_PRK_TIMING
// (-DRAMX_PRK_TIMING_TAIL: only the second half of the columns is accounted)
#ifdef RAMX_PRK_TIMING_TAIL
#define PKB_TICK_ON (2 * r >= a.L)
#else
#define PKB_TICK_ON true
#endif
#define PKB_TICK(k) do { const unsigned long long t_ = wall_clock64(); if (PKB_TICK_ON) tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define PKB_TICK(k) do { } while (0)
#endif

template <int W, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK / 256) void ramx_packed_kernel(const PKArgs a)
{
  constexpr int NP = PkCfg<W>::NP, NW = PkCfg<W>::NW, NG = PkCfg<W>::NG, Q = PkCfg<W>::Q, WPB = BLOCK / 64;
  // 320 threads = four band waves (one per SIMD) and a VOTE WAVE without flanks: with one wave per SIMD all band waves finish
  // their row together, and the poll of the vote (a round trip to the memory fabric, ~0.9 us with the fold and the stop rule)
  // by one of them afterwards was part of every column; the fifth wave has polled and decided by the time they arrive.  (With
  // two waves per SIMD the older wave of SIMD 0 has that time anyway, and a ninth wave has no registers.)
  constexpr bool VW = (BLOCK % 256) != 0;
  constexpr int BW = VW ? WPB - 1 : WPB, VOTER = VW ? BW : 0;
  // rows may run one unconfirmed step ahead of the vote where a second copy of the row fits the registers
  constexpr bool SPEC = W <= 40;
  struct Smem      // tables first: their LDS addresses must fit the 16-bit offset field of the ds_read that uses them
  {
    PkTabs pt;
    long long tot[2][2][4];                            // the workgroup's four sums of the next row: [on the guess | confirmed][column parity]
    int word[2][2];                                    // by column parity: { decision(r): winner | new maximum << 2 | stop << 3 | failure << 4,
                                                       //   guess(r+1): informative << 8 | argmax of the workgroup's totals }
    int cnt[2][2];                                     // waves that have added theirs
    int st[4];                                         // the stop rule's state, kept by the vote wave: max_ext (2 words), max_row, overflow
    int lead[WPB][2 * NP + NW + 1];                    // pkb_leader_rows: a leader's row, chain registers and base words, per wave
    int eb[SPEC ? BLOCK * NP : 1];                     // the chain registers of row r-1 while row r stands on a guess: [thread][pair]
                                                       // (NP is odd: a wave's 64 columns start in different banks)
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * BW + wave;
  const bool live = wave < BW && tile < (a.Np >> 6);
  const int n = (live ? tile : 0) * 64 + lane;
  int4 *S = a.S + (size_t)(live ? tile : 0) * Q * 64 + lane;
  const int2 bd = a.bounds[n];
  // last in-bounds band cell of row r: jhi = bdy - r.  Padding lanes (no flank: an all-N base stream, their vote is masked) are
  // never clipped; an empty flank has no in-bounds cell in any row
  const int bdy = (n >= a.Nx) ? 0x3fffffff : (bd.x > bd.y ? -0x40000000 : bd.y);
#define PKB_BDY9X4(b) ((b) > 0x07ffffff ? 0x3fffffff : ((b) < -0x08000000 ? -0x40000000 : 4 * ((b) + 9)))
  const int shard = blockIdx.x % NSHARD;
  const int go2 = pk_two(a.go), ge2 = pk_two(a.ge);

  pkb_tabs_init<BLOCK>(sm.pt, a.tab);
  if (threadIdx.x < 16) (&sm.tot[0][0][0])[threadIdx.x] = 0;
  if (threadIdx.x < 4) (&sm.cnt[0][0])[threadIdx.x] = 0;
  if (threadIdx.x < 4) (&sm.word[0][0])[threadIdx.x] = 0;

  // ---- stop-rule state and records so far -------------------------------------------------------
  long long max_ext = 0;
  int max_row = -1, rows_done = a.r0, ovf = 0, stopped = 0, failed = 0, carried = 0;
  // the snapshot of the records (ram_extend.c:1203-1207) goes straight to a.trim on every new-maximum column (8 bytes per flank,
  // coalesced): two registers less to carry through the loop
  if (a.r0 > 0)
  {
    // (wave-uniform: through the scalar unit, so that the stop rule's state lives in scalar registers from the start)
    const RamxCtl *cp = a.ctl_in;
    const long long me = cp->max_ext;
    max_ext = ((long long)__builtin_amdgcn_readfirstlane((int)(me >> 32)) << 32) | (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)me);
    max_row = __builtin_amdgcn_readfirstlane(cp->max_row);
    rows_done = __builtin_amdgcn_readfirstlane(cp->rows_done);
    ovf = __builtin_amdgcn_readfirstlane(cp->overflow);
    stopped = (__builtin_amdgcn_readfirstlane(cp->stopped) || rows_done < a.r0) ? 1 : 0;       // the launch before this one stopped (or gave up) early
    const int cpad = __builtin_amdgcn_readfirstlane(cp->pad);
    failed = cpad;
    carried = __builtin_amdgcn_readfirstlane(cp->besta);       // LEAN rows / rows computed twice so far (0 behind the int32 kernel)
  }
  else if (live) a.trim[n] = make_int2(0, 0);
  // (selected under a branch the compiler does not know to be uniform: without this the loop-carried copies live in vector registers)
  stopped = __builtin_amdgcn_readfirstlane(stopped); max_row = __builtin_amdgcn_readfirstlane(max_row);
  rows_done = __builtin_amdgcn_readfirstlane(rows_done); carried = __builtin_amdgcn_readfirstlane(carried);
  const bool skip = stopped || failed || a.r0 >= a.Lseg;
  if (threadIdx.x == 0) { sm.st[0] = (int)max_ext; sm.st[1] = (int)(max_ext >> 32); sm.st[2] = max_row; sm.st[3] = ovf; }

  // ---- the exchanger (multi-rank runs, when the device has a CU to spare): a workgroup WITHOUT flanks behind the last one.
  // Its first wave does what is the device's business rather than a workgroup's -- it waits for the local total of every row,
  // stores it into every rank's box, writes the consensus and clears the vote set three rows ahead -- and nothing else, so
  // the local total crosses the fabric as soon as the last ticket is in.  In workgroup 0 the same duties wait for that
  // workgroup's own row: with rows computed ahead of the vote, everybody's vote then arrives a whole band late
  // (two ranks on one GPU: 8.3 against 5.8 us per column for one rank, profiles/r04_two_rank.log).
  if (blockIdx.x == a.nblocks)
  {
    if (wave != 0 || stopped || failed || a.r0 >= a.Lseg) return;
    for (int r = a.r0; r < a.Lseg; r++)
    {
      PShard *vb = a.vote;
      asm volatile("" : "+s"(vb));
      long long v[4];
      prk_wait_vote(a, vb, a.sums_in, r == a.r0 ? (a.r0 == 0 ? 1 : 2) : 0, r, lane, (a.nblocks - (lane & (NSHARD - 1)) + NSHARD - 1) / NSHARD, failed, v, true);
      const int fl = __any(failed) ? 1 : 0;
      const int dword = pkb_stop_rule(a, r, v, fl, sm.st, lane);
      if (fl) { if (lane == 0) __hip_atomic_fetch_max(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if (lane == 0) a.cons_out[r] = (signed char)(dword & 3);
      if (lane < NSHARD)
      {
        PShard *z = vb + (size_t)((r + 3) & (PRK_NSETS - 1)) * NSHARD + lane;
#pragma unroll
        for (int k = 0; k < 4; k++) __hip_atomic_store(&z->word[k], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((dword & 8) || r == a.L - 1) break;
    }
    return;
  }

  // ---- row state -> packed registers -----------------------------------------------------------
  int R[NP], E[NP];
  int high, pos, pbase = 0;
  {
    // cells above jcut (beyond the flank's end in row r0 - 1) are never read by an in-bounds cell of a later row: pinned at once
    const int jcut = bdy - (a.r0 - 1);
    int bmax = -2147483647 - 1, bmin = 2147483647;
    // each pass over the row walks its own opaque copy of the pointer: 41 slot addresses shared between the two passes here and
    // the write-back at the end would stay alive through the whole kernel (82 registers)
    const int4 *S1 = S;
    asm volatile("" : "+v"(S1));
    // (every access to R / E / w is through compile-time indices: an array indexed by an unrolled loop variable is turned into
    // one wide vector register by the alloca promotion that runs before unrolling, and every branch then copies it whole)
    static_for([&](auto qc) __attribute__((always_inline))
    {
      constexpr int q = decltype(qc)::value;
      // a few slots in flight at a time: the whole row in flight would be the register peak of the kernel
      if constexpr ((q & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const int4 v = S1[(size_t)q * 64];
      const int x0 = 2 * q <= jcut ? v.x : -2147483647 - 1, n0 = 2 * q <= jcut ? v.x : 2147483647;
      bmax = imax(bmax, x0); bmin = n0 < bmin ? n0 : bmin;
      if constexpr (q < W)
      {
        const int x1 = 2 * q + 1 <= jcut ? v.z : -2147483647 - 1, n1 = 2 * q + 1 <= jcut ? v.z : 2147483647;
        bmax = imax(bmax, x1); bmin = n1 < bmin ? n1 : bmin;
      }
      // (both running values here and now: left alone, the minimum is computed after all the maxima, from 2 Q comparison masks
      // kept in scalar registers -- the peak of the kernel's scalar spills, which cost a vector register per 64)
      asm volatile("" : "+v"(bmax), "+v"(bmin));
    }, std::make_integer_sequence<int, Q>{});
    __builtin_amdgcn_sched_barrier(0);
    const bool none = bmax == -2147483647 - 1;
    pbase = none ? 0 : bmax;
    // entry check: every in-bounds cell within the span the host computed (else this representation is not safe: give up loudly)
    if (live && !skip && n < a.Nx && !none && (long long)bmax - (long long)bmin > (long long)a.spread)
      __hip_atomic_fetch_max(a.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int jcut2 = jcut;
    const int4 *S2 = S;
    asm volatile("" : "+v"(jcut2), "+v"(S2));                     // second pass: its own comparisons (the first pass's would be kept in scalar registers)
    auto rel16 = [&](int x, bool inb) __attribute__((always_inline)) -> int
    {
      int d = x - pbase;                                // in-bounds cells: within `spread` of the base; others are pinned below
      d = d < -32768 ? -32768 : (d > 32767 ? 32767 : d);
      return inb ? d : -32768;
    };
    int eprev = -32768;                                 // e[2k-1] relative
    static_for([&](auto qc) __attribute__((always_inline))
    {
      constexpr int q = decltype(qc)::value;
      if constexpr ((q & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const int4 v = S2[(size_t)q * 64];
      const bool in0 = 2 * q <= jcut2, in1 = q < W && 2 * q + 1 <= jcut2;
      const int m0 = rel16(v.x, in0), e0 = rel16(v.y, in0);
      const int m1 = q < W ? rel16(v.z, in1) : -32768, e1 = q < W ? rel16(v.w, in1) : -32768;
      R[q] = (m0 & 0xffff) | (m1 << 16);
      E[q] = (eprev & 0xffff) | (e0 << 16);
      eprev = e1;
      if constexpr (q == W) { high = v.z; pos = v.w; }
    }, std::make_integer_sequence<int, Q>{});
    __builtin_amdgcn_sched_barrier(0);
  }
  if (skip)
  {
    // nothing to do: the control block passes through unchanged (the rows in HBM are the ones of the launch before)
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
      RamxCtl o;
      o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = carried; o.pad = failed;
      *a.ctl_out = o;
    }
    return;
  }

  // ---- base words: carried across columns, one new word every eighth column, far-end nibbles masked when a word comes in ----
  unsigned w[NW], wnext;
  {
    const int k0 = (a.r0 + 8) >> 3;
    const unsigned *bp = a.bases + (size_t)k0 * a.Np + n;
    static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; w[k] = pkb_mask_word(bp[(size_t)k * a.Np], k0 + k, PKB_BDY9X4(bdy)); },
               std::make_integer_sequence<int, NW>{});
    wnext = bp[(size_t)NW * a.Np];
  }
  __syncthreads();

  int prevBest = 0x3fffffff;             // best cell of the previous row (LEAN test): unknown before the first band of this launch
  int full_rows = 0, lean_rows = 0, wrong_rows = 0;      // wave 0 of workgroup 0: rows by variant, rows computed twice (reported)
#ifdef RAMX_PRK_TIMING
  unsigned long long tsum[6] = { 0, 0, 0, 0, 0, 0 }, tlast = wall_clock64();
#endif
  // A column's vote needs two trips through the memory fabric (~2.5 us between the last ticket and the moment every workgroup
  // has seen the sums).  A workgroup therefore does not wait for it when its OWN sums of row r say something about the winner
  // (they differ): every wave computes row r against the argmax of the workgroup's sums (`guess`), derives records and
  // contributions, and only then does the vote wave look at the device's vote -- which has been travelling meanwhile.  Guess
  // right (nearly always while the flanks align, and always in a workgroup that holds the leader flank behind the end of the
  // alignment): the workgroup's ticket for row r+1 leaves at once.  Guess wrong: the saved row r-1 and records come back, row r is
  // computed again with the winner, and the sums are sent then.  Nothing computed from an unconfirmed guess leaves the
  // workgroup, and a row is only ever one unconfirmed step ahead, so results cannot differ from the lock-step order.  A
  // workgroup whose sums are all equal (all its flanks at their caps) knows nothing about the winner: it waits for the vote
  // as before, but its LEAN waves have sent their sums before the band, so the exchange runs beside its band as well.
  int spec = 0, guess = 0;               // this column: compute on the workgroup's own argmax before the vote is known
  for (int r = a.r0; r < a.Lseg; r++)
  {
    PKB_TICK(5);
    // (the per-thread addresses into the vote sets would live across the whole kernel and find their home in scratch: an opaque
    // scalar copy of the base pointer makes every column recompute them, ramx_kernels_resident.h round 3)
    PShard *vb = a.vote;
    asm volatile("" : "+s"(vb));
    if (((r + 8) & 7) == 0 && r > a.r0)
    {
      static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; w[k] = w[k + 1]; }, std::make_integer_sequence<int, NW - 1>{});
      const int kn = ((r + 8) >> 3) + NW;
      int nn = n, bb = bdy;
      asm volatile("" : "+v"(nn), "+v"(bb));           // (address and mask are formed here, not carried through the loop)
      w[NW - 1] = pkb_mask_word(wnext, kn - 1, PKB_BDY9X4(bb));      // the word loaded eight columns ago (unmasked until now: nobody waited for it)
      wnext = a.bases[(size_t)kn * a.Np + nn];
    }
    const int par = r & 1, npar = (r + 1) & 1;
    // the last column of a launch that is not the last of the direction (the host runs a long direction in segments, packing the
    // base words of the next one meanwhile): its sums go to a.sums_next instead of the vote sets
    const bool hand_over = r == a.Lseg - 1 && a.Lseg < a.L;
    int besta = 0, next_guess = 0;
    long long dgw = 0;
    bool new_max = false, last_col = false;

    // ---- the vote of row r: the vote wave waits for it and publishes; everybody applies winner and stop rule ------------------
    // The vote wave alone folds the vote and runs the stop rule (every workgroup's comes to the same result: block 0's goes to
    // ctl_out); what the other waves need of it is ONE word: winner, new maximum, stop, failure.  It sits next to the guess the
    // last arriver of this column leaves for the next one (sm.word[par] = { decision(r), guess(r+1) }: one LDS read per column).
    auto decide = [&]() __attribute__((always_inline)) -> bool
    {
      if (wave == VOTER)
      {
        long long v[4];
        int lv = lane;
        asm volatile("" : "+v"(lv));                   // (per-lane addresses of the poll are formed here, not carried through the loop)
        prk_wait_vote(a, vb, a.sums_in, r == a.r0 ? (a.r0 == 0 ? 1 : 2) : 0, r, lv, (a.nblocks - (lv & (NSHARD - 1)) + NSHARD - 1) / NSHARD, failed, v, blockIdx.x == a.xblock);
        const int fl = __any(failed) ? 1 : 0;
        // the stop rule's state lives in LDS between columns (only this wave touches it): in registers it would be carried by
        // every wave of the workgroup
        const int dword = pkb_stop_rule(a, r, v, fl, sm.st, lane);
        const int bw = dword & 3;
        if (lane == 0)
        {
          sm.word[par][0] = dword;
          if (fl) __hip_atomic_fetch_max(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x == a.xblock && !fl)
        {
          if (lane == 0) a.cons_out[r] = (signed char)bw;
          // block 0 clears the vote set of row r+3 (ramx_kernels_vote.h)
          if (lane < NSHARD)
          {
            PShard *z = vb + (size_t)((r + 3) & (PRK_NSETS - 1)) * NSHARD + lane;
#pragma unroll
            for (int k = 0; k < 4; k++) __hip_atomic_store(&z->word[k], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      PKB_TICK(0);                 // wave 0: vote seen, decision made (other waves: nothing)
      // (s_waitcnt lgkmcnt(0) + s_barrier, not __syncthreads(): that would also wait for the base word and the snapshot in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PKB_TICK(1);                 // released by the block barrier
      int dw, gw;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dgw) : "v"(pkb_lds_off(&sm.word[par][0])) : "memory");
      dw = __builtin_amdgcn_readfirstlane((int)dgw); gw = __builtin_amdgcn_readfirstlane((int)(dgw >> 32));
      next_guess = gw;
      if (dw & 16) { failed = 1; return false; }
      besta = dw & 3;
      new_max = (dw & 4) != 0;
      stopped = (dw >> 3) & 1;
      rows_done = r + 1;
      last_col = stopped || r == a.L - 1;      // the vote of row r+1 will not be consumed
      return true;
    };

    // ---- a wave's four sums of row r+1 join the workgroup's.  Two sets of totals: set 1 takes sums computed from the
    // CONFIRMED row r (its last arriver sends the ticket), set 0 those computed on the guess (sent by the vote wave once the
    // vote has confirmed it).  The last arriver also leaves the next column's guess: the argmax of the totals (same rule as the
    // vote, ram_extend.c:1081-1085), and whether they say anything at all.
    auto arrive = [&](const int set, const long long t0, const long long t1, const long long t2, const long long t3) __attribute__((always_inline))
    {
      const long long mine = lane == 0 ? t0 : lane == 1 ? t1 : lane == 2 ? t2 : t3;
      int old = 0;
      if (lane < 4) asm volatile("ds_add_u64 %0, %1" : : "v"(pkb_lds_off(&sm.tot[set][npar][lane])), "v"(mine) : "memory");
      if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(pkb_lds_off(&sm.cnt[set][npar])), "v"(1) : "memory");
      old = __builtin_amdgcn_readfirstlane(old);
      if (old == WPB - 1)
      {
        long long t = 0;
        if (lane < 4)
        {
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(pkb_lds_off(&sm.tot[set][npar][lane])) : "memory");
          if (set == 1)
          {
            asm volatile("ds_write_b64 %0, %1" : : "v"(pkb_lds_off(&sm.tot[set][npar][lane])), "v"(0LL) : "memory");
            if (hand_over)     // the launch that continues with row r+1 takes the sums as plain words
            {
              int lq = lane;
              asm volatile("" : "+v"(lq));             // (once per launch: the address is not worth two registers through the loop)
              __hip_atomic_fetch_add(&a.sums_next[shard * 4 + lq], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
              PShard *sh = vb + (size_t)((r + 1) & (PRK_NSETS - 1)) * NSHARD + shard;
              __hip_atomic_fetch_add(&sh->word[lane], (unsigned long long)t + PRK_BIAS + PRK_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if (set == 1 && lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(pkb_lds_off(&sm.cnt[set][npar])), "v"(0) : "memory");
        if constexpr (SPEC)
        {
          const long long q0 = prk_readlane_u64((unsigned long long)t, 0), q1 = prk_readlane_u64((unsigned long long)t, 1);
          const long long q2 = prk_readlane_u64((unsigned long long)t, 2), q3 = prk_readlane_u64((unsigned long long)t, 3);
          long long c = 0;
          int g = 0;
          if (q0 > c) { c = q0; g = 0; }
          if (q1 > c) { c = q1; g = 1; }
          if (q2 > c) { c = q2; g = 2; }
          if (q3 > c) { c = q3; g = 3; }
          const int informative = !(q0 == q1 && q1 == q2 && q2 == q3) && a.spec_on;
          if (lane == 0) sm.word[par][1] = (informative << 8) | g;
        }
      }
    };

    // ---- row r: on the guess first (if this workgroup has one), on the winner otherwise / afterwards -----------------------
    int Rb[SPEC ? NP : 1], high_b = 0, pos_b = 0, prev_b = 0, pbase_b = 0;       // row r-1: m in registers, e in LDS (sm.eb)
    int *myEb = sm.eb + (SPEC ? threadIdx.x * NP : 0);
    bool pass_spec = false;
    int wsel = 0;
    if (spec)
    {
      if constexpr (SPEC)
      {
#ifndef PKB_PROBE_NO_BACKUP      // timing probe (wrong results after a wrong guess)
        if (live)         // (a wave without flanks -- the vote wave of the 320-thread shape, padding tiles -- has nothing to keep)
        static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; Rb[k] = R[k]; myEb[k] = E[k]; }, std::make_integer_sequence<int, NP>{});
#else
        Rb[0] = R[0];
#endif
        high_b = high; pos_b = pos; prev_b = prevBest; pbase_b = pbase;
      }
      pass_spec = true;
      wsel = guess;
      if (a.test_wrong_every > 0 && (r % a.test_wrong_every) == 0) wsel = (guess + 1) & 3;      // test hook: a deliberately wrong guess
    }
    else
    {
      if (!decide()) break;
      wsel = besta;
    }
    bool confirmed_arrival = false;        // this wave's sums of row r+1 (from the confirmed row r) have joined set 1
    bool gave_up = false;                  // (wave-uniform, unlike `failed`, which the vote wave's lanes set one by one)
    for (;;)
    {
      const int set = pass_spec ? 0 : 1;
      // ---- LEAN?  (ramx_kernels_resident.h: no lane can contribute more than its cap to the vote of row r+1 or set a record in
      // row r.)  prevBest = best cell of row r-1, high = record after row r-1.  Up to leader_max lanes that fail the test
      // (`leaders`) do not keep the wave from the LEAN row: pkb_leader_rows computes what it skipped for them.
      const int jhi = bdy - r;
      const int capfloor = (high + a.cap) > 0 ? (high + a.cap) : 0;
      bool lean = false, early = false;
      unsigned long long leaders = 0;
      if (live && a.lean_p >= 0)
      {
        const unsigned long long keep = __ballot((n < a.Nx) && !((prevBest + 2 * a.lean_p <= capfloor) && (prevBest + a.lean_p <= high)));
        if (keep == 0) { lean = true; early = true; }
        else if (__popcll(keep) <= a.leader_max) { lean = true; leaders = keep; }
      }
      if (!live) early = true;
      const bool send = pass_spec || !last_col;
      if (early && send)
      {
        // this wave's contribution to row r+1 is the sum of its caps whatever the row holds (ram_extend.c:1042, 1052-1062)
        const long long t = wave_sum_nonneg31((live && n < a.Nx) ? capfloor : 0);
        arrive(set, t, t, t, t);
      }
      PKB_TICK(3);                 // early arrival
      // a wave whose sums come after its row (FULL, or LEAN with leaders) is what the workgroup's ticket waits for: it goes first
      // on its SIMD (behind the end of the alignment that is the one wave holding the leader flank, and the whole device waits
      // for its workgroup)
      if (__builtin_amdgcn_readfirstlane(early ? 1 : 0) == 0) __builtin_amdgcn_s_setprio(2);     // (a scalar branch: under an exec mask it would run in every wave)

      // ---- the band ----------------------------------------------------------------------------
      int contrib[4] = { 0, 0, 0, 0 };       // each in [0, 2^31): clamped at 0 below, capped from below by high + cap
      if (live)
      {
        const int ph4 = 4 * ((r + 8) & 7);
        int best, kg[NG], bA[4], maxE;
        if (lean) pkb_band<W, BLOCK, false>(go2, ge2, sm.pt, wsel, ph4, w, R, E, best, kg, bA, maxE);
        else pkb_band<W, BLOCK, true>(go2, ge2, sm.pt, wsel, ph4, w, R, E, best, kg, bA, maxE);
        int rel, jbest = 0, bestA[4] = { NEG, NEG, NEG, NEG };
        if (lean) rel = pkb_hmax(best);
        else
        {
          // best cell: highest value, lowest group on ties (inside a group the key already prefers the lowest cell)
          int bkey = kg[NG - 1], bg = NG - 1;
          rel = kg[NG - 1] >> 16;
#pragma unroll
          for (int g = NG - 2; g >= 0; g--)
          {
            const int v = kg[g] >> 16;
            const bool take = v >= rel;
            rel = take ? v : rel;
            bkey = take ? kg[g] : bkey;
            bg = take ? g : bg;
          }
          jbest = 32 * bg + 31 - (bkey & 31);
          const int mE = pkb_hmax(maxE);
          int rot[4];
#pragma unroll
          for (int i = 0; i < 4; i++) rot[i] = jhi < 1 ? NEG : pbase + imax(pkb_hmax(bA[i]), mE);
          // bA[i] belongs to candidate (wsel + i) mod 4
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            const int i = (c - wsel) & 3;
            bestA[c] = i == 0 ? rot[0] : i == 1 ? rot[1] : i == 2 ? rot[2] : rot[3];
          }
        }
        const int bestF = jhi < 0 ? SENT : pbase + rel;      // a flank that has run out: every cell holds the sentinel (bnw_extend.c:990-1002)
        for (unsigned long long rest = leaders; rest != 0; rest &= rest - 1)
          pkb_leader_rows<W, BLOCK>(sm.pt, sm.lead[wave], (r + 8) & 7, __builtin_ctzll(rest), pbase, jhi, bestF, w, R, E, bestA, jbest);
        // Every 64th row the span the host computed for this scoring system (ramx_pk_plan: how far an in-bounds cell can lie below
        // its row's best cell) is CHECKED against the row, not only assumed: the arithmetic saturates, so a cell outside it would
        // not wrap but silently stick.  A row outside the span raises error word 2 (3: the rows the launch was handed; the highest word stays); the host repeats the
        // direction on the per-column route (3 instructions per row on average; a violation is a property of the scoring system
        // and the data, not of one row: it does not go away before the next look).
        if ((r & 63) == 63)
        {
          int mn = 0x7fff7fff, jj = jhi;
          asm volatile("" : "+v"(jj));
          static_for([&](auto kc) __attribute__((always_inline))
          {
            constexpr int k = decltype(kc)::value;
            int v = R[k];
            const bool in_hi = 2 * k + 1 < PkCfg<W>::B && 2 * k + 1 <= jj;       // (cell 2W+1 does not exist; cells beyond the flank's end do not count)
            v = in_hi ? v : ((v & 0xffff) | 0x7fff0000);
            v = 2 * k <= jj ? v : 0x7fff7fff;
            mn = pk_i(__builtin_elementwise_min(pk_v(mn), pk_v(v)));
          }, std::make_integer_sequence<int, NP>{});
          const pk_s2 q = pk_v(mn);
          const int lowest = q.x < q.y ? (int)q.x : (int)q.y;
          if (n < a.Nx && jhi >= 0 && rel - lowest > a.spread_rows) __hip_atomic_fetch_max(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the base follows the row's best cell (every 16th row, when some lane's has moved far enough)
        if ((r & 15) == 15 && __any(jhi >= 0 && (rel > a.rebase || rel < -a.rebase)))
        {
          const int mv = jhi >= 0 ? rel : 0;
          const int bb = pk_two(mv);
          static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; R[k] = pkb_sub_sat(R[k], bb); E[k] = pkb_sub_sat(E[k], bb); },
                     std::make_integer_sequence<int, NP>{});
          pbase += mv;
        }
        prevBest = bestF;
        if (bestF > high) { high = bestF; pos = r + jbest - W; }   // ram_extend.c:1140-1150
        if (n < a.Nx)
        {
          const int capv = high + a.cap;
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            const int b = bestA[c] < 0 ? 0 : bestA[c];
            contrib[c] = (b >= capv) ? b : capv;
          }
        }
        if (!pass_spec) { if (lean) lean_rows++; else full_rows++; }
      }
      PKB_TICK(2);                 // band done
      if (!early && send)
      {
#ifndef PKB_PROBE_NO_REDUCE      // timing probe (wrong results)
        const long long t0 = wave_sum_nonneg31(contrib[0]), t1 = wave_sum_nonneg31(contrib[1]);
        const long long t2 = wave_sum_nonneg31(contrib[2]), t3 = wave_sum_nonneg31(contrib[3]);
#else
        const long long t0 = __builtin_amdgcn_readfirstlane(contrib[0]), t1 = __builtin_amdgcn_readfirstlane(contrib[1]);
        const long long t2 = __builtin_amdgcn_readfirstlane(contrib[2]), t3 = __builtin_amdgcn_readfirstlane(contrib[3]);
#endif
        arrive(set, t0, t1, t2, t3);
      }
      PKB_TICK(4);                 // late arrival
      __builtin_amdgcn_s_setprio(0);
      if (!pass_spec) { confirmed_arrival = send; break; }
      // ---- the row was computed on the guess: what does the vote say? -------------------------------------------------
      if (!decide()) { gave_up = true; break; }
      const bool right = besta == wsel;
      if (wave == VOTER)
      {
        // the speculative totals are complete (every wave arrived before the barrier): send them if they stand, clear them
        if (lane < 4)
        {
          long long t;
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(pkb_lds_off(&sm.tot[0][npar][lane])) : "memory");
          asm volatile("ds_write_b64 %0, %1" : : "v"(pkb_lds_off(&sm.tot[0][npar][lane])), "v"(0LL) : "memory");
          if (right && !last_col)
          {
            if (hand_over)
            {
              int lq = lane;
              asm volatile("" : "+v"(lq));
              __hip_atomic_fetch_add(&a.sums_next[shard * 4 + lq], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
              PShard *sh = vb + (size_t)((r + 1) & (PRK_NSETS - 1)) * NSHARD + shard;
              __hip_atomic_fetch_add(&sh->word[lane], (unsigned long long)t + PRK_BIAS + PRK_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(pkb_lds_off(&sm.cnt[0][npar])), "v"(0) : "memory");
      }
      if (right) break;
      // guessed wrong: back to row r-1 and the records behind it
      if constexpr (SPEC)
      {
        if (live)
        static_for([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; R[k] = Rb[k]; E[k] = myEb[k]; }, std::make_integer_sequence<int, NP>{});
        high = high_b; pos = pos_b; prevBest = prev_b; pbase = pbase_b;
      }
      wrong_rows++;
      pass_spec = false;
      wsel = besta;
    }
    if (gave_up) break;
#ifndef PKB_PROBE_NO_TRIM
    if (new_max && live)                                        // ram_extend.c:1203-1207 (the confirmed records)
#else
    if (new_max && live && r < 0)
#endif
    {
      int nn = n;
      asm volatile("" : "+v"(nn));
      a.trim[nn] = make_int2(high, pos);
    }
    if (last_col || hand_over) break;
    // next column: the guess its totals gave.  After a confirmed round the last arriver wrote it just now: one more barrier makes
    // it (and the cleared totals) everybody's; after a guess that stood, the barrier of decide() has done so already
    if constexpr (SPEC)
    {
      if (confirmed_arrival)
      {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int gw;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(gw) : "v"(pkb_lds_off(&sm.word[par][1])) : "memory");
        next_guess = __builtin_amdgcn_readfirstlane(gw);
      }
      spec = next_guess >> 8; guess = next_guess & 3;
    }
  }
#ifdef RAMX_PRK_TIMING
  if (a.dbg != NULL && (threadIdx.x & 63) == 0)
  {
#pragma unroll
    for (int k = 0; k < 6; k++) a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + k] += tsum[k];      // (+=: a direction is several launches)
    a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + 6] += (unsigned long long)full_rows;
    a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + 7] += (unsigned long long)lean_rows;
  }
#endif

  // ---- write back: rows (so that the device state can be inspected / resumed), trim, control ----
  if (live && rows_done > a.r0)
  {
    // the rows are those of row rl = rows_done - 1; cells beyond the flank's end carry the reference's fill (bnw_extend.c:990-1002)
    int bz = bdy;
    asm volatile("" : "+v"(bz));
    const int rl = rows_done - 1, jh = bz - rl;
    const int edge = rl < W ? a.go + (rl + 1) * a.ge : SENT;
    // (recomputed from an opaque copy of the lane index: the pointer would otherwise stay in two registers through the whole loop)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int4 *S3 = a.S + (size_t)(blockIdx.x * BW + (tid >> 6)) * Q * 64 + (tid & 63);
    auto val = [&](int half, int j, bool is_e) __attribute__((always_inline)) -> int
    {
      if (j > jh) { const int f = j < W ? edge : SENT; return is_e ? f + a.ge : f; }
      return pbase + half;
    };
    static_for([&](auto qc) __attribute__((always_inline))
    {
      constexpr int q = decltype(qc)::value;
      const int m0 = val((int)(short)R[q], 2 * q, false), e0 = val(E[q] >> 16, 2 * q, true);
      if constexpr (q < W)
      {
        const int m1 = val(R[q] >> 16, 2 * q + 1, false), e1 = val((int)(short)E[q + 1 < NP ? q + 1 : 0], 2 * q + 1, true);
        S3[(size_t)q * 64] = make_int4(m0, e0, m1, e1);
      }
      else S3[(size_t)q * 64] = make_int4(m0, e0, high, pos);
    }, std::make_integer_sequence<int, Q>{});
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = ((long long)sm.st[1] << 32) | (long long)(unsigned)sm.st[0]; o.max_row = sm.st[2]; o.stopped = stopped; o.rows_done = rows_done;
    o.overflow = sm.st[3];
    o.besta = ((carried & 0xffff) + lean_rows) | ((((carried >> 16) & 0xffff) + wrong_rows) << 16);      // reported: LEAN rows of the first wave, rows it computed twice after a wrong guess
    o.pad = failed;
    *a.ctl_out = o;
  }
}
