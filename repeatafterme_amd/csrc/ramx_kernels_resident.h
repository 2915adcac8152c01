// ramx_kernels_resident.h -- register-resident kernels: the device-wide persistent kernel and the one-workgroup-per-family kernel
// (device code of libramx; included by ramx_device.hip only -- one translation unit so that everything inlines)
#pragma once

#include "ramx_kernels_common.h"
#include "ramx_kernels_vote.h"

// ------------------------------------------------------------------------------------------
// persistent kernel: the whole direction in ONE launch, DP rows resident in registers
// ------------------------------------------------------------------------------------------
//
// For W known at compile time and N <= (resident waves) x 64 the row of a flank (2 x (2W+1) int32) fits the
// lane's registers, so the row never travels: HBM sees only the base words (~12 words per flank per column) and
// the 32-byte vote.  All L columns run inside ONE launch; the dependent-launch boundary of the
// streaming kernel becomes a device-wide barrier that is fused with the vote (protocol, vote shards and the cross-device
// mailboxes: ramx_kernels_vote.h, shared with the packed-row kernel of ramx_kernels_packed.h, which serves the flank sets
// and scoring systems whose rows fit int16 -- this kernel keeps the others, and the first rows of flanks whose cores are
// shorter than the band).

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
  PeerBox *mirror;              // host-memory boxes only: a device-memory copy of `box` that block 0 keeps current, so that
                                // ONE workgroup polls across PCIe and the others poll locally (NULL: everybody polls `box`)
  int rank, nranks;
  int Np, Nx, r0, L, go, ge, cap, minimp, when_to_stop, nblocks;
  int tab[RAMX_NCLASS][4];
  int pack_ok;                  // every reachable score fits 27 bits: the fast path may pack (score, cell) keys
  int lean_p;                   // P = max(0, largest matrix entry) of the LEAN test (prk_band_fast); -1: never LEAN
  int leader_max;               // a wave with at most this many lanes that fail the LEAN test runs LEAN + prk_leader_rows (0: off)
  long long *sums_next;         // != NULL: this launch runs the first L rows of a longer direction -- the sums of row L go here as
                                // NSHARD x 4 plain int64 words (zeroed by the host) for the launch that continues
  unsigned long long *dbg;      // -DRAMX_PRK_TIMING builds only: [block][8] phase sums in 10 ns ticks
};

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
// cells between scheduling barriers of the fast bands, and the LEAN band's look-ahead: two waves per SIMD (512-thread
// workgroups) want short groups -- 8 -> 4 cells with the look-ahead at 3: 7.24 -> 7.00 us per column at N = 100,000, groups of 1 or
// 2 are not better (profiles/r03_ab_sweep*.log) -- a lone wave per SIMD wants the long ones (N = 65,000: 6.2 against 6.9 us,
// profiles/r03_ab_other_shapes.log)
#ifndef PRK_FAST_GROUP
#define PRK_FAST_GROUP(BLOCK) ((BLOCK) >= 512 ? 4 : 8)
#endif
#ifndef PRK_FETCH_AHEAD
#define PRK_FETCH_AHEAD 2
#endif
#ifndef PRK_FETCH_AHEAD_LEAN
#define PRK_FETCH_AHEAD_LEAN(BLOCK) ((BLOCK) >= 512 ? 3 : 2)        // the LEAN band has registers to spare
#endif
__device__ __forceinline__ int vmax3(int x, int y, int z)   // forced v_max3_i32 (keeps the compiler from re-associating)
{
  int d;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
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

template <int HALF>
__device__ __forceinline__ int add_sext_word(int x, int packed)   // x + (int)(short)(packed >> 16*HALF)
{
  int d;
  if (HALF == 0) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(x), "v"(packed));
  else asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(x), "v"(packed));
  return d;
}
__device__ __forceinline__ int pack_halves(int lo, int hi)         // (lo & 0xffff) | (hi << 16)
{
  int d;
  asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "v"(lo), "s"(0x05040100));
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
//
// MASKHI: the same band for a wave in which some flank has run out at its far end (cells j > jhi out of bounds, every
// cell in bounds on the near side, row >= W so that every fill is the sentinel, bnw_extend.c:990-1002).  Out-of-bounds
// cells store (SENT, SENT + ge) exactly as the general band does; their candidate terms only have to stay negative
// (the vote clamps at 0, ram_extend.c:1042) and their keys below every positive score (the best cell only matters when
// it beats high >= 0): five extra VALU per cell instead of the general band's separate formulation.
//
// LEAN: the same row update WITHOUT the four candidate rows and without the best cell's index, for a wave whose lanes can
// neither contribute anything but their cap to the next vote nor set a record in this row.  With go, ge <= 0 every cell of a
// row is at most P = max(0, largest matrix entry) above the best cell of the row before, so with prevBest = best cell of row
// r-1:  best cell of row r <= prevBest + P,  best cell of any candidate row r+1 <= prevBest + 2P.  If
//     prevBest + 2P <= max(0, high + CAPPENALTY)   and   prevBest + P <= high      for every lane of the wave,
// then max(0, candidate best) never exceeds max(0, high + cap) -- the contribution is max(0, high + cap) whatever the candidate
// rows hold (ram_extend.c:1042, 1052-1062) -- and `best > high` (:1140) is false: only the row itself (and its best VALUE,
// for the next row's test) has to be computed.  That is every column behind the end of the alignment, where flanks sit at
// their cap: ~9 instead of ~16.5 VALU per cell.  Exact, not an approximation: the skipped values cannot influence any output.
template <int W, int BLOCK, bool MASKHI, bool LEAN = false>
__device__ __forceinline__ void prk_band_fast(const int go_, const int ge_, const FastTabs &ft, short *sD, const int r, const int jhi,
                                              const unsigned (&w)[(2 * W + 1 + 8) / 8 + 2], int (&M)[2 * W + 1], LaneDP &D)
{
#ifdef RAMX_VGPR_GOGE
  // v_add_u32 with two VGPR operands issues at twice the rate of one with an SGPR operand on gfx950
  // (tools/microbench/valu_rate.hip); the empty asm keeps the copies in vector registers
  int go = go_, ge = ge_;
  asm volatile("" : "+v"(go), "+v"(ge));
#else
  const int go = go_, ge = ge_;
#endif
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
  constexpr int PD = LEAN ? PRK_FETCH_AHEAD_LEAN(BLOCK) : PRK_FETCH_AHEAD;
  unsigned A = 0, Alo = 0;
  int2 rowQ[PD];                                    // {candidate bytes, M[besta][base]} of steps j .. j+PD-1
  // e - m of the previous row, TWO cells per LDS instruction: a lane's dword of cell pair p holds d[2p] | d[2p+1] << 16
  // (the layout the int16 accesses of the other bands use too).  dCurW: pair of the cell whose d this step needs (cell j+1),
  // dNxtW: the pair after it, loaded one pair ahead of its first use; the new row's d is written a pair at a time as well.
  int dCurW = 0, dNxtW = 0, dLo = 0;
  int *myDW = reinterpret_cast<int *>(sD) + threadIdx.x;
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
#ifdef PRK_PROBE_NO_TABLE
    return make_int2((int)off, (int)off >> 3);      // timing probe: no LDS lookup (wrong results)
#else
    if constexpr (LEAN) return make_int2(0, *reinterpret_cast<const int *>(tb + off + 4));    // the winner's score only
    else return *reinterpret_cast<const int2 *>(tb + off);
#endif
  };
  static_for([&](auto kc) __attribute__((always_inline))
  {
    constexpr int k = decltype(kc)::value;
    if constexpr (k <= B) rowQ[k] = fetch_row(std::integral_constant<int, (k <= B ? k : 0)>{});
    else rowQ[k] = make_int2(0, 0);
  }, std::make_integer_sequence<int, PD>{});
#ifdef PRK_PROBE_NO_DROW
#define PRK_DLOAD(x) (r + (x))                       /* timing probe: the d-row stays out of LDS (wrong results) */
#define PRK_DSTORE(p, v) asm volatile("" :: "v"((int)(v)))
#else
#define PRK_DLOAD(x) myDW[x]
#define PRK_DSTORE(p, v) (p) = (v)
#endif
  dCurW = PRK_DLOAD(0);                              // pair 0: cell 1 is step 0's deletion predecessor
  if constexpr (B > 2) dNxtW = PRK_DLOAD(BLOCK);     // pair 1
  auto step = [&](auto jc) __attribute__((always_inline))
  {
    constexpr int j = decltype(jc)::value;
    if constexpr ((j & (PRK_FAST_GROUP(BLOCK) - 1)) == 0)
    {
      // pin the accumulators to their group: nothing but data dependences orders pure arithmetic against
      // sched_barrier during instruction selection, and a sunk accumulation keeps every table row alive
      asm volatile("" ::"v"(bA[0]), "v"(bA[1]), "v"(bA[2]), "v"(bA[3]), "v"(maxE), "v"(kg[(j > 0 ? j - 1 : 0) >> 4]));
      __builtin_amdgcn_sched_barrier(0);
    }
    const int sv = rowQ[0].x, sF = rowQ[0].y;
#pragma unroll
    for (int k = 0; k + 1 < PD; k++) rowQ[k] = rowQ[k + 1];
    if constexpr (j + PD <= B) rowQ[PD - 1] = fetch_row(std::integral_constant<int, (j + PD <= B ? j + PD : 0)>{});
    // cell j+1 is the first cell of pair (j+1)/2 when j is odd: move on to the pair loaded two steps ago, load the next one
    if constexpr ((j & 1) != 0 && j + 1 < B)
    {
      dCurW = dNxtW;
      if constexpr (((j + 1) >> 1) + 1 <= (B - 1) / 2) dNxtW = PRK_DLOAD((((j + 1) >> 1) + 1) * BLOCK);
    }
    const bool inb = MASKHI ? (j <= jhi) : true;     // this step's cell of row r and candidate cell j-1 of row r+1
    // candidates' cell j-1 of row r+1: substitution from m_{j-1} of row r (the base of that cell is this step's)
    if constexpr (j >= 1 && !LEAN)
    {
      const int ms = MASKHI ? (inb ? mPrev : SENT) : mPrev;
      const int t4[4] = { add_sext_byte<0>(ms, sv), add_sext_byte<1>(ms, sv), add_sext_byte<2>(ms, sv), add_sext_byte<3>(ms, sv) };
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
      if constexpr (j + 1 < B) Pe = add_sext_word<((j + 1) & 1)>(M[j + 1], dCurW);
      const int sub = Pm + sF;                       // bnw_extend.c:950-956
      const int mr = vmax3(sub, eC, Pe);             // max(sub, max(ins, del)), :1007-1018
      const int er = vmax3(sub + go, eC, Pe) + ge;
      const int m = MASKHI ? (inb ? mr : SENT) : mr;                 // :990-1002
      const int e = MASKHI ? (inb ? er : SENT + ge) : er;            // max(SENT + go, SENT) + ge with go <= 0
      M[j] = m;
      if constexpr ((j & 1) == 0 && j + 1 < B) dLo = e - m;
      else if constexpr ((j & 1) != 0) PRK_DSTORE(myDW[(j >> 1) * BLOCK], pack_halves(dLo, e - m));
      else PRK_DSTORE(myD[(j >> 1) * (2 * BLOCK)], (short)(e - m));           // cell 2W: the low half of the last pair
      // best cell: packed (value, cell) keys -- LEAN: the value alone (kg[0] is then a plain running maximum of m)
      const int kr = LEAN ? mr : (int)(((unsigned)mr << 4) | (unsigned)(15 - (j & 15)));
      const int key = MASKHI ? (inb ? kr : -2147483647 - 1) : kr;
      constexpr int gk = LEAN ? 0 : (j >> 4);
      if constexpr ((j & 1) == 0 && j + 1 < B) kPend = key;
      else if constexpr ((j & 1) != 0) kg[gk] = imax3(kg[gk], kPend, key);
      else kg[gk] = imax(kg[gk], key);
      // deletion term of candidate cell j-1 is e_j (cells 1..B-1)
      if constexpr (j >= 1 && !LEAN)
      {
        if constexpr ((j & 1) != 0) ePend = e;
        else maxE = imax3(maxE, ePend, e);
      }
      mPrev = m;
      eC = e;
    }
  };
  static_for(step, std::make_integer_sequence<int, B + 1>{});
  if constexpr (LEAN)
  {
    // no candidate rows: the caller has proved that the contribution is max(0, high + cap) whatever they hold
#pragma unroll
    for (int c = 0; c < 4; c++) D.bestA[c] = NEG;
    D.bestF = kg[0];
    D.jbest = 0;
    return;
  }
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

// Leaders.  Behind the end of the alignment nearly every flank sits at its cap, but a FEW keep climbing -- the consensus follows
// whoever still contributes above the cap, so those flanks match it column after column (measured on the N = 100,000 bench
// set: one flank of 100,000, profiles/r03_notes.md).  Their wave must not run the LEAN band for them, and the full band for
// one lane costs the wave (and every workgroup waiting for its ticket) 640 extra instructions per column.  Instead the wave
// runs LEAN -- which updates the leader's row like everybody's -- and then computes what LEAN skipped for that ONE flank with
// all 64 lanes: the leader's lane puts its new row and its base words into LDS, lane i takes cells i, i + 64, .. (candidate
// cell k of row r+1 = m_k + M[a][class of step k+1], deletion terms e_j = m_j + d_j straight from the d-row), five wave
// maxima and the lowest cell holding the row's best.  In-bounds waves only (the in-bounds fast variant's formulation).
template <int W, int BLOCK>
__device__ __forceinline__ void prk_leader_rows(const FastTabs &ft, const short *sD, int *scr /* [B + NW] of this wave */, const int r, const int leader,
                                                const unsigned (&w)[(2 * W + 1 + 8) / 8 + 2], const int (&M)[2 * W + 1], LaneDP &D)
{
  constexpr int B = 2 * W + 1, NW = (B + 8) / 8 + 2;
  constexpr int NH = (B + 63) / 64;                     // cells per lane
  const int lane = threadIdx.x & 63;
  if (lane == leader)
  {
#pragma unroll
    for (int j = 0; j < B; j++) scr[j] = M[j];
#pragma unroll
    for (int k = 0; k < NW; k++) scr[B + k] = (int)w[k];
  }
  __builtin_amdgcn_wave_barrier();          // LDS operations of a wave execute in order; this only pins the compiler's order
  const int ph = (r + 8) & 7;
  const int thr = (threadIdx.x & ~63) + leader;        // the leader's thread: its column of the d-row
  const int bf = __builtin_amdgcn_readlane(D.bestF, leader);
  int ta[4] = { NEG, NEG, NEG, NEG }, me = NEG;
  unsigned long long hit[NH];
#pragma unroll
  for (int h = 0; h < NH; h++)
  {
    const int k = lane + 64 * h;
    const bool ok = k < B;
    const int kk = ok ? k : 0;
    const int m = scr[kk];
    const int g = kk + 1 + ph;                         // nibble of step k+1 in the base words (prk_band_fast: alignbit by 4 * ph)
    const unsigned cls = ((unsigned)scr[B + (g >> 3)] >> (4 * (g & 7))) & 15u;
    const int sv = ft.row[cls][0];
    if (ok)
    {
      ta[0] = imax(ta[0], add_sext_byte<0>(m, sv)); ta[1] = imax(ta[1], add_sext_byte<1>(m, sv));
      ta[2] = imax(ta[2], add_sext_byte<2>(m, sv)); ta[3] = imax(ta[3], add_sext_byte<3>(m, sv));
      if (k >= 1) me = imax(me, m + (int)sD[(kk >> 1) * (2 * BLOCK) + 2 * thr + (kk & 1)]);      // e_k, cells 1 .. B-1
    }
    hit[h] = __ballot(ok && m == bf);
  }
  const int mx = wave_max_i32_dpp(me);
  int best[4];
#pragma unroll
  for (int c = 0; c < 4; c++) best[c] = imax(wave_max_i32_dpp(ta[c]), mx);
  int jb = 0;                                          // lowest cell on ties (bnw_extend.c:1020-1024)
#pragma unroll
  for (int h = NH - 1; h >= 0; h--)
    if (hit[h]) jb = 64 * h + __builtin_ctzll(hit[h]);
  if (lane == leader)
  {
#pragma unroll
    for (int c = 0; c < 4; c++) D.bestA[c] = best[c];
    D.jbest = jb;
  }
}

template <int W, bool OOB, int BLOCK, bool INIT = false>
__device__ __forceinline__ void prk_band(const int go, const int ge, const int *s_tab, const FastTabs &ft, short *sD, const int r,
                                         const unsigned (&w)[(2 * W + 1 + 8) / 8 + 2], const int jlo, const int jhi,
                                         int (&M)[2 * W + 1], LaneDP &D)
{
  constexpr int B = 2 * W + 1;
  if (!OOB && !INIT)
  {
    prk_band_fast<W, BLOCK, false>(go, ge, ft, sD, r, 0, w, M, D);
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

// W = 80 (161 cells per lane): one wave per SIMD only, so that the row has the whole register file of the lane (the
// accumulation registers take what the 256 architectural ones cannot hold)
template <int W, int BLOCK>
__global__ __launch_bounds__(BLOCK, (W > 40 ? 1 : 2)) void ramx_persistent_kernel(const PArgs a)
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
    int lead[WPB][B + NW + 1];                         // prk_leader_rows: a leader's row and base words, per wave
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
  int prevBest = 0x3fffffff;             // best cell of the previous row (LEAN test): unknown before the first band of this launch
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
  unsigned long long lstat[2] = { 0, 0 };
#endif
  for (int r = 0; r < a.L; r++)
  {
    PRK_TICK(5);
    // ---- base words of this column (independent of the vote: issued before the wait) -------
    // (Keeping the NW words across columns and loading one new word every eighth column was tried: they are live across the
    // whole loop then, the allocator spills them, and scratch reloads replace the L2 loads one for one.)
    unsigned w[NW];
    {
      const unsigned *bp = a.bases + (size_t)((r + 8) >> 3) * a.Np + n;
#pragma unroll
      for (int k = 0; k < NW; k++) w[k] = bp[(size_t)k * a.Np];
    }
    // ---- vote of row r -----------------------------------------------------------------------
    if (wave == 0)
    {
      long long v[4];
      prk_wait_vote(a, a.vote, a.sums0, r == 0 ? 1 : 0, r, lane, my_shard_blocks, failed, v, blockIdx.x == 0);
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
    // block 0 clears the vote set of row r+3 (see the protocol above)
    if (blockIdx.x == 0 && threadIdx.x < NSHARD)
    {
      PShard *z = a.vote + (size_t)((r + 3) & (PRK_NSETS - 1)) * NSHARD + threadIdx.x;
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
      // padding lanes (no flank: their base stream is all N, their vote is masked) must not force the whole wave onto
      // the masked path -- with N not a multiple of 64 that one slow wave would gate every column
      const bool all_in = a.pack_ok && __all((n >= a.Nx) || ((jlo <= 0) && (jhi >= B)));
      const bool maskhi_ok = a.pack_ok > 1 && r >= W && __all((n >= a.Nx) || (jlo <= 0));
      // LEAN?  (prk_band_fast: no lane can contribute more than its cap to the vote of row r+1 or set a record in row r, and the
      // wave takes one of the in-bounds fast variants.)  prevBest = best cell of row r-1, high = record after row r-1.  A few
      // lanes that fail the test (`leaders`) do not keep the wave from LEAN: prk_leader_rows computes their candidate rows.
      bool lean = false;
      unsigned long long leaders = 0;
      if (a.lean_p >= 0 && (all_in || maskhi_ok))
      {
        const int capfloor = (high + a.cap) > 0 ? (high + a.cap) : 0;
        const unsigned long long keep = __ballot((n < a.Nx) && !((prevBest + 2 * a.lean_p <= capfloor) && (prevBest + a.lean_p <= high)));
        lean = keep == 0;
        if (all_in && keep != 0 && __popcll(keep) <= a.leader_max) { lean = true; leaders = keep; }
#ifdef RAMX_PRK_TIMING
        // behind the first half of the run: columns in which this wave may not run plain LEAN, and the lanes that keep it from it
        if (2 * r >= a.L && keep != 0) { lstat[0] += 1; lstat[1] += (unsigned)__popcll(keep); }
#endif
      }
#ifdef PRK_PROBE_NO_BAND
      if (r >= 0) { D.bestF = (int)w[0] & 1023; asm volatile("" : "+v"(M[0]), "+v"(M[B - 1])); }      // timing probe: the column without its band
      else
#endif
      if (all_in)
      {
        if (lean)
        {
          prk_band_fast<W, BLOCK, false, true>(a.go, a.ge, s_ft, sD, r, 0, w, M, D);
          for (unsigned long long rest = leaders; rest != 0; rest &= rest - 1)
            prk_leader_rows<W, BLOCK>(s_ft, sD, sm.lead[wave], r, __builtin_ctzll(rest), w, M, D);
        }
        else prk_band<W, false, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
      }
      else if (maskhi_ok)
      {
        if (lean) prk_band_fast<W, BLOCK, true, true>(a.go, a.ge, s_ft, sD, r, jhi, w, M, D);
        else prk_band_fast<W, BLOCK, true>(a.go, a.ge, s_ft, sD, r, jhi, w, M, D);      // some flank has run out at its far end
      }
      else prk_band<W, true, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
      prevBest = D.bestF;
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
    const bool hand_over = a.sums_next != NULL && !stopped && r == a.L - 1;     // another launch continues with row L
    if (!hand_over && (stopped || r == a.L - 1)) break;     // the vote of row r+1 will not be consumed
    {
      long long tot[4];
      // (one reduction instead of four when a plain LEAN column makes the four sums equal was measured: 7.46 against 7.21 us
      // per column -- the allocator's answer to the extra branch costs more than the 45 instructions)
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
    if (threadIdx.x < 4)
    {
      long long t = 0;
#pragma unroll
      for (int wv = 0; wv < WPB; wv++) t += s_red[wv][threadIdx.x];
      if (hand_over)
        __hip_atomic_fetch_add(&a.sums_next[shard * 4 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
      {
        PShard *sh = a.vote + (size_t)((r + 1) & (PRK_NSETS - 1)) * NSHARD + shard;
        __hip_atomic_fetch_add(&sh->word[threadIdx.x], (unsigned long long)t + PRK_BIAS + PRK_TICKET, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    PRK_TICK(4);                 // contribution issued
    if (hand_over) break;
  }
#ifdef RAMX_PRK_TIMING
  if (a.dbg != NULL && (threadIdx.x & 63) == 0)
  {
#pragma unroll
    for (int k = 0; k < 6; k++) a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + k] = tsum[k];
    a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + 6] = lstat[0];
    a.dbg[((size_t)blockIdx.x * WPB + wave) * 8 + 7] = lstat[1];
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
  int lean_p;                   // P of the LEAN test (see PArgs)
  unsigned long long *dbg;      // -DRAMX_PRK_TIMING builds only
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
  int prevBest = 0x3fffffff;             // best cell of the previous row (LEAN test)
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
#ifdef RAMX_PRK_TIMING
  unsigned long long tsum[6] = { 0, 0, 0, 0, 0, 0 }, tlast = wall_clock64();
#endif
  for (int r = -1; r < a.L; r++)
  {
    PRK_TICK(5);
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
      PRK_TICK(0);               // vote folded, stop rule
      fast_tabs_winner(s_ft, s_tab, threadIdx.x);
      __syncthreads();
      PRK_TICK(1);               // winner table + barrier
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
        // padding lanes (no flank: their base stream is all N, their vote is masked) must not force the whole wave onto
        // the masked path -- with N not a multiple of 64 that one slow wave would gate every column
        const bool all_in = a.pack_ok && __all((!active) || ((jlo <= 0) && (jhi >= B)));
        const int capfloor = (high + a.cap) > 0 ? (high + a.cap) : 0;
        const bool lean = a.lean_p >= 0 && __all((!active) || ((prevBest + 2 * a.lean_p <= capfloor) && (prevBest + a.lean_p <= high)));
        if (all_in)
        {
          if (lean) prk_band_fast<W, BLOCK, false, true>(a.go, a.ge, s_ft, sD, r, 0, w, M, D);
          else prk_band<W, false, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
        }
        else if (a.pack_ok > 1 && r >= W && __all(!active || (jlo <= 0)))
        {
          if (lean) prk_band_fast<W, BLOCK, true, true>(a.go, a.ge, s_ft, sD, r, jhi, w, M, D);
          else prk_band_fast<W, BLOCK, true>(a.go, a.ge, s_ft, sD, r, jhi, w, M, D);
        }
        else prk_band<W, true, BLOCK>(a.go, a.ge, s_tab, s_ft, sD, r, w, jlo, jhi, M, D);
        prevBest = D.bestF;
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
    PRK_TICK(2);                 // band + contributions
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
    PRK_TICK(3);                 // wave reduction + LDS write
    __syncthreads();
    PRK_TICK(4);                 // end-of-column barrier
  }
#ifdef RAMX_PRK_TIMING
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && a.dbg != NULL)
  {
#pragma unroll
    for (int k = 0; k < 6; k++) a.dbg[wave * 8 + k] = tsum[k];
    a.dbg[wave * 8 + 6] = (unsigned long long)rows_done;
  }
#endif
  if (live) a.trim[n] = make_int2(thigh, tpos);
  if (threadIdx.x == 0)
  {
    RamxCtl o;
    o.max_ext = max_ext; o.max_row = max_row; o.stopped = stopped; o.rows_done = rows_done; o.overflow = ovf; o.besta = 0; o.pad = 0;
    a.ctl_out[fd.id] = o;
  }
}

