// ramx_kernels_vote.h -- the device-wide vote of the lane-per-flank persistent kernels (the int32 rows of ramx_kernels_resident.h and
// the packed rows of ramx_kernels_packed.h): ticketed vote shards, the mailboxes of the cross-device step, and the code both
// kernels run to wait for a column's vote (one copy, so that a protocol change is made in one place).
// (device code of libramx; included by ramx_device.hip and ramx_packed.hip)
#pragma once

#include "ramx_kernels_common.h"

// ------------------------------------------------------------------------------------------
// protocol
// ------------------------------------------------------------------------------------------
//
//   column c, every block:   4 x int64 atomicAdd into shard (blockIdx % 32) of vote set (c+1) % 4.  Each add
//                            carries its own arrival ticket: value = partial_sum + 2^41 + 2^54, so bits 54..63 of
//                            a shard word count the blocks that have contributed and the low 54 bits hold
//                            sum + count * 2^41 (|partial| <= 512 lanes * 2^31 < 2^41: exact for any input).
//   column c+1, wave 0:      lanes 0..31 poll "their" shard's four words (relaxed agent-scope loads + s_sleep,
//                            bounded) until all four show every block of the shard, decode, shuffle-reduce and
//                            publish the vote through LDS.  One fabric round trip after the last arrival.
//   Four sets rotate.  Set (c+3) % 4 (last used by row c-1) is zeroed by block 0 during column c, once block 0 has seen
//   every ticket of row c -- so everybody has finished reading row c-1.  Nobody adds to it before having seen block 0's
//   ticket for row c+2, which block 0 sends in column c+1, after its own poll for row c+1 -- whose
//   s_waitcnt vmcnt(0) also drains the zeroing stores of column c (same wave).  The zeroing is complete a column before
//   it has to be and nobody stalls for it (with three sets block 0 had to wait for its stores before every add).
//
// Placement independent: only agent-scope atomics / atomic loads touch shared words, no assumption on which
// XCD a block runs.  The launch is a PLAIN one (a cooperative launch made profiled processes crash at exit, DESIGN.md
// section 7): the host launches at most one workgroup per CU and never more workgroups than CUs, which makes the grid
// co-resident on an idle device but is no guarantee next to a co-tenant -- correctness therefore rests on the BOUNDED
// spins (a timeout raises `err`, every block leaves) and on the host repeating the direction with the per-column
// launches when that happens (ramx_dev_run_direction; batch mode: ramx_dev_run_families repeats the affected
// families).  Multi-GPU runs exchange the vote through the mailboxes below, or fall back to RCCL between per-column launches.

#define PRK_NSETS 4        // rotating vote sets: row r uses set r % 4, block 0 clears the set of row r+3 in column r (protocol above)
#ifndef PRK_SHARD_BYTES
#define PRK_SHARD_BYTES 256   // 64 (one line per shard) measured 1.5-2 % slower: neighbouring shards share a memory channel
#endif
struct PShard { unsigned long long word[4]; unsigned long long pad[PRK_SHARD_BYTES / 8 - 4]; };

// sum of a 64-bit value over each row of 16 lanes (every lane of the row gets it): xor-1, xor-2 butterflies inside quads,
// then the mirrored half-row and the mirrored row (sums are uniform below each step, so a mirror reaches the other half)
__device__ __forceinline__ unsigned long long prk_row_sum_u64(unsigned long long x)
{
#define PRK_SUM_STEP(ctrl) do { \
    const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)x, ctrl, 0xf, 0xf, false); \
    const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), ctrl, 0xf, 0xf, false); \
    x += ((unsigned long long)hi_ << 32) | lo_; } while (0)
  PRK_SUM_STEP(0xB1);     // quad_perm [1,0,3,2]
  PRK_SUM_STEP(0x4E);     // quad_perm [2,3,0,1]
  PRK_SUM_STEP(0x141);    // row_half_mirror
  PRK_SUM_STEP(0x140);    // row_mirror
#undef PRK_SUM_STEP
  return x;
}
__device__ __forceinline__ unsigned long long prk_readlane_u64(unsigned long long x, int l)
{
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l);
}
#define PRK_BIAS (1ULL << 41)
#define PRK_TICKET (1ULL << 54)

// Multi-GPU: every rank owns one PeerBox in fine-grained device memory, mapped into all other ranks through
// hipIpc handles.  After a rank's own blocks have all contributed to a column, its block 0 stores the rank's four
// totals into slot [set][rank] of EVERY box (its own included) over xGMI; each word carries the column number in
// its top 16 bits, so a reader knows a word is current without any flag or fence; every block then polls the local
// box until all ranks' words of this column are there.  3 sets rotate exactly like the vote shards.
#define RAMX_MAX_RANKS 16
struct PeerBox { unsigned long long slot[3][RAMX_MAX_RANKS][4]; unsigned long long token[RAMX_MAX_RANKS]; };
// a word's tag is the low 16 bits of (row + 1): a cleared slot (tag 0) is never taken for row 0's word by a rank that
// looks before its peer has written
#ifndef PEER_TAG_OFFSET
#define PEER_TAG_OFFSET 1
#endif
#define PEER_VBIAS (1LL << 46)
#define PEER_VMASK ((1ULL << 48) - 1)

#define PRK_SPIN_LIMIT (1u << 21)

template <class F, int... Js>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, Js...>)
{
  (f(std::integral_constant<int, Js>{}), ...);
}

// Wave 0 of a workgroup, column r: the device's (multi-rank: every device's) four candidate sums of row r, identical in all
// lanes.  `first` != 0: the device's sums come as NSHARD x 4 plain int64 words (`sums0`: written by the launch before this one)
// instead of tickets; 1: they are every device's already (row 0: the host has summed them over the ranks), 2: this device's
// only (a launch that continues a direction: the cross-device step follows as for any other row).  vb: the vote sets; my_shard_blocks: blocks arriving on shard (lane & 31).  A bounded spin that gives up sets
// `failed`.  A: the kernel's argument block (err, nranks, rank, peers, box, mirror).  xb: this workgroup is the device's
// EXCHANGER (multi-rank: the one that waits for the local total and stores it into every rank's box; workgroup 0, or a
// workgroup without flanks launched for the purpose, ramx_kernels_packed.h).
template <class A>
__device__ __forceinline__ void prk_wait_vote(const A &a, const PShard *vb, const long long *sums0, const int first, const int r, const int lane,
                                              const int my_shard_blocks, int &failed, long long (&v)[4], const bool xb)
{
  v[0] = v[1] = v[2] = v[3] = 0;
  if (first)
  {
    if (lane < NSHARD) { const long long *p = sums0 + lane * 4; v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = wave_sum_ll(v[k]);
  }
  else
  {
    // lane = shard + 32 * half polls words 2*half, 2*half+1 of "its" shard: one 16-byte load per lane and round
    // (a quarter of the requests of four 8-byte loads on 32 lanes; the poll competes with the adds it waits for)
    const int sidx = lane & (NSHARD - 1), half = lane >> 5;
    const unsigned long long *src = &vb[(size_t)(r & (PRK_NSETS - 1)) * NSHARD + sidx].word[2 * half];
    unsigned spins = 0;
    bool done = my_shard_blocks <= 0 || (a.nranks > 1 && !xb);   // multi-rank: only the exchanger needs the local total
#ifdef PRK_PROBE_NO_WAIT
    done = true;         // timing probe (wrong results by construction): nobody waits for the vote, the winner rotates
#endif
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
    // fold the 32 shards: the raw words first (sum + bias and ticket fields are both additive: at most 256 tickets,
    // ten bits), rows of 16 lanes with DPP butterflies, the four rows on the scalar unit; one decode per word
    if (my_shard_blocks <= 0 || failed || (a.nranks > 1 && !xb)) { x0 = 0; x1 = 0; }
    x0 = prk_row_sum_u64(x0); x1 = prk_row_sum_u64(x1);
    const unsigned long long t0 = prk_readlane_u64(x0, 0) + prk_readlane_u64(x0, 16), t1 = prk_readlane_u64(x1, 0) + prk_readlane_u64(x1, 16);
    const unsigned long long t2 = prk_readlane_u64(x0, 32) + prk_readlane_u64(x0, 48), t3 = prk_readlane_u64(x1, 32) + prk_readlane_u64(x1, 48);
    v[0] = (long long)(t0 & (PRK_TICKET - 1)) - (long long)(t0 >> 54) * (long long)PRK_BIAS;
    v[1] = (long long)(t1 & (PRK_TICKET - 1)) - (long long)(t1 >> 54) * (long long)PRK_BIAS;
    v[2] = (long long)(t2 & (PRK_TICKET - 1)) - (long long)(t2 >> 54) * (long long)PRK_BIAS;
    v[3] = (long long)(t3 & (PRK_TICKET - 1)) - (long long)(t3 >> 54) * (long long)PRK_BIAS;
#ifdef PRK_PROBE_NO_WAIT
    v[0] = (r & 3) == 0; v[1] = (r & 3) == 1; v[2] = (r & 3) == 2; v[3] = (r & 3) == 3;
#endif
  }
  if (a.nranks > 1 && first != 1 && !failed)
  {
    // ---- cross-device step: v[] is this rank's total (identical in all lanes) -----------
    const unsigned long long tag = (unsigned long long)((r + PEER_TAG_OFFSET) & 0xffff) << 48;
    if (xb && lane < a.nranks)
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
    const PeerBox *pollbox = (a.mirror != NULL && !xb) ? a.mirror : a.box;
    for (;;)
    {
      if (!got)
      {
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = __hip_atomic_load(&pollbox->slot[r % 3][lane][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        got = (y[0] >> 48) == (tag >> 48) && (y[1] >> 48) == (tag >> 48) && (y[2] >> 48) == (tag >> 48) && (y[3] >> 48) == (tag >> 48);
        if (got && a.mirror != NULL && xb)
        {
          // this rank's word of this column has arrived: pass it on to the local pollers
#pragma unroll
          for (int k = 0; k < 4; k++)
            __hip_atomic_store(&a.mirror->slot[r % 3][lane][k], y[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
}

// 16-bit halves of a register (the packed rows of ramx_kernels_packed.h; the leaders of both kernels)
__device__ __forceinline__ int pk_max3_lll(int a, int b, int c) { int d; asm("v_max3_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int pk_max3_hhh(int a, int b, int c) { int d; asm("v_max3_i16 %0, %1, %2, %3 op_sel:[1,1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
typedef short pk_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_s2 pk_v(int x) { return __builtin_bit_cast(pk_s2, x); }
__device__ __forceinline__ int pk_i(pk_s2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk_two(int x) { return (x & 0xffff) | (x << 16); }
template <int BYTE, int SH>
__device__ __forceinline__ unsigned pk_byte_shl(unsigned A)   // ((A >> 8*BYTE) & 0xff) << SH
{
  unsigned d;
  if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "n"(SH), "v"(A));
  else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "n"(SH), "v"(A));
  else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "n"(SH), "v"(A));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "n"(SH), "v"(A));
  return d;
}
template <int HALF, int SH>
__device__ __forceinline__ unsigned pk_word_shl(unsigned A)   // ((A >> 16*HALF) & 0xffff) << SH
{
  unsigned d;
  if (HALF == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "n"(SH), "v"(A));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "n"(SH), "v"(A));
  return d;
}
