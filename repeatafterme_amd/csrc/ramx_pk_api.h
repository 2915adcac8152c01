// ramx_pk_api.h -- host-side interface of the packed-row persistent kernel (ramx_packed.hip), used by ramx_device.hip.
// Internal to libramx (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include "ramx_kernels_common.h"

struct PShard;
struct PeerBox;

struct PKArgs
{
  int4 *S;                      // row state in HBM: read at start (row r0 - 1), written back at the end
  const unsigned *bases;
  const int2 *bounds;
  int2 *trim;                   // r0 > 0: read at start (the snapshot so far)
  const long long *sums_in;     // vote of row r0 as NSHARD x 4 plain int64 words (K(-1), or the int32 kernel that ran rows 0 .. r0-1)
  PShard *vote;                 // [PRK_NSETS][NSHARD], zeroed by the host
  const RamxCtl *ctl_in;        // r0 > 0: stop-rule state after row r0 - 1
  RamxCtl *ctl_out;
  signed char *cons_out;
  unsigned *err;                // != 0: a bounded spin gave up (1), a computed row left the span (2), the entry check refused the rows (3); the highest stays
  PeerBox *const *peers;        // multi-rank: the mailboxes of ramx_kernels_vote.h (NULL on one GPU)
  PeerBox *box;
  PeerBox *mirror;
  int rank, nranks;
  int xblock;                   // the workgroup that does the device's exchange duties: 0, or nblocks (one more workgroup, without flanks)
  long long *sums_next;         // Lseg < L: the sums of row Lseg as NSHARD x 4 plain int64 words (zeroed by the host) for the launch that continues
  int Np, Nx, r0, Lseg, L, go, ge, cap, minimp, when_to_stop, nblocks;      // this launch runs rows r0 .. Lseg-1 of a direction of L rows
  int tab[RAMX_NCLASS][4];
  int lean_p;                   // P = max(0, largest matrix entry) of the LEAN test; -1: never LEAN
  int leader_max;               // a wave with at most this many lanes that fail the LEAN test runs LEAN + pkb_leader_rows (0: off)
  int spread;                   // an in-bounds cell lies at most this far below its row's best cell (entry check)
  int spread_rows;              // the same for the check of every 64th computed row (= spread; a test hook sets them apart)
  int rebase;                   // |best cell - base| that moves the base (looked at every 16th row)
  int spec_on;                  // 1: a workgroup computes a row on its own argmax before the vote is known (RAMX_NO_PK_SPEC=1: 0)
  int test_wrong_every;         // test hook (RAMX_TEST_PK_WRONG_EVERY=n): every n-th guess is replaced by another base; 0 = off
  unsigned long long *dbg;      // -DRAMX_PRK_TIMING builds only: [wave][8] phase sums in 10 ns ticks
};

// Can the packed rows hold this scoring system at this band width (int16 relative to a per-flank base)?  Sets spread / rebase.
// 0: no (the int32 rows serve it).
int ramx_pk_plan(int W, int go, int ge, const int (&tab)[RAMX_NCLASS][4], int *spread, int *rebase);
// launch shape that keeps `tiles` 64-flank tiles resident (at most one workgroup per CU): block = 0 if none
int ramx_pk_capacity(int W, int block, int *cap);
int ramx_pk_shape(int W, int tiles, int *block, int *blocks);
int ramx_pk_launch(hipStream_t st, int W, int block, int blocks, const PKArgs &a);
