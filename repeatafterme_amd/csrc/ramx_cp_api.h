// ramx_cp_api.h -- host-side interface of the cell-parallel kernels (ramx_cp.hip), used by ramx_device.hip.
// Internal to libramx (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include "ramx_kernels_common.h"

// device-wide mode: one entry per workgroup.  A launch may hold several flank sets ("families" too large for one
// workgroup): the workgroups of one set vote among themselves through their own ticket words / error word / outputs.
// device-wide mode: rotating sets of vote shards (row r uses set r % 4; workgroup 0 clears the set of row r+3 during column r,
// a whole column before anybody adds to it -- see publish / wait_vote in ramx_kernels_cp.h)
#define RAMX_CP_NSETS 4
// device-wide mode, blocks of at most RAMX_CP_SYNCW_MAXC cells per lane: wave 0 of every workgroup holds no flank (it runs
// the vote), so a workgroup of T threads holds (T - 64) / K flanks.  The band waves keep the row they may have to restore
// in registers up to RAMX_CP_SYNCW_REGC cells per lane (workgroups of up to 576 threads), in LDS above (up to 512 threads)
#define RAMX_CP_SYNCW_MAXC 21
#define RAMX_CP_SYNCW_REGC 11

struct CpDevDesc
{
  int first;        // first flank of the set in the flank arrays
  int nx;           // flanks of the set
  int b, nb;        // this workgroup's index inside the set, workgroups of the set
  int id;           // index of the set's outputs: ctl_out[id], cons_out + id * L
  int vs;           // index of the set's vote words: vote + vs * RAMX_CP_NSETS * NSHARD, err + vs * 16
  int pad[2];
};

struct CPArgs
{
  const unsigned *bases;        // [KW][Np] packed windows (ramx_pack_kernel)
  const int2 *bounds;
  const FamDesc *fam;
  int2 *trim;                   // per flank
  RamxCtl *ctl_out;             // per family
  signed char *cons_out;        // [family][L]
  int2 *state_out;              // optional (tests): final rows, [flank][2W+1] (m, e), else NULL
  int Np, KW, L, go, ge, cap, minimp, when_to_stop;
  int tab[RAMX_NCLASS][4];
  unsigned long long *dbg;      // -DRAMX_CP_TIMING builds only: [wave of block 0][8] phase sums in shader clocks
  // device-wide mode (one flank set over the whole grid)
  const CpDevDesc *dev;         // [workgroups of the launch]
  struct PShard *vote;          // [3][NSHARD] ticketed vote words (zeroed by the host before the launch)
  unsigned *err;                // != 0: a bounded spin gave up
  int4 *S;                      // optional: final rows in the lane-per-flank layout (ramx_dev_peek_state), else NULL
  // multi-GPU (flanks sharded over ranks): the mailboxes of ramx_kernels_resident.h
  struct PeerBox *const *peers; // [nranks] every rank's box as seen from this device (NULL on one GPU)
  struct PeerBox *box;          // this rank's own box
  struct PeerBox *mirror;       // host-memory boxes only: device copy kept current by workgroup 0 (NULL: everybody polls `box`)
  int rank, nranks;
  int vote_wave;       // device-wide mode, blocks of up to RAMX_CP_SYNCW_MAXC cells: 1 = wave 0 of every workgroup runs the vote (ramx_cp_device_plan)
  int test_drop_row;   // test hook (RAMX_TEST_CP_DROP_TICKET=row): the last workgroup of every set withholds its words for that row; 0 = off
  int lean_p;            // P = max(0, largest matrix entry) of the LEAN test (ramx_kernels_cp.h band()); -1: never LEAN
  int test_wrong_every;  // test hook (RAMX_TEST_CP_WRONG_EVERY=n): vote-wave mode, every n-th row is computed on a deliberately wrong guess; 0 = off
  int test_vote_delay;   // test hook (RAMX_TEST_CP_VOTE_DELAY=units): the vote wave idles that long before every row, so the band waves reach full depth
  int test_drop_id;    // ... of the set whose outputs index (CpDevDesc.id) is this one only; -1 = every set
};

#define RAMX_CP_NCLASS 6
// Largest family (flanks) the cell-parallel family kernel takes for this band width and scoring system, 0 if none:
// non-positive gap penalties (chain-free candidates), int8 scores, every reachable |score| < 2^23 (packed keys).
int ramx_cp_max_family(int W, int go, int ge, const int (&tab)[RAMX_NCLASS][4], int L);
// class of a family of nx flanks (0 .. RAMX_CP_NCLASS-1): lanes per flank and workgroup size; -1 if too large
int ramx_cp_class(int W, int nx, int *lanes_per_flank, int *threads);
int ramx_cp_single_family_max(int W);
int ramx_cp_launch_families(hipStream_t st, int W, int lanes_per_flank, int threads, int n_families, const CPArgs &a);
// Device-wide mode: lanes per flank and workgroup count for n flanks on `cus` compute units (one 512-thread workgroup per
// CU at most), 0 lanes if the set does not fit or the width / scoring system is not supported (ramx_cp_max_family > 0).
// wide: 0 = four band waves per workgroup when the set then fits `cus` workgroups (lowest latency), 1 = the largest workgroup only
int ramx_cp_device_plan(int W, int n_flanks, int cus, int wide, int *lanes_per_flank, int *threads, int *blocks, int *vote_wave);
int ramx_cp_launch_device(hipStream_t st, int W, int lanes_per_flank, int threads, int blocks, const CPArgs &a);
