// ramx_cp.hip -- second translation unit of libramx's device code: the cell-parallel kernels (ramx_kernels_cp.h) and
// their launchers.  Kept apart from ramx_device.hip so that the two compile side by side.
#define RAMX_SECONDARY_TU 1
#include "ramx_kernels_cp.h"

#include <stdlib.h>

static bool cp_has_width(int W) { return W == 14 || W == 20 || W == 40 || W == 80; }
static int cp_cells(int W, int K) { return (2 * W + 1 + K - 1) / K; }
// workgroup size limit of a block of C cells per lane (register budget, CpCfg::MAXT)
static int cp_max_threads(int W, int C) { (void)W; return C <= 41 ? 512 : 0; }

int ramx_cp_max_family(int W, int go, int ge, const int (&tab)[RAMX_NCLASS][4], int L)
{
  if (!cp_has_width(W) || go > 0 || ge > 0 || L <= 0 || getenv("RAMX_NO_CP") != NULL) return 0;
  long long mx = (long long)(-go) + (long long)(-ge);
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
    {
      if (tab[c][k] < -128 || tab[c][k] > 127) return 0;
      const long long v = tab[c][k] < 0 ? -(long long)tab[c][k] : (long long)tab[c][k];
      if (v > mx) mx = v;
    }
  // a row-r cell is at most (r + W + 2) steps of at most mx away from 0; keys hold (score << 8 | cell); block totals are
  // tilted by up to 16 * C * |ge|
  if (((long long)L + 2LL * W + 4) * mx >= (1LL << 23)) return 0;
  if (-(long long)ge * 200 >= (1LL << 23)) return 0;
  int best = 0;
  for (int K = 16; K >= 2; K >>= 1)
  {
    const int t = cp_max_threads(W, cp_cells(W, K));
    if (t / K > best) best = t / K;
  }
  return best;
}

// Largest single family for which ONE workgroup is the faster place (blocks of up to 21 cells per lane): above it the
// device-wide mode with 16 lanes per flank wins (W = 40, 250 flanks: 4.1 us per column in one workgroup at K = 2, 2.6
// device-wide; profiles/r02_cp_family_timing.log).  Batches keep every family that fits in one workgroup: there a family
// per CU is the better use of the chip.
int ramx_cp_single_family_max(int W)
{
  // A family that runs ALONE (seam 1) stays in one workgroup while that keeps sixteen lanes per flank (32 flanks), or eight
  // for the narrow bands (64 flanks; blocks of 2-3 cells); above that the device-wide mode -- several workgroups of four
  // band waves and a vote wave -- is faster (profiles/r02_small_route.log: W = 40, 100 flanks: 2.49 -> 1.51 us per column).
  const char *e = getenv("RAMX_CP_SINGLE_MAX");      // tuning hook
  if (e && atoi(e) > 0) return atoi(e);
  return cp_cells(W, 16) >= 5 ? 32 : 64;
}

// A family takes the most lanes per flank (16, 8, 4, 2) that keep it within one 512-thread workgroup; the classes are
// (lanes per flank, workgroup size): families of up to 16 flanks run 16 lanes per flank in 256 threads.
int ramx_cp_class(int W, int nx, int *K, int *threads)
{
  static const int cls_k[RAMX_CP_NCLASS] = { 16, 16, 8, 4, 2, 0 };
  static const int cls_t[RAMX_CP_NCLASS] = { 256, 512, 512, 512, 512, 0 };
  const char *fk = getenv("RAMX_CP_K");              // test hook: at most this many lanes per flank
  const int force = fk ? atoi(fk) : 0;
  for (int c = 0; c < RAMX_CP_NCLASS; c++)
  {
    const int k = cls_k[c];
    if (k == 0 || (force >= 2 && k > force)) continue;
    int t = cp_max_threads(W, cp_cells(W, k));
    if (t > cls_t[c]) t = cls_t[c];
    if (t == 0 || nx * k > t) continue;
    *K = k;
    *threads = t;
    return c;
  }
  return -1;
}

template <int W, int K>
static int cp_launch(hipStream_t st, int threads, int F, const CPArgs &a)
{
  if constexpr (CpCfg<W, K>::MAXT == 0) return RAMX_ERR_UNSUPPORTED;
  else
  {
    if (threads > CpCfg<W, K>::MAXT) return RAMX_ERR_ARG;
    hipLaunchKernelGGL((ramx_cp_kernel<W, K, false>), dim3(F), dim3(threads), 0, st, a);
    return hipGetLastError() == hipSuccess ? RAMX_OK : RAMX_ERR_HIP;
  }
}

template <int W>
static int cp_launch_w(hipStream_t st, int K, int threads, int F, const CPArgs &a)
{
  switch (K)
  {
    case 16: return cp_launch<W, 16>(st, threads, F, a);
    case 8: return cp_launch<W, 8>(st, threads, F, a);
    case 4: return cp_launch<W, 4>(st, threads, F, a);
    case 2: return cp_launch<W, 2>(st, threads, F, a);
  }
  return RAMX_ERR_UNSUPPORTED;
}

int ramx_cp_launch_families(hipStream_t st, int W, int K, int threads, int F, const CPArgs &a)
{
  if (F <= 0) return RAMX_OK;
  if (threads < 64 || threads > 1024 || (threads & 63)) return RAMX_ERR_ARG;
  switch (W)
  {
    case 14: return cp_launch_w<14>(st, K, threads, F, a);
    case 20: return cp_launch_w<20>(st, K, threads, F, a);
    case 40: return cp_launch_w<40>(st, K, threads, F, a);
    case 80: return cp_launch_w<80>(st, K, threads, F, a);
  }
  return RAMX_ERR_UNSUPPORTED;
}

// ---- device-wide mode ----------------------------------------------------------------------------
// One workgroup (512 threads unless tuned) per CU at most (the vote barrier wants few participants and the launch must be co-resident).
// Lanes per flank: as many as keep the set within `cus` workgroups, but not more than a family of that size would get.
int ramx_cp_device_plan(int W, int n, int cus, int wide, int *K, int *threads, int *blocks, int *vote_wave)
{
  *K = 0; *threads = 0; *blocks = 0; *vote_wave = 0;
  if (n <= 0 || cus <= 0) return RAMX_OK;
  const char *fk = getenv("RAMX_CP_K");
  const int force = fk ? atoi(fk) : 0;
  const char *ft = getenv("RAMX_CP_DEV_THREADS");    // tuning hook: workgroup size of the device-wide launch
  const int ftv = (ft && atoi(ft) >= 64 && (atoi(ft) & 63) == 0) ? atoi(ft) : 0;
  const bool no_vw = getenv("RAMX_CP_NO_VOTE_WAVE") != NULL;   // A/B hook (blocks of 12..21 cells only: shorter blocks have no other mode)
  for (int k = 16; k >= 2; k >>= 1)
  {
    if (force >= 2 && k > force) continue;
    const int cells = cp_cells(W, k);
    const int tmax = cp_max_threads(W, cells);
    if (tmax == 0) continue;
    // Blocks of up to RAMX_CP_SYNCW_MAXC cells: wave 0 of every workgroup is the vote wave and holds no flank.  Four band
    // waves (one per SIMD) when the set then fits the CUs; else the largest workgroup: eight band waves while the saved
    // row lives in registers, seven when it lives in LDS -- and for those, before giving up lanes per flank, the plain
    // order without a vote wave (eight band waves again).
    const bool syncw = cells <= RAMX_CP_SYNCW_MAXC, regs = cells <= RAMX_CP_SYNCW_REGC;
    int cand_t[5], cand_vw[5], nc = 0;
    if (ftv)
    {
      const int vw = syncw && (regs || !no_vw);
      const int tbig = vw ? (regs ? tmax + 64 : tmax) : tmax;
      cand_t[nc] = ftv < (vw ? 128 : 64) ? (vw ? 128 : 64) : (ftv > tbig ? tbig : ftv); cand_vw[nc++] = vw;
    }
    else if (syncw && regs)
    {
      // three band waves and the vote wave: every wave has a SIMD to itself, which shortens the vote wave's chain by more than
      // the larger number of workgroups costs (profiles/r03_ab_threads.log: N = 1,000: 1.43 us per column against 1.58 with
      // four band waves, N = 3,000: 1.50 against 1.62)
      if (!wide) { cand_t[nc] = 256; cand_vw[nc++] = 1; cand_t[nc] = 320; cand_vw[nc++] = 1; }
      // seven band waves + the vote wave before eight + one: the vote wave then shares its SIMD with one band wave instead of
      // two (profiles/r03_ab_k.log: 12,500 flanks, 8 lanes per flank: 2.23 against 2.33 us per column)
      cand_t[nc] = tmax; cand_vw[nc++] = 1;
      cand_t[nc] = tmax + 64; cand_vw[nc++] = 1;
    }
    else if (syncw)
    {
      if (!no_vw) { if (!wide) { cand_t[nc] = 320; cand_vw[nc++] = 1; } cand_t[nc] = tmax; cand_vw[nc++] = 1; }
      cand_t[nc] = tmax; cand_vw[nc++] = 0;
    }
    else { cand_t[nc] = tmax; cand_vw[nc++] = 0; }
    for (int c = 0; c < nc; c++)
    {
      const int t = cand_t[c];
      const int per = (t - (cand_vw[c] ? 64 : 0)) / k;
      if (per <= 0) continue;
      const int nb = (n + per - 1) / per;
      if (nb > cus) continue;
      *K = k; *threads = t; *blocks = nb; *vote_wave = cand_vw[c];
      return RAMX_OK;
    }
  }
  return RAMX_OK;
}

template <int W, int K>
static int cp_launch_dev(hipStream_t st, int threads, int blocks, const CPArgs &a)
{
  if constexpr (CpCfg<W, K>::MAXT == 0) return RAMX_ERR_UNSUPPORTED;
  else
  {
    if (threads > CpCfg<W, K>::MAXT_DEV) return RAMX_ERR_ARG;
    // occupancy of this instantiation at this workgroup size and the CU count: asked once per process and device
    // (the answers do not change; the query costs tens of microseconds per launch)
    static int c_dev = -1, c_cus = 0, c_per_cu[1024 / 64 + 1];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RAMX_ERR_HIP;
    if (dev != c_dev)
    {
      if (hipDeviceGetAttribute(&c_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return RAMX_ERR_HIP;
      for (int i = 0; i <= 1024 / 64; i++) c_per_cu[i] = -1;
      c_dev = dev;
    }
    if (c_per_cu[threads / 64] < 0)
    {
      int q = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, ramx_cp_kernel<W, K, true>, threads, 0) != hipSuccess) return RAMX_ERR_HIP;
      c_per_cu[threads / 64] = q;
    }
    const int per_cu = c_per_cu[threads / 64], cus = c_cus;
    if (per_cu < 1 || blocks > cus) return RAMX_ERR_UNSUPPORTED;      // one workgroup per CU by design
    // plain launch (see prk_launch in ramx_device.hip: the co-residency check is done here, the kernel's barrier is
    // bounded, and a cooperative launch makes rocprofv3-profiled processes crash at exit); RAMX_COOP_LAUNCH=1 restores it
    if (getenv("RAMX_COOP_LAUNCH") != NULL)
    {
      CPArgs copy = a;
      void *args[] = { (void *)&copy };
      hipError_t e = hipLaunchCooperativeKernel((const void *)ramx_cp_kernel<W, K, true>, dim3(blocks), dim3(threads), args, 0, st);
      return e == hipSuccess ? RAMX_OK : RAMX_ERR_HIP;
    }
    hipLaunchKernelGGL((ramx_cp_kernel<W, K, true>), dim3(blocks), dim3(threads), 0, st, a);
    return hipGetLastError() == hipSuccess ? RAMX_OK : RAMX_ERR_HIP;
  }
}

template <int W>
static int cp_launch_dev_w(hipStream_t st, int K, int threads, int blocks, const CPArgs &a)
{
  switch (K)
  {
    case 16: return cp_launch_dev<W, 16>(st, threads, blocks, a);
    case 8: return cp_launch_dev<W, 8>(st, threads, blocks, a);
    case 4: return cp_launch_dev<W, 4>(st, threads, blocks, a);
    case 2: return cp_launch_dev<W, 2>(st, threads, blocks, a);
  }
  return RAMX_ERR_UNSUPPORTED;
}

int ramx_cp_launch_device(hipStream_t st, int W, int K, int threads, int blocks, const CPArgs &a)
{
  if (blocks <= 0 || threads < 64 || threads > 1024 || (threads & 63)) return RAMX_ERR_ARG;
  switch (W)
  {
    case 14: return cp_launch_dev_w<14>(st, K, threads, blocks, a);
    case 20: return cp_launch_dev_w<20>(st, K, threads, blocks, a);
    case 40: return cp_launch_dev_w<40>(st, K, threads, blocks, a);
    case 80: return cp_launch_dev_w<80>(st, K, threads, blocks, a);
  }
  return RAMX_ERR_UNSUPPORTED;
}
