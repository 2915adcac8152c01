// ramx_packed.hip -- third translation unit of libramx's device code: the packed-row persistent kernel (ramx_kernels_packed.h)
// and its launcher.  Kept apart from ramx_device.hip so that the translation units compile side by side.
#define RAMX_SECONDARY_TU 1
#include "ramx_kernels_packed.h"

#include <stdlib.h>

#define PK_REBASE 6000

int ramx_pk_plan(int W, int go, int ge, const int (&tab)[RAMX_NCLASS][4], int *spread, int *rebase)
{
  *spread = 0; *rebase = 0;
  if (!(W == 14 || W == 20 || W == 40 || W == 80) || go > 0 || ge > 0 || getenv("RAMX_NO_PK") != NULL) return 0;
  long long P = 0, mn = 0;
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
    {
      if (tab[c][k] > P) P = tab[c][k];
      if (-(long long)tab[c][k] > mn) mn = -(long long)tab[c][k];
    }
  const long long GO = -(long long)go, GE = -(long long)ge;
  // an in-bounds cell lies at most `sp` below its row's best cell (ramx_kernels_packed.h); the base lags the best cell by at most
  // PK_REBASE + 16 max(P, mn); sub + go of the lowest cell needs mn + GO more, e of it GO + GE
  const long long sp = 3LL * W * (P + mn + GE) + GO + (long long)W * GE;
  const long long lag = PK_REBASE + 16 * (P > mn ? P : mn);
  if (sp + lag + mn + GO + GE + 64 > 32000) return 0;
  if (lag + P + 64 > 32000) return 0;
  *spread = (int)sp; *rebase = PK_REBASE;
  return 1;
}

template <int W, int BLOCK>
static int pk_capacity(int *out)
{
  static int cached = -1;              // (one device per process)
  if (cached >= 0) { *out = cached; return RAMX_OK; }
  int per_cu = 0, dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RAMX_ERR_HIP;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return RAMX_ERR_HIP;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ramx_packed_kernel<W, BLOCK>, BLOCK, 0) != hipSuccess) return RAMX_ERR_HIP;
  // One workgroup per CU (one barrier participant per CU).  Two 256-thread workgroups per CU instead of one of 512 threads (each
  // with one wave per SIMD, drifting apart so that one's barriers fall into the other's band; RAMX_PK_WG_PER_CU=2) were measured
  // at the bench size: aligned phase 7.12 against 7.14 us per column, capped tail 5.70 against 5.53 -- twice the tickets per vote
  // cost more than the drift gave (profiles/r04_notes.md)
  int lim = 1;
  if (const char *e = getenv("RAMX_PK_WG_PER_CU")) lim = atoi(e) >= 2 && BLOCK == 256 ? 2 : 1;
  if (per_cu > lim) per_cu = lim;
  *out = per_cu * cus;
  cached = *out;
  return RAMX_OK;
}

template <int W>
static int pk_shape(int tiles, int *block, int *blocks)
{
  int cap = 0, rc;
  *block = 0; *blocks = 0;
  // one wave per SIMD while the flank set allows it (four band waves and a vote wave without flanks: 320 threads;
  // RAMX_PK_NO_VW=1: the four band waves alone, A/B), two above
  // (W = 80: the row's 162 registers leave no room for a second wave on SIMD 0 -- its 256-thread workgroups have the register
  // file of a SIMD per wave -- so no vote wave there)
  const bool vw = W <= 40 && getenv("RAMX_PK_NO_VW") == NULL;
  if constexpr (W <= 40) rc = vw ? pk_capacity<W, 320>(&cap) : pk_capacity<W, 256>(&cap);
  else rc = pk_capacity<W, 256>(&cap);
  if (rc != RAMX_OK) return rc;
  if (getenv("RAMX_PK_NO_256") != NULL) cap = 0;
  if ((tiles + 3) / 4 <= cap) { *block = vw ? 320 : 256; *blocks = (tiles + 3) / 4; return RAMX_OK; }
  if constexpr (W <= 40)       // W = 80: 162 row registers leave no room for a second wave per SIMD (it would run from scratch memory)
  {
    if ((rc = pk_capacity<W, 512>(&cap)) != RAMX_OK) return rc;
    if ((tiles + 7) / 8 <= cap) { *block = 512; *blocks = (tiles + 7) / 8; }
  }
  return RAMX_OK;
}

// workgroups of `block` threads the device holds at once (what ramx_pk_shape compared `blocks` with)
int ramx_pk_capacity(int W, int block, int *cap)
{
  *cap = 0;
  if (block == 256)
    switch (W) { case 14: return pk_capacity<14, 256>(cap); case 20: return pk_capacity<20, 256>(cap); case 40: return pk_capacity<40, 256>(cap); case 80: return pk_capacity<80, 256>(cap); }
  if (block == 320)
    switch (W) { case 14: return pk_capacity<14, 320>(cap); case 20: return pk_capacity<20, 320>(cap); case 40: return pk_capacity<40, 320>(cap); }
  if (block == 512)
    switch (W) { case 14: return pk_capacity<14, 512>(cap); case 20: return pk_capacity<20, 512>(cap); case 40: return pk_capacity<40, 512>(cap); }
  return RAMX_OK;
}

int ramx_pk_shape(int W, int tiles, int *block, int *blocks)
{
  switch (W)
  {
    case 14: return pk_shape<14>(tiles, block, blocks);
    case 20: return pk_shape<20>(tiles, block, blocks);
    case 40: return pk_shape<40>(tiles, block, blocks);
    case 80: return pk_shape<80>(tiles, block, blocks);
  }
  *block = 0; *blocks = 0;
  return RAMX_OK;
}

template <int W>
static int pk_launch(hipStream_t st, int block, int blocks, const PKArgs &a)
{
  const int grid = blocks + (a.xblock != 0 ? 1 : 0);       // + the exchanger (ramx_kernels_packed.h)
  if (block == 256) hipLaunchKernelGGL((ramx_packed_kernel<W, 256>), dim3(grid), dim3(256), 0, st, a);
  else if (block == 320)
  {
    if constexpr (W <= 40) hipLaunchKernelGGL((ramx_packed_kernel<W, 320>), dim3(grid), dim3(320), 0, st, a);
    else return RAMX_ERR_ARG;
  }
  else if (block == 512)
  {
    if constexpr (W <= 40) hipLaunchKernelGGL((ramx_packed_kernel<W, 512>), dim3(grid), dim3(512), 0, st, a);
    else return RAMX_ERR_ARG;
  }
  else return RAMX_ERR_ARG;
  return hipGetLastError() == hipSuccess ? RAMX_OK : RAMX_ERR_HIP;
}

int ramx_pk_launch(hipStream_t st, int W, int block, int blocks, const PKArgs &a)
{
  switch (W)
  {
    case 14: return pk_launch<14>(st, block, blocks, a);
    case 20: return pk_launch<20>(st, block, blocks, a);
    case 40: return pk_launch<40>(st, block, blocks, a);
    case 80: return pk_launch<80>(st, block, blocks, a);
  }
  return RAMX_ERR_UNSUPPORTED;
}
