// ramx_kernels_stream.h -- streaming kernels: one launch per column (ramx_column_kernel) and the streaming family kernel of batch mode
// (device code of libramx; included by ramx_device.hip only -- one translation unit so that everything inlines)
#pragma once

#include "ramx_kernels_common.h"

template <bool INIT, bool CHAIN, int BLOCK, bool DBG = false>
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
    for (int c = 0; c < 4; c++) { D.eA[c] = NEG; D.bestA[c] = NEG; D.jA[c] = 0; D.gA0[c] = NEG; D.gAL[c] = NEG; }
    D.gF0 = NEG; D.gFL = NEG;
    D.band = (DBG && CHAIN && a.dbg_band != NULL) ? a.dbg_band + (size_t)n * 8 * B : nullptr;
    int high = 0, pos = 0;
    // wave-uniform choice: do all 64 flanks cover every cell of both rows?  (steps 0..B)
    const bool all_in = !INIT && !DBG && __all((n >= a.Nx) || ((jlo <= 0) && (jhi >= B)));   // padding lanes: all-N stream, masked vote
    if (all_in) run_band<INIT, false, CHAIN>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2);
    else run_band<INIT, true, CHAIN, DBG>(a, r, s_tab, Sin, Sout, bp, jlo, jhi, D, high, pos, buf, far, w0, w1, w2,
                                          (DBG && a.dbg_codes != NULL) ? a.dbg_codes + (size_t)n * B : nullptr);
    if (DBG && !INIT && a.dbg_best != NULL) a.dbg_best[n] = make_int2(D.bestF, r + D.jbest - W);   // best_row_score, *max_score_sequence_idx
    if (DBG && CHAIN && a.dbg_cand != NULL)
    {
      int *o = a.dbg_cand + (size_t)n * 16;
#pragma unroll
      for (int c = 0; c < 4; c++) { o[c] = D.bestA[c]; o[4 + c] = D.jA[c]; o[8 + c] = D.gA0[c]; o[12 + c] = D.gAL[c]; }
      a.dbg_gap[n] = make_int2(D.gF0, D.gFL);
    }
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
        const bool all_in = __all(!active || ((jlo <= 0) && (jhi >= B)));                // padding lanes: all-N stream, masked vote
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

