// Round-trip cost of a counter barrier between workgroups: all on one XCD (workgroup i of a launch lands on XCD i % 8)
// or spread over the XCDs, polling with an agent-scope load (sc1) or a workgroup-scope load (sc0: L1 bypass, L2 hit).
// Build and run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/microbench/xcd_sync.hip -o /tmp/xcd_sync && /tmp/xcd_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// a poll whose data lands in LDS (lane i -> slot + 16 * i): no destination register, so several can be in flight
__device__ __forceinline__ void poll_to_lds(const void *src, unsigned lds_byte_offset)
{
  unsigned keep;
  lds_byte_offset = __builtin_amdgcn_readfirstlane(lds_byte_offset);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte_offset) : "memory");
}

__global__ void lds_layout_check(const unsigned long long *p, int *bad)
{
  __shared__ __attribute__((aligned(16))) unsigned long long buf[2][64][2];
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&buf[1][0][0]);
  poll_to_lds(p + 2 * threadIdx.x, base);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (buf[1][threadIdx.x][0] != 1000 + 2 * threadIdx.x || buf[1][threadIdx.x][1] != 1001 + 2 * threadIdx.x) atomicAdd(bad, 1);
}

template <int MODE>   // 0: sc1 poll, 1: sc0 poll, 2: sc1 polls through LDS, four in flight
__global__ void sync_kernel(unsigned long long *ctr, int *xcc, int stride, int nwork, int rounds, unsigned long long *clocks, int *fail)
{
  if (blockIdx.x % stride != 0) return;
  const int w = blockIdx.x / stride;
  if (w >= nwork) return;
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) xcc[w] = x & 15;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; r++)
  {
    if (threadIdx.x == 0)
    {
      __hip_atomic_fetch_add(ctr, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long want = (unsigned long long)(r + 1) * nwork;
      unsigned spins = 0;
      if (MODE == 2)
      {
        __shared__ __attribute__((aligned(16))) unsigned long long buf[4][64][2];
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&buf[0][0][0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        poll_to_lds(ctr, base); __builtin_amdgcn_s_sleep(2);
        poll_to_lds(ctr, base + 1024); __builtin_amdgcn_s_sleep(2);
        poll_to_lds(ctr, base + 2048); __builtin_amdgcn_s_sleep(2);
        poll_to_lds(ctr, base + 3072);
        for (unsigned it = 0;; it++)
        {
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          const unsigned long long v = *(volatile unsigned long long *)&buf[it & 3][0][0];
          if (v >= want) break;
          if (++spins > 2000000u) { *fail = 1; break; }
          poll_to_lds(ctr, base + 1024 * (it & 3));
        }
      }
      else for (;;)
      {
        unsigned long long v;
        if (MODE == 0) v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v >= want) break;
        if (++spins > 2000000u) { *fail = 1; break; }
      }
    }
    __syncthreads();
    if (*(volatile int *)fail) break;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && w == 0) clocks[0] = t1 - t0;
}

int main()
{
  unsigned long long *ctr, *clocks; int *xcc, *fail;
  hipMalloc(&ctr, 64); hipMalloc(&clocks, 64); hipMalloc(&xcc, 4096 * 4); hipMalloc(&fail, 4);
  const int rounds = 2000;
  {
    std::vector<unsigned long long> pat(128); for (int i = 0; i < 128; i++) pat[i] = 1000 + i;
    unsigned long long *dp; hipMalloc(&dp, 1024); hipMemcpy(dp, pat.data(), 1024, hipMemcpyHostToDevice);
    hipMemset(fail, 0, 4);
    hipLaunchKernelGGL(lds_layout_check, dim3(1), dim3(64), 0, 0, dp, fail);
    int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("LDS-direct dwordx4 layout check: %d lanes wrong\n", f);
  }
  for (int mode = 0; mode < 3; mode++)
    for (int stride : { 8, 1 })
      for (int nwork : { 2, 8, 32, 128 })
      {
        hipMemset(ctr, 0, 64); hipMemset(fail, 0, 4); hipMemset(xcc, 0xff, 4096 * 4);
        const int blocks = nwork * stride;
        if (mode == 0) hipLaunchKernelGGL(sync_kernel<0>, dim3(blocks), dim3(64), 0, 0, ctr, xcc, stride, nwork, rounds, clocks, fail);
        else if (mode == 2) hipLaunchKernelGGL(sync_kernel<2>, dim3(blocks), dim3(64), 0, 0, ctr, xcc, stride, nwork, rounds, clocks, fail);
        else hipLaunchKernelGGL(sync_kernel<1>, dim3(blocks), dim3(64), 0, 0, ctr, xcc, stride, nwork, rounds, clocks, fail);
        hipDeviceSynchronize();
        unsigned long long c; int f; std::vector<int> hx(nwork);
        hipMemcpy(&c, clocks, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xcc, nwork * 4, hipMemcpyDeviceToHost);
        int mask = 0; for (int v : hx) mask |= 1 << (v & 15);
        printf("poll %s  stride %d  workgroups %2d  xcc mask 0x%02x  fail %d  %.3f us per round\n", mode == 2 ? "sc1 x4 via LDS" : mode ? "sc0" : "sc1", stride, nwork, mask, f,
               (double)c / 100.0 / rounds);   // s_memrealtime: 100 MHz
      }
  return 0;
}
