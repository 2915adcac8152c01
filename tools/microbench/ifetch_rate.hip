// Is straight-line code limited by instruction fetch?  BODY distinct VALU instructions (no loop inside the body), repeated `iters`
// times by 8 waves per workgroup (two per SIMD), on 1 .. 256 workgroups.  Prints ns per wave-instruction per SIMD; a rate that
// drops when more CUs run the same code, or with the size of the body, is the instruction cache (shared by neighbouring CUs),
// not the VALU.  hipcc -O3 --offload-arch=gfx950 ifetch_rate.hip -o ifetch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int BODY, int KIND>
__global__ __launch_bounds__(512) void k(int *out, unsigned long long *ticks, int iters, int a0, int a1)
{
  int x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 3) + a0;
  int y = a1 + threadIdx.x, z = a0 - threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int r = 0; r < BODY / 8; r++)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        if (KIND == 0) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));                      // 8 bytes
        if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));                                  // 4 bytes
        if (KIND == 2) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x[i]) : "v"(y));   // 8 bytes
        if (KIND == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(y));                                  // 4 bytes, full-rate class
        // dependent chains: every instruction needs the result of the one before it (4), of the one two before it (5), four before it (6)
        if (KIND == 4) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y), "v"(z));
        if (KIND == 5) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i & 1]) : "v"(y), "v"(z));
        if (KIND == 6) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(y), "v"(z));
        if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
        if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i & 1]) : "v"(y));
        // the band's own chain: e = max3(sub + go, e, Pe) + ge  (two dependent instructions per cell), next to independent work
        if (KIND == 9) { asm volatile("v_max3_i32 %0, %1, %0, %2\n\tv_add_u32 %0, %3, %0" : "+v"(x[0]) : "v"(x[1 + (i & 3)]), "v"(z), "s"(a1)); }
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  int s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int BODY, int KIND>
static void run(const char *name, int *out, unsigned long long *ticks)
{
  static const int grids[] = { 1, 196 };
  printf("%-14s body %5d instr:", name, BODY);
  for (int g : grids)
  {
    const int iters = (1 << 22) / BODY;
    hipLaunchKernelGGL((k<BODY, KIND>), dim3(g), dim3(512), 0, 0, out, ticks, 4, 1, 2);       // warm the cache
    hipLaunchKernelGGL((k<BODY, KIND>), dim3(g), dim3(512), 0, 0, out, ticks, iters, 1, 2);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)g * 8);
    hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    // two waves per SIMD each execute iters * BODY instructions; wall_clock64 ticks are 10 ns
    const double ns_avg = 10.0 * (sum / h.size()) / ((double)iters * BODY * 2), ns_max = 10.0 * mx / ((double)iters * BODY * 2);
    printf("  %3d wg: %.3f (max %.3f)", g, ns_avg, ns_max);
  }
  printf("   ns per instruction per SIMD\n");
}

int main()
{
  int *out; unsigned long long *ticks;
  hipMalloc(&out, 256 * 512 * sizeof(int));
  hipMalloc(&ticks, 256 * 8 * sizeof(unsigned long long));
  run<64, 0>("v_max3 (8 B)", out, ticks);   run<1024, 0>("v_max3 (8 B)", out, ticks);   run<4096, 0>("v_max3 (8 B)", out, ticks);
  run<64, 1>("v_add (4 B)", out, ticks);    run<1024, 1>("v_add (4 B)", out, ticks);    run<4096, 1>("v_add (4 B)", out, ticks);
  run<64, 2>("sdwa add (8 B)", out, ticks); run<1024, 2>("sdwa add (8 B)", out, ticks); run<4096, 2>("sdwa add (8 B)", out, ticks);
  run<64, 3>("v_max (4 B)", out, ticks);    run<1024, 3>("v_max (4 B)", out, ticks);    run<4096, 3>("v_max (4 B)", out, ticks);
  run<1024, 4>("max3 chain 1", out, ticks); run<1024, 5>("max3 chain 2", out, ticks);   run<1024, 6>("max3 chain 4", out, ticks);
  run<1024, 7>("add chain 1", out, ticks);  run<1024, 8>("add chain 2", out, ticks);
  run<1024, 9>("max3+add chain (x2 instr)", out, ticks);
  return 0;
}
