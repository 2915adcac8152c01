// Issue rate of the integer VALU instructions the band kernels are made of, wave64 on gfx950:
// cycles per wave-instruction per SIMD at 1, 2, 4 waves per SIMD (s_memtime around an unrolled block of independent ops).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(int *out, unsigned long long *cyc, int iters, int a0, int a1)
{
  int x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 3) + a0;
  int y = a1 + threadIdx.x, z = a0 - threadIdx.x;
  const unsigned long long msk = 0x5555555555555555ULL * (unsigned)a0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 2) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 4) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 5) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[i]));
        if (OP == 6) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[i]));
        if (OP == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 8) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 11) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 15) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 16) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 17) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 19) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(x[i]) : "v"(y));
        if (OP == 20) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 21) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 22) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 24) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        if (OP == 25) asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(x[i]));
        if (OP == 26) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc");
        if (OP == 27) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "s"(msk));
        if (OP == 28) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(x[i]), "v"(y) : "vcc");
        if (OP == 29) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x[i]) : "v"(y));
        if (OP == 30) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
        if (OP == 31) asm volatile("v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
        if (OP == 32) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 33) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 34) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 35) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 36) asm volatile("v_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 37) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 38) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(y) : "vcc");
        if (OP == 39) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 40) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if (OP == 41) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x[i]));
        if (OP == 42) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "s"(a1));
        if (OP == 43) asm volatile("v_max_i32 %0, %1, %0" : "+v"(x[i]) : "s"(a1));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main()
{
  int *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  const char *nm[] = { "v_add_u32", "v_sub_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_max_i32", "v_min_i32", "v_max_u32", "v_max_f32", "v_min_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_max3_i32", "v_med3_i32", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32", "v_mad_i32_i24", "v_mul_i32_i24", "v_mul_lo_u32", "v_alignbit_b32", "v_perm_b32", "v_bfe_u32", "v_cmp+cndmask_vcc", "v_cndmask_e64_sgpr", "v_cmp_gt_i32", "v_add_u32_sdwa", "v_add_u32_dpp", "v_max_i32_dpp", "v_pk_add_i16", "v_pk_max_i16", "v_max_i16", "v_add_u16", "v_max_f16", "v_pk_max_f16", "v_add_co_u32", "v_sub_f32", "v_xor_b32", "v_add_u32 (literal)", "v_add_u32 (sgpr)", "v_max_i32 (sgpr)" };
  const int iters = 2000;
  for (int op = 0; op < 44; op++)
  {
    printf("%-20s", nm[op]);
    for (int wps = 1; wps <= 4; wps *= 2)       // waves per SIMD: block = 256 * wps threads on one CU
    {
      const int threads = 256 * wps;
      std::vector<unsigned long long> h(threads / 64);
      for (int rep = 0; rep < 2; rep++)
      {
        switch (op)
        {
#define L(o) case o: hipLaunchKernelGGL(k<o>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 3, 5); break;
          L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32) L(33) L(34) L(35) L(36) L(37) L(38) L(39) L(40) L(41) L(42) L(43)
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0;
      for (auto v : h) if (v > mx) mx = v;
      // cycles per wave-instruction per SIMD = wall cycles / (instructions per wave * waves per SIMD)
      printf("  %dw/SIMD: %5.2f cyc/instr/SIMD", wps, (double)mx / ((double)iters * REP * wps));
    }
    printf("\n");
  }
  return 0;
}
