// What does ONE generated field multiplication cost in issue time, operands in registers, no memory?  A chain
// x <- x * y (fe_mul), x <- x^2 (fe_sqr) and x <- x*y + y*x (fe_mul2add), 1 and 4 waves per SIMD, against the sum
// of the instructions' own issue times (tools/probe/vgpr_bank_probe.hip, profiles/r01_ubench_valu_more.txt).
// Build: hipcc --offload-arch=gfx950 -O3 -I crypto_amd/csrc tools/probe/fe_mul_probe.hip -o tools/probe/fe_mul_probe
#include "fe29.cuh"
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template <int KIND>
__global__ void __launch_bounds__(256) kern(i32* out, int iters, u32 seed) {
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  fe x, y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.v[i] = (i32)((seed * 2654435761u + tid * (i + 1)) & FE_MASK);
    y.v[i] = (i32)(((seed ^ 0x9e3779b9u) + tid * 7u * (i + 3)) & FE_MASK);
  }
  x.v[8] &= 0xfffff;
  y.v[8] &= 0xfffff;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (KIND == 0) x = fe_mul(x, y);
      if (KIND == 1) x = fe_sqr(x);
      if (KIND == 2) x = fe_mul2add(x, y, y, x);
    }
  }
  i32 acc = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) acc ^= x.v[i];
  if (acc == 0x12345678) out[tid] = acc;
}

template <int KIND>
static void run(const char* name, int ncu, int wg_per_cu, int instr) {
  i32* out; CK(hipMalloc(&out, 64 << 20));
  const int iters = 400;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern<KIND>, dim3(ncu * wg_per_cu), dim3(256), 0, 0, out, 4, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<KIND>, dim3(ncu * wg_per_cu), dim3(256), 0, 0, out, iters, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double ops_per_simd = (double)iters * 8 * wg_per_cu;
  const double clk = best * 1e-3 * 2.4e9 / ops_per_simd;
  printf("%-12s waves/SIMD=%d  %.3f ms  %.0f clk per operation @2.4GHz = %.2f clk per instruction (%d)\n", name, wg_per_cu, best, clk, clk / instr, instr);
  CK(hipFree(out));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  for (int w : {1, 2, 4}) {
    run<0>("fe_mul", ncu, w, 205);
    run<1>("fe_sqr", ncu, w, 178);
    run<2>("fe_mul2add", ncu, w, 287);
  }
  return 0;
}
