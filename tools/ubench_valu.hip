// Issue-rate microbenchmark for the integer-VALU roofline ("verify with a
// register-resident v_mad_u64_u32 issue-rate microbenchmark on the box").
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

#define REP 64          // instructions per accumulator per loop body
#define NACC 8          // independent accumulators

template <int KIND>
__global__ void __launch_bounds__(256) kern(u32* out, int iters, u32 seed) {
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  u64 acc[NACC]; u32 t32[NACC]; double d[NACC];
  u32 a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  double fa = 1.0000001 + tid * 1e-9, fb = 0.9999999;
#pragma unroll
  for (int i = 0; i < NACC; i++) { acc[i] = tid + i; t32[i] = tid ^ i; d[i] = i + 0.5; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / NACC * NACC / NACC; r++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 3) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(t32[i]) : "v"(a) : "vcc");
        if (KIND == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(fa), "v"(fb));
        if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % NACC]));
        if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                                    : "+v"(acc[i]), "+v"(t32[i]) : "v"(a), "v"(b) : "vcc");
        if (KIND == 7) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(t32[i]) : "v"(a), "v"(b));
        if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");  // dependent chain
        if (KIND == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(t32[i]) : "v"(a), "v"(b));
        if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(t32[i]) : "v"(t32[(i + 1) % NACC]));
        if (KIND == 12) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (KIND == 13) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[i]));
        if (KIND == 14) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(t32[i]) : "v"(a));
        if (KIND == 15) asm volatile("v_and_b32 %0, %1, %0" : "+v"(t32[i]) : "v"(a));
        if (KIND == 16) asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(t32[i]));
        if (KIND == 17) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 18) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (KIND == 19) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(seed) : "vcc");
        if (KIND == 20) asm volatile("v_bfe_u32 %0, %0, 0, 29" : "+v"(t32[i]));
        if (KIND == 21) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(t32[i]));
        if (KIND == 22) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t32[i]) : "v"(a) : "vcc");
        if (KIND == 23) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 24) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_and_b32 %3, %4, %3" : "+v"(acc[i]), "+v"(t32[i]) : "v"(a), "v"(b), "v"(a) : "vcc");
        if (KIND == 25) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(t32[i]) : "v"(a));
        if (KIND == 26) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(t32[i]) : "v"(a), "v"(b));
      }
    }
  }
  u32 x = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) x ^= (u32)acc[i] ^ (u32)(acc[i] >> 32) ^ t32[i] ^ (u32)__double_as_longlong(d[i]);
  if (x == 0x12345678u) out[tid] = x;   // keep live, (almost) never stores
}

template <int KIND>
double run(const char* name, int blocks_per_cu, int ncu, int instr_per_body) {
  u32* out; CK(hipMalloc(&out, 4 << 20));
  int iters = 2000;
  dim3 g(ncu * blocks_per_cu), b(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern<KIND>, g, b, 0, 0, out, 10, 1u);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<KIND>, g, b, 0, 0, out, iters, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  double ms = ts[2];
  double wave_instr = (double)g.x * 4 /*waves per block*/ * iters * (double)(REP / NACC * NACC) * instr_per_body;
  double lane_ops = wave_instr * 64;
  double per_simd_clk = lane_ops / (ms * 1e-3) / (ncu * 4.0) / 2.4e9;
  printf("%-28s waves/SIMD=%d  %8.3f ms  %9.3e lane-ops/s  %6.2f lanes/clk/SIMD@2.4GHz  (%5.2f clk per wave-instr)\n",
         name, blocks_per_cu, ms, lane_ops / (ms * 1e-3), per_simd_clk, 64.0 / per_simd_clk);
  CK(hipFree(out));
  return lane_ops / (ms * 1e-3);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  printf("device %s  CUs=%d  clock=%d kHz\n", p.name, ncu, p.clockRate);
  for (int w : {4}) {
    run<0>("v_mad_u64_u32", w, ncu, 1);
    run<9>("v_mad_u64_u32 (dep chain)", w, ncu, 1);
    run<6>("v_mad_u64_u32+v_addc", w, ncu, 1);
    run<1>("v_mul_lo_u32", w, ncu, 1);
    run<2>("v_mul_hi_u32", w, ncu, 1);
    run<7>("v_mad_u32_u24", w, ncu, 1);
    run<3>("v_addc_co_u32", w, ncu, 1);
    run<8>("v_add_u32", w, ncu, 1);
    run<10>("v_add3_u32", w, ncu, 1);
    run<11>("v_mov_b32", w, ncu, 1);
    run<5>("v_lshl_add_u64", w, ncu, 1);
    run<4>("v_fma_f64", w, ncu, 1);
    run<12>("v_mad_i64_i32", w, ncu, 1);
    run<19>("v_mad_i64_i32 (sgpr src)", w, ncu, 1);
    run<18>("v_mad_i64_i32 + s_nop 0", w, ncu, 1);
    run<24>("v_mad_i64_i32 + v_and", w, ncu, 1);
    run<13>("v_ashrrev_i64", w, ncu, 1);
    run<14>("v_alignbit_b32", w, ncu, 1);
    run<15>("v_and_b32", w, ncu, 1);
    run<16>("v_ashrrev_i32", w, ncu, 1);
    run<17>("v_sub_u32", w, ncu, 1);
    run<20>("v_bfe_u32", w, ncu, 1);
    run<21>("v_lshlrev_b32", w, ncu, 1);
    run<22>("v_cndmask_b32", w, ncu, 1);
    run<23>("v_mul_i32_i24", w, ncu, 1);
    run<25>("v_pk_mul_lo_u16", w, ncu, 1);
    run<26>("v_dot4_i32_i8", w, ncu, 1);
  }
  return 0;
}
