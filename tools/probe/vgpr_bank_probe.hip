// Does the VGPR bank of a v_mad_i64_i32's multiplier operands matter?  The generated field multiplication
// keeps its 64-bit accumulator in v[0:1] and takes its two 32-bit factors wherever the register allocator put
// them.  Same dependent chain, four waves per SIMD, factors pinned to chosen physical registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/vgpr_bank_probe.hip -o tools/probe/vgpr_bank_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

#define M16(A, B) \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t" \
  "v_mad_i64_i32 v[0:1], vcc, " A ", " B ", v[0:1]\n\t" "v_mad_i64_i32 v[0:1], vcc, " B ", " A ", v[0:1]\n\t"
#define M128(A, B) M16(A, B) M16(A, B) M16(A, B) M16(A, B) M16(A, B) M16(A, B) M16(A, B) M16(A, B)

#define KERN(NAME, A, B)                                                                          \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters, uint32_t seed) {          \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, x = seed + tid, y = seed * 7u + tid;     \
    uint32_t lo, hi;                                                                              \
    asm volatile("v_mov_b32 " A ", %0\n\tv_mov_b32 " B ", %1\n\tv_mov_b32 v0, %0\n\tv_mov_b32 v1, %1" \
                 :: "v"(x), "v"(y) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11"); \
    for (int it = 0; it < iters; it++)                                                            \
      asm volatile(M128(A, B) ::: "vcc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11"); \
    asm volatile("v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1" : "=v"(lo), "=v"(hi) :: "v0", "v1");       \
    if ((lo ^ hi) == 0x12345678u) out[tid] = lo;                                                  \
  }
// the same chain with a loop body as long as k_accumulate's (2048 multiply-adds = 16 KB of code): does
// instruction fetch keep up when sixteen waves of a CU stream through a body that long?
#define M2048(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) \
                    M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B) M128(A, B)
__global__ void __launch_bounds__(256) k_long(uint32_t* out, int iters, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, x = seed + tid, y = seed * 7u + tid;
  uint32_t lo, hi;
  asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v0, %0\n\tv_mov_b32 v1, %1"
               :: "v"(x), "v"(y) : "v0", "v1", "v2", "v3");
  for (int it = 0; it < iters / 16; it++)
    asm volatile(M2048("v2", "v3") ::: "vcc", "v0", "v1", "v2", "v3");
  asm volatile("v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1" : "=v"(lo), "=v"(hi) :: "v0", "v1");
  if ((lo ^ hi) == 0x12345678u) out[tid] = lo;
}
KERN(k_23, "v2", "v3")     // banks 2, 3: clear of the accumulator's 0, 1
KERN(k_48, "v4", "v8")     // both bank 0 (the accumulator's low half)
KERN(k_45, "v4", "v5")     // banks 0, 1: the accumulator's two
KERN(k_610, "v6", "v10")   // both bank 2
KERN(k_67, "v6", "v7")     // banks 2, 3 again (other registers)
KERN(k_59, "v5", "v9")     // both bank 1 (the accumulator's high half)

typedef void (*kfn)(uint32_t*, int, uint32_t);
static void run(const char* name, kfn k, int ncu, int wg_per_cu) {
  uint32_t* out; CK(hipMalloc(&out, 64 << 20));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(ncu * wg_per_cu), dim3(256), 0, 0, out, 10, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(ncu * wg_per_cu), dim3(256), 0, 0, out, iters, (uint32_t)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  // waves per SIMD = wg_per_cu (256 threads = 4 waves, one per SIMD)
  const double instr_per_simd = (double)iters * 128 * wg_per_cu;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f clk per wave-instr @2.4GHz\n", name, wg_per_cu, best, best * 1e-3 * 2.4e9 / instr_per_simd);
  CK(hipFree(out));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  for (int w : {1, 4}) {
    run("factors v2,v3  (banks 2,3)", k_23, ncu, w);
    run("factors v6,v7  (banks 2,3)", k_67, ncu, w);
    run("factors v4,v8  (banks 0,0)", k_48, ncu, w);
    run("factors v5,v9  (banks 1,1)", k_59, ncu, w);
    run("factors v6,v10 (banks 2,2)", k_610, ncu, w);
    run("factors v4,v5  (banks 0,1)", k_45, ncu, w);
    run("v2,v3, 2048-instruction body", k_long, ncu, w);
  }
  return 0;
}
