// Clocks per dependent instruction for ONE wave (the regime of every latency chain of the library: quad folds,
// generator folds, heavy buckets, small calls): which multiply instructions a lone wave can issue back to back.
//   hipcc --offload-arch=gfx950 -O2 -o lone_wave_ops lone_wave_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void chain(unsigned long long* out, unsigned long long* clk, unsigned a, unsigned b, double da, double db) {
  unsigned long long acc = threadIdx.x + 1;
  unsigned x = threadIdx.x + 3;
  double d = 1.0 + threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 512; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (OP == 0) acc = (unsigned long long)(unsigned)acc * b + acc;          // v_mad_u64_u32
      if (OP == 1) x = x * b + a;                                              // v_mul_lo_u32 (+ add) or v_mad_u32
      if (OP == 2) x = __umulhi(x, b) + a;                                     // v_mul_hi_u32 + add
      if (OP == 3) x = __umul24(x, b) + a;                                     // v_mad_u32_u24
      if (OP == 4) d = __builtin_fma(d, db, da);                               // v_fma_f64
      if (OP == 5) x = (x ^ a) + b;                                            // two simple ops
      if (OP == 6) acc = (unsigned long long)((long long)(int)acc * (int)b + (long long)acc);  // v_mad_i64_i32
      if (OP == 7) x = __umul24(x, b) ^ a;                     // v_mul_u32_u24 + xor
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc + x + (unsigned long long)d;
  if (threadIdx.x == 0) *clk = t1 - t0;
}
int main() {
  unsigned long long *out, *clk, h;
  (void)hipMalloc(&out, 64 * 8);
  (void)hipMalloc(&clk, 8);
  const char* names[8] = {"v_mad_u64_u32", "x*b+a (32-bit)", "mul_hi_u32 + add", "mul24 + add", "v_fma_f64", "xor + add", "v_mad_i64_i32", "mul_u24 ^ a"};
  for (int rep = 0; rep < 2; rep++)
    for (int op = 0; op < 8; op++) {
      switch (op) {
#define L(N) case N: hipLaunchKernelGGL(chain<N>, dim3(1), dim3(64), 0, 0, out, clk, 0x9e3779b9u, 0x85ebca6bu, 0.25, 0.999999); break;
        L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
      }
      (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      if (rep) printf("%-18s %.2f clk per step\n", names[op], (double)h / 8192.0);
    }
  return 0;
}
