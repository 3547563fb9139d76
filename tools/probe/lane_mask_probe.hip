// Does a wave64 whose active lanes all sit in one 16-lane group issue its multiply-adds faster than a full wave?
// One wave, a dependent chain of 8192 v_mad_u64_u32, lanes [0, L) active; clocks per multiply-add from
// s_memtime.   hipcc --offload-arch=gfx950 -O2 -o lane_mask_probe lane_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(unsigned long long* out, unsigned long long* clk, int L, unsigned a, unsigned b) {
  if ((int)threadIdx.x >= L) return;
  unsigned long long acc = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 512; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) acc = (unsigned long long)((unsigned)acc ^ a) * b + acc;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) *clk = t1 - t0;
}
int main() {
  unsigned long long *out, *clk, h;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&clk, 8);
  for (int rep = 0; rep < 2; rep++)
    for (int L : {64, 48, 32, 17, 16, 8, 4, 1}) {
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, clk, L, 0x9e3779b9u, 0x85ebca6bu);
      hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      if (rep) printf("active lanes %2d: %.2f clk per multiply-add (+xor)\n", L, (double)h / 8192.0);
    }
  return 0;
}
