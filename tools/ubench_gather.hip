// Calibration of FETCH_SIZE for k_accumulate's access pattern (MI355X_MICROARCH.md: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern"): every lane
// reads one random, 64-byte-aligned 64-byte record as four 16-byte loads, exactly like aff_load.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
//   ./ubench_gather [log2 records=26] [gathers per thread=16]
// Prints the gather rate (the bandwidth-bound rate tells how many HBM bytes one gather really
// costs); run it under `rocprofv3 --pmc FETCH_SIZE` for the counter's bytes per gather.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint64_t mask, int per,
                                                uint32_t* __restrict__ out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int j = 0; j < per; j++) {
    uint64_t idx = mix64(t * 1315423911ull + (uint64_t)j * 0x9E3779B97F4A7C15ULL) & mask;
    const uint4* p = table + idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  out[t] = acc;
}
int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 26, per = argc > 2 ? atoi(argv[2]) : 16;
  size_t recs = (size_t)1 << lg, threads = (size_t)1 << 22;
  uint4* table;
  uint32_t* out;
  if (hipMalloc(&table, recs * 64) != hipSuccess || hipMalloc(&out, threads * 4) != hipSuccess) return 1;
  hipMemset(table, 1, recs * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather, dim3(threads / 256), dim3(256), 0, 0, table, (uint64_t)recs - 1, per, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double g = (double)threads * per;
  printf("table %.1f GiB, %.0f M gathers of 64 B: %.3f ms -> %.1f G gathers/s, %.2f TB/s at 64 B, %.2f TB/s at 128 B per gather\n",
         recs * 64.0 / (1ull << 30), g / 1e6, best, g / best / 1e6, g * 64 / best / 1e9, g * 128 / best / 1e9);
  return 0;
}
