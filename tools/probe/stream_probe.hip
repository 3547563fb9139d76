// Which pairs of HIP streams let a kernel of one start while the other still has kernels queued behind an
// event record?  Stream A: spin 300 us, record E, spin 300 us.  Stream B: wait E, tiny kernel.  If B's kernel
// ends ~300 us after the start it ran at E; ~600 us: it ran after ALL of A (the two streams are served in
// order, e.g. by one hardware queue).   hipcc --offload-arch=gfx950 -O2 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int NS = argc > 1 ? atoi(argv[1]) : 6;
  std::vector<hipStream_t> st(NS);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e, done_b;
  CK(hipEventCreate(&e));
  CK(hipEventCreate(&done_b));
  const long long us300 = 300 * 100;  // wall_clock64 ticks at 100 MHz
  for (int rep = 0; rep < 2; rep++)
    for (int a = 0; a < NS; a++)
      for (int b = 0; b < NS; b++) {
        if (a == b) continue;
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[a], us300, nullptr);
        CK(hipEventRecord(e, st[a]));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[a], us300, nullptr);
        CK(hipStreamWaitEvent(st[b], e, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[b], 100, nullptr);
        CK(hipStreamSynchronize(st[b]));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st[a]));
        auto t2 = std::chrono::steady_clock::now();
        if (rep == 1)
          printf("A = stream %d, B = stream %d: B done after %6.0f us, A after %6.0f us\n", a, b,
                 std::chrono::duration<double, std::micro>(t1 - t0).count(),
                 std::chrono::duration<double, std::micro>(t2 - t0).count());
      }
  return 0;
}
