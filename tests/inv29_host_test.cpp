// CPU check of crypto_amd/csrc/inv29.h (the inversion the device's normalisation kernels use, compiled here for
// the host from the same source): inverse(x) * x == 1 in Fp (host_g1.h arithmetic) for edge values and N random
// ones; prints "ok <count>" or the first failure.   g++ -O2 -o inv29_host_test inv29_host_test.cpp && ./inv29_host_test N
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../crypto_amd/csrc/host_g1.h"
#include "../crypto_amd/csrc/inv29.h"

using namespace hostg1;

static void to_limbs(int32_t l[9], const uint64_t w[4]) {
  for (int i = 0; i < 9; i++) {
    int bit = 29 * i, j = bit / 64, s = bit % 64;
    uint64_t x = w[j] >> s;
    if (s > 35 && j + 1 < 4) x |= w[j + 1] << (64 - s);
    l[i] = (int32_t)(x & 0x1fffffff);
  }
}
static void from_limbs(uint64_t w[4], const int32_t l[9]) {
  unsigned __int128 acc = 0;
  int have = 0, k = 0;
  for (int i = 0; i < 4; i++) w[i] = 0;
  for (int i = 0; i < 9; i++) {
    acc |= (unsigned __int128)(uint32_t)l[i] << have;
    have += 29;
    while (have >= 64 && k < 4) {
      w[k++] = (uint64_t)acc;
      acc >>= 64;
      have -= 64;
    }
  }
  if (k < 4) w[k] = (uint64_t)acc;
}

static int check(const uint64_t w[4], long idx) {
  int32_t x[9], y[9];
  to_limbs(x, w);
  bool ok = inv29::inverse(y, x);
  bool zero = (w[0] | w[1] | w[2] | w[3]) == 0;
  if (zero) {
    if (ok) { printf("FAIL %ld: inverse(0) reported success\n", idx); return 1; }
    return 0;
  }
  if (!ok) { printf("FAIL %ld: did not converge for %016llx%016llx%016llx%016llx\n", idx, (unsigned long long)w[3], (unsigned long long)w[2], (unsigned long long)w[1], (unsigned long long)w[0]); return 1; }
  for (int i = 0; i < 8; i++)
    if (y[i] < 0 || y[i] > 0x1fffffff) { printf("FAIL %ld: limb %d not normalised\n", idx, i); return 1; }
  uint64_t yw[4];
  from_limbs(yw, y);
  if (y[8] < 0 || raw_geq_p(yw)) { printf("FAIL %ld: result not below p\n", idx); return 1; }
  Fp a = from_raw(w), b = from_raw(yw);
  if (!equal(mul(a, b), kOne)) { printf("FAIL %ld: x * inverse(x) != 1 for %016llx%016llx%016llx%016llx\n", idx, (unsigned long long)w[3], (unsigned long long)w[2], (unsigned long long)w[1], (unsigned long long)w[0]); return 1; }
  return 0;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 1000000;
  long cnt = 0;
  // edge values: 0, small, powers of two and their neighbours, p - small
  for (uint64_t s = 0; s < 70; s++) {
    uint64_t w[4] = {s, 0, 0, 0};
    if (check(w, cnt++)) return 1;
    uint64_t q[4] = {kP[0] - 1 - s, kP[1], kP[2], kP[3]};
    if (check(q, cnt++)) return 1;
  }
  for (int bit = 0; bit < 256; bit++)
    for (int d = -1; d <= 1; d++) {
      uint64_t w[4] = {0, 0, 0, 0};
      w[bit / 64] = 1ull << (bit % 64);
      if (d == 1) w[0] |= 1;
      if (d == -1) {  // 2^bit - 1
        for (int i = 0; i < 4; i++) w[i] = 0;
        for (int i = 0; i < bit; i++) w[i / 64] |= 1ull << (i % 64);
      }
      if (raw_geq_p(w)) continue;
      if (check(w, cnt++)) return 1;
    }
  std::mt19937_64 g(0x1297);
  for (long i = 0; i < n; i++) {
    uint64_t w[4] = {g(), g(), g(), g()};
    int shape = (int)(i % 11);
    if (shape == 1) w[3] = 0;                       // 192-bit
    if (shape == 2) w[3] = w[2] = 0;                // 128-bit
    if (shape == 3) w[3] = w[2] = w[1] = 0;         // 64-bit
    if (shape == 4) { w[3] &= 0xffff; }             // sparse top
    if (shape == 5) { w[0] &= ~0xfffffffull; }      // many trailing zeros
    if (shape == 6) { w[0] = w[1] = 0; }            // 128 trailing zeros
    if (shape == 7) { w[1] = w[2] = ~0ull; }        // long runs of ones
    w[3] %= kP[3];                                   // below p
    if (check(w, cnt++)) return 1;
  }
  printf("ok %ld\n", cnt);
  return 0;
}
