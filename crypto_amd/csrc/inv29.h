// x^-1 mod p for the BN256 base field in radix 2^29, by the optimized binary GCD of T. Pornin ("Optimized
// Binary GCD for Modular Inversion", 2020, Algorithm 2) with k = 30: an outer iteration runs k - 1 = 29
// steps of the binary GCD on 60-bit approximations of a and b (their 29 low bits glued to the 31 bits
// below the top of the longer one), collects the steps in four update factors |f|, |g| <= 2^29, and applies
// them to the full values with one pass each: (a, b) exactly (the low limb of the sums is zero), (u, v) modulo
// p with one Montgomery step -- radix 2^29 makes "divide by 2^(k-1)" a one-limb shift, and the factors are
// operands of the 32 x 32 + 64 multiply-add.  ceil((2 * 256 - 1) / 29) = 18 outer iterations, ~20 000
// instructions, no data-dependent branch; a^(p-2) is 383 field multiplications, ~78 000 instructions.  The
// normalisation of a handful of points (a prover's small rounds) is one inversion's latency on a lone wave:
// 173 -> 60 us per k_batch_to_affine call.
//
// Plain C++ with no dependency on the device headers: bn256msm.hip compiles it for the GPU (INV29_FN =
// __device__), tests/inv29_host_test.cpp for the CPU, where x * inverse(x) = 1 is checked in the library's host
// field arithmetic on edge values and millions of random ones (6 * 10^7 once; 3 * 10^5 in every CPU test run).
// The caller checks the result's flag and falls back to a^(p-2): inverse() returns false when b did not reach 1
// (x = 0, or, against the paper's proof, an input that needs more iterations).
#pragma once
#include <stdint.h>

#ifndef INV29_FN
#define INV29_FN inline
#endif

namespace inv29 {

// p in radix 2^29 (fe29.cuh: FE_P_LIMBS) and -p^-1 mod 2^29 (FE_NP)
#define INV29_P_LIMBS \
  { 0x1e089667, 0x02e56362, 0x0d6d6786, 0x1711a241, 0x0dc21ee5, 0x165c30c2, 0x1fe6a9bf, 0x1c695470, 0x008fb501 }
#define INV29_NP 0x1f17daa9u
#define INV29_MASK 0x1fffffff

INV29_FN constexpr int32_t p_limb(int i) {
  constexpr int32_t p[9] = INV29_P_LIMBS;
  return p[i];
}

// -x for limbs with a signed top limb (lower limbs end in [0, 2^29))
INV29_FN void negate(int32_t x[9]) {
  int32_t c = 0;
  for (int i = 0; i < 8; i++) {
    int32_t t = c - x[i];
    x[i] = t & INV29_MASK;
    c = t >> 29;
  }
  x[8] = c - x[8];
}

// the 60-bit approximation of x for the current outer iteration: low 29 bits | 31 bits from position `lt` of the
// top limb triple (xj, xj1, xj2 = limbs j, j-1, j-2)
INV29_FN uint64_t approx(int32_t x0, int32_t xj, int32_t xj1, int32_t xj2, int lt) {
  uint64_t hi = ((uint64_t)(uint32_t)xj << 29) | (uint32_t)xj1;                 // 58 bits
  uint64_t y = (hi << 2) | ((uint32_t)xj2 >> 27);                              // limbs j .. j-2 without their low 27 bits
  return ((y >> lt) << 29) | (uint32_t)x0;
}

// x: canonical limbs of 0 <= x < p.  out: canonical limbs of x^-1 mod p.  false: not inverted (see above).
INV29_FN bool inverse(int32_t out[9], const int32_t x[9]) {
  int32_t a[9], b[9], u[9], v[9];
  for (int i = 0; i < 9; i++) {
    a[i] = x[i];
    b[i] = p_limb(i);
    u[i] = 0;
    v[i] = 0;
  }
  u[0] = 1;
  for (int it = 0; it < 18; it++) {
    // the top limb that a or b still uses, never below limb 2 (values of at most 60 bits are their own approximation)
    int32_t aj = a[2], aj1 = a[1], aj2 = a[0], bj = b[2], bj1 = b[1], bj2 = b[0];
    bool low = true;
    for (int i = 3; i < 9; i++) {
      const bool nz = (a[i] | b[i]) != 0;
      aj = nz ? a[i] : aj;
      aj1 = nz ? a[i - 1] : aj1;
      aj2 = nz ? a[i - 2] : aj2;
      bj = nz ? b[i] : bj;
      bj1 = nz ? b[i - 1] : bj1;
      bj2 = nz ? b[i - 2] : bj2;
      low = nz ? false : low;
    }
    const uint32_t top = (uint32_t)(aj | bj);
    int lt = top ? 32 - __builtin_clz(top) : 0;
    lt = (low && lt < 2) ? 2 : lt;
    uint64_t xa = approx(a[0], aj, aj1, aj2, lt), xb = approx(b[0], bj, bj1, bj2, lt);
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    for (int s = 0; s < 29; s++) {
      const bool odd = (xa & 1u) != 0;
      const bool sw = odd && xa < xb;
      const uint64_t ta = sw ? xb : xa, tb = sw ? xa : xb;
      const int32_t tf0 = sw ? f1 : f0, tf1 = sw ? f0 : f1, tg0 = sw ? g1 : g0, tg1 = sw ? g0 : g1;
      xa = (odd ? ta - tb : ta) >> 1;
      xb = tb;
      f0 = odd ? tf0 - tf1 : tf0;
      g0 = odd ? tg0 - tg1 : tg0;
      f1 = tf1 * 2;
      g1 = tg1 * 2;
    }
    // (a, b) <- ((a f0 + b g0) / 2^29, (a f1 + b g1) / 2^29): exact, the low limb of both sums is zero
    int32_t na[9], nb[9];
    int64_t ca = 0, cb = 0;
    for (int i = 0; i < 9; i++) {
      ca += (int64_t)a[i] * f0 + (int64_t)b[i] * g0;
      cb += (int64_t)a[i] * f1 + (int64_t)b[i] * g1;
      if (i > 0) {
        na[i - 1] = (int32_t)ca & INV29_MASK;
        nb[i - 1] = (int32_t)cb & INV29_MASK;
      }
      ca >>= 29;
      cb >>= 29;
    }
    na[8] = (int32_t)ca;
    nb[8] = (int32_t)cb;
    if (na[8] < 0) {
      negate(na);
      f0 = -f0;
      g0 = -g0;
    }
    if (nb[8] < 0) {
      negate(nb);
      f1 = -f1;
      g1 = -g1;
    }
    // (u, v) <- the same combinations divided by 2^29 modulo p (one Montgomery step), kept in (-p, p)
    int32_t nu[9], nv[9];
    const int32_t mu = (int32_t)((((uint32_t)u[0] * (uint32_t)f0 + (uint32_t)v[0] * (uint32_t)g0) * INV29_NP) & INV29_MASK);
    const int32_t mv = (int32_t)((((uint32_t)u[0] * (uint32_t)f1 + (uint32_t)v[0] * (uint32_t)g1) * INV29_NP) & INV29_MASK);
    int64_t cu = 0, cv = 0;
    for (int i = 0; i < 9; i++) {
      cu += (int64_t)u[i] * f0 + (int64_t)v[i] * g0 + (int64_t)mu * p_limb(i);
      cv += (int64_t)u[i] * f1 + (int64_t)v[i] * g1 + (int64_t)mv * p_limb(i);
      if (i > 0) {
        nu[i - 1] = (int32_t)cu & INV29_MASK;
        nv[i - 1] = (int32_t)cv & INV29_MASK;
      }
      cu >>= 29;
      cv >>= 29;
    }
    nu[8] = (int32_t)cu;
    nv[8] = (int32_t)cv;
    // results are in (-p, 2p): back below p
    int32_t du[9], dv[9], bu = 0, bv = 0;
    for (int i = 0; i < 8; i++) {
      int32_t t = nu[i] - p_limb(i) + bu;
      du[i] = t & INV29_MASK;
      bu = t >> 29;
      t = nv[i] - p_limb(i) + bv;
      dv[i] = t & INV29_MASK;
      bv = t >> 29;
    }
    du[8] = nu[8] - p_limb(8) + bu;
    dv[8] = nv[8] - p_limb(8) + bv;
    const bool su = du[8] >= 0, sv = dv[8] >= 0;
    for (int i = 0; i < 9; i++) {
      a[i] = na[i];
      b[i] = nb[i];
      u[i] = su ? du[i] : nu[i];
      v[i] = sv ? dv[i] : nv[i];
    }
  }
  int32_t rest = b[0] ^ 1;
  for (int i = 1; i < 9; i++) rest |= b[i];
  // v in (-p, p) -> [0, p)
  int32_t c = 0;
  const bool negv = v[8] < 0;
  for (int i = 0; i < 8; i++) {
    int32_t t = v[i] + (negv ? p_limb(i) : 0) + c;
    out[i] = t & INV29_MASK;
    c = t >> 29;
  }
  out[8] = v[8] + (negv ? p_limb(8) : 0) + c;
  return rest == 0;
}

}  // namespace inv29
