// Host-side BN256 G1 arithmetic used by the library for the short serial tails that would
// be latency-bound on the GPU: the Horner combination of the per-bit bucket sums, the
// window combination, the final normalisation to the reference's Marshal() encoding
// (64-byte big-endian affine, infinity = zeros; cf. go/bp/main.go:220,278 which compare
// Marshal() bytes) and the combination of per-GPU partial sums.
//
// This is product code: it does not share anything with oracle/.
#pragma once
#include <cstdint>
#include <cstring>

namespace hostg1 {

typedef unsigned __int128 u128;

struct Fp {  // Montgomery form, R = 2^256, 4 x 64-bit little-endian limbs, value in [0, p)
  uint64_t l[4];
};

static const uint64_t kP[4] = {0x185cac6c5e089667ULL, 0xee5b88d120b5b59eULL, 0xaa6fecb86184dc21ULL,
                               0x8fb501e34aa387f9ULL};
static const Fp kR2 = {{0x9c21c3ff7e444f56ULL, 0x409ed151b2efb0c2ULL, 0x0c6dc37b80fb1651ULL,
                        0x7c36e0e62c2380b7ULL}};
static const Fp kOne = {{0xe7a35393a1f76999ULL, 0x11a4772edf4a4a61ULL, 0x559013479e7b23deULL,
                         0x704afe1cb55c7806ULL}};
static const uint64_t kNInv = 0x2387f9007f17daa9ULL;  // -p^-1 mod 2^64

inline bool is_zero(const Fp& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool equal(const Fp& a, const Fp& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}
inline bool raw_geq_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] != kP[i]) return a[i] > kP[i];
  }
  return true;
}
inline Fp add(const Fp& a, const Fp& b) {
  Fp r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if ((uint64_t)c || raw_geq_p(r.l)) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)r.l[i] - kP[i] - br;
      r.l[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  return r;
}
inline Fp sub(const Fp& a, const Fp& b) {
  Fp r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.l[i] - b.l[i] - br;
    r.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.l[i] + kP[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
// CIOS Montgomery product, fully unrolled (p's top limb is >= 2^63, so the extra carry word stays).
inline Fp mul(const Fp& a, const Fp& b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define HOSTG1_ROUND(bi)                                      \
  {                                                           \
    u128 cur = (u128)a.l[0] * (bi) + t0;                      \
    uint64_t u0 = (uint64_t)cur;                              \
    cur = (u128)a.l[1] * (bi) + t1 + (uint64_t)(cur >> 64);   \
    uint64_t u1 = (uint64_t)cur;                              \
    cur = (u128)a.l[2] * (bi) + t2 + (uint64_t)(cur >> 64);   \
    uint64_t u2 = (uint64_t)cur;                              \
    cur = (u128)a.l[3] * (bi) + t3 + (uint64_t)(cur >> 64);   \
    uint64_t u3 = (uint64_t)cur;                              \
    cur = (u128)t4 + (uint64_t)(cur >> 64);                   \
    uint64_t u4 = (uint64_t)cur, u5 = (uint64_t)(cur >> 64);  \
    uint64_t m = u0 * kNInv;                                  \
    cur = (u128)m * kP[0] + u0;                               \
    cur = (u128)m * kP[1] + u1 + (uint64_t)(cur >> 64);       \
    t0 = (uint64_t)cur;                                       \
    cur = (u128)m * kP[2] + u2 + (uint64_t)(cur >> 64);       \
    t1 = (uint64_t)cur;                                       \
    cur = (u128)m * kP[3] + u3 + (uint64_t)(cur >> 64);       \
    t2 = (uint64_t)cur;                                       \
    cur = (u128)u4 + (uint64_t)(cur >> 64);                   \
    t3 = (uint64_t)cur;                                       \
    t4 = u5 + (uint64_t)(cur >> 64);                          \
  }
  HOSTG1_ROUND(b.l[0])
  HOSTG1_ROUND(b.l[1])
  HOSTG1_ROUND(b.l[2])
  HOSTG1_ROUND(b.l[3])
#undef HOSTG1_ROUND
  Fp r = {{t0, t1, t2, t3}};
  if (t4 || raw_geq_p(r.l)) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)r.l[i] - kP[i] - br;
      r.l[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  return r;
}
inline Fp sqr(const Fp& a) { return mul(a, a); }
inline Fp from_raw(const uint64_t w[4]) {  // w < p, plain integer -> Montgomery
  Fp t = {{w[0], w[1], w[2], w[3]}};
  return mul(t, kR2);
}
inline void to_raw(uint64_t w[4], const Fp& a) {
  Fp one = {{1, 0, 0, 0}};
  Fp t = mul(a, one);
  memcpy(w, t.l, 32);
}
inline Fp inv_fermat(const Fp& a) {  // a^(p-2): 16 us here; kept as the check of inv() in tests
  uint64_t e[4] = {kP[0] - 2, kP[1], kP[2], kP[3]};
  Fp acc = kOne;
  for (int i = 255; i >= 0; i--) {
    acc = sqr(acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a);
  }
  return acc;
}
// x / 2 mod p for x in [0, p)
inline void halve_mod_p(uint64_t x[4]) {
  uint64_t top = 0;
  if (x[0] & 1) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)x[i] + kP[i];
      x[i] = (uint64_t)c;
      c >>= 64;
    }
    top = (uint64_t)c;
  }
  for (int i = 0; i < 3; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
  x[3] = (x[3] >> 1) | (top << 63);
}
inline bool raw_geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
inline void raw_sub(uint64_t a[4], const uint64_t b[4]) {  // a -= b, a >= b
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
// 1 / a by the binary extended Euclidean algorithm for an odd modulus (the invariants u = x1 A, v = x2 A
// mod p with A = a R the representation itself): ~1.5 us against 16 us for a^(p-2), and the one inversion
// of marshal() is on every call's critical path.  inv(0) = 0, as a^(p-2) gives.
inline Fp inv(const Fp& a) {
  if (is_zero(a)) return a;
  uint64_t u[4] = {a.l[0], a.l[1], a.l[2], a.l[3]}, v[4] = {kP[0], kP[1], kP[2], kP[3]};
  Fp x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  auto is_one = [](const uint64_t w[4]) { return w[0] == 1 && (w[1] | w[2] | w[3]) == 0; };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) {
      for (int i = 0; i < 3; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 63);
      u[3] >>= 1;
      halve_mod_p(x1.l);
    }
    while (!(v[0] & 1)) {
      for (int i = 0; i < 3; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
      v[3] >>= 1;
      halve_mod_p(x2.l);
    }
    if (raw_geq(u, v)) {
      raw_sub(u, v);
      x1 = sub(x1, x2);
    } else {
      raw_sub(v, u);
      x2 = sub(x2, x1);
    }
  }
  // x = A^-1 = a^-1 R^-1 as a plain integer; times R^3 / R gives a^-1 R
  static const Fp kR3 = mul(kR2, kR2);
  return mul(is_one(u) ? x1 : x2, kR3);
}

struct Pt {  // XYZZ; infinity <=> zz == 0
  Fp x, y, zz, zzz;
};
inline Pt infinity() {
  Pt r;
  memset(&r, 0, sizeof r);
  return r;
}
inline bool is_inf(const Pt& a) { return is_zero(a.zz); }

inline Pt dbl(const Pt& a) {
  if (is_inf(a)) return a;
  Fp u = add(a.y, a.y);
  Fp v = sqr(u);
  Fp w = mul(u, v);
  Fp s = mul(a.x, v);
  Fp xx = sqr(a.x);
  Fp m = add(add(xx, xx), xx);
  Pt r;
  r.x = sub(sub(sqr(m), s), s);
  r.y = sub(mul(m, sub(s, r.x)), mul(w, a.y));
  r.zz = mul(v, a.zz);
  r.zzz = mul(w, a.zzz);
  return r;
}
inline Pt padd(const Pt& a, const Pt& b) {
  if (is_inf(a)) return b;
  if (is_inf(b)) return a;
  Fp u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz);
  Fp s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
  Fp h = sub(u2, u1), rr = sub(s2, s1);
  if (is_zero(h)) {
    if (is_zero(rr)) return dbl(a);
    return infinity();
  }
  Fp pp = sqr(h), ppp = mul(h, pp), q = mul(u1, pp);
  Pt r;
  r.x = sub(sub(sub(sqr(rr), ppp), q), q);
  r.y = sub(mul(rr, sub(q, r.x)), mul(s1, ppp));
  r.zz = mul(mul(a.zz, b.zz), pp);
  r.zzz = mul(mul(a.zzz, b.zzz), ppp);
  return r;
}

// Device hand-off format: X, Y, ZZ, ZZZ as canonical non-Montgomery integers, 8 x u32 LE each.
inline Pt from_device_words(const uint32_t w[32]) {
  Pt r;
  Fp* f[4] = {&r.x, &r.y, &r.zz, &r.zzz};
  for (int k = 0; k < 4; k++) {
    uint64_t q[4];
    for (int i = 0; i < 4; i++) q[i] = (uint64_t)w[8 * k + 2 * i] | ((uint64_t)w[8 * k + 2 * i + 1] << 32);
    *f[k] = from_raw(q);
  }
  return r;
}

inline void be_store(uint8_t* out, const uint64_t w[4]) {
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) out[(3 - i) * 8 + k] = (uint8_t)(w[i] >> (56 - 8 * k));
}
inline void be_load(uint64_t w[4], const uint8_t* in) {
  for (int i = 0; i < 4; i++) {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v = (v << 8) | in[(3 - i) * 8 + k];
    w[i] = v;
  }
}

// Marshal(): normalised affine x || y, 32-byte big-endian each; infinity = 64 zero bytes.
inline void marshal(uint8_t out[64], const Pt& a) {
  if (is_inf(a)) {
    memset(out, 0, 64);
    return;
  }
  Fp zi3 = inv(a.zzz);
  Fp zi = mul(a.zz, zi3);
  Fp zi2 = sqr(zi);
  uint64_t w[4];
  to_raw(w, mul(a.x, zi2));
  be_store(out, w);
  to_raw(w, mul(a.y, zi3));
  be_store(out + 32, w);
}

// Unmarshal(): 0 ok, -3 coordinate >= p, -4 not on the curve.
inline int unmarshal(Pt& r, const uint8_t in[64]) {
  uint64_t x[4], y[4];
  be_load(x, in);
  be_load(y, in + 32);
  if (raw_geq_p(x) || raw_geq_p(y)) return -3;
  bool allz = true;
  for (int i = 0; i < 4; i++) allz = allz && x[i] == 0 && y[i] == 0;
  if (allz) {
    r = infinity();
    return 0;
  }
  r.x = from_raw(x);
  r.y = from_raw(y);
  r.zz = kOne;
  r.zzz = kOne;
  uint64_t three[4] = {3, 0, 0, 0};
  Fp rhs = add(mul(sqr(r.x), r.x), from_raw(three));
  if (!equal(sqr(r.y), rhs)) return -4;
  return 0;
}

}  // namespace hostg1
