// Device arithmetic in the BN256 base field Fp (the "p" of cloudflare/bn256; the gfP
// arithmetic the reference reaches through bn256.G1.ScalarMult / G1.Add at
// go/bp/util.go:170-172).
//
// Representation: 9 signed 32-bit limbs, radix 2^29 ("digits of 29 bits in 32-bit
// limbs"), value = sum v[i] * 2^(29 i), Montgomery form with R = 2^261, *lazily reduced*:
// a value is any integer congruent to the element with |value| < 16 p.
//
// Why not saturated 8 x 32: p is a full 256-bit prime, so saturated limbs need a
// carry-out after every v_mad_u64_u32, and the on-box issue-rate measurement
// (profiles/r01_ubench_valu.txt) shows v_addc_co_u32 costs the same issue time as the
// multiply itself (4.5 vs 4.7 clk per wave64 instruction).  With 29-bit digits a whole
// product-scanning column (9 a*b terms + 9 m*p terms, each < 2^58) fits a signed 64-bit
// accumulator, so a Montgomery product is 162 v_mad_i64_i32 + 9 v_mul_lo_u32 with no
// carry instructions at all, and add/sub are 9 independent v_add_u32/v_sub_u32.
//
// Limb discipline ("norm" = lower 8 limbs in [0, 2^29), top limb signed):
//   fe_mul inputs a, b: |a.v[i]| * |b.v[j]| <= 2 * 2^58   (column sum < 2^63)
//   fe_mul output: norm, value in (-|ab|/R, |ab|/R + p), R = 57.005 p
//   fe_add / fe_sub of two norm values: limbs in (-2^29, 2^30)
// Callers renormalise with fe_norm where the products of limb magnitudes would exceed
// the bound; see g1.cuh for the per-formula accounting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#define INV29_FN __device__ __forceinline__
#include "inv29.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define FE_MASK 0x1fffffff
#define FE_NP 0x1f17daa9  // -p^-1 mod 2^29

struct fe {
  i32 v[9];
};

// p, R mod p, R^2 mod p, 3R mod p in radix 2^29 (top limb unreduced)
#define FE_P_LIMBS \
  { 0x1e089667, 0x02e56362, 0x0d6d6786, 0x1711a241, 0x0dc21ee5, 0x165c30c2, 0x1fe6a9bf, 0x1c695470, 0x008fb501 }
#define FE_ONE_LIMBS \
  { 0x10168311, 0x1aecdef8, 0x02a3f324, 0x1d12df6f, 0x0fc71ed9, 0x057924b5, 0x05a43451, 0x0c8c32d7, 0x0000b294 }
#define FE_R2_LIMBS \
  { 0x1d8d65ed, 0x157f5abf, 0x058bb993, 0x0a97ab55, 0x043b578d, 0x0a4d6606, 0x085e0882, 0x1df0d5b9, 0x000ec411 }
// beta: the cube root of unity in Fp with (beta x, y) = lambda (x, y) on G1 (GLV endomorphism; lambda in
// bn256msm.hip), Montgomery form
#define FE_BETA_LIMBS \
  { 0x14d7f6f4, 0x084a8b7d, 0x1ea4e338, 0x10ea9d96, 0x06ff6659, 0x1f509ce9, 0x127412d9, 0x1736b20a, 0x000755ae }
#define FE_B3_LIMBS \
  { 0x10438933, 0x10c69ce9, 0x07ebd96e, 0x17389e4d, 0x0f555c8d, 0x106b6e20, 0x10ec9cf3, 0x05a49885, 0x000217bd }

__device__ __forceinline__ constexpr i32 fe_p_limb(int i) {
  constexpr i32 p[9] = FE_P_LIMBS;
  return p[i];
}

__device__ __forceinline__ fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ fe fe_one() {
  fe r = {FE_ONE_LIMBS};
  return r;
}
__device__ __forceinline__ bool fe_limbs_all_zero(const fe& a) {
  i32 o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.v[i];
  return o == 0;
}

__device__ __forceinline__ fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
__device__ __forceinline__ fe fe_sub(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
__device__ __forceinline__ fe fe_neg(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = -a.v[i];
  return r;
}
// limb-wise multiple by a small constant (caller renormalises)
__device__ __forceinline__ fe fe_small_mul(const fe& a, i32 k) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * k;
  return r;
}

// Signed carry propagation: lower limbs -> [0, 2^29), top limb keeps the rest.
__device__ __forceinline__ fe fe_norm(const fe& a) {
  fe r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    i32 t = a.v[i] + c;
    r.v[i] = t & FE_MASK;
    c = t >> 29;
  }
  r.v[8] = a.v[8] + c;
  return r;
}

// Montgomery product a*b/R mod p, finely integrated product scanning, radix 2^29:
// 162 v_mad_i64_i32 (signed 32x32+64 -> 64, no carry handling) + 9 v_mul_lo_u32; the square
// uses pre-doubled cross terms (126 + 9).
#include "fe_mul_gen.inc"

// Portable restatement of fe_mul (no inline asm) used by the on-device self test.
__device__ __forceinline__ fe fe_mul_portable(const fe& a, const fe& b) {
  i64 acc = 0;
  i32 m[9];
  fe r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (i64)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (i64)m[i] * fe_p_limb(k - i);
    m[k] = (i32)(((u32)acc * (u32)FE_NP) & FE_MASK);
    acc += (i64)m[k] * fe_p_limb(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (i64)a.v[i] * b.v[k - i];
      acc += (i64)m[i] * fe_p_limb(k - i);
    }
    r.v[k - 9] = (i32)acc & FE_MASK;
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// (a*b + c*d) / R mod p with ONE Montgomery reduction (the reduction is 81 of a product's 162
// multiply-adds).  fe_mul2add is the generated single-statement asm (fe_mul_gen.inc: 45 operands -- hipcc
// takes them; the 30-operand limit is GCC's): 243 multiply-adds + 44 other instructions = 287, where the
// compiler's code for the C++ below is ~320 (extra 64-bit moves and v_lshl_add_u64 around its
// v_mad_u64_u32 quotient terms).  The C++ form stays as the portable restatement the self test compares with.
// Bounds: every |a.v[i]*b.v[j]|, |c.v[i]*d.v[j]| < 2^58 (all four operands with limb magnitude
// < 2^29: normalised values or differences of two normalised values), so a column holds at most
// 27 terms < 2^58 plus a carry < 2^35: below 2^63.  Output: norm, value in
// (-(|ab|+|cd|)/R, (|ab|+|cd|)/R + p).
__device__ __forceinline__ fe fe_mul2add(const fe& a, const fe& b, const fe& c, const fe& d) {
  return fe_mul2add_asm(a, b, c, d);
}
__device__ __forceinline__ fe fe_mul2add_portable(const fe& a, const fe& b, const fe& c, const fe& d) {
  i64 acc = 0;
  i32 m[9];
  fe r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc += (i64)a.v[i] * b.v[k - i];
      acc += (i64)c.v[i] * d.v[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; i++) acc += (i64)m[i] * fe_p_limb(k - i);
    m[k] = (i32)(((u32)acc * (u32)FE_NP) & FE_MASK);
    acc += (i64)m[k] * fe_p_limb(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (i64)a.v[i] * b.v[k - i];
      acc += (i64)c.v[i] * d.v[k - i];
      acc += (i64)m[i] * fe_p_limb(k - i);
    }
    r.v[k - 9] = (i32)acc & FE_MASK;
    acc >>= 29;
  }
  r.v[8] = (i32)acc;
  return r;
}

// normalised limbs of p * 2^S (compile-time)
struct fe_kp {
  i32 v[9];
};
__host__ __device__ constexpr fe_kp fe_make_kp(int s) {
  constexpr i32 p[9] = FE_P_LIMBS;
  fe_kp r{};
  i64 c = 0;
  for (int i = 0; i < 8; i++) {
    i64 t = ((i64)p[i] << s) + c;
    r.v[i] = (i32)(t & FE_MASK);
    c = t >> 29;
  }
  r.v[8] = (i32)(((i64)p[8] << s) + c);
  return r;
}
template <int S>
__device__ __forceinline__ void fe_canon_step(fe& t) {
  constexpr fe_kp kp = fe_make_kp(S);
  fe d;
#pragma unroll
  for (int i = 0; i < 9; i++) d.v[i] = t.v[i] - kp.v[i];
  d = fe_norm(d);
  bool ok = d.v[8] >= 0;
#pragma unroll
  for (int i = 0; i < 9; i++) t.v[i] = ok ? d.v[i] : t.v[i];
}

// Full reduction to the canonical representative in [0, p), normalised limbs.
// Requires |value| < 16 p.
__device__ __noinline__ fe fe_canon(fe a) {  // by value: arguments travel in VGPRs, no scratch
  constexpr fe_kp p16 = fe_make_kp(4);
  fe t;
#pragma unroll
  for (int i = 0; i < 9; i++) t.v[i] = a.v[i] + p16.v[i];
  t = fe_norm(t);  // value in (0, 32p)
  fe_canon_step<4>(t);
  fe_canon_step<3>(t);
  fe_canon_step<2>(t);
  fe_canon_step<1>(t);
  fe_canon_step<0>(t);
  return t;
}

__device__ __forceinline__ bool fe_is_zero_mod_p(const fe& a) { return fe_limbs_all_zero(fe_canon(a)); }

// Cheap necessary condition for "a == 0 mod p" when a is a fe_mul output of a square
// (value in [0, 3p)): the low limb must be that of 0, p or 2p.
__device__ __forceinline__ bool fe_maybe_zero_sq(const fe& a) {
  i32 l = a.v[0];
  return l == 0 || l == fe_p_limb(0) || l == ((2 * fe_p_limb(0)) & FE_MASK);
}

// canonical limbs <-> 8 x 32-bit words (little-endian words of the 256-bit integer)
__device__ __forceinline__ void fe_pack256(u32 w[8], const fe& c) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int bit = 32 * j;
    const int i = bit / 29;
    const int s = bit - 29 * i;
    u32 x = (u32)c.v[i] >> s;
    if (i + 1 < 9) x |= (u32)c.v[i + 1] << (29 - s);
    if (58 - s < 32 && i + 2 < 9) x |= (u32)c.v[i + 2] << (58 - s);
    w[j] = x;
  }
}
__device__ __forceinline__ fe fe_unpack256(const u32 w[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i;
    const int j = bit / 32;
    const int s = bit - 32 * j;
    u32 x = w[j] >> s;
    if (s > 3 && j + 1 < 8) x |= w[j + 1] << (32 - s);
    r.v[i] = (i32)(x & FE_MASK);
  }
  return r;
}

// to / from Montgomery form (input of to_mont: canonical integer limbs)
__device__ __forceinline__ fe fe_to_mont(const fe& a) {
  fe r2 = {FE_R2_LIMBS};
  return fe_mul(a, r2);
}
__device__ __forceinline__ fe fe_from_mont(const fe& a) {
  fe one = fe_zero();
  one.v[0] = 1;
  return fe_mul(a, one);
}

// a^(p-2).  Exponent is a compile-time constant: the multiply branch is wave-uniform.
__device__ __noinline__ fe fe_inv_fermat(fe a_in) {
  const u32 e[8] = {0x5e089667u - 2u, 0x185cac6cu, 0x20b5b59eu, 0xee5b88d1u,
                    0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
  fe a = fe_norm(a_in);
  fe acc = a;  // bit 255 of p-2 is set
  for (int i = 254; i >= 0; i--) {
    acc = fe_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) acc = fe_mul(acc, a);
  }
  return acc;
}

// 1 / a by the binary GCD of inv29.h (~20 000 instructions against ~78 000 for a^(p-2), no data-dependent
// branch): a's Montgomery residue A = a R is inverted as a plain integer, A^-1 = a^-1 R^-1, and two products
// with R^2 bring it to a^-1 R.  a^(p-2) remains the answer for 0 (inverse 0, as every caller expects) and for
// any input the GCD reports as not inverted.
__device__ __noinline__ fe fe_inv(fe a_in) {
  const fe c = fe_canon(a_in);
  fe y;
  if (!inv29::inverse(y.v, c.v)) return fe_inv_fermat(a_in);
  const fe r2 = {FE_R2_LIMBS};
  return fe_mul(fe_mul(y, r2), r2);
}

// 256-bit words: raw compare with p (validation of incoming coordinates)
__device__ __forceinline__ bool u256_geq_p(const u32 w[8]) {
  const u32 p[8] = {0x5e089667u, 0x185cac6cu, 0x20b5b59eu, 0xee5b88d1u,
                    0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)w[i] - p[i] - br;
    br = (t >> 32) & 1;
  }
  return br == 0;
}
