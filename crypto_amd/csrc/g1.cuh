// Device group law for BN256 G1 (y^2 = x^3 + 3 over Fp), the operations the reference
// performs through bn256.G1.Add / G1.ScalarMult (go/bp/util.go:170-172, :180, :192).
//
// Accumulators use XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2); bases are affine.
// Infinity is flagged by ZZ having all limbs zero (a sentinel that is only ever written
// explicitly: a computed ZZ is == 0 mod p only in the P == +-Q cases, which are
// intercepted before the formulas run).
//
// Limb/value accounting (see fe29.cuh): every coordinate that leaves a function is
// fe_norm'ed (lower limbs in [0,2^29)) with |X|,|Y| <= 8p and |ZZ|,|ZZZ| <= 4.2p, so all
// fe_mul operands below have limb-magnitude products <= 2 * 2^58, and every value that
// reaches fe_canon / fe_is_zero_mod_p is below 16p in magnitude.
#pragma once
#include "fe29.cuh"

struct xyzz {
  fe x, y, zz, zzz;
};

// memory images
struct aff_mem {  // 64 B: canonical Montgomery (R = 2^261) x, y as 256-bit LE words; zeros = infinity
  u32 x[8];
  u32 y[8];
};
struct xyzz_mem {  // 144 B: four lazily-reduced 9-limb coordinates
  i32 c[36];
};

__device__ __forceinline__ bool xyzz_is_inf(const xyzz& a) { return fe_limbs_all_zero(a.zz); }
__device__ __forceinline__ void xyzz_set_inf(xyzz& a) {
  a.x = fe_zero();
  a.y = fe_zero();
  a.zz = fe_zero();
  a.zzz = fe_zero();
}
__device__ __forceinline__ void xyzz_from_affine(xyzz& a, const fe& x, const fe& y) {
  a.x = x;
  a.y = y;
  a.zz = fe_one();
  a.zzz = fe_one();
}

__device__ __forceinline__ void xyzz_store(xyzz_mem* p, const xyzz& a) {
  int4* q = reinterpret_cast<int4*>(p->c);
  const i32* s0 = a.x.v;
  const i32* s1 = a.y.v;
  const i32* s2 = a.zz.v;
  const i32* s3 = a.zzz.v;
  i32 t[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    t[i] = s0[i];
    t[9 + i] = s1[i];
    t[18 + i] = s2[i];
    t[27 + i] = s3[i];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_int4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}
__device__ __forceinline__ void xyzz_load(xyzz& a, const xyzz_mem* p) {
  const int4* q = reinterpret_cast<const int4*>(p->c);
  i32 t[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int4 v = q[i];
    t[4 * i] = v.x;
    t[4 * i + 1] = v.y;
    t[4 * i + 2] = v.z;
    t[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 9; i++) {
    a.x.v[i] = t[i];
    a.y.v[i] = t[9 + i];
    a.zz.v[i] = t[18 + i];
    a.zzz.v[i] = t[27 + i];
  }
}

// Load an affine base; returns false for the point at infinity.  neg selects -P.
__device__ __forceinline__ bool aff_load(fe& x, fe& y, const aff_mem* p, bool neg) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  u32 wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32 wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  u32 any = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) any |= wx[i] | wy[i];
  x = fe_unpack256(wx);
  y = fe_unpack256(wy);
  if (neg) y = fe_neg(y);  // limbs in (-2^29, 0]: still magnitude < 2^29
  return any != 0;
}
// The same for a table that holds no point at infinity.
__device__ __forceinline__ void aff_load_nz(fe& x, fe& y, const aff_mem* p, bool neg) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  u32 wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32 wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  x = fe_unpack256(wx);
  y = fe_unpack256(wy);
  if (neg) y = fe_neg(y);
}
__device__ __forceinline__ void aff_store(aff_mem* p, const fe& x_canon, const fe& y_canon) {
  u32 wx[8], wy[8];
  fe_pack256(wx, x_canon);
  fe_pack256(wy, y_canon);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(wx[0], wx[1], wx[2], wx[3]);
  q[1] = make_uint4(wx[4], wx[5], wx[6], wx[7]);
  q[2] = make_uint4(wy[0], wy[1], wy[2], wy[3]);
  q[3] = make_uint4(wy[4], wy[5], wy[6], wy[7]);
}
__device__ __forceinline__ void aff_store_inf(aff_mem* p) {
  uint4* q = reinterpret_cast<uint4*>(p);
  uint4 z = make_uint4(0, 0, 0, 0);
  q[0] = z;
  q[1] = z;
  q[2] = z;
  q[3] = z;
}

// 2 * (x, y) for an affine point (mdbl-2008-s-1).  BN256 G1 has prime order, so y != 0.
__device__ __forceinline__ void xyzz_dbl_affine(xyzz& r, const fe& x, const fe& y_in) {
  fe y = fe_norm(y_in);
  fe yy = fe_sqr(y);
  fe v = fe_norm(fe_small_mul(yy, 4));
  fe t = fe_mul(y, v);
  fe w = fe_norm(fe_add(t, t));
  fe s = fe_mul(x, v);
  fe m = fe_norm(fe_small_mul(fe_sqr(x), 3));
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sqr(m), s), s));
  fe y3 = fe_mul2add(m, fe_sub(s, x3), fe_neg(w), y);
  r.x = x3;
  r.y = y3;
  r.zz = v;
  r.zzz = w;
}

// r = 2a (dbl-2008-s-1)
__device__ __forceinline__ void xyzz_dbl(xyzz& r, const xyzz& a) {
  if (xyzz_is_inf(a)) {
    xyzz_set_inf(r);
    return;
  }
  fe yy = fe_sqr(a.y);
  fe v = fe_norm(fe_small_mul(yy, 4));
  fe t = fe_mul(a.y, v);
  fe w = fe_norm(fe_add(t, t));
  fe s = fe_mul(a.x, v);
  fe m = fe_norm(fe_small_mul(fe_sqr(a.x), 3));
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sqr(m), s), s));
  fe y3 = fe_mul2add(m, fe_sub(s, x3), fe_neg(w), a.y);
  fe zz3 = fe_mul(v, a.zz);
  fe zzz3 = fe_mul(w, a.zzz);
  r.x = x3;
  r.y = y3;
  r.zz = zz3;
  r.zzz = zzz3;
}

// Rare paths (operands equal or opposite).  Out of line and BY VALUE: arguments and results
// travel in VGPRs / the callee's frame, so the callers' accumulators never need an address
// (an accumulator passed by reference to a noinline function is demoted to scratch memory
// for the whole kernel).
__device__ __noinline__ xyzz xyzz_madd_special(fe x2, fe y2, fe rr) {
  xyzz r;
  if (fe_is_zero_mod_p(rr)) {
    xyzz_dbl_affine(r, x2, y2);  // P == Q
  } else {
    xyzz_set_inf(r);  // P == -Q
  }
  return r;
}
// The doubling inside a general addition (P == Q) goes through the caller's stack, one out-of-line
// multiplication at a time: passed by value, its 72 argument and result registers (and, inlined, its nine
// multiplications' live ranges) set the register budget of every kernel that can reach it -- 161 VGPRs,
// three waves per SIMD instead of four -- for a path that random inputs never take.  The caller copies
// the operand into a temporary inside the rare branch only, so its own operands stay in registers.
// Same formulas and operand ranges as xyzz_dbl_quad (dbl-2008-s-1).
__device__ __noinline__ void fe_mul_mem(fe* r, const fe* a, const fe* b) {
  fe x = *a, y = *b;
  *r = fe_mul(x, y);
}
__device__ __noinline__ void xyzz_dbl_outofline(xyzz* io) {
  if (xyzz_is_inf(*io)) return;
  fe yy, v, t, w, s, xx, m, mm, x3, d, t1, t2;
  fe_mul_mem(&yy, &io->y, &io->y);
  v = fe_norm(fe_small_mul(yy, 4));
  fe_mul_mem(&t, &io->y, &v);
  w = fe_norm(fe_add(t, t));
  fe_mul_mem(&s, &io->x, &v);
  fe_mul_mem(&xx, &io->x, &io->x);
  m = fe_norm(fe_small_mul(xx, 3));
  fe_mul_mem(&mm, &m, &m);
  x3 = fe_norm(fe_sub(fe_sub(mm, s), s));
  d = fe_sub(s, x3);
  fe_mul_mem(&t1, &m, &d);
  fe_mul_mem(&t2, &w, &io->y);
  fe_mul_mem(&io->zz, &v, &io->zz);
  fe_mul_mem(&io->zzz, &w, &io->zzz);
  io->x = x3;
  io->y = fe_norm(fe_sub(t1, t2));
}

// acc += (x2, y2), affine operand not at infinity (madd-2008-s): 8M + 2S.
__device__ __forceinline__ void xyzz_madd(xyzz& acc, const fe& x2, const fe& y2) {
  if (xyzz_is_inf(acc)) {
    xyzz_from_affine(acc, x2, fe_norm(y2));
    return;
  }
  fe u2 = fe_mul(x2, acc.zz);
  fe s2 = fe_mul(y2, acc.zzz);
  fe h = fe_sub(u2, acc.x);
  fe rr = fe_sub(s2, acc.y);
  fe pp = fe_sqr(h);
  if (__builtin_expect(fe_maybe_zero_sq(pp), 0)) {
    if (fe_is_zero_mod_p(h)) {
      acc = xyzz_madd_special(x2, y2, rr);
      return;
    }
  }
  fe ppp = fe_mul(h, pp);
  fe q = fe_mul(acc.x, pp);
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sub(fe_sqr(rr), ppp), q), q));
  // rr (q - x3) - Y1 ppp under one reduction; |Y3| < 0.2p + p, limbs normalised by fe_mul2add
  fe y3 = fe_mul2add(rr, fe_sub(q, x3), fe_neg(acc.y), ppp);
  acc.zz = fe_mul(acc.zz, pp);
  acc.zzz = fe_mul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

// r = a + b, both XYZZ (add-2008-s): 12M + 2S, all special cases.
__device__ __forceinline__ void xyzz_add(xyzz& r, const xyzz& a, const xyzz& b) {
  if (xyzz_is_inf(a)) {
    r = b;
    return;
  }
  if (xyzz_is_inf(b)) {
    r = a;
    return;
  }
  fe u1 = fe_mul(a.x, b.zz);
  fe u2 = fe_mul(b.x, a.zz);
  fe s1 = fe_mul(a.y, b.zzz);
  fe s2 = fe_mul(b.y, a.zzz);
  fe h = fe_sub(u2, u1);
  fe rr = fe_sub(s2, s1);
  fe pp = fe_sqr(h);
  if (fe_maybe_zero_sq(pp)) {
    if (fe_is_zero_mod_p(h)) {
      if (fe_is_zero_mod_p(rr)) {
        xyzz tmp = a;
        xyzz_dbl_outofline(&tmp);
        r = tmp;
      } else {
        xyzz_set_inf(r);
      }
      return;
    }
  }
  fe ppp = fe_mul(h, pp);
  fe q = fe_mul(u1, pp);
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sub(fe_sqr(rr), ppp), q), q));
  fe y3 = fe_mul2add(rr, fe_sub(q, x3), fe_neg(s1), ppp);
  fe zz3 = fe_mul(fe_mul(a.zz, b.zz), pp);
  fe zzz3 = fe_mul(fe_mul(a.zzz, b.zzz), ppp);
  r.x = x3;
  r.y = y3;
  r.zz = zz3;
  r.zzz = zzz3;
}

// ------------------------------------------------------------------ Jacobian accumulators
// The element-wise chains (generator folds, the h1 loop: go/bp/main.go:156-159,355-356) are doublings with
// an addition of an AFFINE table entry now and then.  There Jacobian coordinates (x = X/Z^2, y = Y/Z^3) are
// cheaper than XYZZ: the doubling is 3 squarings + 3 products with 5 reductions (XYZZ: 3 + 6 with 8), the
// mixed addition one squaring more (Z^2 is not carried along).  Bucket accumulation (additions only) stays
// XYZZ.  Infinity: Z has all limbs zero (only ever written explicitly, as for XYZZ).
struct jac {
  fe x, y, z;
};
__device__ __forceinline__ bool jac_is_inf(const jac& a) { return fe_limbs_all_zero(a.z); }
__device__ __forceinline__ void jac_set_inf(jac& a) {
  a.x = fe_zero();
  a.y = fe_zero();
  a.z = fe_zero();
}
// r = 2a (dbl-2009-l with S = 4 X YY as a product): XX, YY, MM squarings; S, Z3 products; Y3 = M (S - T) - 8 YYYY
// as M (S - T) - (4 YY)(2 YY) under one reduction.  Inputs normalised with |X|, |Y|, |Z| < 6p; outputs
// normalised, |X3| < 4p, |Y3|, |Z3| < 2p.
__device__ __forceinline__ void jac_dbl(jac& r, const jac& a) {
  if (jac_is_inf(a)) {
    jac_set_inf(r);
    return;
  }
  fe xx = fe_sqr(a.x);
  fe yy = fe_sqr(a.y);
  fe w = fe_norm(fe_add(yy, yy));                    // 2 YY
  fe w2 = fe_norm(fe_small_mul(yy, 4));              // 4 YY (normalised: every term of the double product < 2^58)
  fe s = fe_mul(a.x, w2);                            // 4 X YY
  fe m = fe_norm(fe_small_mul(xx, 3));               // 3 XX
  fe t = fe_norm(fe_sub(fe_sub(fe_sqr(m), s), s));   // X3 = M^2 - 2S
  fe y3 = fe_mul2add(m, fe_sub(s, t), fe_neg(w2), w);  // M (S - X3) - 8 YYYY
  fe z3 = fe_mul(fe_add(a.y, a.y), a.z);             // 2 Y Z   (2Y: limbs < 2^30, times normalised: < 2 * 2^58)
  r.x = t;
  r.y = y3;
  r.z = z3;
}
// Rare paths of the mixed addition, out of line and by value (see xyzz_madd_special)
__device__ __noinline__ jac jac_madd_special(fe x2, fe y2, fe rr) {
  jac r;
  if (fe_is_zero_mod_p(rr)) {  // P == Q: 2 (x2, y2, 1)
    jac a;
    a.x = x2;
    a.y = fe_norm(y2);
    a.z = fe_one();
    jac_dbl(r, a);
  } else {
    jac_set_inf(r);  // P == -Q
  }
  return r;
}
// acc += (x2, y2), affine operand not at infinity (the madd-2007-bl flow without its factors of two):
// 3 squarings + 6 products + one double product, 10 reductions.  x2, y2: limb magnitudes < 2^29.
__device__ __forceinline__ void jac_madd(jac& acc, const fe& x2, const fe& y2) {
  if (jac_is_inf(acc)) {
    acc.x = x2;
    acc.y = fe_norm(y2);
    acc.z = fe_one();
    return;
  }
  fe z1z1 = fe_sqr(acc.z);
  fe u2 = fe_mul(x2, z1z1);
  fe s2 = fe_mul(y2, fe_mul(acc.z, z1z1));
  fe h = fe_sub(u2, acc.x);
  fe rr = fe_sub(s2, acc.y);
  fe hh = fe_sqr(h);
  if (__builtin_expect(fe_maybe_zero_sq(hh), 0)) {
    if (fe_is_zero_mod_p(h)) {
      acc = jac_madd_special(x2, y2, rr);
      return;
    }
  }
  fe hhh = fe_mul(h, hh);
  fe v = fe_mul(acc.x, hh);
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sub(fe_sqr(rr), hhh), v), v));
  fe y3 = fe_mul2add(rr, fe_sub(v, x3), fe_neg(acc.y), hhh);
  acc.z = fe_mul(acc.z, h);
  acc.x = x3;
  acc.y = y3;
}
__device__ __forceinline__ void jac_to_xyzz(xyzz& r, const jac& a) {
  if (jac_is_inf(a)) {
    xyzz_set_inf(r);
    return;
  }
  r.x = a.x;
  r.y = a.y;
  r.zz = fe_sqr(a.z);
  r.zzz = fe_mul(a.z, r.zz);
}

// ------------------------------------------------------------------ quad-cooperative addition
// The bucket-reduction tail is a chain of dependent XYZZ additions executed by very few
// waves, and one wave alone issues a VALU instruction only every ~9.5 clk, so an addition is
// pure latency (14 dependent multiplies).  Here the four lanes of a quad hold the same
// operands and each lane runs one multiply of a four-wide stage; results are exchanged with
// DPP quad_perm broadcasts (no LDS).  14 sequential multiplies become 4.
template <int K>
__device__ __forceinline__ fe fe_quad_bcast(const fe& v) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, v.v[i], K * 0x55, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ fe fe_sel4(u32 role, const fe& a0, const fe& a1, const fe& a2, const fe& a3) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    i32 lo = (role & 1u) ? a1.v[i] : a0.v[i];
    i32 hi = (role & 1u) ? a3.v[i] : a2.v[i];
    r.v[i] = (role & 2u) ? hi : lo;
  }
  return r;
}
// r = a + b.  Must be called by all four lanes of a quad with identical a and b; every lane
// returns the full result.  role = lane & 3.
__device__ __forceinline__ void xyzz_add_quad(xyzz& r, const xyzz& a, const xyzz& b, u32 role) {
  if (xyzz_is_inf(a)) {
    r = b;
    return;
  }
  if (xyzz_is_inf(b)) {
    r = a;
    return;
  }
  fe m1 = fe_mul(fe_sel4(role, a.x, b.x, a.y, b.y), fe_sel4(role, b.zz, a.zz, b.zzz, a.zzz));
  fe u1 = fe_quad_bcast<0>(m1), u2 = fe_quad_bcast<1>(m1), s1 = fe_quad_bcast<2>(m1), s2 = fe_quad_bcast<3>(m1);
  fe h = fe_sub(u2, u1);
  fe rr = fe_sub(s2, s1);
  fe m2 = fe_mul(fe_sel4(role, a.zz, a.zzz, h, rr), fe_sel4(role, b.zz, b.zzz, h, rr));
  fe zz12 = fe_quad_bcast<0>(m2), zzz12 = fe_quad_bcast<1>(m2), pp = fe_quad_bcast<2>(m2), rr2 = fe_quad_bcast<3>(m2);
  if (fe_maybe_zero_sq(pp)) {
    if (fe_is_zero_mod_p(h)) {  // P == +-Q: rare, every lane takes the serial path
      if (fe_is_zero_mod_p(rr)) {
        xyzz tmp = a;
        xyzz_dbl_outofline(&tmp);
        r = tmp;
      } else {
        xyzz_set_inf(r);
      }
      return;
    }
  }
  fe m3 = fe_mul(fe_sel4(role, h, u1, zz12, h), pp);
  fe ppp = fe_quad_bcast<0>(m3), q = fe_quad_bcast<1>(m3), zz3 = fe_quad_bcast<2>(m3);
  fe x3 = fe_norm(fe_sub(fe_sub(fe_sub(rr2, ppp), q), q));
  fe m4 = fe_mul(fe_sel4(role, rr, s1, zzz12, rr), fe_sel4(role, fe_sub(q, x3), ppp, ppp, ppp));
  fe t1 = fe_quad_bcast<0>(m4), t2 = fe_quad_bcast<1>(m4), zzz3 = fe_quad_bcast<2>(m4);
  r.x = x3;
  r.y = fe_norm(fe_sub(t1, t2));
  r.zz = zz3;
  r.zzz = zzz3;
}

// r = 2a, quad-cooperative (same contract as xyzz_add_quad): 9 sequential multiplies become 3.
__device__ __forceinline__ void xyzz_dbl_quad(xyzz& r, const xyzz& a, u32 role) {
  if (xyzz_is_inf(a)) {
    xyzz_set_inf(r);
    return;
  }
  fe m1 = fe_sqr(fe_sel4(role, a.y, a.x, a.y, a.x));
  fe yy = fe_quad_bcast<0>(m1), xx = fe_quad_bcast<1>(m1);
  fe v = fe_norm(fe_small_mul(yy, 4));
  fe m = fe_norm(fe_small_mul(xx, 3));
  fe m2 = fe_mul(fe_sel4(role, a.y, a.x, a.zz, m), fe_sel4(role, v, v, v, m));
  fe t = fe_quad_bcast<0>(m2), s = fe_quad_bcast<1>(m2), zz3 = fe_quad_bcast<2>(m2), mm = fe_quad_bcast<3>(m2);
  fe w = fe_norm(fe_add(t, t));
  fe x3 = fe_norm(fe_sub(fe_sub(mm, s), s));
  fe m3 = fe_mul(fe_sel4(role, m, w, w, w), fe_sel4(role, fe_sub(s, x3), a.y, a.zzz, a.zzz));
  fe t1 = fe_quad_bcast<0>(m3), t2 = fe_quad_bcast<1>(m3), zzz3 = fe_quad_bcast<2>(m3);
  r.x = x3;
  r.y = fe_norm(fe_sub(t1, t2));
  r.zz = zz3;
  r.zzz = zzz3;
}

// Normalise to affine; outputs canonical Montgomery limbs.  Returns false for infinity.
__device__ __forceinline__ bool xyzz_to_affine(fe& x, fe& y, const xyzz& a) {
  if (xyzz_is_inf(a)) return false;
  fe zi3 = fe_inv(a.zzz);          // 1 / Z^3
  fe zi = fe_mul(a.zz, zi3);       // 1 / Z
  fe zi2 = fe_sqr(zi);             // 1 / Z^2
  x = fe_canon(fe_mul(a.x, zi2));
  y = fe_canon(fe_mul(a.y, zi3));
  return true;
}
