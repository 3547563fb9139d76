// Fr = integers mod bn256.Order on the device: the prover's scalar algebra
// (add/sub/mul of go/bp/util.go:197-207, vectorMul / weightVectorMul / vector folds of
// go/bp/util.go:106-163 and go/bppp/util.go:143-176,237-277).  These kernels are HBM-bound
// (32 bytes per element, a handful of multiplications), so the representation is the plain
// one: 8 x 32-bit words, Montgomery R = 2^256, CIOS with 64-bit accumulators.  In memory an
// Fr vector is the reference's own scalar encoding (32-byte big-endian, canonical), which is
// exactly what the MSM kernels consume.
#pragma once
#include "fe29.cuh"

struct fr {
  u32 w[8];  // little-endian words
};

__device__ __forceinline__ constexpr u32 fr_mod_word(int i) {
  constexpr u32 n[8] = {0x57ac7261u, 0x1a2ef45bu, 0xf82b3924u, 0x2e8d8e12u,
                        0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
  return n[i];
}
__device__ __forceinline__ constexpr u32 fr_r2_word(int i) {  // 2^512 mod n
  constexpr u32 r2[8] = {0x2affbc35u, 0xb5f03013u, 0x0792e95du, 0x85a1f7dau,
                         0xa6ee4895u, 0x26841e5fu, 0x765aefd5u, 0x3d8f6c73u};
  return r2[i];
}
#define FR_NINV32 0x2d284e5fu  // -n^-1 mod 2^32

__device__ __forceinline__ fr fr_zero() {
  fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = 0;
  return r;
}
// r = a - n if a >= n (a < 2n)
__device__ __forceinline__ void fr_cond_sub(fr& a, u32 top_carry) {
  u32 t[8];
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 d = (u64)a.w[i] - fr_mod_word(i) - br;
    t[i] = (u32)d;
    br = (d >> 32) & 1;
  }
  bool take = top_carry != 0 || br == 0;
#pragma unroll
  for (int i = 0; i < 8; i++) a.w[i] = take ? t[i] : a.w[i];
}
__device__ __forceinline__ fr fr_add(const fr& a, const fr& b) {
  fr r;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (u64)a.w[i] + b.w[i];
    r.w[i] = (u32)c;
    c >>= 32;
  }
  fr_cond_sub(r, (u32)c);
  return r;
}
// Montgomery product a*b/2^256 mod n, inputs < n (or one input < 2^256 and the other < n), output < n
__device__ __forceinline__ fr fr_mont(const fr& a, const fr& b) {
  u32 t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      u64 cur = (u64)a.w[j] * b.w[i] + t[j] + c;
      t[j] = (u32)cur;
      c = cur >> 32;
    }
    u64 cur = (u64)t[8] + c;
    t[8] = (u32)cur;
    t[9] = (u32)(cur >> 32);
    u32 m = t[0] * FR_NINV32;
    c = ((u64)m * fr_mod_word(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      u64 cur2 = (u64)m * fr_mod_word(j) + t[j] + c;
      t[j - 1] = (u32)cur2;
      c = cur2 >> 32;
    }
    cur = (u64)t[8] + c;
    t[7] = (u32)cur;
    t[8] = t[9] + (u32)(cur >> 32);
  }
  fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = t[i];
  fr_cond_sub(r, t[8]);
  return r;
}
__device__ __forceinline__ fr fr_r2() {
  fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = fr_r2_word(i);
  return r;
}
__device__ __forceinline__ fr fr_to_mont(const fr& a) { return fr_mont(a, fr_r2()); }

// 32-byte big-endian in memory (16-byte aligned) -> canonical words (reduced once: 2^256 < 2n)
__device__ __forceinline__ fr fr_load_be(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  fr r;
  r.w[7] = __builtin_bswap32(a.x);
  r.w[6] = __builtin_bswap32(a.y);
  r.w[5] = __builtin_bswap32(a.z);
  r.w[4] = __builtin_bswap32(a.w);
  r.w[3] = __builtin_bswap32(b.x);
  r.w[2] = __builtin_bswap32(b.y);
  r.w[1] = __builtin_bswap32(b.z);
  r.w[0] = __builtin_bswap32(b.w);
  fr_cond_sub(r, 0);
  return r;
}
__device__ __forceinline__ void fr_store_be(uint8_t* p, const fr& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(__builtin_bswap32(a.w[7]), __builtin_bswap32(a.w[6]), __builtin_bswap32(a.w[5]),
                    __builtin_bswap32(a.w[4]));
  q[1] = make_uint4(__builtin_bswap32(a.w[3]), __builtin_bswap32(a.w[2]), __builtin_bswap32(a.w[1]),
                    __builtin_bswap32(a.w[0]));
}

struct fr_arg {  // kernel-argument form of one scalar (canonical words)
  u32 w[8];
};
__device__ __forceinline__ fr fr_from_arg(const fr_arg& a) {
  fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = a.w[i];
  return r;
}

// wave priority of the Fr kernels (experiments; 3 beside the MSM's sort and tail kernels changed neither WNLA nor go/bp)
#ifndef FR_PRIO
#define FR_PRIO 0
#endif
#define FR_KERNEL_PRIO() do { if (FR_PRIO) __builtin_amdgcn_s_setprio(FR_PRIO); } while (0)

// out[i] = s_a * a[a_off + i*stride] + s_b * b[b_off + i*stride]; an index past n_a / n_b reads 0
// (BP++'s pad-tolerant vectorAdd, go/bppp/util.go:245-260; IPA halves go/bp/main.go:357-358;
// WNLA l' = l0 + y l1, n' = rho^-1 n0 + y n1, go/bppp/docs_test.go:1318-1319).
// Up to FR_BATCH folds per launch (blockIdx.y picks the fold): a WNLA round has three (c', l', n'), and in a
// proof's small rounds a launch costs the host more than the arithmetic costs the device.
#define FR_BATCH 8
struct fr_fold_desc {
  const uint8_t* a;
  const uint8_t* b;
  uint8_t* out;
  u32 n_a, n_b, stride, n;
  fr_arg s_a, s_b;
};
struct fr_fold_batch {
  fr_fold_desc d[FR_BATCH];
};
__global__ void __launch_bounds__(256) k_fr_fold(fr_fold_batch batch) {
  FR_KERNEL_PRIO();
  const fr_fold_desc& D = batch.d[blockIdx.y];
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.n) return;
  fr sa = fr_to_mont(fr_from_arg(D.s_a)), sb = fr_to_mont(fr_from_arg(D.s_b));
  fr r = fr_zero();
  if (i < D.n_a) r = fr_mont(fr_load_be(D.a + (size_t)i * D.stride * 32), sa);
  if (i < D.n_b) r = fr_add(r, fr_mont(fr_load_be(D.b + (size_t)i * D.stride * 32), sb));
  fr_store_be(D.out + (size_t)i * 32, r);
}

// partials[block] = sum_i a_i * b_i * (weighted ? mu^(i+1) : 1) over the block's share of
// [0, n): vectorMul (go/bp/util.go:132-142, go/bppp/util.go:143-157) and weightVectorMul
// (go/bppp/util.go:159-176).  Thread t owns `per` consecutive indices.
// Up to FR_DOT_BATCH inner products per launch (blockIdx.y picks the product): a WNLA round has five
// (go/bppp/docs_test.go:1297-1311), and in its small rounds a launch costs more than the arithmetic.
#define FR_DOT_BATCH 8
struct fr_dot_desc {
  const uint8_t* a;
  const uint8_t* b;
  fr* partials;  // one per workgroup
  u32 a_stride, n_a, b_stride, n_b, n, per, weighted, nblk;
  fr_arg mu;
};
struct fr_dot_batch {
  fr_dot_desc d[FR_DOT_BATCH];
};
__global__ void __launch_bounds__(256) k_fr_dot(fr_dot_batch batch) {
  FR_KERNEL_PRIO();
  __shared__ fr lds[256];
  const fr_dot_desc& D = batch.d[blockIdx.y];
  if (blockIdx.x >= D.nblk) return;  // uniform over the workgroup
  const uint8_t* __restrict__ a = D.a;
  const uint8_t* __restrict__ b = D.b;
  const u32 n = D.n, per = D.per, weighted = D.weighted, a_stride = D.a_stride, b_stride = D.b_stride;
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo64 = (u64)t * per;
  fr acc = fr_zero();
  if (lo64 < n) {
    u32 lo = (u32)lo64;
    u32 hi = (n - lo > per) ? lo + per : n;
    u32 lim = D.n_a < D.n_b ? D.n_a : D.n_b;  // beyond the shorter vector the padded product is 0
    if (hi > lim) hi = lim;
    fr mu = fr_to_mont(fr_from_arg(D.mu)), w = fr_zero();
    if (weighted && lo < hi) {
      // w = mu^(lo+1) in Montgomery form, left-to-right square and multiply
      u32 e = lo + 1;
      w = mu;
      for (int bit = 30 - __builtin_clz(e); bit >= 0; bit--) {
        w = fr_mont(w, w);
        if ((e >> bit) & 1u) w = fr_mont(w, mu);
      }
    }
    const fr r2 = fr_r2();
    for (u32 i = lo; i < hi; i++) {
      fr x = fr_mont(fr_load_be(a + (size_t)i * a_stride * 32), r2);
      fr p = fr_mont(x, fr_load_be(b + (size_t)i * b_stride * 32));
      if (weighted) {
        p = fr_mont(p, w);
        w = fr_mont(w, mu);
      }
      acc = fr_add(acc, p);
    }
  }
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (u32 d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) lds[threadIdx.x] = fr_add(lds[threadIdx.x], lds[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) D.partials[blockIdx.x] = lds[0];
}
// one workgroup per product: out_be[32 * blockIdx.x ..] = sum of its partials
__global__ void __launch_bounds__(256) k_fr_sum(fr_dot_batch batch, uint8_t* __restrict__ out_be) {
  FR_KERNEL_PRIO();
  __shared__ fr lds[256];
  const fr_dot_desc& D = batch.d[blockIdx.x];
  fr acc = fr_zero();
  for (u32 i = threadIdx.x; i < D.nblk; i += 256) acc = fr_add(acc, D.partials[i]);
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (u32 d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) lds[threadIdx.x] = fr_add(lds[threadIdx.x], lds[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) fr_store_be(out_be + 32 * (size_t)blockIdx.x, lds[0]);
}

// Scalar vectors for MSMs over generators that were NOT folded.  After t folding rounds G' = lo G_a (+) hi G_b
// the round-t generator is a fixed combination of the frozen ones, G_t[I] = sum_j W(j) G_0[m(I, j)], with W the
// tensor of the rounds' fold scalars, so <G_t, a> = <G_0, W (x) a>: one MSM over the frozen vector instead of t
// latency-bound generator folds.  Two index layouts:
//   layout 0, even / odd folds (WNLA, go/bppp/docs_test.go:1325-1326: G'[i] = lo G[2i] + hi G[2i+1]):
//             m = 2^t I + j, a new round is the new TOP bit of j;
//   layout 1, half folds (inner-product argument, go/bp/main.go:355-356: G'[i] = lo G[i] + hi G[i + n/2]):
//             m = I + n_t j, a new round is the new LOW bit of j.
// k_fr_tensor_step: w' = the weights after one more round (w = [1] before the first).
struct fr_step_desc {
  const uint8_t* w;
  uint8_t* out;
  u32 len;
  fr_arg lo, hi;
};
struct fr_step_batch {
  fr_step_desc d[FR_BATCH];
};
__global__ void __launch_bounds__(256) k_fr_tensor_step(fr_step_batch batch, u32 layout) {
  FR_KERNEL_PRIO();
  const fr_step_desc& D = batch.d[blockIdx.y];
  u32 m = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 len = D.len;
  if (m >= 2 * len) return;
  const u32 b = layout ? (m & 1u) : (m >= len ? 1u : 0u), j = layout ? (m >> 1) : (b ? m - len : m);
  const fr f = fr_to_mont(fr_from_arg(b ? D.hi : D.lo));
  fr x;
  if (D.w) {
    x = fr_load_be(D.w + (size_t)j * 32);
  } else {
    x = fr_zero();
    x.w[0] = 1;
  }
  fr_store_be(D.out + (size_t)m * 32, fr_mont(x, f));
}
// out[m] = w[j] * s(I) * v[I ^ (swap << sel)], with (I, j) = (m >> t, m mod 2^t) in layout 0 and
// (m mod n_t, m / n_t) in layout 1 (n_t = n_out >> t, a power of two there); s(I) = s_set when bit `sel` of I is
// set, s_clear otherwise; a source index past n_v reads 0 (the BP++ zero padding).  swap = 1 pairs each half
// with the OTHER half's scalars (<H0, l1> + <H1, l0>; <G[n1:], a[:n1]>).  Up to FR_BATCH vectors per launch.
struct fr_expand_desc {
  const uint8_t* v;
  const uint8_t* w;
  uint8_t* out;
  u32 n_v, n_out, t, sel, swap;
  fr_arg s_clear, s_set;
};
struct fr_expand_batch {
  fr_expand_desc d[FR_BATCH];
};
__global__ void __launch_bounds__(256) k_fr_tensor_expand(fr_expand_batch batch, u32 layout) {
  FR_KERNEL_PRIO();
  const fr_expand_desc& D = batch.d[blockIdx.y];
  u32 m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= D.n_out) return;
  const u32 t = D.t, sel = D.sel;
  const u32 nI = D.n_out >> t;
  const u32 I = layout ? (m & (nI - 1u)) : (m >> t), j = layout ? (m / nI) : (m & ((1u << t) - 1u));
  const u32 src = I ^ (D.swap << sel);
  fr r = fr_zero();
  if (src < D.n_v) {
    r = fr_mont(fr_load_be(D.v + (size_t)src * 32), fr_to_mont(fr_from_arg(((I >> sel) & 1u) ? D.s_set : D.s_clear)));
    if (D.w) r = fr_mont(r, fr_to_mont(fr_load_be(D.w + (size_t)j * 32)));
  }
  fr_store_be(D.out + (size_t)m * 32, r);
}

// Gathers one (points view, scalars view) segment into contiguous arrays so that several
// segments become ONE MSM: dst_pts[i] = src_pts[i*pt_stride], dst_sc[i] = scale * src_sc[i*sc_stride]
// (scale optional).  <rho * n1, G0> of the WNLA round (go/bppp/docs_test.go:1296-1306) is a
// segment with scale = rho.
#define GATHER_BATCH 8
struct gather_desc {
  const uint4* src_pts;
  uint4* dst_pts;
  const uint8_t* src_sc;  // null: points only, the caller supplies the scalars itself
  uint8_t* dst_sc;
  u32 pt_stride, sc_stride, has_scale, n;
  fr_arg scale;
};
struct gather_batch {
  gather_desc d[GATHER_BATCH];
};
// up to GATHER_BATCH segments per launch (blockIdx.y): the WNLA X commitment has four
__global__ void __launch_bounds__(256) k_gather_segment(gather_batch batch) {
  FR_KERNEL_PRIO();
  const gather_desc& D = batch.d[blockIdx.y];
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.n) return;
  const uint4* sp = D.src_pts + (size_t)i * D.pt_stride * 4;
  uint4* dp = D.dst_pts + (size_t)i * 4;
  uint4 a = sp[0], b = sp[1], c = sp[2], d = sp[3];
  dp[0] = a;
  dp[1] = b;
  dp[2] = c;
  dp[3] = d;
  if (!D.src_sc) return;
  fr x = fr_load_be(D.src_sc + (size_t)i * D.sc_stride * 32);
  if (D.has_scale) x = fr_mont(x, fr_to_mont(fr_from_arg(D.scale)));
  fr_store_be(D.dst_sc + (size_t)i * 32, x);
}
