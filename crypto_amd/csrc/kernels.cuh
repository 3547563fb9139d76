// HIP kernels of the BN256 G1 MSM engine (gfx950).  Pipeline of one MSM
// (vectorPointScalarMul, go/bp/util.go:165-175, go/bppp/util.go:178-196):
//
//   k_part<16|20>, k_group_sort(_tl)  two-pass LDS-staged bucket sort (>= 2^15 points with a table, 2^17 without; _tl: the 20-bit-window
//                     path's tile-local records); k_group_count/subscan/place: groups larger than LDS.  Otherwise:
//   k_tile_sort<0>    signed c-bit digits of every scalar, per-tile LDS bucket histograms
//   k_tile_scan       per-bucket prefix over tiles;  k_scan1/2/3 or k_scan_small: bucket totals -> offsets
//   k_tile_sort<1>    LDS cursors, scatter of (base index|sign) into bucket order (counting sort)
//   k_recode<0|1>     the same two steps with global atomics, for bucket sets that do not fit LDS
//   k_accumulate      balanced segmented bucket accumulation, XYZZ mixed additions
//   k_bucket_fix      joins the partial sums of buckets that straddle thread chunks (quad per bucket);
//   k_fix_boundary    the same seen from the chunk boundaries (lane per boundary; 2^19 buckets);
//   k_heavy           buckets with many partial sums (skewed scalars)
//   k_fold_lane2, k_fold_lane, k_fold   log-depth bucket reduction: T = sum B_b and G_k = sum_{bit k of b} B_b
//                     (lane per four inputs / per addition for the wide first levels, quad per addition after)
//   k_pack_results    canonical XYZZ words of T, G_0..G_{c-2} for the host Horner tail
//   k_fold_small      all levels and the packing of a <= 512-bucket set in one launch
//
// Element-wise companions: k_points_from_be / k_points_to_be (Unmarshal / Marshal),
// k_precompute_level (2^c * P), k_scalar_mul (vectorPointMulOnScalar and the h1 loop,
// go/bp/util.go:177-183, go/bp/main.go:156-159), k_add_points (hadamardPointMul,
// go/bp/util.go:185-195), k_fold_points (generator folding, go/bp/main.go:355-356).
#pragma once
#include "g1.cuh"
#include "fr.cuh"
// Wave priorities (s_setprio, 0..3; a SIMD issues from its highest-priority ready wave, oldest first among equals).
// With several calls on the chip (bn256_msm_submit lanes, bench.py keeps four in flight) the kernels of one call's
// sort and tail are short dependent chains beside another call's accumulation, which can use every issue slot it
// is given: left at equal priority the chains wait their turn behind 2000-instruction loop bodies and every call
// takes longer to leave its lane.  Tail kernels (fix-up, folds, packing) run at 3, sort kernels at 2, k_accumulate
// at 0 (or stepping 3 -> 0 when the call is alone, below): four calls in flight 872-885 -> 926-947 Mpts/s on one
// box (tools/ab_prio_tp.sh; 1, 2 or 3 for either class measured the same within 1 %, tail-only 910, none 880).
#ifndef ACC_PRIO_BASE
#define ACC_PRIO_BASE 3
#endif
#ifndef SMALL_PRIO
#define SMALL_PRIO 3
#endif
#ifndef SORT_PRIO
#define SORT_PRIO 2
#endif
#define SMALL_KERNEL_PRIO() do { if (SMALL_PRIO) __builtin_amdgcn_s_setprio(SMALL_PRIO); } while (0)
#define SORT_KERNEL_PRIO() do { if (SORT_PRIO) __builtin_amdgcn_s_setprio(SORT_PRIO); } while (0)

#define NOKEY 0xffffffffu

// ------------------------------------------------------------------ scalars
// group order n, 32-bit little-endian words
__device__ __forceinline__ constexpr u32 fr_n_word(int i) {
  constexpr u32 n[8] = {0x57ac7261u, 0x1a2ef45bu, 0xf82b3924u, 0x2e8d8e12u,
                        0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
  return n[i];
}
__device__ __forceinline__ constexpr u32 fr_half_word(int i) {  // (n - 1) / 2
  return (fr_n_word(i) >> 1) | (i < 7 ? (fr_n_word(i < 7 ? i + 1 : 7) << 31) : 0u);
}

// k = big-endian 32 bytes at p (16-byte aligned)
__device__ __forceinline__ void scalar_load_be(u32 k[8], const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  k[7] = __builtin_bswap32(a.x);
  k[6] = __builtin_bswap32(a.y);
  k[5] = __builtin_bswap32(a.z);
  k[4] = __builtin_bswap32(a.w);
  k[3] = __builtin_bswap32(b.x);
  k[2] = __builtin_bswap32(b.y);
  k[1] = __builtin_bswap32(b.z);
  k[0] = __builtin_bswap32(b.w);
}
__device__ __forceinline__ void scalar_store_be(uint8_t* p, const u32 k[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(__builtin_bswap32(k[7]), __builtin_bswap32(k[6]), __builtin_bswap32(k[5]),
                    __builtin_bswap32(k[4]));
  q[1] = make_uint4(__builtin_bswap32(k[3]), __builtin_bswap32(k[2]), __builtin_bswap32(k[1]),
                    __builtin_bswap32(k[0]));
}
// r = a - b, returns borrow
__device__ __forceinline__ u32 u256_sub(u32 r[8], const u32 a[8], const u32 b[8]) {
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)a[i] - b[i] - br;
    r[i] = (u32)t;
    br = (t >> 32) & 1;
  }
  return (u32)br;
}
// Reduce an arbitrary 256-bit scalar to the signed representative: k mod n, then if
// k > (n-1)/2 replace by n - k and report neg (k.P = -(n-k).P).  Result < 2^255.
__device__ __forceinline__ bool scalar_reduce_signed(u32 k[8]) {
  u32 nn[8], hh[8], t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    nn[i] = fr_n_word(i);
    hh[i] = fr_half_word(i);
  }
  u32 br = u256_sub(t, k, nn);  // k >= n  <=> no borrow
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = br ? k[i] : t[i];
  br = u256_sub(t, hh, k);  // half < k <=> borrow
  bool neg = br != 0;
  u256_sub(t, nn, k);
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = neg ? t[i] : k[i];
  return neg;
}
__device__ __forceinline__ u32 sel8(const u32 k[8], u32 j) {
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r = (j == (u32)i) ? k[i] : r;
  return r;
}
// ---- GLV split on the device (the host's glv_split in bn256msm.hip restated with 32-bit words) ----
// m < n  ->  m = k1 + k2 lambda (mod n), |k1|, |k2| < 2^129; (beta x, y) = lambda (x, y) on G1.
// c1 = (m G1) >> 384 ~ m B2 / n, c2 = (m G2) >> 384 ~ m B1 / n;  k1 = m - c1 A1 - c2 B1,
// k2 = c2 B2 - c1 B1, both exact in two's complement; a quotient that is off by one only moves
// (k1, k2) by one (short) basis vector.
template <int NA, int NB, int NR>
__device__ __forceinline__ void words_mul(u32 (&r)[NR], const u32 (&a)[NA], const u32 (&b)[NB]) {
#pragma unroll
  for (int i = 0; i < NR; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (i + j < NR) {
        c += (u64)a[i] * b[j] + r[i + j];
        r[i + j] = (u32)c;
        c >>= 32;
      }
    }
#pragma unroll
    for (int t = i + NB; t < NR; t++) {
      c += r[t];
      r[t] = (u32)c;
      c >>= 32;
    }
  }
}
template <int N>
__device__ __forceinline__ void words_sub(u32 (&r)[N], const u32 (&a)[N]) {  // r -= a (mod 2^(32 N))
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u64 d = (u64)r[i] - a[i] - br;
    r[i] = (u32)d;
    br = (d >> 32) & 1;
  }
}
// two's complement (N words) -> magnitude in 5 words, returns the sign
template <int N>
__device__ __forceinline__ bool words_abs5(u32 (&m)[5], const u32 (&v)[N]) {
  bool neg = (v[N - 1] >> 31) != 0;
  u64 c = neg ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    c += neg ? (u32)~v[i] : v[i];
    m[i] = (u32)c;
    c >>= 32;
  }
  return neg;
}
__device__ __forceinline__ void glv_split_dev(const u32 (&m)[8], u32 (&k1)[5], bool& n1, u32 (&k2)[5], bool& n2) {
  const u32 A1[4] = {0xb7be300bu, 0x6808727cu, 0x288a7c7bu, 0xbfcdfabeu};
  const u32 B1[2] = {0xd8b10603u, 0xb4ed5d35u};
  const u32 B2[4] = {0xdf0d2a08u, 0xb31b1546u, 0x288a7c7au, 0xbfcdfabeu};
  const u32 G1[9] = {0xbdbf1528u, 0x60ee1bc8u, 0x8a2a05c0u, 0xe21d18f2u, 0x7c4a4445u, 0xae13553cu,
                     0xffd56e3du, 0x55ae596cu, 1u};
  const u32 G2[7] = {0x8c26a0b7u, 0x369d8675u, 0x653eef7bu, 0x287e2fc7u, 0x740d1660u, 0x424dd4afu, 1u};
  u32 t17[17], t15[15];
  words_mul<8, 9, 17>(t17, m, G1);
  words_mul<8, 7, 15>(t15, m, G2);
  const u32 c1[5] = {t17[12], t17[13], t17[14], t17[15], t17[16]};
  const u32 c2[3] = {t15[12], t15[13], t15[14]};
  u32 v1[9], v2[9], u[9];
#pragma unroll
  for (int i = 0; i < 9; i++) v1[i] = i < 8 ? m[i] : 0u;
  words_mul<5, 4, 9>(u, c1, A1);
  words_sub<9>(v1, u);
  words_mul<3, 2, 9>(u, c2, B1);
  words_sub<9>(v1, u);
  words_mul<3, 4, 9>(v2, c2, B2);
  words_mul<5, 2, 9>(u, c1, B1);
  words_sub<9>(v2, u);
  n1 = words_abs5<9>(k1, v1);
  n2 = words_abs5<9>(k2, v2);
}
__device__ __forceinline__ u32 sel5(const u32 (&k)[5], u32 j) {
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) r = (j == (u32)i) ? k[i] : r;
  return r;
}

__device__ __forceinline__ u32 scalar_bits(const u32 k[8], u32 bit, u32 width) {
  u32 j = bit >> 5, s = bit & 31;
  u64 v = (u64)sel8(k, j) | ((u64)sel8(k, j + 1) << 32);
  return (u32)(v >> s) & ((1u << width) - 1u);
}

// ------------------------------------------------------------------ recode / count / scatter
// Signed digits d_w in [-2^(c-1), 2^(c-1)], sum d_w 2^(cw) = +-k.  Bucket key within a set =
// |d| - 1; entry value = (table index << 1) | sign.  A "set" is a group of windows sharing one
// bucket array: all W windows with a fixed-base table (precomp), one window per set without.
//
// The sort is a counting sort staged through LDS (160 KB per CU holds a whole 2^15-bucket
// histogram): workgroup (tile, set) owns a contiguous tile of scalars.
//   k_tile_count   : LDS histogram of the tile's digits -> hist[set][tile][bucket]
//   k_tile_scan    : per (set, bucket) exclusive prefix over tiles, totals[set*nb + bucket]
//   k_scan1/2/3    : exclusive scan of totals -> offsets[bucket]  (offsets[nbt] = #entries)
//   k_tile_scatter : LDS cursors = offsets + tile prefix; pos = ds_add_rtn(cursor[key], 1);
//                    sval[pos] = entry.  No global atomics anywhere.
// digit of window w (call with w = 0, 1, 2, ... in order; carry threads through)
__device__ __forceinline__ void recode_digit(const u32 k[8], bool neg, u32 w, u32 c, u32& carry, u32& mag, u32& sign) {
  const u32 half = 1u << (c - 1);
  u32 raw = scalar_bits(k, w * c, c) + carry;
  bool over = raw > half;
  mag = over ? ((1u << c) - raw) : raw;
  carry = over ? 1u : 0u;
  sign = (over != neg) ? 1u : 0u;
}

// carry entering window w without walking the windows below it: the window below decides unless
// its raw value is exactly half (probability 2^-c), in which case the one below that does.
__device__ __forceinline__ u32 recode_carry_into(const u32 k[8], u32 w, u32 c) {
  const u32 half = 1u << (c - 1);
  while (w > 0) {
    u32 raw = scalar_bits(k, (w - 1) * c, c);
    if (raw != half) return raw > half ? 1u : 0u;
    w--;
  }
  return 0u;
}

#define HEAVY_SLOTS 1024u      // buckets reduced by HEAVY_PARTS workgroups per pass (2T / 4096 <= 256 at the default chunk length)
// The context's counter block: [0] heavy queue length, [1] split buckets, [2, 2 + HEAVY_SLOTS) tickets, 1024
// group totals of the two-pass sort, the medium queue's length, then the two MSM_FLAG words of the call;
// behind the block a second array of group totals (HEAVY_GTOTAL2).
#define HEAVY_GTOTAL (2u + HEAVY_SLOTS)
#define HEAVY_MEDIUM (HEAVY_GTOTAL + 1024u)  // length of the medium queue (k_fix_boundary)
#define HEAVY_BASE (HEAVY_MEDIUM + 2u)
#define HEAVY_WORDS (HEAVY_BASE + 2u)
#define HEAVY_GTOTAL2 HEAVY_WORDS
#define HEAVY_ALLOC_WORDS (HEAVY_WORDS + 1024u)
// No memset in front of a pass (hipMemsetAsync is two fill kernels; the device starts ~21 us after the call is
// entered, a kernel launch ~7 us): block 0 of the pass's FIRST kernel clears what only later kernels of the
// pass use -- queue lengths, tickets, on a call's first pass the flags -- and, for the two-pass sort, the
// group totals of the NEXT pass: they alternate between two arrays, this pass's was cleared by the pass
// before it (both are zero when the context is made).
__device__ __forceinline__ void heavy_clear(u32* __restrict__ heavy_count, u32* __restrict__ gtotal_next, u32 first_pass) {
  for (u32 j = threadIdx.x; j < HEAVY_GTOTAL; j += blockDim.x) heavy_count[j] = 0;
  if (threadIdx.x < 2u) heavy_count[HEAVY_MEDIUM + threadIdx.x] = 0;
  if (first_pass && threadIdx.x < 2u) heavy_count[HEAVY_BASE + threadIdx.x] = 0;
  if (gtotal_next)
    for (u32 j = threadIdx.x; j < 1024u; j += blockDim.x) gtotal_next[j] = 0;
}

// Wide-window tables (bases_precompute): the top table level of point i holds (1 / s_i) 2^(c (W-1)) P_i with
// s_i = top_scale - (i mod TOP_SCALES), and the point's top digit is multiplied by s_i.  top_scale = 1: nothing of this.
#define TOP_SCALES 8u
__device__ __forceinline__ u32 top_scale_of(u32 top_scale, u32 table_index) {
  return top_scale > 1u ? top_scale - (table_index & (TOP_SCALES - 1u)) : top_scale;
}

template <bool SCATTER>
__global__ void __launch_bounds__(1024) k_tile_sort(const uint8_t* __restrict__ scalars_be, u32 n, u32 c, u32 W, u32 nb,
                                                    u32 precomp, u32 base_off, u32 pt_stride, u32 level_stride, u32 tile_len,
                                                    u32 sc_stride, u32 top_scale, u32* __restrict__ hist,
                                                    const u32* __restrict__ offsets, u32* __restrict__ sval,
                                                    u32* __restrict__ heavy_count, u32 first_pass) {
  SORT_KERNEL_PRIO();
  extern __shared__ u32 lds[];
  const u32 tile = blockIdx.x, set = blockIdx.y, ntiles = gridDim.x;
  if (!SCATTER && tile == 0 && set == 0) heavy_clear(heavy_count, nullptr, first_pass);
  u32* h = hist + ((size_t)set * ntiles + tile) * nb;
  if (!SCATTER) {
    for (u32 j = threadIdx.x; j < nb; j += blockDim.x) lds[j] = 0;
  } else {
    const u32* o = offsets + (size_t)set * nb;
    for (u32 j = threadIdx.x; j < nb; j += blockDim.x) lds[j] = o[j] + h[j];
  }
  __syncthreads();
  u32 lo = tile * tile_len;
  u32 hi = (n - lo > tile_len) ? lo + tile_len : n;
  for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    u32 k[8];
    scalar_load_be(k, scalars_be + (size_t)i * sc_stride * 32);
    bool neg = scalar_reduce_signed(k);
    // with a table one workgroup bins every window of its scalars; without, only window `set`
    const u32 w_lo = precomp ? 0u : set, w_hi = precomp ? W : set + 1u;
    u32 carry = precomp ? 0u : recode_carry_into(k, set, c);
    for (u32 w = w_lo; w < w_hi; w++) {
      u32 mag, sign;
      recode_digit(k, neg, w, c, carry, mag, sign);
      if (w == W - 1) mag *= top_scale_of(top_scale, base_off + i * pt_stride);
      if (mag == 0) continue;
      if (!SCATTER) {
        atomicAdd(&lds[mag - 1], 1u);
      } else {
        u32 pos = atomicAdd(&lds[mag - 1], 1u);
        u32 tidx = (precomp ? w * level_stride : 0u) + base_off + i * pt_stride;
        sval[pos] = (tidx << 1) | sign;
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (u32 j = threadIdx.x; j < nb; j += blockDim.x) h[j] = lds[j];
  }
}

// ------------------------------------------------------------------ two-pass sort (c = 16)
// The single-pass scatter above writes every 4-byte entry to its own place in a 64 MiB array:
// one 32-byte HBM sector per entry (PMC WRITE_SIZE 526 MB for 67 MB of entries), and even with
// perfect merging in L2 one L2 write transaction per entry (~300 G/s: 54 us for 16.7 M entries).
// With 2^15 buckets per set the sort is split in two passes, both staged through LDS so that
// every global store is part of a long contiguous run:
//   pass A  k_part         tile of 2048 scalars: recode once (records stay in registers), counting
//                          sort by *group* (G = keys >> LB groups = high key bits) inside LDS, then
//                          the tile's records go out as one contiguous block rec[tile][..] together
//                          with the group boundaries loc[tile][0..G]; group totals by global atomics.
//                          record = ((window, index, sign) << LB) | low key bits
//   pass B  k_group_sort   one workgroup per group: gathers the group's run from every tile (a
//                          contiguous ~256-byte read per tile), counting sort by the low key bits in
//                          LDS, coalesced copy-out of the final entries, offsets[] for k_accumulate
//   split path for a group larger than SORT2_CAP (skewed scalars put nearly every entry into one
//   group): k_group_count / k_group_subscan / k_group_place spread the group over S workgroups with
//   direct stores; they return at once for ordinary groups.
#define SORT2_S 32u          // sub-parts per oversized group
#define SORT2_TILE 2048u     // scalars per tile of pass A
#define SORT2_TCAP (16u * SORT2_TILE)  // record slots per tile (W <= 16 windows)
#define SORT2_CAP 35840u     // records of one group that pass B sorts inside LDS (35 * 1024)

// infinity (ZZ = 0) into a bucket slot
__device__ __forceinline__ void xyzz_mem_set_inf(xyzz_mem* p) {
  int4* q = reinterpret_cast<int4*>(p->c);
  const int4 z = make_int4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = z;
}
// Flag word shared by the sort and accumulation kernels of one call: set when a group is larger than
// pass B's LDS capacity while the split kernels were not enqueued (see msm_enqueue_srcs).
#define MSM_FLAG_OVERSIZED 1u

// Exclusive scan of v over the first B (<= blockDim <= 1024) threads; every thread of the block must
// call it.  Wave-level shuffles, then the 16 wave totals through LDS: two barriers in all.
// tmp[j] (j < B) holds the inclusive value afterwards.
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32 B, u32* tmp) {
  __shared__ u32 wsum[16];
  const u32 j = threadIdx.x, lane = j & 63u, wave = j >> 6;
  u32 x = j < B ? v : 0u;
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    u32 y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  u32 base = 0;
  for (u32 w = 0; w < wave; w++) base += wsum[w];
  x += base;
  if (j < B) tmp[j] = x;
  __syncthreads();
  return j < B ? x - v : 0u;
}

// key = w * key_stride + |digit| - 1: key_stride = 0 with a table (one bucket set, the window is
// kept in the record), nb without (one bucket set per window, the window is implied by the group)
// The window width is fixed at 16 bits (the only width with 2^15 buckets per set): with the window
// loop unrolled every digit is a static half-word of the scalar, no dynamic word selects.
// TL (tile-local records, the c = 20 path): the record carries the scalar's index inside its tile
// (11 bits) instead of the pass-wide index; pass B knows the tile from the run it reads.  With
// 4 window bits, the sign and LB = 10 low key bits that is 26 bits, where (window, 20-bit index, sign,
// low key bits) would not fit a 32-bit record.
template <int C, bool TL>
__global__ void __launch_bounds__(1024) k_part(const uint8_t* __restrict__ scalars_be, u32 n, u32 W, u32 sc_stride,
                                               u32 G, u32 LB, u32 ibits, u32 key_stride, u32 top_scale,
                                               u32 base_off, u32 pt_stride, u32* __restrict__ gtotal,
                                               u32* __restrict__ loc, u32* __restrict__ rec,
                                               u32* __restrict__ heavy_count, u32* __restrict__ gtotal_next,
                                               u32 first_pass) {
  extern __shared__ u32 lds[];  // cnt[G], cur[G], then per-wave histograms [16][G] aliased by stage[TCAP]
  SORT_KERNEL_PRIO();
  constexpr u32 c = C;
  const u32 tile = blockIdx.x, wave = threadIdx.x >> 6;
  if (tile == 0) heavy_clear(heavy_count, gtotal_next, first_pass);
  u32* cnt = lds;
  u32* cur = lds + G;
  u32* hist = lds + 2 * G;
  u32* stage = hist;
  for (u32 j = threadIdx.x; j < 16 * G; j += blockDim.x) hist[j] = 0;
  __syncthreads();
  const u32 lo = tile * SORT2_TILE;
  const u32 hi = lo >= n ? lo : ((n - lo > SORT2_TILE) ? lo + SORT2_TILE : n);
  const u32 lmask = (1u << LB) - 1u;
  u32 rr[SORT2_TILE / 1024][16], gg[SORT2_TILE / 1024][16];
#pragma unroll
  for (u32 sidx = 0; sidx < SORT2_TILE / 1024; sidx++) {
    const u32 i = lo + sidx * 1024 + threadIdx.x;
    u32 k[8];
    bool neg = false;
    if (i < hi) {
      scalar_load_be(k, scalars_be + (size_t)i * sc_stride * 32);
      neg = scalar_reduce_signed(k);
    }
    u32 carry = 0;
#pragma unroll
    for (u32 w = 0; w < 16; w++) {
      gg[sidx][w] = 0xffffffffu;
      rr[sidx][w] = 0;
      if (w < 255 / C + 1 && i < hi && w < W) {
        u32 mag, sign;
        recode_digit(k, neg, w, c, carry, mag, sign);
        if (w == W - 1) mag *= top_scale_of(top_scale, base_off + i * pt_stride);
        if (mag != 0) {
          u32 key = w * key_stride + mag - 1, grp = key >> LB;
          u32 wfield = key_stride ? 0u : (w << (TL ? 11u : ibits));
          rr[sidx][w] = ((((wfield | (TL ? i - lo : i)) << 1) | sign) << LB) | (key & lmask);
          gg[sidx][w] = grp;
          atomicAdd(&hist[wave * G + grp], 1u);
        }
      }
    }
  }
  __syncthreads();
  u32 t = 0;
  if (threadIdx.x < G) {
#pragma unroll
    for (u32 v = 0; v < 16; v++) t += hist[v * G + threadIdx.x];
    cnt[threadIdx.x] = t;
    if (t) atomicAdd(&gtotal[threadIdx.x], t);
  }
  __syncthreads();  // every histogram read is done before stage[] (same memory) becomes the scan scratch
  u32 excl = block_exclusive_scan(t, G, stage);
  if (threadIdx.x < G) {
    cur[threadIdx.x] = excl;
    loc[(size_t)tile * (G + 1) + threadIdx.x] = excl;
    if (threadIdx.x == G - 1) loc[(size_t)tile * (G + 1) + G] = excl + t;
  }
  __syncthreads();
#pragma unroll
  for (u32 sidx = 0; sidx < SORT2_TILE / 1024; sidx++) {
#pragma unroll
    for (u32 w = 0; w < 16; w++) {
      if (w < 255 / C + 1 && gg[sidx][w] != 0xffffffffu) {
        u32 pos = atomicAdd(&cur[gg[sidx][w]], 1u);
        stage[pos] = rr[sidx][w];
      }
    }
  }
  __syncthreads();
  const u32 m = cur[G - 1];  // end of the last group = number of records of the tile
  u32* dst = rec + (size_t)tile * SORT2_TCAP;
  for (u32 j = threadIdx.x; j < m; j += blockDim.x) dst[j] = stage[j];
}

// One workgroup per group.  B = 2^LB keys per group (<= 512); NH histogram copies spread the LDS
// atomics.  LDS: wcnt[NH][B], cur[B], tmp[1024], tile_a[1024], stage[SORT2_CAP].
__global__ void __launch_bounds__(1024) k_group_sort(const u32* __restrict__ rec, const u32* __restrict__ loc,
                                                     const u32* __restrict__ gtotal, u32 ntiles, u32 nbt, u32 LB, u32 NH,
                                                     u32 ibits, u32 base_off, u32 pt_stride, u32 level_stride,
                                                     u32* __restrict__ offsets, u32* __restrict__ sval,
                                                     xyzz_mem* __restrict__ buckets, u32 init_inf,
                                                     u32* __restrict__ flags) {
  SORT_KERNEL_PRIO();
  extern __shared__ u32 lds[];
  __shared__ u32 sh_gs, sh_m;
  const u32 B = 1u << LB, G = gridDim.x, g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  u32* wcnt = lds;
  u32* cur = lds + NH * B;
  u32* tmp = cur + B;
  u32* tile_a = tmp + 1024;
  u32* stage = tile_a + 1024;
  // group start = prefix of the group totals
  u32 tot = threadIdx.x < G ? gtotal[threadIdx.x] : 0;
  u32 pre = block_exclusive_scan(tot, G, tmp);
  if (threadIdx.x == g) {
    sh_gs = pre;
    sh_m = tot;
  }
  if (g == G - 1 && threadIdx.x == G - 1) offsets[nbt] = pre + tot;
  __syncthreads();
  const u32 gs = sh_gs, m = sh_m;
  if (m > SORT2_CAP) {  // left to the split kernels below
    if (threadIdx.x == 0) flags[0] = MSM_FLAG_OVERSIZED;
    return;
  }
  // gather: the group's run of every tile, placed back to back in stage[]
  u32 a = 0, len = 0;
  if (threadIdx.x < ntiles) {
    a = loc[(size_t)threadIdx.x * (G + 1) + g];
    len = loc[(size_t)threadIdx.x * (G + 1) + g + 1] - a;
    tile_a[threadIdx.x] = a;
  }
  u32 toff = block_exclusive_scan(len, ntiles, tmp);  // tmp[t] = end of tile t's run afterwards
  (void)toff;
  // a run is ~64 records (one load per lane); 8 tiles per step keep 8 loads in flight per wave
  for (u32 t0 = wave; t0 < ntiles; t0 += 16 * 8) {
    u32 beg[8], len8[8], v[8];
    const u32* src[8];
#pragma unroll
    for (u32 u = 0; u < 8; u++) {
      const u32 t = t0 + 16 * u;
      beg[u] = 0;
      len8[u] = 0;
      src[u] = rec;
      if (t < ntiles) {
        beg[u] = t ? tmp[t - 1] : 0;
        len8[u] = tmp[t] - beg[u];
        src[u] = rec + (size_t)t * SORT2_TCAP + tile_a[t];
      }
    }
#pragma unroll
    for (u32 u = 0; u < 8; u++) v[u] = lane < len8[u] ? src[u][lane] : 0u;
#pragma unroll
    for (u32 u = 0; u < 8; u++)
      if (lane < len8[u]) stage[beg[u] + lane] = v[u];
#pragma unroll
    for (u32 u = 0; u < 8; u++)
      for (u32 j = lane + 64; j < len8[u]; j += 64) stage[beg[u] + j] = src[u][j];
  }
  for (u32 j = threadIdx.x; j < NH * B; j += blockDim.x) wcnt[j] = 0;
  __syncthreads();
  // the thread's records move to registers for both LDS passes (m <= SORT2_CAP = 35 * 1024)
  u32 r[SORT2_CAP / 1024];
#pragma unroll
  for (u32 it = 0; it < SORT2_CAP / 1024; it++) {
    u32 e = it * 1024 + threadIdx.x;
    r[it] = e < m ? stage[e] : 0u;
  }
  const u32 hrow = (wave & (NH - 1)) * B;
#pragma unroll
  for (u32 it = 0; it < SORT2_CAP / 1024; it++)
    if (it * 1024 + threadIdx.x < m) atomicAdd(&wcnt[hrow + (r[it] & (B - 1))], 1u);
  __syncthreads();
  u32 t = 0;
  if (threadIdx.x < B)
    for (u32 v = 0; v < NH; v++) t += wcnt[v * B + threadIdx.x];
  u32 excl = block_exclusive_scan(t, B, tmp);
  if (threadIdx.x < B) {
    cur[threadIdx.x] = excl;
    offsets[g * B + threadIdx.x] = gs + excl;
    // a bucket without entries: infinity on the first pass of a call (later passes leave the sum alone)
    if (init_inf && t == 0) xyzz_mem_set_inf(&buckets[g * B + threadIdx.x]);
  }
  __syncthreads();
  const u32 imask = (1u << ibits) - 1u;
#pragma unroll
  for (u32 it = 0; it < SORT2_CAP / 1024; it++) {
    if (it * 1024 + threadIdx.x < m) {
      u32 pos = atomicAdd(&cur[r[it] & (B - 1)], 1u);
      u32 v = r[it] >> LB, idx = v >> 1;
      u32 tidx = (idx >> ibits) * level_stride + base_off + (idx & imask) * pt_stride;
      stage[pos] = (tidx << 1) | (v & 1u);
    }
  }
  __syncthreads();
  for (u32 e = threadIdx.x; e < m; e += blockDim.x) sval[gs + e] = stage[e];
}

// Pass B for tile-local records (c = 20: 2^19 buckets, G = 512 groups of B = 1024 keys, ~26.6 k records per
// group at 2^20 points).  Same structure as k_group_sort; a record's tile is recovered from its
// position in the gathered runs (binary search over the run ends), so that registers hold
// (window, tile * 2048 + local index, sign) and the low key bits as two words per record, and LDS
// only ever holds 32-bit words: the gathered records first, the final entries after placement.
// LDS: wcnt[NH][B], cur[B], tmp[1024], tile_a[1024], run_end[1024], stage[SORT20_CAP].
#define SORT20_CAP 32768u
__global__ void __launch_bounds__(1024) k_group_sort_tl(const u32* __restrict__ rec, const u32* __restrict__ loc,
                                                        const u32* __restrict__ gtotal, u32 ntiles, u32 ntiles_data,
                                                        u32 nbt, u32 LB, u32 NH, u32 base_off, u32 pt_stride,
                                                        u32 level_stride, u32* __restrict__ offsets,
                                                        u32* __restrict__ sval, xyzz_mem* __restrict__ buckets,
                                                        u32 init_inf, u32* __restrict__ flags, u32 K,
                                                        u32* __restrict__ chunk_bucket) {
  SORT_KERNEL_PRIO();
  // ntiles: tiles of pass A (a multiple of SORT2_S, the last ones may be empty); ntiles_data: those with scalars
  extern __shared__ u32 lds[];
  __shared__ u32 sh_gs, sh_m, sh_far;
  const u32 B = 1u << LB, G = gridDim.x, g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  u32* wcnt = lds;
  u32* cur = lds + NH * B;
  u32* tmp = cur + B;
  u32* tile_a = tmp + 1024;
  u32* run_end = tile_a + 1024;
  u32* stage = run_end + 1024;
  if (threadIdx.x == 0) sh_far = 0;
  u32 tot = threadIdx.x < G ? gtotal[threadIdx.x] : 0;
  u32 pre = block_exclusive_scan(tot, G, tmp);
  if (threadIdx.x == g) {
    sh_gs = pre;
    sh_m = tot;
  }
  if (g == G - 1 && threadIdx.x == G - 1) offsets[nbt] = pre + tot;
  __syncthreads();
  const u32 gs = sh_gs, m = sh_m;
  if (m > SORT20_CAP) {  // left to the split kernels
    if (threadIdx.x == 0) flags[0] = MSM_FLAG_OVERSIZED;
    return;
  }
  u32 a = 0, len = 0;
  if (threadIdx.x < ntiles) {
    a = loc[(size_t)threadIdx.x * (G + 1) + g];
    len = loc[(size_t)threadIdx.x * (G + 1) + g + 1] - a;
    tile_a[threadIdx.x] = a;
  }
  (void)block_exclusive_scan(len, ntiles, tmp);  // tmp[t] = end of tile t's run afterwards
  if (threadIdx.x < 1024) run_end[threadIdx.x] = threadIdx.x < ntiles ? tmp[threadIdx.x] : m;
  const u32 tagmask = LB == 10 ? 0xfc000000u : 0u;  // 26-bit records (13 windows, 1024 keys a group): the host's only use
  for (u32 t0 = wave; t0 < ntiles; t0 += 16 * 8) {
    u32 beg[8], len8[8], v[8];
    const u32* src[8];
#pragma unroll
    for (u32 u = 0; u < 8; u++) {
      const u32 t = t0 + 16 * u;
      beg[u] = 0;
      len8[u] = 0;
      src[u] = rec;
      if (t < ntiles) {
        beg[u] = t ? tmp[t - 1] : 0;
        len8[u] = tmp[t] - beg[u];
        src[u] = rec + (size_t)t * SORT2_TCAP + tile_a[t];
      }
    }
#pragma unroll
    for (u32 u = 0; u < 8; u++) v[u] = lane < len8[u] ? src[u][lane] : 0u;
    // a record is 26 bits (window, index in the tile, sign, 10 key bits): the six above them take the tile number
    // mod 64, from which the decoding below gets the tile back without looking at run_end[]
#pragma unroll
    for (u32 u = 0; u < 8; u++)
      if (lane < len8[u]) stage[beg[u] + lane] = v[u] | ((((t0 + 16 * u) & 63u) << 26) & tagmask);
#pragma unroll
    for (u32 u = 0; u < 8; u++)
      for (u32 j = lane + 64; j < len8[u]; j += 64) stage[beg[u] + j] = src[u][j] | ((((t0 + 16 * u) & 63u) << 26) & tagmask);
  }
  for (u32 j = threadIdx.x; j < NH * B; j += blockDim.x) wcnt[j] = 0;
  __syncthreads();
  // the thread's records move to registers: (window | tile * 2048 + local | sign) and the low key bits
  u32 r[SORT20_CAP / 1024], kk[SORT20_CAP / 1024];
  // (over the tiles that hold scalars: with the empty tiles at the end counted in, the guess ran up to 29 tiles
  // ahead for a pass of 547 tiles padded to 576, and four records in five fell back to the search -- pass B
  // 126 us against 73 us for 7 % more records)
  const float tiles_per_record = m ? (float)ntiles_data / (float)m : 0.f;
  // tile of the record at place e = first t with run_end[t] > e.  Uniform scalars give every tile the same share of
  // the group, so g(e) = e * ntiles / m is right to within a tile or two, and the tile is the number within 31 of
  // g(e) that has the six bits stored with the record: a handful of instructions, no LDS read (a walk along
  // run_end[] from the guess was ~40 instructions a record, and this kernel is half issue-bound: 16 waves x 2650
  // VALU instructions a workgroup).  Whether "within 31" holds is checked per RUN, at both its ends (g is
  // monotonic): if any run of the group fails (skewed scalars), every record takes the binary search.
  if (threadIdx.x < ntiles && len) {
    const u32 t = threadIdx.x, first = t ? tmp[t - 1] : 0u, last = tmp[t] - 1;
    u32 g1 = (u32)((float)first * tiles_per_record), g2 = (u32)((float)last * tiles_per_record);
    if (g1 > ntiles - 1) g1 = ntiles - 1;
    if (g2 > ntiles - 1) g2 = ntiles - 1;
    if (g1 + 31 < t || t + 31 < g1 || g2 + 31 < t || t + 31 < g2) sh_far = 1;
  }
  __syncthreads();
  const bool near = sh_far == 0 && tagmask != 0;
#pragma unroll
  for (u32 it = 0; it < SORT20_CAP / 1024; it++) {
    const u32 e = it * 1024 + threadIdx.x;
    r[it] = 0;
    kk[it] = 0;
    if (e < m) {
      const u32 xt = stage[e], x = xt & ~tagmask;
      u32 lo_t = (u32)((float)e * tiles_per_record);
      if (lo_t > ntiles - 1) lo_t = ntiles - 1;
      if (near) {
        u32 d = ((xt >> 26) - lo_t) & 63u;  // tile - guess mod 64, in [-32, 31]
        lo_t += d;
        if (d >= 32u) lo_t -= 64u;
      } else {
        u32 hi_t = ntiles - 1;
        lo_t = 0;
        while (lo_t < hi_t) {
          const u32 mid = (lo_t + hi_t) >> 1;
          if (run_end[mid] <= e) lo_t = mid + 1; else hi_t = mid;
        }
      }
      const u32 v = x >> LB;                       // (w << 11 | local) << 1 | sign
      const u32 local = (v >> 1) & 2047u, w = v >> 12;
      r[it] = (((w << 21) | (lo_t * SORT2_TILE + local)) << 1) | (v & 1u);  // index in the pass < 2^21 (1024 tiles)
      kk[it] = x & (B - 1);
    }
  }
  const u32 hrow = (wave & (NH - 1)) * B;
#pragma unroll
  for (u32 it = 0; it < SORT20_CAP / 1024; it++)
    if (it * 1024 + threadIdx.x < m) atomicAdd(&wcnt[hrow + kk[it]], 1u);
  __syncthreads();  // also: every stage[] word has been read, stage[] may be overwritten below
  u32 t = 0;
  if (threadIdx.x < B)
    for (u32 v = 0; v < NH; v++) t += wcnt[v * B + threadIdx.x];
  u32 excl = block_exclusive_scan(t, B, tmp);
  if (threadIdx.x < B) {
    cur[threadIdx.x] = excl;
    offsets[g * B + threadIdx.x] = gs + excl;
    if (init_inf && t == 0) xyzz_mem_set_inf(&buckets[g * B + threadIdx.x]);
    // k_accumulate's chunk j starts at entry j K: the bucket that holds that entry, as a hint (the kernel checks
    // it against offsets[] and searches when it does not fit -- chunks inside a bucket of more than four chunks,
    // the split-sort path, anything stale)
    if (chunk_bucket && t) {
      const u32 o = gs + excl;
      u32 j = (o + K - 1) / K;
      for (u32 k = 0; k < 4 && (u64)j * K < (u64)o + t; k++, j++) chunk_bucket[j] = g * B + threadIdx.x;
    }
  }
  __syncthreads();
#pragma unroll
  for (u32 it = 0; it < SORT20_CAP / 1024; it++) {
    if (it * 1024 + threadIdx.x < m) {
      u32 pos = atomicAdd(&cur[kk[it]], 1u);
      u32 idx = (r[it] >> 1) & 0x1fffffu, w = r[it] >> 22;
      u32 tidx = w * level_stride + base_off + idx * pt_stride;
      stage[pos] = (tidx << 1) | (r[it] & 1u);
    }
  }
  __syncthreads();
  for (u32 e = threadIdx.x; e < m; e += blockDim.x) sval[gs + e] = stage[e];
}

// ---- split path for oversized groups.  Block (g, sp) owns the group's runs of tiles [sp*TS, (sp+1)*TS).
__global__ void __launch_bounds__(256) k_group_count(const u32* __restrict__ rec, const u32* __restrict__ loc,
                                                     const u32* __restrict__ gtotal, u32 G, u32 TS, u32 LB, u32 cap,
                                                     u32* __restrict__ subhist) {
  SORT_KERNEL_PRIO();
  extern __shared__ u32 lds[];  // h[4][B]
  const u32 B = 1u << LB;
  const u32 g = blockIdx.x / SORT2_S, sp = blockIdx.x % SORT2_S, wave = threadIdx.x >> 6;
  if (gtotal[g] <= cap) return;
  for (u32 j = threadIdx.x; j < 4 * B; j += blockDim.x) lds[j] = 0;
  __syncthreads();
  for (u32 t = sp * TS; t < (sp + 1) * TS; t++) {
    const u32 a = loc[(size_t)t * (G + 1) + g], e = loc[(size_t)t * (G + 1) + g + 1];
    const u32* src = rec + (size_t)t * SORT2_TCAP;
    for (u32 j = a + threadIdx.x; j < e; j += blockDim.x) atomicAdd(&lds[wave * B + (src[j] & (B - 1))], 1u);
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < B; b += blockDim.x)
    subhist[((size_t)g * B + b) * SORT2_S + sp] = lds[b] + lds[B + b] + lds[2 * B + b] + lds[3 * B + b];
}
// exclusive scan, in place, of an oversized group's B*S counters (bucket-major), offset by the group
// start; one block of 1024 threads per group, thread b owns bucket b's S counters
__global__ void __launch_bounds__(1024) k_group_subscan(const u32* __restrict__ gtotal, u32 LB, u32 cap,
                                                        u32* __restrict__ subhist, u32* __restrict__ offsets,
                                                        xyzz_mem* __restrict__ buckets, u32 init_inf) {
  SORT_KERNEL_PRIO();
  __shared__ u32 tmp[1024];
  __shared__ u32 sh_gs;
  const u32 B = 1u << LB, G = gridDim.x, g = blockIdx.x, b = threadIdx.x;
  if (gtotal[g] <= cap) return;
  u32 tot = b < G ? gtotal[b] : 0;
  u32 pre = block_exclusive_scan(tot, G, tmp);
  if (b == g) sh_gs = pre;
  __syncthreads();
  const u32 gs = sh_gs;
  u32* row = subhist + ((size_t)g * B + b) * SORT2_S;
  u32 t = 0;
  if (b < B)
    for (u32 v = 0; v < SORT2_S; v++) t += row[v];
  u32 excl = block_exclusive_scan(t, B, tmp);
  if (b < B) {
    u32 run = gs + excl;
    offsets[g * B + b] = run;
    if (init_inf && t == 0) xyzz_mem_set_inf(&buckets[g * B + b]);
    for (u32 v = 0; v < SORT2_S; v++) {
      u32 cnt = row[v];
      row[v] = run;
      run += cnt;
    }
  }
}
__global__ void __launch_bounds__(256) k_group_place(const u32* __restrict__ rec, const u32* __restrict__ loc,
                                                     const u32* __restrict__ gtotal, const u32* __restrict__ subscan,
                                                     u32 G, u32 TS, u32 LB, u32 ibits, u32 base_off, u32 pt_stride,
                                                     u32 level_stride, u32 cap, u32 tile_local,
                                                     u32* __restrict__ sval) {
  SORT_KERNEL_PRIO();
  extern __shared__ u32 cur[];  // [B]
  const u32 B = 1u << LB;
  const u32 g = blockIdx.x / SORT2_S, sp = blockIdx.x % SORT2_S;
  if (gtotal[g] <= cap) return;
  for (u32 b = threadIdx.x; b < B; b += blockDim.x) cur[b] = subscan[((size_t)g * B + b) * SORT2_S + sp];
  __syncthreads();
  const u32 imask = (1u << ibits) - 1u;
  for (u32 t = sp * TS; t < (sp + 1) * TS; t++) {
    const u32 a = loc[(size_t)t * (G + 1) + g], e = loc[(size_t)t * (G + 1) + g + 1];
    const u32* src = rec + (size_t)t * SORT2_TCAP;
    for (u32 j = a + threadIdx.x; j < e; j += blockDim.x) {
      u32 r = src[j];
      u32 pos = atomicAdd(&cur[r & (B - 1)], 1u);
      u32 v = r >> LB, idx = v >> 1;
      // tile-local records: (window << 11 | index inside tile t)
      u32 w = tile_local ? (idx >> 11) : (idx >> ibits);
      u32 i = tile_local ? t * SORT2_TILE + (idx & 2047u) : (idx & imask);
      u32 tidx = w * level_stride + base_off + i * pt_stride;
      sval[pos] = (tidx << 1) | (v & 1u);
    }
  }
}

// thread per (set, bucket): exclusive prefix of hist over tiles (in place) and the bucket total
__global__ void __launch_bounds__(256) k_tile_scan(u32* __restrict__ hist, u32 nsets, u32 ntiles, u32 nb,
                                                   u32* __restrict__ totals) {
  SORT_KERNEL_PRIO();
  u32 id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nsets * nb) {
    if (id == nsets * nb) totals[id] = 0;  // sentinel so that the scan yields offsets[nbt] = total
    return;
  }
  u32 set = id / nb, b = id % nb;
  u32* h = hist + (size_t)set * ntiles * nb + b;
  u32 run = 0;
  u32 t = 0;
  for (; t + 8 <= ntiles; t += 8) {
    u32 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = h[(size_t)(t + j) * nb];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      h[(size_t)(t + j) * nb] = run;
      run += v[j];
    }
  }
  for (; t < ntiles; t++) {
    u32 v = h[(size_t)t * nb];
    h[(size_t)t * nb] = run;
    run += v;
  }
  totals[id] = run;
}

// Fallback for bucket arrays too large for LDS (nb * 4 > 128 KB): global-atomic counting sort.
//   SCATTER = false: rank[w*n+i] = atomicAdd(count[key], 1);  SCATTER = true: sval[offsets[key] + rank] = entry
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_recode(const uint8_t* __restrict__ scalars_be, u32 n, u32 c, u32 W,
                                                u32 nb, u32 precomp, u32 base_off, u32 pt_stride, u32 level_stride,
                                                u32 sc_stride, u32 top_scale, u32* __restrict__ counts,
                                                u32* __restrict__ rank,
                                                const u32* __restrict__ offsets, u32* __restrict__ sval) {
  SORT_KERNEL_PRIO();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8];
  scalar_load_be(k, scalars_be + (size_t)i * sc_stride * 32);
  bool neg = scalar_reduce_signed(k);
  u32 carry = 0;
  for (u32 w = 0; w < W; w++) {
    u32 mag, sign;
    recode_digit(k, neg, w, c, carry, mag, sign);
    if (w == W - 1) mag *= top_scale_of(top_scale, base_off + i * pt_stride);
    if (mag == 0) continue;
    u32 key = (precomp ? 0u : w * nb) + mag - 1u;
    if (!SCATTER) {
      rank[(size_t)w * n + i] = atomicAdd(&counts[key], 1u);
    } else {
      u32 pos = offsets[key] + rank[(size_t)w * n + i];
      u32 tidx = (precomp ? w * level_stride : 0u) + base_off + i * pt_stride;
      sval[pos] = (tidx << 1) | sign;
    }
  }
}

// ------------------------------------------------------------------ exclusive scan (3 phase)
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* lds, u32& total) {
  // lds: 256 entries
  u32 tid = threadIdx.x;
  lds[tid] = v;
  __syncthreads();
  for (u32 d = 1; d < 256; d <<= 1) {
    u32 t = (tid >= d) ? lds[tid - d] : 0;
    __syncthreads();
    lds[tid] += t;
    __syncthreads();
  }
  total = lds[255];
  u32 incl = lds[tid];
  __syncthreads();
  return incl - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan1(const u32* __restrict__ in, u32* __restrict__ out, u32 n,
                                                      u32* __restrict__ blocksum) {
  SORT_KERNEL_PRIO();
  __shared__ u32 lds[256];
  u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    s += v[j];
  }
  u32 total;
  u32 ex = block_exclusive_scan_256(s, lds, total);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
  if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}
// single block: exclusive scan of blocksum[0..nblk) in place; writes grand total to *total_out
__global__ void __launch_bounds__(256) k_scan2(u32* __restrict__ blocksum, u32 nblk, u32* __restrict__ total_out) {
  SORT_KERNEL_PRIO();
  __shared__ u32 lds[256];
  u32 per = (nblk + 255) / 256;
  u32 lo = threadIdx.x * per;
  u32 s = 0;
  for (u32 j = 0; j < per; j++)
    if (lo + j < nblk) s += blocksum[lo + j];
  u32 total;
  u32 ex = block_exclusive_scan_256(s, lds, total);
  for (u32 j = 0; j < per; j++) {
    if (lo + j < nblk) {
      u32 t = blocksum[lo + j];
      blocksum[lo + j] = ex;
      ex += t;
    }
  }
  if (threadIdx.x == 0) *total_out = total;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan3(u32* __restrict__ out, u32 n, const u32* __restrict__ blocksum) {
  SORT_KERNEL_PRIO();
  u32 add = blocksum[blockIdx.x];
  u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    if (base + j < n) out[base + j] += add;
}

// The same scan for up to SCAN_SMALL_MAX elements as ONE launch (small calls are bound by the number of
// launches the host has to enqueue, ~4 us each, not by device time): 1024 threads x 16 elements.
#define SCAN_SMALL_ITEMS 16
#define SCAN_SMALL_MAX (1024 * SCAN_SMALL_ITEMS)
__global__ void __launch_bounds__(1024) k_scan_small(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
  SORT_KERNEL_PRIO();
  __shared__ u32 tmp[1024];
  const u32 base = threadIdx.x * SCAN_SMALL_ITEMS;
  u32 v[SCAN_SMALL_ITEMS];
  u32 sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_SMALL_ITEMS; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    sum += v[j];
  }
  u32 ex = block_exclusive_scan(sum, 1024, tmp);
#pragma unroll
  for (int j = 0; j < SCAN_SMALL_ITEMS; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// ------------------------------------------------------------------ bucket accumulation
#define FIX_CAP 32
#define HEAVY_SPLIT_MIN 4096u  // partials above which one bucket is reduced by HEAVY_PARTS workgroups
#define HEAVY_PARTS 64u
// Thread t owns the sorted entries [t*K, (t+1)*K); bucket boundaries come from offsets[].
// Runs of one bucket are summed with mixed additions.  A run that neither continues from
// thread t-1 nor into thread t+1 is a complete bucket; otherwise it is a partial: slot 2t if
// it continues from the previous thread (possibly also into the next), slot 2t+1 if it only
// continues into the next.
// add_prev (second and later passes of one call): a run that STARTS its bucket in this thread begins
// from the bucket's current content instead of infinity, so the passes' sums meet in one bucket array
// with no extra addition (k_bucket_fix then leaves buckets without entries alone).
// NOINF: the table is known to hold no point at infinity (generated bases; uploads without a zero
// point): the 16-instruction test per gathered base goes.
template <bool NOINF>
__global__ void __launch_bounds__(256) k_accumulate(const aff_mem* __restrict__ table, const u32* __restrict__ sval,
                                                    const u32* __restrict__ offsets, u32 nbt, u32 K, u32 add_prev,
                                                    xyzz_mem* __restrict__ buckets,
                                                    xyzz_mem* __restrict__ partials, const u32* __restrict__ flags,
                                                    u32* __restrict__ lastb, u32 prio_steps,
                                                    const u32* __restrict__ chunk_bucket) {
  // an oversized group that no split kernel handled left part of sval[] unwritten: do not gather through
  // it (the host repeats the call with the split kernels, msm_finish)
  if (flags[0] & MSM_FLAG_OVERSIZED) return;
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 M = offsets[nbt];
  u64 start64 = (u64)t * K;
  if (start64 >= M) return;
  u32 start = (u32)start64;
  u32 end = (M - start > K) ? start + K : M;
  // cur = the bucket containing entry `start`: first b with offsets[b+1] > start.  The two-pass sort leaves it
  // as a hint (k_group_sort_tl); nineteen dependent loads of a binary search otherwise -- every wave of the
  // kernel's single round stands in them at once, nothing to hide them behind
  u32 lo = 0, hi = nbt - 1;
  if (chunk_bucket) {
    const u32 h = chunk_bucket[t];
    if (h < nbt) {
      const u32 ob = offsets[h], oe = offsets[h + 1];
      if (ob <= start && start < oe) lo = hi = h;
    }
  }
  while (lo < hi) {
    u32 mid = (lo + hi) >> 1;
    if (offsets[mid + 1] <= start) lo = mid + 1; else hi = mid;
  }
  u32 cur = lo;
  u32 bend = offsets[cur + 1];
  bool head = offsets[cur] < start;  // first run continues from the previous thread
  xyzz acc;
  xyzz_set_inf(acc);
  if (add_prev && !head) xyzz_load(acc, &buckets[cur]);
  // The four waves of a SIMD are served oldest first, so left alone they finish one after the other -- a quarter
  // of the kernel runs with ONE wave per SIMD (PMC: mean occupancy 10 of 16 waves per CU, wave lifetime 63 % of
  // the kernel = (1 + 2 + 3 + 4) / 16), at a lone wave's issue rate and with its memory waits exposed.  Every wave
  // therefore starts at the highest priority and steps down as it gets through its chunk: whoever is behind is
  // served first, the four stay within a quarter of a chunk of each other and finish together.
  // prio_steps: the three step-down points as iterations (bytes 0..2; see the launch in bn256msm.hip, which puts
  // all three at iteration 0 -- priority 0 throughout -- when other calls of the context are in flight)
  const u32 s1 = prio_steps & 255u, s2 = (prio_steps >> 8) & 255u, s3 = (prio_steps >> 16) & 255u;
  __builtin_amdgcn_s_setprio(ACC_PRIO_BASE);
  for (u32 i = 0; i < K; i++) {
    if (i == s1) __builtin_amdgcn_s_setprio(ACC_PRIO_BASE > 1 ? ACC_PRIO_BASE - 1 : 0);
    if (i == s2) __builtin_amdgcn_s_setprio(ACC_PRIO_BASE > 2 ? ACC_PRIO_BASE - 2 : 0);
    if (i == s3) __builtin_amdgcn_s_setprio(0);
    const u32 e = start + i;
    if (e >= end) break;
    u32 val = sval[e];
    if (e >= bend) {
      xyzz_store(head ? &partials[2 * (size_t)t] : &buckets[cur], acc);
      head = false;
      u32 steps = 0;
      do {
        cur++;
        bend = offsets[cur + 1];
      } while (e >= bend && ++steps < 8);
      if (e >= bend) {
        // many empty buckets in between (a sparse pass: a few points over 2^19 buckets would walk them
        // one dependent load at a time -- 11 ms for a 9-point pass): first b > cur with offsets[b+1] > e
        u32 lo = cur + 1, hi = nbt - 1;
        while (lo < hi) {
          u32 mid = (lo + hi) >> 1;
          if (offsets[mid + 1] <= e) lo = mid + 1; else hi = mid;
        }
        cur = lo;
        bend = offsets[cur + 1];
      }
      xyzz_set_inf(acc);
      if (add_prev) xyzz_load(acc, &buckets[cur]);
    }
    fe x2, y2;
    if (NOINF) {
      aff_load_nz(x2, y2, &table[val >> 1], (val & 1u) != 0);
      xyzz_madd(acc, x2, y2);
    } else if (aff_load(x2, y2, &table[val >> 1], (val & 1u) != 0)) {
      xyzz_madd(acc, x2, y2);
    }
  }
  bool tail = bend > end;
  xyzz_mem* dst = head ? &partials[2 * (size_t)t] : (tail ? &partials[2 * (size_t)t + 1] : &buckets[cur]);
  xyzz_store(dst, acc);
  if (lastb) lastb[t] = cur | (tail ? 0x80000000u : 0u);  // for k_fix_boundary
}

// The fix-up seen from the chunks instead of the buckets (dense passes: far fewer chunk boundaries than
// buckets): lane t looks at the bucket its chunk's last run belongs to; if that bucket starts in
// chunk t and runs on into the next chunks, this lane joins its partial sums (slot 2 t + 1, then
// 2 t' for the chunks it covers).  With ~26 entries per bucket and 52 per chunk nearly every chunk
// boundary falls inside a bucket: 2^18 independent additions, throughput work, one LANE each (the quad
// form spends 2.3x the machine time on them, which shows once several calls share the chip).  Chains of
// more than FIX_LANE_CAP partial sums (skewed scalars) go to the medium queue, which k_heavy's quads work
// off as serial chains, those above FIX_CAP to the heavy queue.  Buckets without entries were set to
// infinity by the sort.
#define FIX_LANE_CAP 4u
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_fix_boundary(const u32* __restrict__ offsets, u32 nbt, u32 K, const u32* __restrict__ lastb,
               xyzz_mem* __restrict__ buckets, const xyzz_mem* __restrict__ partials, u32* __restrict__ heavy_count,
               u32* __restrict__ heavy_list, const u32* __restrict__ flags) {
  SMALL_KERNEL_PRIO();
  if (flags[0] & MSM_FLAG_OVERSIZED) return;
  const u32 M = offsets[nbt];
  const u32 nchunks = M / K + (M % K ? 1u : 0u), step = gridDim.x * blockDim.x;
  for (u32 t0 = blockIdx.x * blockDim.x + threadIdx.x; t0 < nchunks; t0 += step) {
    u32 v = lastb[t0];
    if (!(v >> 31)) continue;             // the chunk's last run ends inside the chunk
    u32 b = v & 0x7fffffffu;
    u32 lo = offsets[b], hi = offsets[b + 1];
    if ((u64)lo < (u64)t0 * K) continue;  // the bucket started in an earlier chunk: that chunk's lane has it
    u32 t1 = (hi - 1) / K, np = t1 - t0 + 1;
    if (np > FIX_CAP) {
      u32 ent = b;
      if (np > HEAVY_SPLIT_MIN && b < (1u << 20)) {
        u32 slot = atomicAdd(heavy_count + 1, 1u);
        if (slot < HEAVY_SLOTS) ent = b | 0x80000000u | (slot << 20);
      }
      heavy_list[atomicAdd(heavy_count, 1u)] = ent;
      continue;
    }
    if (np > FIX_LANE_CAP) {  // the medium queue grows from the end of the list
      heavy_list[nbt - 1u - atomicAdd(heavy_count + HEAVY_MEDIUM, 1u)] = b;
      continue;
    }
    xyzz acc;
    xyzz_load(acc, &partials[2 * (size_t)t0 + 1]);
    for (u32 t = t0 + 1; t <= t1; t++) {
      xyzz p, r;
      xyzz_load(p, &partials[2 * (size_t)t]);
      xyzz_add(r, acc, p);
      acc = r;
    }
    xyzz_store(&buckets[b], acc);
  }
}

// One thread per bucket: empty buckets become infinity; buckets that straddle chunks
// t0 < t1 are the sum of partials 2*t0+1, 2*t (t0 < t < t1), 2*t1.  Buckets with more than
// FIX_CAP partials (heavy buckets: skewed scalar distributions such as the Bulletproofs
// 0/1 bit vectors, go/bp/main.go:84-85, or a narrow top window) are queued for k_heavy.
__global__ void __launch_bounds__(64) k_bucket_fix(const u32* __restrict__ offsets, u32 nbt, u32 K, u32 add_prev,
                                                   xyzz_mem* __restrict__ buckets,
                                                   const xyzz_mem* __restrict__ partials,
                                                   u32* __restrict__ heavy_count, u32* __restrict__ heavy_list,
                                                   const u32* __restrict__ flags) {
  SMALL_KERNEL_PRIO();
  // an oversized group that no split kernel handled left part of offsets[] unwritten: chunk numbers
  // derived from it would index partials[] out of bounds (the host repeats the call, msm_finish)
  if (flags[0] & MSM_FLAG_OVERSIZED) return;
  // one quad (4 lanes) per bucket: the serial chain of additions runs quad-cooperatively
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 b = tid >> 2, role = tid & 3u;
  if (b >= nbt) return;
  u32 lo = offsets[b], hi = offsets[b + 1];
  xyzz acc;
  if (lo == hi) {
    if (add_prev) return;  // no entry in this pass: the earlier passes' sum stays
    xyzz_set_inf(acc);
    if (role == 0) xyzz_store(&buckets[b], acc);
    return;
  }
  u32 t0 = lo / K, t1 = (hi - 1) / K;
  if (t0 == t1) return;
  if (t1 - t0 + 1 > FIX_CAP) {
    if (role == 0) {
      // heavy_count[0]: queue length; heavy_count[1]: buckets so heavy that k_heavy splits them
      // over HEAVY_PARTS workgroups (bit 31 of the queue entry, slot in bits 20..30)
      u32 ent = b;
      if (t1 - t0 + 1 > HEAVY_SPLIT_MIN && b < (1u << 20)) {
        u32 slot = atomicAdd(heavy_count + 1, 1u);
        if (slot < HEAVY_SLOTS) ent = b | 0x80000000u | (slot << 20);
      }
      heavy_list[atomicAdd(heavy_count, 1u)] = ent;
    }
    return;
  }
  // the chain of additions is latency-bound: the next partial is fetched while the current one is added
  xyzz p;
  xyzz_load(acc, &partials[2 * (size_t)t0 + 1]);
  xyzz_load(p, &partials[2 * (size_t)(t0 + 1)]);
  for (u32 t = t0 + 1; t <= t1; t++) {
    xyzz pn, r;
    if (t < t1) xyzz_load(pn, &partials[2 * (size_t)(t + 1)]);
    xyzz_add_quad(r, acc, p, role);
    acc = r;
    p = pn;
  }
  if (role == 0) xyzz_store(&buckets[b], acc);
}

// Heavy buckets.  A work item is (queue entry, part): one part for an ordinary heavy bucket,
// HEAVY_PARTS parts for a bucket with more than HEAVY_SPLIT_MIN partials (a 0/1 bit vector puts
// 3/4 of all entries into one bucket).  The 64 quads of a workgroup each add a strided share of
// the item's partials quad-cooperatively, then a 6-level tree through LDS.  A split bucket's part
// results go to tmp[slot][part]; the workgroup that draws the last ticket adds them up.
__device__ __forceinline__ void heavy_reduce(xyzz& acc, const xyzz_mem* __restrict__ src, u32 lo, u32 hi, u32 first_slot,
                                             bool raw, xyzz_mem* sh) {
  // partial index i -> slot: raw: src[i]; else slot 2*t0+1 for i = 0 and 2*(t0+i) after (first_slot = t0)
  const u32 quad = threadIdx.x >> 2, role = threadIdx.x & 3u;
  xyzz_set_inf(acc);
  for (u32 i = lo + quad; i < hi; i += 64) {
    size_t slot = raw ? (size_t)i : (i == 0 ? 2 * (size_t)first_slot + 1 : 2 * ((size_t)first_slot + i));
    xyzz p, r;
    xyzz_load(p, &src[slot]);
    xyzz_add_quad(r, acc, p, role);
    acc = r;
  }
  if (role == 0) xyzz_store(&sh[quad], acc);
  __syncthreads();
  for (u32 s = 32; s >= 1; s >>= 1) {
    if (quad < s) {
      xyzz p, r;
      xyzz_load(p, &sh[quad + s]);
      xyzz_add_quad(r, acc, p, role);
      acc = r;
      if (role == 0) xyzz_store(&sh[quad], acc);
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_heavy(const u32* __restrict__ offsets, u32 nbt, u32 K,
                                               xyzz_mem* __restrict__ buckets,
                                               const xyzz_mem* __restrict__ partials, u32* __restrict__ heavy_count,
                                               const u32* __restrict__ heavy_list, xyzz_mem* __restrict__ tmp) {
  SMALL_KERNEL_PRIO();
  __shared__ xyzz_mem sh[64];
  __shared__ u32 ticket;
  const u32 nmed = heavy_count[HEAVY_MEDIUM];
  if (nmed) {  // the medium queue (k_fix_boundary): one quad per bucket, its partial sums as a serial chain
    const u32 role = threadIdx.x & 3u, nquads = (gridDim.x * blockDim.x) >> 2;
    for (u32 i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < nmed; i += nquads) {
      const u32 b = heavy_list[nbt - 1u - i];
      const u32 t0 = offsets[b] / K, t1 = (offsets[b + 1] - 1) / K;
      xyzz acc, p;
      xyzz_load(acc, &partials[2 * (size_t)t0 + 1]);
      xyzz_load(p, &partials[2 * (size_t)(t0 + 1)]);
      for (u32 t = t0 + 1; t <= t1; t++) {
        xyzz pn, r;
        if (t < t1) xyzz_load(pn, &partials[2 * (size_t)(t + 1)]);
        xyzz_add_quad(r, acc, p, role);
        acc = r;
        p = pn;
      }
      if (role == 0) xyzz_store(&buckets[b], acc);
    }
  }
  const u32 nheavy = heavy_count[0];
  u32* tickets = heavy_count + 2;
  // items are part-major (part 0 of every entry first) so that ordinary heavy buckets spread over
  // all workgroups; parts >= 1 exist only when some bucket was split
  const u32 nitems = nheavy * (heavy_count[1] ? HEAVY_PARTS : 1u);
  for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
    const u32 ent = heavy_list[item % nheavy], part = item / nheavy;
    const bool split = (ent & 0x80000000u) != 0;
    if (!split && part != 0) continue;  // uniform over the workgroup
    const u32 b = split ? (ent & 0xfffffu) : ent, slot = (ent >> 20) & 0x7ffu;
    const u32 t0 = offsets[b] / K, t1 = (offsets[b + 1] - 1) / K, np = t1 - t0 + 1;
    xyzz acc;
    if (!split) {
      heavy_reduce(acc, partials, 0, np, t0, false, sh);
      if (threadIdx.x == 0) xyzz_store(&buckets[b], acc);
      __syncthreads();
      continue;
    }
    const u32 per = (np + HEAVY_PARTS - 1) / HEAVY_PARTS;
    const u32 lo = part * per, hi = lo + per < np ? lo + per : np;
    heavy_reduce(acc, partials, lo < np ? lo : np, hi, t0, false, sh);
    if (threadIdx.x == 0) {
      xyzz_store(&tmp[(size_t)slot * HEAVY_PARTS + part], acc);
      __threadfence();
      ticket = atomicAdd(&tickets[slot], 1u);
    }
    __syncthreads();
    if (ticket == HEAVY_PARTS - 1) {  // every part of this bucket is in tmp
      __threadfence();
      heavy_reduce(acc, tmp + (size_t)slot * HEAVY_PARTS, 0, HEAVY_PARTS, 0, true, sh);
      if (threadIdx.x == 0) xyzz_store(&buckets[b], acc);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ bucket reduction
// sum_b (b+1) B_b = T + sum_k 2^k G_k with T = sum_b B_b, G_k = sum_{b: bit k} B_b.
// Level l (width w = nb >> l) folds the main row [0,w) and the l rows at bases nb>>j
// (j = 1..l) in half: B[base+i] += B[base+i+w/2].  After c-1 levels T = B[0], G_k = B[2^k].
__global__ void __launch_bounds__(64) k_fold(xyzz_mem* __restrict__ buckets, u32 nsets, u32 nb, u32 level) {
  SMALL_KERNEL_PRIO();
  // one quad per output element (quad-cooperative addition, see g1.cuh)
  u32 half = (nb >> level) >> 1;
  u32 rows = level + 1;
  u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 tid = gid >> 2;
  u32 role = (u32)gid & 3u;
  u64 per_set = (u64)rows * half;
  if (tid >= per_set * nsets) return;
  u32 set = (u32)(tid / per_set);
  u32 rem = (u32)(tid % per_set);
  u32 row = rem / half, i = rem % half;
  u32 base = row == 0 ? 0u : (nb >> row);
  xyzz_mem* p = buckets + (size_t)set * nb + base + i;
  xyzz a, b, r;
  xyzz_load(a, p);
  xyzz_load(b, p + half);
  xyzz_add_quad(r, a, b, role);
  if (role == 0) xyzz_store(p, r);
}

// The same level with one LANE per output element: the first levels of a 2^19-bucket reduction are
// 2^17..2^18 independent additions each -- throughput work, not a latency chain, so the quad form
// (4 lanes per addition, every lane executing every multiplication's instructions) is the wrong shape
// there.  The host picks this kernel for levels of >= 2^17 additions.  It is issue-bound like the
// accumulation (~3300 instructions per general addition; moving the 144-byte records through LDS with
// coalesced 16-byte accesses instead of nine strided loads per lane changed nothing: measured).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_fold_lane(xyzz_mem* __restrict__ buckets, u32 nsets, u32 nb, u32 level) {
  SMALL_KERNEL_PRIO();
  u32 half = (nb >> level) >> 1;
  u32 rows = level + 1;
  u64 per_set = (u64)rows * half;
  const u64 total = per_set * nsets, step = (u64)gridDim.x * blockDim.x;
  // grid-stride: the host launches at most one full round of four waves per SIMD (128 VGPRs by the attribute:
  // 0.225 -> 0.221 ms for the reduction against 134 VGPRs, three waves and two rounds)
  for (u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += step) {
    u32 set = (u32)(tid / per_set);
    u32 rem = (u32)(tid % per_set);
    u32 row = rem / half, i = rem % half;
    u32 base = row == 0 ? 0u : (nb >> row);
    xyzz_mem* p = buckets + (size_t)set * nb + base + i;
    xyzz a, b, r;
    xyzz_load(a, p);
    xyzz_load(b, p + half);
    xyzz_add(r, a, b);
    xyzz_store(p, r);
  }
}

// Levels l and l + 1 in one launch, one lane per FOUR inputs of a row: with h = half the row width of level
// l and q = h / 2, lane i of a row at `base` reads a = B[i], b = B[i+q], c = B[i+h], d = B[i+h+q] and leaves
// B[i] = (a + c) + (b + d).  In the main row the upper halves of both levels live on as rows of their own:
// B[i+q] = b + d (level l + 1's upper half) and B[i+h] = c + d (level l's upper half, folded once).
// Against two k_fold_lane launches: 4 loads + 3 stores per 4 additions instead of 8 + 4 (1 store per 3
// additions in the other rows), and a lane's next operands load while it computes -- the one-addition
// kernel's waves all load, then all compute, then all store.
__global__ void __launch_bounds__(256) k_fold_lane2(xyzz_mem* __restrict__ buckets, u32 nsets, u32 nb, u32 level) {
  SMALL_KERNEL_PRIO();
  const u32 h = (nb >> level) >> 1, q = h >> 1;
  const u32 rows = level + 1;
  const u64 per_set = (u64)rows * q;
  const u64 total = per_set * nsets, step = (u64)gridDim.x * blockDim.x;
  for (u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += step) {
    const u32 set = (u32)(tid / per_set);
    const u32 rem = (u32)(tid % per_set);
    const u32 row = rem / q, i = rem % q;
    xyzz_mem* p = buckets + (size_t)set * nb + (row == 0 ? 0u : (nb >> row)) + i;
    // (wave priorities stepping down with progress, as in k_accumulate, were measured here: no change)
    xyzz c, d, r, a;
    xyzz_load(c, p + h);
    xyzz_load(d, p + h + q);
    if (row == 0) {
      xyzz_add(r, c, d);
      xyzz_store(p + h, r);
    }
    xyzz_load(a, p);
    xyzz_add(r, a, c);      // a + c
    xyzz_load(a, p + q);
    xyzz_add(c, a, d);      // b + d
    if (row == 0) xyzz_store(p + q, c);
    xyzz_add(d, r, c);
    xyzz_store(p, d);
  }
}

// Canonical, non-Montgomery X,Y,ZZ,ZZZ (8 x u32 LE each) of one bucket for the host tail.
__device__ __forceinline__ void pack_point(u32* __restrict__ o, const xyzz_mem* src) {
  xyzz a;
  xyzz_load(a, src);
  fe* cs[4] = {&a.x, &a.y, &a.zz, &a.zzz};
  bool inf = xyzz_is_inf(a);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    u32 w[8];
    fe c = fe_canon(fe_from_mont(*cs[k]));
    fe_pack256(w, c);
#pragma unroll
    for (int m = 0; m < 8; m++) o[8 * k + m] = inf ? 0u : w[m];
  }
}
// T and G_0..G_{q-1} of every bucket set for the host tail (out[set][j]: j = 0 T, j = k + 1 G_k), followed
// by the call's two MSM_FLAG words: one device-to-host copy brings the host everything it needs.
// (Folding the last ten levels of a 2^19-bucket reduction and this packing into one workgroup-resident
// launch was measured: 99 us against 65 + 8 us for the launches.  A level is a quad addition's latency,
// ~5 us for a lone wave that issues one v_mad_i64_i32 per ~9.6 clk, not launch overhead, and a single
// workgroup needs two or three rounds for the wider levels.  Round 3, the same question from the other side: two
// narrow levels per launch with sixteen lanes per item -- quads 0..2 add a + c, b + d, c + d side by side, quad 0
// takes b + d over a DPP row shift and adds -- gave the same 0.206-0.209 ms for the reduction as fifteen single
// levels: two dependent quad additions cost what two launches of one cost.)
__global__ void __launch_bounds__(64) k_pack_results(const xyzz_mem* __restrict__ buckets, u32 nsets, u32 nb,
                                                     u32 q, u32* __restrict__ out, const u32* __restrict__ flags) {
  SMALL_KERNEL_PRIO();
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) {
    out[(size_t)nsets * (q + 1) * 32] = flags[0];
    out[(size_t)nsets * (q + 1) * 32 + 1] = flags[1];
  }
  if (tid >= nsets * (q + 1)) return;
  u32 set = tid / (q + 1), j = tid % (q + 1);
  pack_point(out + (size_t)tid * 32, buckets + (size_t)set * nb + (j == 0 ? 0u : (1u << (j - 1))));
}

// Small bucket sets (nb <= 512: the window widths of calls below ~2^16 points): every level of the
// reduction and the packing in ONE launch, one workgroup per set, a __syncthreads() per level.  On the
// device that is the same chain of quad additions as the q launches (the widest levels of a 512-bucket
// set take two rounds of the 192 quads); the host enqueues one launch instead of q + 1, and a small call
// is bound by exactly that.
#define FOLD_SMALL_MAX_NB 512u
__global__ void __launch_bounds__(768) k_fold_small(xyzz_mem* buckets, u32 nsets, u32 nb, u32 q, u32* __restrict__ out,
                                                    const u32* __restrict__ flags) {
  const u32 set = blockIdx.x;
  SMALL_KERNEL_PRIO();
  xyzz_mem* B = buckets + (size_t)set * nb;
  const u32 quad = threadIdx.x >> 2, role = threadIdx.x & 3u, nquads = blockDim.x >> 2;
  for (u32 l = 0; l < q; l++) {
    const u32 half = (nb >> l) >> 1;
    const u32 items = (l + 1) * half;
    for (u32 it = quad; it < items; it += nquads) {
      const u32 row = it / half, i = it % half;
      xyzz_mem* p = B + (row == 0 ? 0u : (nb >> row)) + i;
      xyzz a, b, sum;
      xyzz_load(a, p);
      xyzz_load(b, p + half);
      xyzz_add_quad(sum, a, b, role);
      if (role == 0) xyzz_store(p, sum);
    }
    __syncthreads();  // the level's sums are in memory before the next level reads them (one workgroup, one CU)
  }
  if (set == 0 && threadIdx.x == 0) {
    out[(size_t)nsets * (q + 1) * 32] = flags[0];
    out[(size_t)nsets * (q + 1) * 32 + 1] = flags[1];
  }
  if (threadIdx.x <= q) {
    const u32 j = threadIdx.x;
    pack_point(out + ((size_t)set * (q + 1) + j) * 32, B + (j == 0 ? 0u : (1u << (j - 1))));
  }
}

// ------------------------------------------------------------------ Marshal / Unmarshal
// err bits: 1 coordinate >= p, 2 not on curve; 4 (no error) some point is the point at infinity
__global__ void __launch_bounds__(256) k_points_from_be(const uint8_t* __restrict__ in, aff_mem* __restrict__ out,
                                                        u32 n, u32* __restrict__ err) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wx[8], wy[8];
  scalar_load_be(wx, in + (size_t)i * 64);
  scalar_load_be(wy, in + (size_t)i * 64 + 32);
  u32 any = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) any |= wx[j] | wy[j];
  if (any == 0) {
    aff_store_inf(&out[i]);
    atomicOr(err, 4u);  // not an error: the handle holds a point at infinity (k_accumulate then tests for it)
    return;
  }
  if (u256_geq_p(wx) || u256_geq_p(wy)) {
    atomicOr(err, 1u);
    aff_store_inf(&out[i]);
    return;
  }
  fe x = fe_to_mont(fe_unpack256(wx));
  fe y = fe_to_mont(fe_unpack256(wy));
  fe b3 = {FE_B3_LIMBS};
  fe lhs = fe_sqr(y);
  fe rhs = fe_add(fe_mul(fe_sqr(x), x), b3);
  if (!fe_is_zero_mod_p(fe_sub(lhs, rhs))) {
    atomicOr(err, 2u);
    aff_store_inf(&out[i]);
    return;
  }
  aff_store(&out[i], fe_canon(x), fe_canon(y));
}
__global__ void __launch_bounds__(256) k_points_to_be(const aff_mem* __restrict__ in, uint8_t* __restrict__ out,
                                                      u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x, y;
  u32 wx[8], wy[8];
  if (aff_load(x, y, &in[i], false)) {
    fe_pack256(wx, fe_canon(fe_from_mont(x)));
    fe_pack256(wy, fe_canon(fe_from_mont(y)));
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) wx[j] = wy[j] = 0;
  }
  scalar_store_be(out + (size_t)i * 64, wx);
  scalar_store_be(out + (size_t)i * 64 + 32, wy);
}

// Batch normalisation XYZZ -> affine (Montgomery's trick): thread t owns points [tB, (t+1)B),
// multiplies their ZZZ together (prefix products to `prefix`), inverts once, and walks back.
// 7 multiplications per point + one ~380-multiplication inversion per B points, instead of one
// inversion per point.  Used by every element-wise kernel below.
__global__ void __launch_bounds__(256) k_batch_to_affine(const xyzz_mem* __restrict__ in, aff_mem* __restrict__ out,
                                                         i32* __restrict__ prefix, u32 n, u32 B) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo64 = (u64)t * B;
  if (lo64 >= n) return;
  u32 lo = (u32)lo64;
  u32 hi = (n - lo > B) ? lo + B : n;
  fe run = fe_one();
  for (u32 i = lo; i < hi; i++) {
    xyzz a;
    xyzz_load(a, &in[i]);
    if (!xyzz_is_inf(a)) run = fe_mul(run, a.zzz);
#pragma unroll
    for (int j = 0; j < 9; j++) prefix[(size_t)i * 9 + j] = run.v[j];
  }
  fe inv = fe_inv(run);
  for (u32 i = hi; i-- > lo;) {
    xyzz a;
    xyzz_load(a, &in[i]);
    if (xyzz_is_inf(a)) {
      aff_store_inf(&out[i]);
      continue;
    }
    fe prev = fe_one();
    if (i > lo) {
#pragma unroll
      for (int j = 0; j < 9; j++) prev.v[j] = prefix[(size_t)(i - 1) * 9 + j];
    }
    fe zi3 = fe_mul(inv, prev);   // 1 / ZZZ_i
    inv = fe_mul(inv, a.zzz);
    fe zi = fe_mul(a.zz, zi3);    // 1 / Z
    fe zi2 = fe_sqr(zi);          // 1 / ZZ
    fe x = fe_canon(fe_mul(a.x, zi2));
    fe y = fe_canon(fe_mul(a.y, zi3));
    aff_store(&out[i], x, y);
  }
}

// next[i] = 2^c * prev[i]
__global__ void __launch_bounds__(256) k_precompute_level(const aff_mem* __restrict__ prev, xyzz_mem* __restrict__ next,
                                                          u32 n, u32 c) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x, y;
  xyzz a, r;
  xyzz_set_inf(a);
  if (aff_load(x, y, &prev[i], false)) {
    xyzz_dbl_affine(a, x, y);
    for (u32 j = 1; j < c; j++) {
      xyzz_dbl(r, a);
      a = r;
    }
  }
  xyzz_store(&next[i], a);
}

// ------------------------------------------------------------------ element-wise point ops
// acc = k * (x, y).  The signed representative of k is split by the endomorphism,
// |k| = k1 + k2 lambda with |k1|, |k2| < 2^129 (glv_split_dev), and one MSB-first joint
// double-and-add runs over A = +-P, B = +-phi(P) = (beta x, +-y) and A + B: 130 doublings and ~97
// additions (A, B mixed; A + B general) instead of 255 and ~128.
__device__ __forceinline__ void g1_scalar_mul(xyzz& acc, const fe& x, const fe& y_in, u32 (&k)[8]) {
  const bool neg = scalar_reduce_signed(k);
  u32 k1[5], k2[5];
  bool n1, n2;
  glv_split_dev(k, k1, n1, k2, n2);
  const fe beta = {FE_BETA_LIMBS};
  const fe bx = fe_mul(x, beta);
  const fe ya = (neg != n1) ? fe_neg(y_in) : y_in, yb = (neg != n2) ? fe_neg(y_in) : y_in;
  xyzz ab;
  xyzz_from_affine(ab, x, fe_norm(ya));
  xyzz_madd(ab, bx, yb);  // A + B (A == -B cannot happen: beta != 1; A == B neither)
  xyzz_set_inf(acc);
  const fe one = fe_one();
  for (int bit = 129; bit >= 0; bit--) {
    if (!xyzz_is_inf(acc)) {
      xyzz r;
      xyzz_dbl(r, acc);
      acc = r;
    }
    const u32 b1 = (sel5(k1, (u32)bit >> 5) >> (bit & 31)) & 1u, b2 = (sel5(k2, (u32)bit >> 5) >> (bit & 31)) & 1u;
    // the lanes of a wave need different operands: one general addition with a selected operand
    // (A, B as XYZZ with ZZ = ZZZ = 1) instead of three divergent paths
    if (b1 | b2) {
      xyzz T, r;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        T.x.v[j] = (b1 & b2) ? ab.x.v[j] : (b1 ? x.v[j] : bx.v[j]);
        T.y.v[j] = (b1 & b2) ? ab.y.v[j] : (b1 ? ya.v[j] : yb.v[j]);
        T.zz.v[j] = (b1 & b2) ? ab.zz.v[j] : one.v[j];
        T.zzz.v[j] = (b1 & b2) ? ab.zzz.v[j] : one.v[j];
      }
      xyzz_add(r, acc, T);
      acc = r;
    }
  }
}

// out[i] = s_i * in[i]   (broadcast_scalar = p > 0: s_(i mod p), so 1 is s_0 for all; broadcast_point: in[0] for all)
__global__ void __launch_bounds__(256) k_scalar_mul(const aff_mem* __restrict__ in, const uint8_t* __restrict__ scalars_be,
                                                    xyzz_mem* __restrict__ out, u32 n, u32 broadcast_scalar,
                                                    u32 broadcast_point) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x, y;
  xyzz acc;
  xyzz_set_inf(acc);
  if (aff_load(x, y, &in[broadcast_point ? 0 : i], false)) {
    u32 k[8];
    scalar_load_be(k, scalars_be + (size_t)(broadcast_scalar ? i % broadcast_scalar : i) * 32);
    g1_scalar_mul(acc, x, y, k);
  }
  xyzz_store(&out[i], acc);
}

// ---- the same for MANY points (h1[i] = yinvn[i] * H[i] over a whole generator vector, go/bp/main.go:156-159;
// the synthetic generators): one lane per point, throughput work.  The joint binary chain above makes a wave
// run an addition at EVERY bit (three of four lanes need one), 130 general additions of 14 products.  Here:
// signed fixed 4-bit windows over both GLV halves -- every lane adds at the same steps, 33 windows x 2
// additions -- with a per-lane table T[m] = m P, m = 1..8, normalised to AFFINE with one inversion (Montgomery's
// trick over the seven Z), so that every addition is a mixed one, and a Jacobian accumulator (g1.cuh: cheaper
// doublings): 130 doublings + 66 mixed additions + the table (7 additions, one inversion) instead of
// 130 x (doubling + general addition) -- 2^20 points 17.5 -> see DESIGN.md 6.
// Table memory: SMW_ENTRY_WORDS 32-bit words per entry, 8 entries per lane, in HBM (`table`; the caller
// launches at most SMW_MAX_POINTS points at a time): while it is being built an entry holds the Jacobian
// (X, Y, Z), afterwards the affine (x, y) in its first 18 words; a lane reads its entries as whole 128-byte
// lines.  The prefix products of the seven Z live in LDS (63 KiB per workgroup of 256).
#define SMW_ENTRY_WORDS 32u
#define SMW_ENTRIES 8u
#define SMW_MAX_POINTS ((size_t)1 << 20)
__device__ __forceinline__ void smw_store3(i32* e, const fe& a, const fe& b, const fe& c) {
  int4* q = reinterpret_cast<int4*>(e);
  i32 t[28];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    t[k] = a.v[k];
    t[9 + k] = b.v[k];
    t[18 + k] = c.v[k];
  }
  t[27] = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) q[k] = make_int4(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
}
__device__ __forceinline__ void smw_load2(fe& a, fe& b, const i32* e) {
  const int4* q = reinterpret_cast<const int4*>(e);
  i32 t[20];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    int4 v = q[k];
    t[4 * k] = v.x;
    t[4 * k + 1] = v.y;
    t[4 * k + 2] = v.z;
    t[4 * k + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
    a.v[k] = t[k];
    b.v[k] = t[9 + k];
  }
}
__device__ __forceinline__ void smw_load3(fe& a, fe& b, fe& c, const i32* e) {
  const int4* q = reinterpret_cast<const int4*>(e);
  i32 t[28];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    int4 v = q[k];
    t[4 * k] = v.x;
    t[4 * k + 1] = v.y;
    t[4 * k + 2] = v.z;
    t[4 * k + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
    a.v[k] = t[k];
    b.v[k] = t[9 + k];
    c.v[k] = t[18 + k];
  }
}
// signed 4-bit windows of a 130-bit magnitude, LSB first: digit j in [-8, 8] as (|d| | sign << 4), six digits
// of five bits per word
__device__ __forceinline__ void smw_recode(u32 (&pack)[6], const u32 (&k)[5]) {
#pragma unroll
  for (int w = 0; w < 6; w++) pack[w] = 0;
  u32 carry = 0;
#pragma unroll
  for (int j = 0; j < 33; j++) {
    const int bit = 4 * j;
    u32 raw = ((k[bit >> 5] >> (bit & 31)) & 15u) + carry;  // 4 j never straddles a word
    const bool big = raw > 8u;
    const u32 dig = big ? 16u - raw : raw;
    carry = big ? 1u : 0u;
    pack[j / 6] |= (dig | (big ? 16u : 0u)) << (5 * (j % 6));
  }
}
__device__ __forceinline__ u32 smw_digit(const u32 (&pack)[6], u32 j) {
  const u32 w = j / 6u;
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) r = (w == (u32)i) ? pack[i] : r;
  return (r >> (5u * (j - 6u * w))) & 31u;
}
#define SMW_BLOCK 128
__global__ void __launch_bounds__(SMW_BLOCK, 3) k_scalar_mul_win(const aff_mem* __restrict__ in,
                                                           const uint8_t* __restrict__ scalars_be,
                                                           xyzz_mem* __restrict__ out, u32 n, u32 broadcast_scalar,
                                                           u32 broadcast_point, i32* __restrict__ table) {
  __shared__ i32 pre[5][9][SMW_BLOCK];  // prefix products Z_2 ... Z_m for m = 3..7: [m - 3][limb][thread] (Z_2 itself is in the table)
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
  // Exactly THREE waves per SIMD: the per-lane tables of twelve waves per CU are 201 MB over the chip and stay in
  // the 256 MB last-level cache; with sixteen (LDS cut to fit eight workgroups) 2^20 points took 11.9 ms instead of
  // 10.8.  Six workgroups per CU fit the LDS either way, but once the kernel needed <= 128 VGPRs (127 after the
  // multiplier's constants moved to scalar registers) nothing held a SIMD to three waves and the hardware spread
  // the twelve unevenly: 12.3 ms.  Naming v135 makes the kernel's register count 136 = three waves per SIMD.
  asm volatile("v_mov_b32 v135, 0" ::: "v135");
  if (i >= n) return;  // no barrier below: every lane works on its own LDS column and table lines
  fe px, py;
  xyzz res;
  xyzz_set_inf(res);
  if (!aff_load(px, py, &in[broadcast_point ? 0 : i], false)) {
    xyzz_store(&out[i], res);
    return;
  }
  u32 pk1[6], pk2[6];
  bool s1, s2;  // signs of the two halves' base points: +-P, +-phi(P)
  {
    u32 k[8], k1[5], k2[5];
    scalar_load_be(k, scalars_be + (size_t)(broadcast_scalar ? i % broadcast_scalar : i) * 32);
    const bool neg = scalar_reduce_signed(k);
    bool n1, n2;
    glv_split_dev(k, k1, n1, k2, n2);
    s1 = neg != n1;
    s2 = neg != n2;
    smw_recode(pk1, k1);
    smw_recode(pk2, k2);
  }
  i32* const tab = table + (size_t)i * (SMW_ENTRIES * SMW_ENTRY_WORDS);
  // ---- table: m P for m = 1..8, Jacobian first
  {
    jac t;
    t.x = px;
    t.y = py;
    t.z = fe_one();
    smw_store3(tab, t.x, t.y, t.z);
    jac d;
    jac_dbl(d, t);  // 2P; P has prime order: no m P, m <= 8, is infinity and (m - 1) P != +-P
    t = d;
    fe run = t.z;
#pragma unroll 1
    for (u32 m = 2; m <= SMW_ENTRIES; m++) {
      if (m > 2) {
        jac_madd(t, px, py);
        run = fe_mul(run, t.z);
      }
      smw_store3(tab + (m - 1) * SMW_ENTRY_WORDS, t.x, t.y, t.z);
      if (m >= 3 && m < SMW_ENTRIES) {
#pragma unroll
        for (int k = 0; k < 9; k++) pre[m - 3][k][tid] = run.v[k];
      }
    }
    fe inv = fe_inv(run);  // 1 / (Z_2 ... Z_8)
#pragma unroll 1
    for (u32 m = SMW_ENTRIES; m >= 2; m--) {
      fe X, Y, Z;
      smw_load3(X, Y, Z, tab + (m - 1) * SMW_ENTRY_WORDS);
      fe zi = inv;
      if (m > 2) {
        fe prev;
        if (m == 3) {
          fe x2, y2;
          smw_load3(x2, y2, prev, tab + SMW_ENTRY_WORDS);  // Z_2
        } else {
#pragma unroll
          for (int k = 0; k < 9; k++) prev.v[k] = pre[m - 4][k][tid];
        }
        zi = fe_mul(inv, prev);  // 1 / Z_m
        inv = fe_mul(inv, Z);
      }
      const fe zi2 = fe_sqr(zi);
      const fe ax = fe_mul(X, zi2), ay = fe_mul(Y, fe_mul(zi2, zi));
      smw_store3(tab + (m - 1) * SMW_ENTRY_WORDS, ax, ay, Z);
    }
  }
  // ---- 33 windows, most significant first: 4 doublings, then +-T[|d1|] and +-phi(T[|d2|])
  const fe beta = {FE_BETA_LIMBS};
  jac acc;
  jac_set_inf(acc);
#pragma unroll 1
  for (int j = 32; j >= 0; j--) {
    // (wave priorities stepping down with the window number: 10.9 -> 10.75 ms at 2^20 alone, nothing in WNLA, go/bp slower)
    if (!jac_is_inf(acc)) {
#pragma unroll 1
      for (int r = 0; r < 4; r++) {
        jac t;
        jac_dbl(t, acc);
        acc = t;
      }
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const u32 f = h ? smw_digit(pk2, (u32)j) : smw_digit(pk1, (u32)j);
      const u32 dig = f & 15u;
      if (dig == 0) continue;
      fe tx, ty;
      smw_load2(tx, ty, tab + (dig - 1u) * SMW_ENTRY_WORDS);
      if (h) tx = fe_mul(tx, beta);
      const bool neg = ((f >> 4) != 0) != (h ? s2 : s1);
      jac_madd(acc, tx, neg ? fe_neg(ty) : ty);
    }
  }
  jac_to_xyzz(res, acc);
  xyzz_store(&out[i], res);
}

// The same for few points (the provers' small vectors: a call is one latency chain, not throughput
// work): one quad per point and GLV half, quad-cooperative doubling / addition, 2-bit windows over
// {P, 2P, 3P} -- 130 doublings + <= 65 additions in 3 + 4 multiplication stages each instead of
// 255 x (doubling + addition) executed serially by one lane.
__global__ void __launch_bounds__(256) k_scalar_mul_quad(const aff_mem* __restrict__ in,
                                                         const uint8_t* __restrict__ scalars_be,
                                                         xyzz_mem* __restrict__ out, u32 n, u32 broadcast_scalar,
                                                         u32 broadcast_point) {
  // GLV as in k_fold_points_quad: blockIdx.y = 0 computes k1 * P into out[i], blockIdx.y = 1 computes
  // k2 * P into out[n + i] (|k| = k1 + k2 lambda, glv_split_dev); k_glv_combine adds out[i] + phi(out[n + i])
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = gid >> 2, role = gid & 3u;
  if (i >= n) return;  // whole quads leave together
  out += (size_t)blockIdx.y * n;
  fe x, y;
  xyzz acc;
  xyzz_set_inf(acc);
  if (aff_load(x, y, &in[broadcast_point ? 0 : i], false)) {
    u32 k[8], k1[5], k2[5];
    bool n1, n2;
    scalar_load_be(k, scalars_be + (broadcast_scalar ? 0 : (size_t)i * 32));
    bool neg = scalar_reduce_signed(k);
    glv_split_dev(k, k1, n1, k2, n2);
    if (blockIdx.y) {
#pragma unroll
      for (int j = 0; j < 5; j++) k1[j] = k2[j];
      n1 = n2;
    }
    neg = neg != n1;
    xyzz P1, P2, P3;
    xyzz_from_affine(P1, x, fe_norm(neg ? fe_neg(y) : y));
    xyzz_dbl_quad(P2, P1, role);
    xyzz_add_quad(P3, P2, P1, role);
    for (int w = 64; w >= 0; w--) {  // 130 bits
      xyzz t;
      xyzz_dbl_quad(t, acc, role);
      xyzz_dbl_quad(acc, t, role);
      u32 d = (sel5(k1, (u32)(2 * w) >> 5) >> ((2 * w) & 31)) & 3u;
      if (d == 0) continue;  // uniform within the quad
      xyzz T = d == 1 ? P1 : (d == 2 ? P2 : P3);
      xyzz_add_quad(t, acc, T, role);
      acc = t;
    }
  }
  if (role == 0) xyzz_store(&out[i], acc);
}

__global__ void __launch_bounds__(256) k_add_points(const aff_mem* __restrict__ a, const aff_mem* __restrict__ b,
                                                    xyzz_mem* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x1, y1, x2, y2;
  bool f1 = aff_load(x1, y1, &a[i], false);
  bool f2 = aff_load(x2, y2, &b[i], false);
  xyzz acc;
  xyzz_set_inf(acc);
  if (f1) xyzz_madd(acc, x1, y1);
  if (f2) xyzz_madd(acc, x2, y2);
  xyzz_store(&out[i], acc);
}

// Joint sparse form (Solinas) of the two fold scalars, computed once on the host: digit pairs
// (d_lo, d_hi) in {0, +1, -1}^2, 4 bits per position (0 -> 0, +1 -> 1, -1 -> 3), at most 257
// positions, about half of them non-zero.
struct jsf_arg {
  u32 w[34];
  u32 len;
};

// out[i] = s_lo * lo[i*stride] + s_hi * hi[i*stride], one QUAD per output point.  Both scalars
// are the same for every point (generator folding, go/bp/main.go:355-356,
// go/bppp/docs_test.go:1325-1326; vectorPointMulOnScalar go/bp/util.go:177-183 with s_hi = 0),
// so the whole grid follows one joint double-and-add chain without divergence: per digit one
// quad-cooperative doubling (3 multiply stages) and, for a non-zero pair, one quad-cooperative
// addition of +-P, +-Q, +-(P+Q) or +-(P-Q) (4 stages).  out has max(n_lo, n_hi) entries; a
// missing side counts as infinity (vectorPointsAdd padding, go/bppp/util.go:279-293).
//
// GLV: the call is one latency chain, so the scalars arrive split by the endomorphism,
// s = s' + s'' lambda with |s'|, |s''| < 2^129 (glv_split in bn256msm.hip): blockIdx.y = 0 runs the
// chain of (s_lo', s_hi') into out[i], blockIdx.y = 1 the chain of (s_lo'', s_hi'') into
// out[n + i] -- the same P, Q tables, half the digits, side by side on otherwise idle SIMDs -- and
// k_glv_combine adds out[i] + phi(out[n + i]).
__global__ void __launch_bounds__(256) k_fold_points_quad(const aff_mem* __restrict__ lo, const aff_mem* __restrict__ hi,
                                                          u32 stride, jsf_arg jsf0, jsf_arg jsf1,
                                                          xyzz_mem* __restrict__ out, u32 n_lo, u32 n_hi) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = gid >> 2, role = gid & 3u;
  u32 n = n_lo > n_hi ? n_lo : n_hi;
  if (i >= n) return;  // a quad leaves together
  const jsf_arg& jsf = blockIdx.y ? jsf1 : jsf0;
  out += (size_t)blockIdx.y * n;
  fe x, y;
  xyzz P, Q, S, D, acc;
  xyzz_set_inf(P);
  xyzz_set_inf(Q);
  if (i < n_lo && aff_load(x, y, &lo[(size_t)i * stride], false)) xyzz_from_affine(P, x, y);
  if (i < n_hi && aff_load(x, y, &hi[(size_t)i * stride], false)) xyzz_from_affine(Q, x, y);
  xyzz_add_quad(S, P, Q, role);
  {
    xyzz Qn = Q;
    Qn.y = fe_neg(Q.y);
    xyzz_add_quad(D, P, Qn, role);
  }
  xyzz_set_inf(acc);
  for (int j = (int)jsf.len - 1; j >= 0; j--) {
    xyzz t;
    xyzz_dbl_quad(t, acc, role);
    acc = t;
    u32 d = (jsf.w[j >> 3] >> ((j & 7) * 4)) & 15u;
    if (d == 0) continue;
    u32 d0 = d & 3u, d1 = d >> 2;
    xyzz T;
    bool neg;
    if (d1 == 0) {
      T = P;
      neg = d0 == 3u;
    } else if (d0 == 0) {
      T = Q;
      neg = d1 == 3u;
    } else if (d0 == d1) {
      T = S;
      neg = d0 == 3u;
    } else {
      T = D;
      neg = d0 == 3u;
    }
    if (neg) T.y = fe_neg(T.y);
    xyzz_add_quad(t, acc, T, role);
    acc = t;
  }
  if (role == 0) xyzz_store(&out[i], acc);
}

// out[i] += phi(out[n + i]), phi(X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ): joins the two GLV chains of
// k_fold_points_quad.  One quad per output.
__global__ void __launch_bounds__(256) k_glv_combine(xyzz_mem* __restrict__ out, u32 n) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = gid >> 2, role = gid & 3u;
  if (i >= n) return;
  xyzz a, b, r;
  xyzz_load(a, &out[i]);
  xyzz_load(b, &out[(size_t)n + i]);
  const fe beta = {FE_BETA_LIMBS};
  b.x = fe_mul(b.x, beta);  // infinity (all-zero limbs) stays infinity
  xyzz_add_quad(r, a, b, role);
  if (role == 0) xyzz_store(&out[i], r);
}

// The same fold for many outputs (the first rounds of a 2^20-point WNLA / inner-product argument):
// ONE LANE per output.  The quad kernel needs 256 VGPRs (one wave per SIMD, every multiplication
// waits out its own latency) and spends 20 lane-multiplications per digit; here the chain is the
// same joint sparse form, still without divergence (the digits are the grid's), at ~12
// multiplications per digit and enough waves per SIMD to issue back to back.  P, Q stay affine
// (mixed additions), P+Q and P-Q stay XYZZ (general additions).
__global__ void __launch_bounds__(256, 2) k_fold_points_xyzz(const aff_mem* __restrict__ lo, const aff_mem* __restrict__ hi,
                                                             u32 stride, jsf_arg jsf0, jsf_arg jsf1,
                                                             xyzz_mem* __restrict__ out, u32 n_lo, u32 n_hi) {
  // (round 2's form, kept for A/B runs: BN256_FOLD_XYZZ=1)
  // GLV with ONE shared doubling chain (throughput, not latency, is what counts here): jsf0 holds
  // the digits of (s_lo', s_hi'), jsf1 those of (s_lo'', s_hi''), s = s' + s'' lambda; a digit of
  // jsf1 adds phi(T) = (beta X, Y, ZZ, ZZZ) of the same table entry T.  130 doublings instead of 257.
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = n_lo > n_hi ? n_lo : n_hi;
  if (i >= n) return;
  // P+Q and P-Q live in LDS ([set][word][thread]: conflict-free), not in 72 more registers
  __shared__ i32 sd[2][36][256];
  fe px, py, qx, qy;
  bool hp = i < n_lo && aff_load(px, py, &lo[(size_t)i * stride], false);
  bool hq = i < n_hi && aff_load(qx, qy, &hi[(size_t)i * stride], false);
  xyzz acc;
#pragma unroll
  for (int which = 0; which < 2; which++) {
    xyzz_set_inf(acc);
    if (hp) xyzz_from_affine(acc, px, py);
    if (hq) xyzz_madd(acc, qx, which ? fe_neg(qy) : qy);
#pragma unroll
    for (int k = 0; k < 9; k++) {
      sd[which][k][threadIdx.x] = acc.x.v[k];
      sd[which][9 + k][threadIdx.x] = acc.y.v[k];
      sd[which][18 + k][threadIdx.x] = acc.zz.v[k];
      sd[which][27 + k][threadIdx.x] = acc.zzz.v[k];
    }
  }
  const fe beta = {FE_BETA_LIMBS};
  xyzz_set_inf(acc);
  const int len = (int)(jsf0.len > jsf1.len ? jsf0.len : jsf1.len);
  for (int j = len - 1; j >= 0; j--) {
    {
      xyzz t;
      xyzz_dbl(t, acc);
      acc = t;
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {  // h = 1: the phi half
      const u32 d = ((h ? jsf1.w[j >> 3] : jsf0.w[j >> 3]) >> ((j & 7) * 4)) & 15u;  // words beyond len are zero
      if (d == 0) continue;  // uniform over the grid
      const u32 d0 = d & 3u, d1 = d >> 2;
      const bool neg = (d0 ? d0 : d1) == 3u;
      if (d1 == 0 || d0 == 0) {
        const bool useq = d1 != 0;
        if (useq ? hq : hp) {
          fe tx = useq ? qx : px, ty = useq ? qy : py;
          if (h) tx = fe_mul(tx, beta);
          xyzz_madd(acc, tx, neg ? fe_neg(ty) : ty);
        }
      } else {
        const u32 which = d0 == d1 ? 0u : 1u;
        xyzz T, r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          T.x.v[k] = sd[which][k][threadIdx.x];
          T.y.v[k] = sd[which][9 + k][threadIdx.x];
          T.zz.v[k] = sd[which][18 + k][threadIdx.x];
          T.zzz.v[k] = sd[which][27 + k][threadIdx.x];
        }
        if (h) T.x = fe_mul(T.x, beta);
        if (neg) T.y = fe_neg(T.y);
        xyzz_add(r, acc, T);
        acc = r;
      }
    }
  }
  xyzz_store(&out[i], acc);
}

// Round 3's form of the same fold: P + Q and P - Q are made AFFINE -- lambda = (yQ -+ yP) / (xQ - xP), both from
// ONE inversion per output (~20 k instructions against ~550 k for the chain) -- so that all ~130 additions of
// the chain are mixed ones (10 products; the XYZZ table entries cost a general addition of 14 for half of
// them), and the accumulator is Jacobian (g1.cuh: the doubling is 5 reductions instead of 8).  LDS holds
// (x, y) of P + Q and P - Q and beta x of all four table points: [entry][limb][thread], 72 KiB as before.
enum { FT_XS, FT_YS, FT_XD, FT_YD, FT_BPX, FT_BQX, FT_BXS, FT_BXD, FT_ENTRIES };
__global__ void __launch_bounds__(256, 2) k_fold_points(const aff_mem* __restrict__ lo, const aff_mem* __restrict__ hi,
                                                        u32 stride, jsf_arg jsf0, jsf_arg jsf1,
                                                        xyzz_mem* __restrict__ out, u32 n_lo, u32 n_hi) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 tid = threadIdx.x;
  u32 n = n_lo > n_hi ? n_lo : n_hi;
  if (i >= n) return;
  __shared__ i32 sd[FT_ENTRIES][9][256];
  auto put = [&](int e, const fe& v) {
#pragma unroll
    for (int k = 0; k < 9; k++) sd[e][k][tid] = v.v[k];
  };
  auto get = [&](int e) {
    fe v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.v[k] = sd[e][k][tid];
    return v;
  };
  fe px, py, qx, qy;
  const bool hp = i < n_lo && aff_load(px, py, &lo[(size_t)i * stride], false);
  const bool hq = i < n_hi && aff_load(qx, qy, &hi[(size_t)i * stride], false);
  bool hs = false, hd = false;  // P + Q, P - Q are not infinity
  {
    fe xs = fe_zero(), ys = fe_zero(), xd = fe_zero(), yd = fe_zero();
    if (hp && hq) {
      const fe dx = fe_sub(qx, px);
      if (!fe_is_zero_mod_p(dx)) {
        const fe inv = fe_inv(dx);
        const fe ls = fe_mul(fe_sub(qy, py), inv), ld = fe_mul(fe_neg(fe_add(qy, py)), inv);
        const fe sx = fe_add(px, qx);
        xs = fe_norm(fe_sub(fe_sqr(ls), sx));
        ys = fe_norm(fe_sub(fe_mul(ls, fe_sub(px, xs)), py));
        xd = fe_norm(fe_sub(fe_sqr(ld), sx));
        yd = fe_norm(fe_sub(fe_mul(ld, fe_sub(px, xd)), py));
        hs = hd = true;
      } else {  // P == +-Q: one of the two is 2P, the other infinity
        xyzz t;
        xyzz_dbl_affine(t, px, py);
        if (fe_is_zero_mod_p(fe_sub(qy, py)))
          hs = xyzz_to_affine(xs, ys, t);
        else
          hd = xyzz_to_affine(xd, yd, t);
      }
    } else if (hp) {  // Q is infinity: P + Q = P - Q = P
      xs = xd = px;
      ys = yd = py;
      hs = hd = true;
    } else if (hq) {  // P is infinity: P + Q = Q, P - Q = -Q
      xs = xd = qx;
      ys = qy;
      yd = fe_neg(qy);
      hs = hd = true;
    }
    const fe beta = {FE_BETA_LIMBS};
    put(FT_XS, xs);
    put(FT_YS, ys);
    put(FT_XD, xd);
    put(FT_YD, yd);
    put(FT_BPX, hp ? fe_mul(px, beta) : fe_zero());
    put(FT_BQX, hq ? fe_mul(qx, beta) : fe_zero());
    put(FT_BXS, fe_mul(xs, beta));
    put(FT_BXD, fe_mul(xd, beta));
  }
  jac acc;
  jac_set_inf(acc);
  const int len = (int)(jsf0.len > jsf1.len ? jsf0.len : jsf1.len);
#pragma unroll 1
  for (int j = len - 1; j >= 0; j--) {
    // (wave priorities stepping down with j: 6.70 -> 6.56 ms at 2^19 outputs alone, nothing in WNLA; not kept)
    {
      jac t;
      jac_dbl(t, acc);
      acc = t;
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {  // h = 1: the phi half, (beta x, y) of the same table entry
      const u32 d = ((h ? jsf1.w[j >> 3] : jsf0.w[j >> 3]) >> ((j & 7) * 4)) & 15u;  // words beyond len are zero
      if (d == 0) continue;  // uniform over the grid, as every branch on d below
      const u32 d0 = d & 3u, d1 = d >> 2;
      const bool neg = (d0 ? d0 : d1) == 3u;
      fe tx, ty;
      bool ok;
      if (d1 == 0) {
        ok = hp;
        ty = py;
        tx = h ? get(FT_BPX) : px;
      } else if (d0 == 0) {
        ok = hq;
        ty = qy;
        tx = h ? get(FT_BQX) : qx;
      } else if (d0 == d1) {
        ok = hs;
        ty = get(FT_YS);
        tx = get(h ? FT_BXS : FT_XS);
      } else {
        ok = hd;
        ty = get(FT_YD);
        tx = get(h ? FT_BXD : FT_XD);
      }
      if (ok) jac_madd(acc, tx, neg ? fe_neg(ty) : ty);
    }
  }
  xyzz res;
  jac_to_xyzz(res, acc);
  xyzz_store(&out[i], res);
}

// out[i] = a_i * g + b_i * h: n independent 2-term Pedersen commitments with per-output scalars --
// com() of go/bp/util.go:92-94 batched, and X[i] = x0[i]*GenH + x1[i]*GenG of go/ve-ca/main.go:37-40.
// One quad per output, joint binary double-and-add over {g, h, g+h}; gh holds g, h (affine).
__global__ void __launch_bounds__(256) k_com_batch_quad(const aff_mem* __restrict__ gh, const uint8_t* __restrict__ a_be,
                                                        const uint8_t* __restrict__ b_be, xyzz_mem* __restrict__ out,
                                                        u32 n) {
  // GLV as in k_scalar_mul_quad: blockIdx.y = 0 runs (a1, b1) into out[i], blockIdx.y = 1 runs (a2, b2)
  // into out[n + i], a = a1 + a2 lambda, b = b1 + b2 lambda; k_glv_combine joins them.  The signs
  // of the two halves go into the points: P = +-g, Q = +-h.
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = gid >> 2, role = gid & 3u;
  if (i >= n) return;
  out += (size_t)blockIdx.y * n;
  u32 k[8], ka[5], kb[5], t5[5];
  bool na, nb, nt;
  scalar_load_be(k, a_be + (size_t)i * 32);
  bool neg = scalar_reduce_signed(k);
  glv_split_dev(k, ka, na, t5, nt);
  if (blockIdx.y) {
#pragma unroll
    for (int j = 0; j < 5; j++) ka[j] = t5[j];
    na = nt;
  }
  na = na != neg;
  scalar_load_be(k, b_be + (size_t)i * 32);
  neg = scalar_reduce_signed(k);
  glv_split_dev(k, kb, nb, t5, nt);
  if (blockIdx.y) {
#pragma unroll
    for (int j = 0; j < 5; j++) kb[j] = t5[j];
    nb = nt;
  }
  nb = nb != neg;
  fe x, y;
  xyzz P, Q, S, acc;
  xyzz_set_inf(P);
  xyzz_set_inf(Q);
  if (aff_load(x, y, &gh[0], na)) xyzz_from_affine(P, x, y);
  if (aff_load(x, y, &gh[1], nb)) xyzz_from_affine(Q, x, y);
  xyzz_add_quad(S, P, Q, role);
  xyzz_set_inf(acc);
  for (int bit = 129; bit >= 0; bit--) {
    xyzz t;
    xyzz_dbl_quad(t, acc, role);
    acc = t;
    u32 da = (sel5(ka, (u32)bit >> 5) >> (bit & 31)) & 1u;
    u32 db = (sel5(kb, (u32)bit >> 5) >> (bit & 31)) & 1u;
    if ((da | db) == 0) continue;
    xyzz T = (da & db) ? S : (da ? P : Q);
    xyzz_add_quad(t, acc, T, role);
    acc = t;
  }
  if (role == 0) xyzz_store(&out[i], acc);
}

// ------------------------------------------------------------------ synthetic workload
__device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// element i: four mix64 words (little-endian 256-bit value), reduced mod n; zero_to_one maps 0 -> 1
__global__ void __launch_bounds__(256) k_synth_scalars(uint8_t* __restrict__ out_be, u64 seed, u64 first, u32 n,
                                                       u32 zero_to_one) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 e = first + i;
  u32 k[8], t[8], nn[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u64 v = mix64(seed + (4 * e + j + 1) * 0x9E3779B97F4A7C15ULL);
    k[2 * j] = (u32)v;
    k[2 * j + 1] = (u32)(v >> 32);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) nn[j] = fr_n_word(j);
  u32 br = u256_sub(t, k, nn);
  u32 any = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    k[j] = br ? k[j] : t[j];
    any |= k[j];
  }
  if (zero_to_one && any == 0) k[0] = 1;
  scalar_store_be(out_be + (size_t)i * 32, k);
}

// ------------------------------------------------------------------ self test
__global__ void __launch_bounds__(256) k_selftest(u32* __restrict__ errs, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wa[8], wb[8], wc[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u64 a = mix64(0x1111 + 12ull * i + j), b = mix64(0x2222 + 12ull * i + 4 + j), c = mix64(0x3333 + 12ull * i + 8 + j);
    wa[2 * j] = (u32)a; wa[2 * j + 1] = (u32)(a >> 32);
    wb[2 * j] = (u32)b; wb[2 * j + 1] = (u32)(b >> 32);
    wc[2 * j] = (u32)c; wc[2 * j + 1] = (u32)(c >> 32);
  }
  wa[7] &= 0x7fffffffu; wb[7] &= 0x7fffffffu; wc[7] &= 0x7fffffffu;  // < 2^255 < p
  if (i == 0) { for (int j = 0; j < 8; j++) wa[j] = 0; }
  if (i == 1) { for (int j = 0; j < 8; j++) wa[j] = (j == 0); }
  fe a = fe_unpack256(wa), b = fe_unpack256(wb), c = fe_unpack256(wc);
  u32 bad = 0;
  // pack/unpack round trip
  u32 w2[8];
  fe_pack256(w2, a);
  for (int j = 0; j < 8; j++) bad |= (w2[j] != wa[j]) ? 1u : 0u;
  // asm multiplier == portable multiplier (compare canonical forms)
  fe m1 = fe_canon(fe_mul(a, b)), m2 = fe_canon(fe_mul_portable(a, b));
  for (int j = 0; j < 9; j++) bad |= (m1.v[j] != m2.v[j]) ? 2u : 0u;
  // a*b + c*d under one reduction == the two products added (signed un-normalised operands)
  {
    fe d1 = fe_sub(b, c), d2 = fe_sub(c, a);
    fe f1 = fe_canon(fe_mul2add(d1, d2, fe_neg(a), b)), f2 = fe_canon(fe_sub(fe_mul(d1, d2), fe_mul(a, b)));
    for (int j = 0; j < 9; j++) bad |= (f1.v[j] != f2.v[j]) ? 128u : 0u;
    fe g1 = fe_mul2add(d1, d2, fe_neg(a), b), g2 = fe_mul2add_portable(d1, d2, fe_neg(a), b);  // limb-identical
    for (int j = 0; j < 9; j++) bad |= (g1.v[j] != g2.v[j]) ? 128u : 0u;
  }
  // device GLV split: |k1|, |k2| < 2^129 and k1 + k2 lambda == m (mod n), m the signed representative
  {
    u32 m[8], k1[5], k2[5];
#pragma unroll
    for (int j = 0; j < 8; j++) m[j] = wb[j];
    if (i == 2) { for (int j = 0; j < 8; j++) m[j] = 0xffffffffu; }
    if (i == 3) { for (int j = 0; j < 8; j++) m[j] = fr_n_word(j) - (j == 0 ? 1u : 0u); }
    if (i == 4) { for (int j = 0; j < 8; j++) m[j] = fr_half_word(j); }
    if (i == 5) { for (int j = 0; j < 8; j++) m[j] = 0; }
    scalar_reduce_signed(m);
    bool n1, n2;
    glv_split_dev(m, k1, n1, k2, n2);
    if (k1[4] > 1u || k2[4] > 1u) bad |= 512u;
    const u32 lamw[8] = {0x264ac23du, 0x3be518b5u, 0xf8bfcd51u, 0xcf9176e4u, 0x07e9f9eeu, 0x96ac037fu, 0x1u, 0u};
    fr f1 = fr_zero(), f2 = fr_zero(), lam, fm;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      lam.w[j] = lamw[j];
      fm.w[j] = m[j];
      if (j < 5) {
        f1.w[j] = k1[j];
        f2.w[j] = k2[j];
      }
    }
    fr t = fr_mont(fr_to_mont(f2), lam);  // k2 * lambda mod n
    auto fneg = [](const fr& v) {
      u32 nn[8], o[8], any = 0;
      for (int j = 0; j < 8; j++) { nn[j] = fr_n_word(j); any |= v.w[j]; }
      u256_sub(o, nn, v.w);
      fr q;
      for (int j = 0; j < 8; j++) q.w[j] = any ? o[j] : 0u;
      return q;
    };
    if (n1) f1 = fneg(f1);
    if (n2) t = fneg(t);
    fr sum = fr_add(f1, t);
    for (int j = 0; j < 8; j++) bad |= (sum.w[j] != fm.w[j]) ? 1024u : 0u;
  }
  // associativity and distributivity
  fe l = fe_canon(fe_mul(fe_mul(a, b), c)), r = fe_canon(fe_mul(a, fe_mul(b, c)));
  for (int j = 0; j < 9; j++) bad |= (l.v[j] != r.v[j]) ? 4u : 0u;
  l = fe_canon(fe_mul(a, fe_norm(fe_add(b, c))));
  r = fe_canon(fe_add(fe_mul(a, b), fe_mul(a, c)));
  for (int j = 0; j < 9; j++) bad |= (l.v[j] != r.v[j]) ? 8u : 0u;
  // subtraction and negative values
  l = fe_canon(fe_mul(a, fe_norm(fe_sub(b, c))));
  r = fe_canon(fe_sub(fe_mul(a, b), fe_mul(a, c)));
  for (int j = 0; j < 9; j++) bad |= (l.v[j] != r.v[j]) ? 16u : 0u;
  // inversion (b != 0 with overwhelming probability)
  fe bm = fe_to_mont(b);
  fe one = fe_canon(fe_mul(bm, fe_inv(bm)));
  fe ref1 = fe_canon(fe_one());
  for (int j = 0; j < 9; j++) bad |= (one.v[j] != ref1.v[j]) ? 32u : 0u;
  // to/from Montgomery round trip
  fe rt = fe_canon(fe_from_mont(fe_to_mont(c)));
  fe cc = fe_canon(c);
  for (int j = 0; j < 9; j++) bad |= (rt.v[j] != cc.v[j]) ? 64u : 0u;
  if (bad) atomicOr(errs, bad);
}
