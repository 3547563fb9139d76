// libbn256msm: C ABI (include/bn256msm.h) + host orchestration of the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/bn256msm.h"
#include "host_g1.h"
#include "kernels.cuh"

#define HIPCHK(x)                         \
  do {                                    \
    hipError_t e_ = (x);                  \
    if (e_ != hipSuccess) {               \
      ctx_note_hip(e_, __LINE__);         \
      return BN256_EHIP;                  \
    }                                     \
  } while (0)

static thread_local char g_hip_msg[256];
static void ctx_note_hip(hipError_t e, int line) {
  snprintf(g_hip_msg, sizeof g_hip_msg, "HIP error: %s (bn256msm.hip:%d)", hipGetErrorString(e), line);
}

// ---------------------------------------------------------------------------- static scratch
// bn256_msm_dev_noalloc (include/bn256msm.h) runs the MSM pipeline without hipMalloc / hipFree: its
// workspaces are carved from this array instead of the context's grow-only hipMalloc blocks.  Every
// GPU the library runs on holds its own copy (at its own address) until the process ends.  In host
// code the array's NAME is only a stand-in: the device address comes from hipGetSymbolAddress, once
// per device (scratch_resolve, called by bn256_ctx_create with that device current), and only that
// pointer is ever handed to kernels, memsets or copies.
static __device__ __align__(256) unsigned char scratch[BN256_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static unsigned char* g_scratch_ptr[16];  // per HIP device ordinal; nullptr = not resolved yet

static int scratch_resolve(int device) {
  if (device < 0 || device >= 16) return BN256_OK;  // such a device simply has no allocation-free entry
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return BN256_OK;
  void* ptr = nullptr;
  hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));  // the current device's copy
  if (e != hipSuccess || !ptr) {
    ctx_note_hip(e != hipSuccess ? e : hipErrorInvalidSymbol, __LINE__);
    return BN256_EHIP;
  }
  g_scratch_ptr[device] = (unsigned char*)ptr;
  return BN256_OK;
}

// ---------------------------------------------------------------------------- handle memory
// Handles (point sets, Fr vectors) come and go once per round in the provers' folding loops
// (go/bp/main.go:355-358, go/bppp/docs_test.go:1318-1326): hipMalloc / hipFree cost 0.1 ms each and
// hipFree waits for the whole device.  Freed blocks are therefore kept per device and handed out
// again (smallest cached block between 1x and 4x the request); the cache is bounded and is released
// when the last context of the process is destroyed.
namespace pool {
constexpr size_t kMaxCached = (size_t)8 << 30;
struct Block {
  void* p;
  size_t cap;
};
static std::mutex g_mu;
static std::multimap<size_t, void*> g_free[16];  // per device: capacity -> block
static size_t g_cached[16];
static int g_contexts;

// BN256_FAIL_ALLOC_ABOVE=<bytes> (environment, read once): every device allocation larger than that
// fails with hipErrorOutOfMemory -- lets the tests walk the out-of-memory paths without filling a GPU
static size_t alloc_limit() {
  static const size_t v = [] {
    const char* e = getenv("BN256_FAIL_ALLOC_ABOVE");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0;
  }();
  return v;
}
static hipError_t dev_malloc(void** out, size_t bytes) {
  if (alloc_limit() && bytes > alloc_limit()) {
    *out = nullptr;
    return hipErrorOutOfMemory;
  }
  return hipMalloc(out, bytes);
}
static hipError_t alloc(void** out, size_t bytes, size_t* cap, int* dev) {
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return e;
  *dev = d;
  bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  if (d >= 0 && d < 16) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_free[d].lower_bound(bytes);
    if (it != g_free[d].end() && it->first <= 4 * bytes) {
      *out = it->second;
      *cap = it->first;
      g_cached[d] -= it->first;
      g_free[d].erase(it);
      return hipSuccess;
    }
  }
  *cap = bytes;
  e = dev_malloc(out, bytes);
  if (e == hipErrorOutOfMemory && d >= 0 && d < 16) {  // give the cache back and try once more
    {
      std::lock_guard<std::mutex> lk(g_mu);
      for (auto& kv : g_free[d]) (void)hipFree(kv.second);
      g_free[d].clear();
      g_cached[d] = 0;
    }
    e = dev_malloc(out, bytes);
  }
  return e;
}
// The caller guarantees that no enqueued work still uses the block (every element-wise entry point
// synchronises its stream before it returns; the free functions wait for submitted MSM calls).
static void release(void* p, size_t cap, int dev) {
  if (!p) return;
  if (dev >= 0 && dev < 16) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_contexts > 0 && g_cached[dev] + cap <= kMaxCached) {
      g_free[dev].emplace(cap, p);
      g_cached[dev] += cap;
      return;
    }
  }
  (void)hipFree(p);
}
static void context_created() {
  std::lock_guard<std::mutex> lk(g_mu);
  g_contexts++;
}
static void context_destroyed() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (--g_contexts > 0) return;
  for (int d = 0; d < 16; d++) {
    if (g_free[d].empty()) continue;
    (void)hipSetDevice(d);
    for (auto& kv : g_free[d]) (void)hipFree(kv.second);
    g_free[d].clear();
    g_cached[d] = 0;
  }
}
}  // namespace pool

struct bn256_bases {
  aff_mem* d;     // [levels][n]
  size_t n;       // points per level
  int levels;     // 1, or W when precomputed
  int c;          // window bits the levels were built for (0 when levels == 1)
  bool no_inf;    // proven to hold no point at infinity (k_accumulate skips the test)
  u32 top_scale;  // the top level holds (1 / top_scale) * 2^(c (W-1)) * P and its digits are multiplied by top_scale
  size_t cap;     // bytes of the block behind d (pool)
  int dev;
  // handle of a multi-device context (bn256_ctx_create_multi): d == nullptr, shard s holds the
  // contiguous slice [shard_off[s], shard_off[s + 1]) as an ordinary handle on its device
  int nshard;
  bn256_bases* shard[BN256_MAX_DEVICES];
  size_t shard_off[BN256_MAX_DEVICES + 1];
};

enum { WS_SCALARS, WS_RANK, WS_SKEY, WS_SVAL, WS_COUNTS, WS_OFFSETS, WS_BLOCKSUM, WS_BUCKETS, WS_PARTIALS,
       WS_PACK, WS_TMP, WS_GATHER, WS_LASTB, WS_GBASE, WS_SUBHIST, WS_HEAVYTMP, WS_PEER_A, WS_PEER_B, WS_SMTABLE, WS_CHUNKB, WS_COUNT };

struct bn256_ctx {
  int device;
  hipStream_t stream;
  hipStream_t copy_stream;  // scalar uploads of bn256_msm, pass by pass, beside the stream's kernels
  hipEvent_t ev_copy[2];
  void* ws[WS_COUNT];
  size_t cap[WS_COUNT];
  hipEvent_t ev[8];
  float last_ms[7];
  int tune_c, tune_k;
  int phase_timing;  // bn256_ctx_set_phase_timing: record the per-phase events of bn256_last_timing
  size_t msm_chunk;  // 0 = MSM_CHUNK; BN256_MSM_CHUNK_LOG2 in the environment overrides (experiments)
  int sort1;         // BN256_SORT1=1 forces the single-pass sort (experiments)
  // bn256_msm_dev_noalloc: while set, ws_reserve carves ws[] from the device's static scratch array
  bool fixed_ws;
  size_t fixed_used;
  u32* h_pack;      // pinned, mapped: the packing kernel writes the call's results straight into it
  u32* d_hpack;     // h_pack as the device sees it
  size_t h_pack_cap;
  u32* d_err;
  u32* d_heavy;     // the counter block of kernels.cuh (HEAVY_*): queue lengths, tickets, the group totals of the
                    // two-pass sort (two arrays, used alternately); zeroed once, then kept clear by heavy_clear
  int gtotal_parity;  // which of the two the next two-pass sort uses
  u32* h_err;       // pinned
  u32* d_flags;     // MSM_FLAG_* of the call in flight (kernels.cuh): the last two words of d_heavy's block; the
                    // packing kernel appends them to the packed results, so they reach the host with h_pack
  int fr_defer;     // bn256_ctx_fr_defer_sync: Fr calls without a host-visible result do not wait for the stream
  int force_split;  // enqueue the split-sort kernels (set after a call met an oversized group, cleared after one that did not)
  int used_split;   // whether the pipeline enqueued last had them
  // bn256_msm_submit / bn256_msm_wait: each lane is a whole child context (own stream, workspaces,
  // pinned result buffer) on the same device, created on first use
  bn256_ctx* lane[BN256_ASYNC_LANES];
  struct lane_worker* worker[BN256_ASYNC_LANES];  // the lane's host thread: enqueue, wait and host tail of a submitted call
  struct lane_ticket* pending[BN256_ASYNC_LANES];
  hipEvent_t lane_ev[BN256_ASYNC_LANES];  // orders a submitted call behind this context's deferred Fr kernels (lane_start)
  // calls submitted to this context's lanes and not finished yet; a lane's context points at its parent's counter
  // (`peers`), so a call can tell whether it has the chip to itself (k_accumulate's wave priorities, msm enqueue)
  std::atomic<int> lanes_inflight{0};
  std::atomic<int>* peers = nullptr;
  // bn256_ctx_create_multi: this context is only the parent of nshard ordinary contexts, one per
  // entry of the device list (it has no stream or workspace of its own)
  int nshard;
  bn256_ctx* shard[BN256_MAX_DEVICES];
};

// entry points that a multi-device context serves on its first device (Fr vectors, synthetic scalars,
// the self test) and those it does not serve at all
#define PRIMARY_SHARD(ctx)                                   \
  do {                                                       \
    if ((ctx) && (ctx)->nshard) (ctx) = (ctx)->shard[0];     \
  } while (0)
#define SINGLE_DEVICE_ONLY(ctx)                              \
  do {                                                       \
    if ((ctx) && (ctx)->nshard) return BN256_ENOTSUP;        \
  } while (0)
// multi-device paths (end of this file)
static int multi_bases_make(bn256_ctx* ctx, const uint8_t* xy_be, bool synth, uint64_t seed, size_t first, size_t n,
                            int flags, bn256_bases** out);
static void multi_bases_free(bn256_ctx* ctx, bn256_bases* b);
static int multi_bases_download(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, uint8_t* xy_be);
static int multi_msm(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const uint8_t* scalars, bool on_device,
                     uint8_t out_xy_be[64]);
static int multi_scalar_mul_batch(bn256_ctx* ctx, const bn256_bases* in, size_t off, size_t n, const uint8_t* scalars_be,
                                  int broadcast, bn256_bases** out);
static int multi_add_batch(bn256_ctx* ctx, const bn256_bases* a, size_t aoff, const bn256_bases* b, size_t boff, size_t n,
                           bn256_bases** out);
static int multi_fold2(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                       size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                       bn256_bases** out);

static int ws_reserve(bn256_ctx* ctx, int which, size_t bytes) {
  if (ctx->cap[which] >= bytes) return BN256_OK;
  if (ctx->fixed_ws) {  // no hipMalloc / hipFree on this path: the next 256-byte aligned piece of the array
    const size_t at = (ctx->fixed_used + 255) & ~(size_t)255;
    if (bytes > BN256_NOALLOC_SCRATCH_BYTES || at > BN256_NOALLOC_SCRATCH_BYTES - bytes) return BN256_ESCRATCH;
    ctx->ws[which] = g_scratch_ptr[ctx->device] + at;
    ctx->cap[which] = bytes;
    ctx->fixed_used = at + bytes;
    return BN256_OK;
  }
  if (ctx->ws[which]) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return BN256_EHIP;
    (void)hipFree(ctx->ws[which]);
    ctx->ws[which] = nullptr;
    ctx->cap[which] = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = pool::dev_malloc(&ctx->ws[which], want);
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    return e == hipErrorOutOfMemory ? BN256_ENOMEM : BN256_EHIP;
  }
  ctx->cap[which] = want;
  return BN256_OK;
}

static inline unsigned nblocks(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" const char* bn256_strerror(int code) {
  switch (code) {
    case BN256_OK: return "ok";
    case BN256_EINVAL: return "invalid argument";
    case BN256_ELENGTH: return "invalid length for scalar mul";
    case BN256_ECOORD: return "bn256: coordinate exceeds modulus";
    case BN256_ENOTONCURVE: return "bn256: malformed point";
    case BN256_EHIP: return g_hip_msg[0] ? g_hip_msg : "HIP runtime error";
    case BN256_ENOMEM: return "out of device memory";
    case BN256_ETOOLARGE: return "problem too large (n * windows >= 2^31)";
    case BN256_EBUSY: return "every asynchronous lane holds a submitted call: wait for one first";
    case BN256_ENOTSUP: return "not supported on a multi-device context / sharded handle (see bn256_ctx_create_multi)";
    case BN256_ESCRATCH: return "bn256_msm_dev_noalloc: the call needs more workspace than the static scratch array holds (BN256_NOALLOC_SCRATCH_BYTES)";
    default: return "unknown error";
  }
}

// Everything bn256_ctx_create made, in reverse; `counted` = pool::context_created() has run.  Also the
// failure path of bn256_ctx_create, so that a half-built context is not leaked.
static void lane_shutdown(bn256_ctx* ctx, int l);
static void ctx_teardown(bn256_ctx* ctx, bool counted) {
  if (ctx->nshard) {  // a multi-device parent owns nothing but its shards
    for (int s = 0; s < ctx->nshard; s++) bn256_ctx_destroy(ctx->shard[s]);
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < BN256_ASYNC_LANES; i++) {
    lane_shutdown(ctx, i);  // waits for a submitted call, ends the lane's thread
    bn256_ctx_destroy(ctx->lane[i]);
    if (ctx->lane_ev[i]) (void)hipEventDestroy(ctx->lane_ev[i]);
  }
  for (int i = 0; i < WS_COUNT; i++)
    if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->d_err) (void)hipFree(ctx->d_err);
  if (ctx->d_heavy) (void)hipFree(ctx->d_heavy);
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
  for (auto& e : ctx->ev_copy)
    if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (counted) pool::context_destroyed();
  delete ctx;
}

static int ctx_init(bn256_ctx* ctx) {
  HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
  for (int i = 0; i < 2; i++) {
    hipError_t e = i == 0 ? pool::dev_malloc((void**)&ctx->d_err, 256)
                          : pool::dev_malloc((void**)&ctx->d_heavy, HEAVY_ALLOC_WORDS * sizeof(u32));
    if (e != hipSuccess) {
      ctx_note_hip(e, __LINE__);
      return e == hipErrorOutOfMemory ? BN256_ENOMEM : BN256_EHIP;
    }
  }
  ctx->d_flags = ctx->d_heavy + HEAVY_BASE;
  // the kernels keep it clear from here on (heavy_clear).  On the context's own stream and waited for: a
  // hipMemset runs on the null stream, which a non-blocking stream does not wait for -- the first pass of a
  // fresh lane could start before it and have its group totals zeroed under it (seen once the lanes had a
  // hardware queue each)
  HIPCHK(hipMemsetAsync(ctx->d_heavy, 0, HEAVY_ALLOC_WORDS * sizeof(u32), ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipHostMalloc(&ctx->h_err, 256));
  ctx->h_pack_cap = 64 * 32 * sizeof(u32) * 32;
  HIPCHK(hipHostMalloc(&ctx->h_pack, ctx->h_pack_cap, hipHostMallocMapped | hipHostMallocCoherent));
  HIPCHK(hipHostGetDevicePointer((void**)&ctx->d_hpack, ctx->h_pack, 0));
  // the sort kernels' dynamic LDS above 48 KiB (per function and device; set here so that no MSM
  // call carries set-up work)
  HIPCHK(hipFuncSetAttribute((const void*)k_part<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_part<20, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_group_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_group_sort_tl, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_tile_sort<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_tile_sort<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  return scratch_resolve(ctx->device);  // this device's copy of the static scratch array
}

extern "C" int bn256_ctx_create(int device, bn256_ctx** out) {
  if (!out) return BN256_EINVAL;
  *out = nullptr;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return BN256_EINVAL;
  HIPCHK(hipSetDevice(device));
  bn256_ctx* ctx = new (std::nothrow) bn256_ctx();
  if (!ctx) return BN256_ENOMEM;
  memset(ctx, 0, sizeof *ctx);
  ctx->device = device;
  if (const char* e = getenv("BN256_SORT1")) ctx->sort1 = atoi(e);

  if (const char* e = getenv("BN256_MSM_CHUNK_LOG2")) {
    int lg = atoi(e);
    if (lg >= 10 && lg <= 30) ctx->msm_chunk = (size_t)1 << lg;
  }
  int rc = ctx_init(ctx);
  if (rc) {
    ctx_teardown(ctx, false);
    return rc;
  }
  pool::context_created();
  *out = ctx;
  return BN256_OK;
}

extern "C" void bn256_ctx_destroy(bn256_ctx* ctx) {
  if (!ctx) return;
  ctx_teardown(ctx, true);
}

extern "C" int bn256_ctx_create_multi(const int* devices, int ndev, bn256_ctx** out) {
  if (!out) return BN256_EINVAL;
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > BN256_MAX_DEVICES) return BN256_EINVAL;
  bn256_ctx* ctx = new (std::nothrow) bn256_ctx();
  if (!ctx) return BN256_ENOMEM;
  memset(ctx, 0, sizeof *ctx);
  ctx->device = devices[0];
  for (int s = 0; s < ndev; s++) {
    int rc = bn256_ctx_create(devices[s], &ctx->shard[s]);
    if (rc) {
      ctx->nshard = s;  // destroy what exists
      if (s == 0) delete ctx; else ctx_teardown(ctx, false);
      return rc;
    }
  }
  ctx->nshard = ndev;
  *out = ctx;
  return BN256_OK;
}

extern "C" void* bn256_ctx_stream(bn256_ctx* ctx) {
  PRIMARY_SHARD(ctx);
  return ctx ? (void*)ctx->stream : nullptr;
}

extern "C" int bn256_ctx_set_tuning(bn256_ctx* ctx, int window_bits, int chunk) {
  if (!ctx) return BN256_EINVAL;
  if (window_bits != 0 && (window_bits < 2 || window_bits > 22)) return BN256_EINVAL;
  if (chunk < 0 || chunk > 65536) return BN256_EINVAL;
  for (int s = 0; s < ctx->nshard; s++) {
    ctx->shard[s]->tune_c = window_bits;
    ctx->shard[s]->tune_k = chunk;
  }
  ctx->tune_c = window_bits;
  ctx->tune_k = chunk;
  return BN256_OK;
}

extern "C" int bn256_ctx_set_phase_timing(bn256_ctx* ctx, int on) {
  if (!ctx) return BN256_EINVAL;
  for (int s = 0; s < ctx->nshard; s++) ctx->shard[s]->phase_timing = on != 0;
  ctx->phase_timing = on != 0;
  return BN256_OK;
}

extern "C" int bn256_ctx_fr_defer_sync(bn256_ctx* ctx, int on) {
  if (!ctx) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  ctx->fr_defer = on != 0;
  return BN256_OK;
}

extern "C" int bn256_last_timing(bn256_ctx* ctx, float* ms, int cap) {
  if (!ctx || !ms) return 0;
  int k = std::min(cap, 7);
  if (ctx->nshard) {  // the shards run side by side: the slowest one
    for (int i = 0; i < k; i++) {
      ms[i] = 0.f;
      for (int sh = 0; sh < ctx->nshard; sh++) ms[i] = std::max(ms[i], ctx->shard[sh]->last_ms[i]);
    }
    return k;
  }
  for (int i = 0; i < k; i++) ms[i] = ctx->last_ms[i];
  return k;
}

static int ilog2_floor(size_t n) {
  int r = 0;
  while (n >>= 1) r++;
  return r;
}
// window width: fixed-base tables have a single bucket set, so wide windows are cheap
static int pick_c(const bn256_ctx* ctx, size_t n, bool precomp) {
  if (ctx->tune_c) return ctx->tune_c;
  // Measured on MI355X (tools/perf_msm.py sweeps, DESIGN.md "window size"): c = 16 keeps the
  // whole 2^15-bucket histogram in LDS and leaves a full-width top window (255 = 15*16 + 15);
  // the smaller choices are the ones whose top window is not a few-bucket hot spot.
  int lg = ilog2_floor(n ? n : 1);
  // from 2^20 points a table of 13 levels of 20-bit windows (2^19 buckets, DESIGN.md "wide windows"):
  // 19 % fewer additions; its longer bucket reduction is paid once per call, so the gain grows with
  // the number of passes (2^20: +2 %, 2^21: +5..10 %, 2^24: +13 %)
  if (precomp && lg >= 20) return 20;
  if (precomp) return lg >= 13 ? 16 : (lg >= 9 ? 13 : 10);
  return lg >= 17 ? 16 : (lg >= 15 ? 10 : 8);
}
static inline int windows_for(int c) { return 255 / c + 1; }

static bn256_bases* bases_alloc(size_t n, int levels, int c, int* rc) {
  bn256_bases* b = new (std::nothrow) bn256_bases();
  if (!b) {
    *rc = BN256_ENOMEM;
    return nullptr;
  }
  b->n = n;
  b->levels = levels;
  b->c = c;
  b->top_scale = 1;
  b->no_inf = false;
  b->d = nullptr;
  size_t bytes = std::max<size_t>(1, n) * (size_t)levels * sizeof(aff_mem);
  hipError_t e = pool::alloc((void**)&b->d, bytes, &b->cap, &b->dev);
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    delete b;
    *rc = e == hipErrorOutOfMemory ? BN256_ENOMEM : BN256_EHIP;
    return nullptr;
  }
  *rc = BN256_OK;
  return b;
}

// Element-wise kernels leave XYZZ results in a scratch array; xyzz_scratch() reserves it (plus the
// prefix-product array) and normalize_into() converts to affine with one inversion per chunk.
static int xyzz_scratch(bn256_ctx* ctx, size_t n, xyzz_mem** out) {
  int rc;
  if ((rc = ws_reserve(ctx, WS_BUCKETS, n * sizeof(xyzz_mem))) || (rc = ws_reserve(ctx, WS_PARTIALS, n * 9 * sizeof(i32))))
    return rc;
  *out = (xyzz_mem*)ctx->ws[WS_BUCKETS];
  return BN256_OK;
}
static void normalize_into(bn256_ctx* ctx, const xyzz_mem* in, aff_mem* out, size_t n) {
  u32 B = (u32)std::min<size_t>(32, std::max<size_t>(1, n / 65536));
  size_t threads = (n + B - 1) / B;
  hipLaunchKernelGGL(k_batch_to_affine, dim3(nblocks(threads, 256)), dim3(256), 0, ctx->stream, in, out,
                     (i32*)ctx->ws[WS_PARTIALS], (u32)n, B);
}

// xs[i] = s_(i mod bs or i) * in[i or 0] for i < n, per-point scalars already on the device: one quad per point and
// GLV half up to SCALAR_MUL_QUAD_MAX points (xs then needs 2n entries; k_glv_combine joins the halves), above it
// one lane per point with a per-lane window table in HBM (k_scalar_mul_win), SMW_MAX_POINTS points per launch.
// BN256_SCALAR_MUL_V1=1 (environment): round 2's joint binary chain instead (A/B runs).
#define SCALAR_MUL_QUAD_MAX ((size_t)1 << 14)
static int launch_scalar_mul(bn256_ctx* ctx, const aff_mem* in, const uint8_t* d_scalars, xyzz_mem* xs, size_t n,
                             u32 broadcast_scalar, u32 broadcast_point) {
  static const bool v1 = getenv("BN256_SCALAR_MUL_V1") && atoi(getenv("BN256_SCALAR_MUL_V1")) != 0;
  hipStream_t s = ctx->stream;
  if (n <= SCALAR_MUL_QUAD_MAX && (broadcast_scalar == 0 || broadcast_scalar == 1)) {
    hipLaunchKernelGGL(k_scalar_mul_quad, dim3(nblocks(n * 4, 256), 2), dim3(256), 0, s, in, d_scalars, xs, (u32)n,
                       broadcast_scalar, broadcast_point);
    hipLaunchKernelGGL(k_glv_combine, dim3(nblocks(n * 4, 256)), dim3(256), 0, s, xs, (u32)n);
    return BN256_OK;
  }
  if (v1) {
    hipLaunchKernelGGL(k_scalar_mul, dim3(nblocks(n, 256)), dim3(256), 0, s, in, d_scalars, xs, (u32)n, broadcast_scalar,
                       broadcast_point);
    return BN256_OK;
  }
  // 1 KiB of table per point of a launch; when memory is short the launches shrink (multiples of 256 points)
  size_t per = std::min(n, SMW_MAX_POINTS);
  int rc;
  while ((rc = ws_reserve(ctx, WS_SMTABLE, per * SMW_ENTRIES * SMW_ENTRY_WORDS * sizeof(i32))) == BN256_ENOMEM && per > 4096)
    per = std::max<size_t>(4096, (per / 2) & ~(size_t)255);
  if (rc) return rc;
  for (size_t at = 0; at < n; at += per) {
    const size_t m = std::min(per, n - at);
    // scalar index of point at + j: (at + j) mod broadcast_scalar -- `at` is a multiple of 256, hence of any
    // small power-of-two period (TOP_SCALES); other periods only ever come with one launch
    hipLaunchKernelGGL(k_scalar_mul_win, dim3(nblocks(m, SMW_BLOCK)), dim3(SMW_BLOCK), 0, s, broadcast_point ? in : in + at,
                       broadcast_scalar ? d_scalars : d_scalars + at * 32, xs + at, (u32)m, broadcast_scalar, broadcast_point,
                       (i32*)ctx->ws[WS_SMTABLE]);
  }
  return BN256_OK;
}

// With wide windows the top window is much narrower than the others (c = 20: 13 windows, the top one
// holds bits 240..254 of a scalar below n/2, values <= 18395 of 2^19): its 2^20 digits per 2^20 points
// would all land in the first 3.5 % of the buckets and make the sort's first groups three times as
// large as the rest.  The top table level is therefore stored as (1 / s_i) * 2^(c (W-1)) * P_i and the top
// digit multiplied by s_i = s - (i & 7) (TOP_SCALES = 8, kernels.cuh), s = floor(2^(c-1) / 18396) = 28: the
// same sum, with the top window's entries spread over the whole bucket range (18396 * 28 < 2^19).  One s
// for every point would put them on every 28th bucket only, 57 extra entries each: buckets three times the
// average, each across two or three chunk boundaries of the accumulation, so that every wave of
// k_fix_boundary ran a chain of two additions where most lanes have one.  Eight scales hit eight times as
// many buckets with an eighth each.
// s = 1 (nothing changes) for every other window width.
static u32 top_scale_for(int c) {
  if (c != 20) return 1;
  const int W = windows_for(c);
  // (n - 1) / 2 >> (c (W - 1)), + 1 for the carry out of the window below
  static const uint64_t half[4] = {0x0d177a2dabd63930ULL, 0x9746c7097c159c92ULL, 0xd537f65c30c26e10ULL,
                                   0x47da80f1a551c3fcULL};
  const int sh = c * (W - 1);
  uint64_t top = half[sh >> 6] >> (sh & 63);
  if ((sh & 63) && (sh >> 6) < 3) top |= half[(sh >> 6) + 1] << (64 - (sh & 63));
  top += 1;
  return (u32)((1ull << (c - 1)) / top);
}
// 1 / s mod n for a small s, 32 bytes big-endian: (n t + 1) / s with n t = -1 (mod s)
static void small_inverse_mod_n(uint8_t out_be[32], u32 s_) {
  static const uint64_t nw[4] = {0x1a2ef45b57ac7261ULL, 0x2e8d8e12f82b3924ULL, 0xaa6fecb86184dc21ULL,
                                 0x8fb501e34aa387f9ULL};
  const uint64_t sdiv = s_;
  uint64_t nmod = 0;
  for (int i = 3; i >= 0; i--) nmod = (uint64_t)((((unsigned __int128)nmod << 64) | nw[i]) % sdiv);
  uint64_t t = 0;
  while ((nmod * t + 1) % sdiv) t++;
  uint64_t prod[5];
  unsigned __int128 cy = 1;
  for (int i = 0; i < 4; i++) {
    cy += (unsigned __int128)nw[i] * t;
    prod[i] = (uint64_t)cy;
    cy >>= 64;
  }
  prod[4] = (uint64_t)cy;
  uint64_t q[5], rem = 0;
  for (int i = 4; i >= 0; i--) {
    unsigned __int128 cur = ((unsigned __int128)rem << 64) | prod[i];
    q[i] = (uint64_t)(cur / sdiv);
    rem = (uint64_t)(cur % sdiv);
  }
  hostg1::be_store(out_be, q);
}
static int fold_points(bn256_ctx* ctx, const aff_mem* lo, const aff_mem* hi, u32 stride, const uint8_t s_lo_be[32],
                       const uint8_t s_hi_be[32], u32 n_lo, u32 n_hi, xyzz_mem** xs_out);

static int bases_precompute(bn256_ctx* ctx, bn256_bases* b) {
  xyzz_mem* xs;
  int rc = xyzz_scratch(ctx, b->n, &xs);
  if (rc) return rc;
  for (int l = 1; l < b->levels; l++) {
    hipLaunchKernelGGL(k_precompute_level, dim3(nblocks(b->n, 256)), dim3(256), 0, ctx->stream,
                       b->d + (size_t)(l - 1) * b->n, xs, (u32)b->n, (u32)b->c);
    normalize_into(ctx, xs, b->d + (size_t)l * b->n, b->n);
  }
  b->top_scale = top_scale_for(b->c);
  if (b->top_scale > 1) {  // top level of point i *= 1 / (top_scale - i mod TOP_SCALES), in place (top_scale_of, kernels.cuh)
    uint8_t inv_be[TOP_SCALES * 32];
    for (u32 r = 0; r < TOP_SCALES; r++) small_inverse_mod_n(inv_be + 32 * r, b->top_scale - r);
    aff_mem* top = b->d + (size_t)(b->levels - 1) * b->n;
    if ((rc = ws_reserve(ctx, WS_SCALARS, sizeof(inv_be)))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->ws[WS_SCALARS], inv_be, sizeof(inv_be), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // inv_be is on this stack
    if ((rc = launch_scalar_mul(ctx, (const aff_mem*)top, (const uint8_t*)ctx->ws[WS_SCALARS], xs, b->n, TOP_SCALES, 0u)))
      return rc;
    normalize_into(ctx, xs, top, b->n);
  }
  HIPCHK(hipGetLastError());
  return BN256_OK;
}

extern "C" int bn256_bases_upload(bn256_ctx* ctx, const uint8_t* xy_be, size_t n, int flags, bn256_bases** out) {
  if (!ctx || !out || (!xy_be && n)) return BN256_EINVAL;
  *out = nullptr;
  if (n >= (1ull << 31)) return BN256_ETOOLARGE;
  if (ctx->nshard) return multi_bases_make(ctx, xy_be, false, 0, 0, n, flags, out);
  HIPCHK(hipSetDevice(ctx->device));
  bool precomp = (flags & BN256_PRECOMPUTE) != 0 && n > 0;
  int c = precomp ? pick_c(ctx, n, true) : 0;
  int levels = precomp ? windows_for(c) : 1;
  if (precomp && (unsigned long long)n * levels >= (1ull << 31)) return BN256_ETOOLARGE;
  int rc;
  bn256_bases* b = bases_alloc(n, levels, c, &rc);
  if (!b) return rc;
  if (n) {
    rc = ws_reserve(ctx, WS_TMP, n * 64);
    if (rc) { bn256_bases_free(ctx, b); return rc; }
    hipError_t e = hipMemcpyAsync(ctx->ws[WS_TMP], xy_be, n * 64, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_err, 0, 4, ctx->stream);
    if (e != hipSuccess) { ctx_note_hip(e, __LINE__); bn256_bases_free(ctx, b); return BN256_EHIP; }
    hipLaunchKernelGGL(k_points_from_be, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream,
                       (const uint8_t*)ctx->ws[WS_TMP], b->d, (u32)n, ctx->d_err);
    e = hipMemcpyAsync(ctx->h_err, ctx->d_err, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx_note_hip(e, __LINE__); bn256_bases_free(ctx, b); return BN256_EHIP; }
    u32 err = ctx->h_err[0];
    if (err & 3u) {
      bn256_bases_free(ctx, b);
      return (err & 1u) ? BN256_ECOORD : BN256_ENOTONCURVE;
    }
    b->no_inf = !(err & 4u);  // multiples 2^(c w) P of a point of prime order are not infinity either
    if (precomp) {
      rc = bases_precompute(ctx, b);
      if (rc == BN256_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BN256_EHIP;
      if (rc) { bn256_bases_free(ctx, b); return rc; }
    }
  }
  *out = b;
  return BN256_OK;
}

// A submitted call (bn256_msm_submit, bn256_fold2_submit) may still read the handle: wait for the lanes
// that hold one (the call itself, not only its stream: the lane's thread may not have enqueued it yet).
static void lane_join(bn256_ctx* ctx, int l);
static void wait_for_submitted(bn256_ctx* ctx) {
  for (int i = 0; i < BN256_ASYNC_LANES; i++)
    if (ctx->lane[i] && ctx->pending[i]) lane_join(ctx, i);
}

extern "C" void bn256_bases_free(bn256_ctx* ctx, bn256_bases* b) {
  if (!b) return;
  if (b->nshard) return multi_bases_free(ctx, b);
  PRIMARY_SHARD(ctx);
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    wait_for_submitted(ctx);
  }
  pool::release(b->d, b->cap, b->dev);
  delete b;
}

extern "C" size_t bn256_bases_len(const bn256_bases* b) { return b ? b->n : 0; }

extern "C" int bn256_bases_download(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, uint8_t* xy_be) {
  if (!ctx || !b || (!xy_be && n)) return BN256_EINVAL;
  if (off > b->n || n > b->n - off) return BN256_ELENGTH;
  if (!n) return BN256_OK;
  if (b->nshard) return multi_bases_download(ctx, b, off, n, xy_be);
  PRIMARY_SHARD(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ws_reserve(ctx, WS_TMP, n * 64);
  if (rc) return rc;
  hipLaunchKernelGGL(k_points_to_be, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, b->d + off,
                     (uint8_t*)ctx->ws[WS_TMP], (u32)n);
  HIPCHK(hipMemcpyAsync(xy_be, ctx->ws[WS_TMP], n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return BN256_OK;
}

// ---------------------------------------------------------------------------- MSM core
// Fixed-base calls above this many points run as several passes over sub-ranges whose bucket
// arrays are added before the reduction: at 2^22+ points the gathers range over a 4-16 GiB table
// and both the scatter and the accumulation slow down per entry ("large n").
#define MSM_CHUNK ((size_t)1 << 21)

// What msm_finish needs to know about a pipeline that msm_enqueue put on the context's stream
struct msm_pending {
  bool empty;   // n == 0: nothing was enqueued, the result is the identity
  bool timed;   // phase events were recorded (see MSM_MARK)
  int c;
  u32 q, nsets;
};

// ---------------------------------------------------------------------------- lanes
// bn256_msm_submit and its relatives hand the whole call -- enqueueing ~30 launches (~0.1 ms of host
// time), waiting for the stream, the host tail (~0.03-0.1 ms) -- to the lane's own host thread, so the
// caller's thread is free at once: a prover's round has two or three such calls plus host work of its
// own (go/bp/main.go:340-347, go/bppp/docs_test.go:1312-1320), and done one after the other on one
// thread they were ~0.6 ms per WNLA round of pure host latency.  A lane runs one call at a time.
struct lane_ticket {
  bool fold = false;  // a bn256_fold2_submit ticket (finished by bn256_fold_wait)
  int rc = BN256_OK;
  uint8_t xy[64] = {0};
  float ms[7] = {0};
  char msg[sizeof g_hip_msg] = {0};
};
struct lane_worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool busy = false, quit = false;
};
static void lane_worker_main(lane_worker* w, int device) {
  (void)hipSetDevice(device);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->quit || w->job; });
    if (w->quit && !w->job) return;
    std::function<void()> job = std::move(w->job);
    w->job = nullptr;
    lk.unlock();
    job();
    lk.lock();
    w->busy = false;
    w->cv.notify_all();
  }
}
// a free lane (context and thread are created on first use)
static int lane_acquire(bn256_ctx* ctx, int* out) {
  int l = 0, rc;
  while (l < BN256_ASYNC_LANES && ctx->pending[l]) l++;
  if (l == BN256_ASYNC_LANES) return BN256_EBUSY;
  if (!ctx->lane[l] && (rc = bn256_ctx_create(ctx->device, &ctx->lane[l]))) return rc;
  if (!ctx->worker[l]) {
    lane_worker* w = new (std::nothrow) lane_worker();
    if (!w) return BN256_ENOMEM;
    w->th = std::thread(lane_worker_main, w, ctx->device);
    ctx->worker[l] = w;
  }
  ctx->lane[l]->tune_c = ctx->tune_c;
  ctx->lane[l]->tune_k = ctx->tune_k;
  ctx->lane[l]->phase_timing = ctx->phase_timing;
  ctx->lane[l]->peers = &ctx->lanes_inflight;
  *out = l;
  return BN256_OK;
}
// runs fn(lane context, ticket) -> rc on the lane's thread; the ticket keeps rc, message and timings
static int lane_start(bn256_ctx* ctx, int l, bool fold, std::function<int(bn256_ctx*, lane_ticket*)> fn) {
  lane_ticket* t = new (std::nothrow) lane_ticket();
  if (!t) return BN256_ENOMEM;
  t->fold = fold;
  bn256_ctx* lane = ctx->lane[l];
  lane_worker* w = ctx->worker[l];
  if (ctx->fr_defer) {
    // Fr kernels enqueued on this context's stream without a wait (bn256_ctx_fr_defer_sync) may produce what the
    // submitted call reads: the lane's stream waits for them on the device, the host does not.  The lane is free
    // (no ticket pending), so nothing else touches its stream now; the job's launches come after this wait.
    hipError_t e = hipSuccess;
    if (!ctx->lane_ev[l]) e = hipEventCreateWithFlags(&ctx->lane_ev[l], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ctx->lane_ev[l], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(lane->stream, ctx->lane_ev[l], 0);
    if (e != hipSuccess) {
      ctx_note_hip(e, __LINE__);
      delete t;
      return BN256_EHIP;
    }
  }
  ctx->lanes_inflight.fetch_add(1, std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->busy = true;
    w->job = [ctx, lane, t, fn] {
      g_hip_msg[0] = 0;
      t->rc = fn(lane, t);
      if (t->rc) (void)hipStreamSynchronize(lane->stream);  // whatever was enqueued has left the workspaces
      memcpy(t->msg, g_hip_msg, sizeof t->msg);
      memcpy(t->ms, lane->last_ms, sizeof t->ms);
      ctx->lanes_inflight.fetch_sub(1, std::memory_order_relaxed);
    };
  }
  w->cv.notify_all();
  ctx->pending[l] = t;
  return BN256_OK;
}
static void lane_join(bn256_ctx* ctx, int l) {
  lane_worker* w = ctx->worker[l];
  if (!w) return;
  std::unique_lock<std::mutex> lk(w->mu);
  w->cv.wait(lk, [&] { return !w->busy; });
}
static void lane_shutdown(bn256_ctx* ctx, int l) {
  lane_worker* w = ctx->worker[l];
  if (w) {
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return !w->busy; });
      w->quit = true;
    }
    w->cv.notify_all();
    w->th.join();
    delete w;
    ctx->worker[l] = nullptr;
  }
  delete ctx->pending[l];
  ctx->pending[l] = nullptr;
}

// One term of a sum of MSMs: a range of a bases handle (view: off, stride) and its device-resident
// scalars (view: sc_stride).
// segments to gather, launched GATHER_BATCH at a time (k_gather_segment, fr.cuh)
struct gather_list {
  hipStream_t s;
  gather_batch batch;
  int m = 0;
  u32 max_n = 0;
  explicit gather_list(hipStream_t st) : s(st) { memset(&batch, 0, sizeof batch); }
  void add(const gather_desc& d) {
    batch.d[m++] = d;
    max_n = std::max(max_n, d.n);
    if (m == GATHER_BATCH) flush();
  }
  void flush() {
    if (m) hipLaunchKernelGGL(k_gather_segment, dim3(nblocks(max_n, 256), m), dim3(256), 0, s, batch);
    m = 0;
    max_n = 0;
  }
};

struct msm_src {
  const bn256_bases* b;
  size_t off, stride, n;
  const uint8_t* sc;
  size_t sc_stride;
  // bn256_msm: the term's scalars are still in HOST memory here (n x 32 bytes, sc_stride 1) and `sc` is
  // where they go.  Every pass uploads its own slice on the context's copy stream, the next pass's while
  // this pass's kernels run, and a fixed-base term of >= 2^20 points is cut in at least two passes for
  // that: the call's PCIe time hides behind its own compute instead of preceding it.
  const uint8_t* host_sc = nullptr;
};

// Enqueues the device pipeline for the sum of `nsrc` such terms on ctx->stream, up to the copy of the
// packed bucket sums into ctx->h_pack.  Several terms need fixed-base tables of one window width: they
// share the single bucket set, every term contributes its passes (sub-ranges of <= MSM_CHUNK points),
// later passes add into the first pass's bucket sums (k_accumulate's add_prev) and the reduction and host
// tail run ONCE -- the BP++
// commitments <HVec, r||l> + <GVec, n> (go/bppp/docs_test.go:560-564) are one pipeline this way.
// A table-less call has one term (callers gather several into one first).  With hold_stream the host
// never waits for the stream here (a call of several passes then reports the phase times of its last
// pass only).
static int msm_enqueue_srcs_impl(bn256_ctx* ctx, const msm_src* srcs, int nsrc, bool hold_stream, msm_pending* pend);
// The counter block is kept clear by the kernels themselves (heavy_clear, kernels.cuh); a call that fails
// part-way may leave it otherwise, so a failed enqueue drains the stream and zeroes it again.
static int msm_enqueue_srcs(bn256_ctx* ctx, const msm_src* srcs, int nsrc, bool hold_stream, msm_pending* pend) {
  const int rc = msm_enqueue_srcs_impl(ctx, srcs, nsrc, hold_stream, pend);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemsetAsync(ctx->d_heavy, 0, HEAVY_ALLOC_WORDS * sizeof(u32), ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->gtotal_parity = 0;
  }
  return rc;
}
static int msm_enqueue_srcs_impl(bn256_ctx* ctx, const msm_src* srcs, int nsrc, bool hold_stream, msm_pending* pend) {
  using namespace hostg1;
  pend->empty = true;
  for (float& f : ctx->last_ms) f = 0.f;
  const bn256_bases* b0 = nullptr;
  size_t n_single = 0;
  int live = 0;
  for (int i = 0; i < nsrc; i++) {
    if (!srcs[i].n) continue;
    if (!b0) b0 = srcs[i].b;
    n_single = srcs[i].n;
    live++;
  }
  if (!live) return BN256_OK;
  const bool precomp = b0->levels > 1;
  if (!precomp && live > 1) return BN256_EINVAL;
  const int c = precomp ? b0->c : pick_c(ctx, n_single, false);
  const int W = windows_for(c);
  const u32 nb = 1u << (c - 1);
  const u32 nsets = precomp ? 1u : (u32)W;
  const size_t nbt = (size_t)nb * nsets;
  const u32 q = (u32)(c - 1);
  // c = 20 ("wide" fixed-base path): 13 windows instead of 16, one set of 2^19 buckets; its sort keeps
  // tile-local 11-bit indices in the records, which limits a pass to 2^20 points
  const bool wide = precomp && c == 20;
  const size_t chunk_max = wide ? ((size_t)1 << 20) : ((size_t)1 << 30);
  const size_t chunk = std::min(ctx->msm_chunk ? ctx->msm_chunk : MSM_CHUNK, chunk_max);
  // The passes: every term in sub-ranges of about `chunk` points (table-less: the one term, one pass).
  // Passes are balanced -- ceil(n / np) points each, rounded up to whole sort tiles -- and the wide path
  // lets a pass grow to 9/8 of the chunk (its groups then average 30 k of the 32 k records pass B sorts in
  // LDS): 2^20 + 9 points (the BP++ commitments' HVec, go/bppp/docs_test.go:560-564) are one pass, not a
  // full one plus a 9-point one that pays every kernel's fixed cost again.
  // Table-less calls: passes of at most 2^20 points.  A signed-reduced scalar is below 2^254.2, so the top one of
  // the 16 windows holds 14.2 bits: its n digits land in 36 of that window's 64 sort groups, n / 36 each, and
  // beyond 36 x 32768 = 1.18 M points those groups outgrow pass B's LDS -- the call then ran TWICE (oversized flag,
  // retry with the split-sort kernels, which stayed on for the lane's next call as well): the 1.57 M-point X
  // commitment of a 2^20 WNLA's first round paid ~1.3 ms for it.  Two balanced passes into the same 16 bucket
  // sets (add_prev) cost one more sort and fix-up instead.
  const size_t pass_max = wide ? chunk + chunk / 8 : (precomp ? chunk : std::min(chunk, (size_t)1 << 20));
  // the passes of a term as start offsets (cuts[j] .. cuts[j + 1]), whole sort tiles each
  auto plan = [&](const msm_src& t) -> std::vector<size_t> {
    const size_t n_term = t.n;
    const bool split_upload = t.host_sc && precomp && n_term >= ((size_t)1 << 20);  // below, more passes cost what they hide
    std::vector<size_t> cuts{0};
    auto round_tile = [&](size_t v) { return std::max<size_t>(SORT2_TILE, (v + SORT2_TILE - 1) / SORT2_TILE * SORT2_TILE); };
    if (split_upload) {
      // Host scalars: a pass cannot start before its scalars have crossed PCIe (~0.6 ms for the 32 MiB of 2^20
      // points), and pass j + 1's copy runs under pass j's kernels.  Two equal passes left the chip idle for the
      // first half's copy (0.3 ms); growing passes -- 3/16, 5/16, the rest in full passes -- wait for a 0.11 ms
      // copy and hide the others.  The gain is small (pinned 1.69 -> 1.6 ms at 2^20): every extra pass pays its
      // fixed costs and one more addition per bucket (add_prev), DESIGN.md 6.
      static const int fa = getenv("BN256_UPLOAD_P0") ? atoi(getenv("BN256_UPLOAD_P0")) : 3;   // sixteenths (experiments)
      static const int fb = getenv("BN256_UPLOAD_P1") ? atoi(getenv("BN256_UPLOAD_P1")) : 5;
      size_t p0 = std::min(round_tile(fa * (n_term / 16)), pass_max);
      size_t p1 = fb ? std::min(round_tile(fb * (n_term / 16)), pass_max) : 0;
      if (p0 < n_term) cuts.push_back(p0);
      if (p1 && p0 + p1 < n_term) cuts.push_back(p0 + p1);
    }
    size_t at = cuts.back(), rest = n_term - at;
    if (rest > 0) {
      const size_t np = (rest + pass_max - 1) / pass_max;
      const size_t len = std::min(round_tile((rest + np - 1) / np), pass_max);
      for (size_t k = 1; k < np; k++) cuts.push_back(at + k * len);
    }
    cuts.push_back(n_term);
    return cuts;
  };
  size_t npasses = 0, nmax = 0;
  std::vector<std::vector<size_t>> plans(nsrc);
  for (int i = 0; i < nsrc; i++) {
    const msm_src& t = srcs[i];
    if (!t.n) continue;
    if ((t.b->levels > 1) != precomp || (precomp && (t.b->c != c || t.b->levels != W))) return BN256_EINVAL;
    if ((unsigned long long)t.b->n * t.b->levels >= (1ull << 31)) return BN256_ETOOLARGE;
    if (t.host_sc && t.sc_stride != 1) return BN256_EINVAL;
    plans[i] = plan(t);
    npasses += plans[i].size() - 1;
    for (size_t j = 0; j + 1 < plans[i].size(); j++) nmax = std::max(nmax, plans[i][j + 1] - plans[i][j]);
  }
  const size_t nchunks = npasses;
  const unsigned long long entries_max = (unsigned long long)nmax * W;
  if (entries_max >= (1ull << 31)) return BN256_ETOOLARGE;
  const u32 nscan = nblocks(nbt + 1, SCAN_TILE);
  const size_t npack = (size_t)nsets * (q + 1);

  int rc;
  // LDS-staged sort when one set's histogram fits the 160 KB LDS (nb <= 2^15)
  const bool use_lds = (size_t)nb * 4 <= 128 * 1024;
  const size_t lds_bytes = (size_t)nb * 4;
  const u32 want_tiles = std::max<u32>(1, 256 / nsets);
  // two-pass sort (kernels.cuh "two-pass sort"): 2^15 buckets per set, <= 16 windows, <= 1024 groups and
  // <= 1024 tiles per pass (k_group_sort scans either with one thread each), groups
  // that average <= 2^15 records, and a record that fits 32 bits: (window, index, sign, low key
  // bits) with a table; (index, sign, low key bits) without (the group implies the window)
  u32 ibits = 0, wbits = 0;
  while (((size_t)1 << ibits) < nmax) ibits++;
  while ((1u << wbits) < (u32)W) wbits++;
  const u32 LB2 = wide ? 10u : precomp ? (entries_max <= (1ull << 24) ? 6u : 5u) : 9u;  // log2 keys per group
  const u32 NH2 = wide ? 2u : LB2 <= 6 ? 16u : 4u;                                       // histogram copies in pass B
  const u32 cap2 = wide ? SORT20_CAP : SORT2_CAP;  // records of one group that pass B sorts inside LDS
  const u32 G2 = (u32)(nbt >> LB2);
  const u32 key_stride = precomp ? 0u : nb;
  auto tiles2_for = [&](size_t m) -> u32 {
    u32 t = (u32)((m + SORT2_TILE - 1) / SORT2_TILE);
    return (t + SORT2_S - 1) / SORT2_S * SORT2_S;
  };
  const u32 tiles2_max = tiles2_for(nmax);
  // a pass uses it from 2^15 points on with a table and from 2^17 (where 16-bit windows start) without: swept with
  // BN256_TWO_MIN_LOG2 -- 50 000 points over a table 0.323 -> 0.308 ms, 2^17 0.417 -> 0.383, 2^18 - 1 0.554 ->
  // 0.485; table-less 2^17 0.715 -> 0.622, 2^19 - 1 1.179 -> 1.001 (the thresholds used to be 2^18 and 2^19)
  static const int two_min_env = getenv("BN256_TWO_MIN_LOG2") ? atoi(getenv("BN256_TWO_MIN_LOG2")) : 0;
  const size_t two_min = (size_t)1 << (two_min_env ? two_min_env : precomp ? 15 : 17);
  const bool two_pass =
      wide ? (!ctx->sort1 && G2 == 512 && tiles2_max <= 1024 && nmax >= two_min && nmax <= ((size_t)1 << 20) + ((size_t)1 << 17) &&
              entries_max <= (unsigned long long)G2 * 32768)
           : (c == 16 && nb == (1u << 15) && !ctx->sort1 && W <= 16 && G2 >= 1 && G2 <= 1024 && tiles2_max <= 1024 &&
              nmax >= two_min && entries_max <= (unsigned long long)G2 * 32768 &&
              (precomp ? wbits + ibits + 1 + LB2 <= 32 : ibits + 1 + LB2 <= 32));
  const size_t lds_part = ((size_t)2 * G2 + std::max<size_t>((size_t)16 * G2, SORT2_TCAP)) * 4;
  const size_t lds_gsort = wide ? ((size_t)(NH2 + 1) * (1u << LB2) + 3072 + SORT20_CAP) * 4
                                : ((size_t)(NH2 + 1) * (1u << LB2) + 2048 + SORT2_CAP) * 4;
  // WS_RANK: the records of pass A (one SORT2_TCAP block per tile), or the per-tile histograms of the
  // single-pass LDS sort (also for a short last pass of a two-pass call), or the ranks of the
  // global-atomic sort -- whichever this call's passes may need
  size_t rank_bytes = use_lds ? (size_t)nsets * want_tiles * nb * 4 : (size_t)entries_max * 4;
  if (two_pass) {
    rank_bytes = std::max(rank_bytes, (size_t)tiles2_max * SORT2_TCAP * 4);
    if ((rc = ws_reserve(ctx, WS_GBASE, (size_t)tiles2_max * (G2 + 1) * 4))) return rc;  // loc[tile][0..G]
    // only the split-sort kernels use it (skewed scalars; see `split` below)
    if (ctx->force_split && (rc = ws_reserve(ctx, WS_SUBHIST, ((size_t)nbt * SORT2_S + 1) * 4))) return rc;
  }
  if ((rc = ws_reserve(ctx, WS_RANK, rank_bytes))) return rc;
  // chunk length K of the accumulation kernel and its thread count for a pass of m points
  auto pick_k = [&](size_t m) -> u32 {
    unsigned long long e = (unsigned long long)m * W;
    if (ctx->tune_k) return (u32)ctx->tune_k;
    static const int env_k = getenv("BN256_TUNE_K") ? atoi(getenv("BN256_TUNE_K")) : 0;  // experiments
    if (env_k > 0 && e > (1ull << 22)) return (u32)env_k;
    // >= 2^18 threads when there is work for them (rounded up: 2^18 threads are one round of waves, a
    // 2^18 + 1st thread would run alone after them).  Up to 2^19 entries the whole call is latency: a thread's K
    // mixed additions are a serial chain of ~7 us each (a lone wave), and a bucket of b entries then needs
    // b / K quad additions of ~5 us in the fix-up, so K ~ 0.85 sqrt(b), as a power of two in [2, 16]
    // (tools/ksweep_small.py: 64 points 0.27 -> 0.22 ms, 0.16 -> 0.14 with a table; 2^14 points without a
    // table 0.43 -> 0.38).  Above that more partial sums cost more than they save: 16.
    unsigned long long kmin = 16;
    if (e <= (1ull << 19)) {
      const double per_bucket = (double)e / (double)nbt;
      kmin = 2;
      while (kmin < 16 && (double)kmin < 0.85 * std::sqrt(per_bucket)) kmin <<= 1;
    }
    return (u32)std::min<unsigned long long>(4096, std::max<unsigned long long>(kmin, (e + 262143) / 262144));
  };
  auto threads_for = [&](size_t m) -> size_t {
    unsigned long long e = (unsigned long long)m * W;
    u32 K = pick_k(m);
    return (size_t)((e + K - 1) / K);
  };
  size_t Tmax = threads_for(nmax);
  for (int i = 0; i < nsrc; i++)  // a shorter pass has a shorter chunk length and may have more threads
    for (size_t j = 0; j + 1 < plans[i].size(); j++) Tmax = std::max(Tmax, threads_for(plans[i][j + 1] - plans[i][j]));
  if ((rc = ws_reserve(ctx, WS_SVAL, entries_max * 4))) return rc;
  if ((rc = ws_reserve(ctx, WS_SKEY, nbt * 4))) return rc;  // heavy-bucket queue
  if ((rc = ws_reserve(ctx, WS_HEAVYTMP, (size_t)HEAVY_SLOTS * HEAVY_PARTS * sizeof(xyzz_mem)))) return rc;
  const size_t ncount = nbt + 1;
  const size_t nscan_max = nscan;
  if ((rc = ws_reserve(ctx, WS_COUNTS, ncount * 4))) return rc;
  if ((rc = ws_reserve(ctx, WS_OFFSETS, (nbt + 2) * 4))) return rc;
  if ((rc = ws_reserve(ctx, WS_BLOCKSUM, (size_t)(nscan_max + 1) * 4))) return rc;
  if ((rc = ws_reserve(ctx, WS_BUCKETS, nbt * sizeof(xyzz_mem)))) return rc;
  if ((rc = ws_reserve(ctx, WS_PARTIALS, 2 * Tmax * sizeof(xyzz_mem)))) return rc;
  if ((rc = ws_reserve(ctx, WS_LASTB, Tmax * 4))) return rc;
  if (two_pass && wide && (rc = ws_reserve(ctx, WS_CHUNKB, (Tmax + 8) * 4))) return rc;  // k_group_sort_tl's hints for k_accumulate
  if ((npack * 32 + 2) * 4 > ctx->h_pack_cap) return BN256_EINVAL;

  u32* counts = (u32*)ctx->ws[WS_COUNTS];
  u32* offsets = (u32*)ctx->ws[WS_OFFSETS];
  u32* rank = (u32*)ctx->ws[WS_RANK];
  u32* hist = rank;
  u32* sval = (u32*)ctx->ws[WS_SVAL];
  u32* blocksum = (u32*)ctx->ws[WS_BLOCKSUM];
  xyzz_mem* buckets = (xyzz_mem*)ctx->ws[WS_BUCKETS];
  xyzz_mem* sum_buckets = buckets;  // later passes add into the same array (k_accumulate's add_prev)
  xyzz_mem* partials = (xyzz_mem*)ctx->ws[WS_PARTIALS];
  u32* heavy_count = ctx->d_heavy;
  u32* heavy_list = (u32*)ctx->ws[WS_SKEY];
  hipStream_t s = ctx->stream;
  // Phase events (bn256_last_timing) are recorded only on request (bn256_ctx_set_phase_timing, or
  // BN256_PHASE_TIMING=1 in the environment): an event between two kernels costs 6-8 us of stream time
  // (rocprofv3 kernel trace, 64 points: kernels that follow one another directly start with no gap) --
  // seven of them are a third of a small call and 1-2.5 % of a 2^20-point one (tools/event_cost.py) -- and a
  // call of several passes has to wait for every pass to read them.
  static const int env_timing = getenv("BN256_PHASE_TIMING") ? atoi(getenv("BN256_PHASE_TIMING")) : -1;
  const bool timed = env_timing >= 0 ? env_timing > 0 : ctx->phase_timing != 0;
  pend->timed = timed;
#define MSM_MARK(i) do { if (timed) HIPCHK(hipEventRecord(ctx->ev[i], s)); } while (0)
  // The split-sort kernels (groups larger than pass B's LDS capacity: skewed scalars) are three
  // launches that do nothing for ordinary inputs.  They are enqueued only while ctx->force_split is
  // set; without them an oversized group raises MSM_FLAG_OVERSIZED, the accumulation kernels then
  // stand down and msm_finish asks for the call to be repeated with them (and keeps them on until a
  // call sees no oversized group).  flags[0]: oversized and unhandled; flags[1]: oversized, handled.
  const bool split = ctx->force_split != 0;
  ctx->used_split = split;
  u32* d_flags = ctx->d_flags;  // cleared with the first pass's counters below

  size_t ch = 0;  // running pass number over all terms
  for (int si = 0; si < nsrc; si++) {
   const msm_src& t = srcs[si];
   if (!t.n) continue;
   const bn256_bases* b = t.b;
   const size_t stride = t.stride, sc_stride = t.sc_stride;
   const u32 top_scale = precomp ? b->top_scale : 1u;
   const std::vector<size_t>& cuts = plans[si];
   const size_t np_t = cuts.size() - 1;
   // the upload of pass j's scalars (host_sc terms): issued once, at the latest when the pass starts
   size_t uploaded = 0;
   auto upload_pass = [&](size_t j) -> int {
     if (!t.host_sc || j >= np_t || j < uploaded) return BN256_OK;
     if (!ctx->copy_stream) {  // made at the first call with host scalars (lanes and device-pointer callers never need it)
       HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
       for (auto& e : ctx->ev_copy) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
     }
     const size_t first = cuts[j], len = cuts[j + 1] - first;
     // the call's very first copy has nothing to hide behind: on the compute stream itself, so that the first
     // kernel follows it directly (an event between two streams costs ~50 us before the dependent kernel starts)
     const bool inline_copy = ch == 0 && j == 0;
     HIPCHK(hipMemcpyAsync((uint8_t*)t.sc + first * 32, t.host_sc + first * 32, len * 32, hipMemcpyHostToDevice,
                           inline_copy ? s : ctx->copy_stream));
     if (!inline_copy) HIPCHK(hipEventRecord(ctx->ev_copy[j & 1], ctx->copy_stream));
     uploaded = j + 1;
     return BN256_OK;
   };
   for (size_t cj = 0; cj < np_t; cj++, ch++) {
    const size_t n_ch = cuts[cj + 1] - cuts[cj];
    const size_t off_ch = t.off + cuts[cj] * stride;
    const uint8_t* sc_ch = t.sc + cuts[cj] * sc_stride * 32;
    const unsigned long long entries = (unsigned long long)n_ch * W;
    const u32 K = pick_k(n_ch);
    const size_t T = threads_for(n_ch);
    if (2 * T * sizeof(xyzz_mem) > ctx->cap[WS_PARTIALS] || entries > entries_max) return BN256_EINVAL;
    u32 ntiles = 1, tile_len = (u32)n_ch;
    if (use_lds) {
      ntiles = (u32)std::min<size_t>(want_tiles, (n_ch + 1023) / 1024);
      if (ntiles < 1) ntiles = 1;
      tile_len = (u32)((n_ch + ntiles - 1) / ntiles);
      ntiles = (u32)((n_ch + tile_len - 1) / tile_len);
    }

    if (t.host_sc) {  // this pass's scalars: uploaded on the copy stream (issued during the previous pass)
      const bool first_of_call = ch == 0 && cj == 0;  // copied on this stream (upload_pass)
      if ((rc = upload_pass(cj))) return rc;
      if (!first_of_call) HIPCHK(hipStreamWaitEvent(s, ctx->ev_copy[cj & 1], 0));
    }
    MSM_MARK(0);
    const u32 first_pass = ch == 0 ? 1u : 0u;  // the pass's first kernel clears the counters (heavy_clear), the flags with it here
    // a pass uses the two-pass sort only with all 256 tiles (>= 2^18 points); the ragged last pass
    // of a chunked call may fall back to the single-pass kernels
    const bool two = two_pass && n_ch >= two_min;
    auto scan = [&](const u32* in, u32* out, size_t m) {
      if (m <= SCAN_SMALL_MAX) {  // one launch instead of three
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, in, out, (u32)m);
        return;
      }
      u32 nblk = nblocks(m, SCAN_TILE);
      hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(SCAN_BLOCK), 0, s, in, out, (u32)m, blocksum);
      hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, s, blocksum, nblk, blocksum + nblk);
      hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(SCAN_BLOCK), 0, s, out, (u32)m, (const u32*)blocksum);
    };
    if (two) {
      u32* gtotal = heavy_count + (ctx->gtotal_parity ? HEAVY_GTOTAL2 : HEAVY_GTOTAL);
      u32* gtotal_next = heavy_count + (ctx->gtotal_parity ? HEAVY_GTOTAL : HEAVY_GTOTAL2);
      ctx->gtotal_parity ^= 1;
      u32* loc = (u32*)ctx->ws[WS_GBASE];
      u32* subhist = (u32*)ctx->ws[WS_SUBHIST];
      const u32 nt2 = tiles2_for(n_ch), TS = nt2 / SORT2_S;
      if (wide)
        hipLaunchKernelGGL((k_part<20, true>), dim3(nt2), dim3(1024), lds_part, s, sc_ch, (u32)n_ch, (u32)W, (u32)sc_stride,
                           G2, LB2, ibits, key_stride, top_scale, (u32)off_ch, (u32)stride, gtotal, loc, rank, heavy_count,
                           gtotal_next, first_pass);
      else
        hipLaunchKernelGGL((k_part<16, false>), dim3(nt2), dim3(1024), lds_part, s, sc_ch, (u32)n_ch, (u32)W,
                           (u32)sc_stride, G2, LB2, ibits, key_stride, top_scale, (u32)off_ch, (u32)stride, gtotal, loc, rank,
                           heavy_count, gtotal_next, first_pass);
      MSM_MARK(1);
      MSM_MARK(2);
      const u32 init_inf = ch == 0 ? 1u : 0u;  // buckets without entries: infinity on the call's first pass
      u32* oversized = d_flags + (split ? 1 : 0);
      if (wide)
        hipLaunchKernelGGL(k_group_sort_tl, dim3(G2), dim3(1024), lds_gsort, s, (const u32*)rank, (const u32*)loc,
                           (const u32*)gtotal, nt2, (u32)((n_ch + SORT2_TILE - 1) / SORT2_TILE), (u32)nbt, LB2, NH2, (u32)off_ch,
                           (u32)stride, (u32)b->n, offsets, sval, buckets, init_inf, oversized, K, (u32*)ctx->ws[WS_CHUNKB]);
      else
        hipLaunchKernelGGL(k_group_sort, dim3(G2), dim3(1024), lds_gsort, s, (const u32*)rank, (const u32*)loc,
                           (const u32*)gtotal, nt2, (u32)nbt, LB2, NH2, ibits, (u32)off_ch, (u32)stride, (u32)b->n, offsets,
                           sval, buckets, init_inf, oversized);
      if (split) {  // groups too large for LDS (skewed scalars): split over SORT2_S workgroups each
        hipLaunchKernelGGL(k_group_count, dim3(G2 * SORT2_S), dim3(256), (size_t)4 * (1u << LB2) * 4, s, (const u32*)rank,
                           (const u32*)loc, (const u32*)gtotal, G2, TS, LB2, cap2, subhist);
        hipLaunchKernelGGL(k_group_subscan, dim3(G2), dim3(1024), 0, s, (const u32*)gtotal, LB2, cap2, subhist, offsets,
                           buckets, init_inf);
        hipLaunchKernelGGL(k_group_place, dim3(G2 * SORT2_S), dim3(256), (size_t)(1u << LB2) * 4, s, (const u32*)rank,
                           (const u32*)loc, (const u32*)gtotal, (const u32*)subhist, G2, TS, LB2, ibits, (u32)off_ch,
                           (u32)stride, (u32)b->n, cap2, (u32)wide, sval);
      }
    } else {
    if (use_lds) {
      hipLaunchKernelGGL(k_tile_sort<false>, dim3(ntiles, nsets), dim3(1024), lds_bytes, s, sc_ch, (u32)n_ch, (u32)c,
                         (u32)W, nb, (u32)precomp, (u32)off_ch, (u32)stride, (u32)b->n, tile_len, (u32)sc_stride, top_scale,
                         hist, (const u32*)nullptr, (u32*)nullptr, heavy_count, first_pass);
      MSM_MARK(1);
      hipLaunchKernelGGL(k_tile_scan, dim3(nblocks(nbt + 1, 256)), dim3(256), 0, s, hist, nsets, ntiles, nb, counts);
    } else {
      // bucket arrays beyond LDS without the two-pass sort: a forced c >= 17, and by default every call of
      // fewer than 2^15 points over a 20-bit-window table (a short sub-range of a big handle): global-atomic
      // counting sort, then the fold over all 2^19 buckets -- ~0.3 ms of fixed cost, see DESIGN.md 4
      HIPCHK(hipMemsetAsync(heavy_count, 0, (first_pass ? HEAVY_WORDS : HEAVY_BASE) * sizeof(u32), s));
      HIPCHK(hipMemsetAsync(counts, 0, (nbt + 1) * 4, s));
      hipLaunchKernelGGL(k_recode<false>, dim3(nblocks(n_ch, 256)), dim3(256), 0, s, sc_ch, (u32)n_ch, (u32)c, (u32)W, nb,
                         (u32)precomp, (u32)off_ch, (u32)stride, (u32)b->n, (u32)sc_stride, top_scale, counts, rank,
                         (const u32*)nullptr, (u32*)nullptr);
      MSM_MARK(1);
    }
    // offsets[0..nbt] = exclusive scan of counts[0..nbt] (counts[nbt] == 0) -> offsets[nbt] = total
    scan(counts, offsets, nbt + 1);
    MSM_MARK(2);
    if (use_lds) {
      hipLaunchKernelGGL(k_tile_sort<true>, dim3(ntiles, nsets), dim3(1024), lds_bytes, s, sc_ch, (u32)n_ch, (u32)c,
                         (u32)W, nb, (u32)precomp, (u32)off_ch, (u32)stride, (u32)b->n, tile_len, (u32)sc_stride, top_scale,
                         hist, (const u32*)offsets, sval, heavy_count, first_pass);
    } else {
      hipLaunchKernelGGL(k_recode<true>, dim3(nblocks(n_ch, 256)), dim3(256), 0, s, sc_ch, (u32)n_ch, (u32)c, (u32)W, nb,
                         (u32)precomp, (u32)off_ch, (u32)stride, (u32)b->n, (u32)sc_stride, top_scale, counts, rank,
                         (const u32*)offsets, sval);
    }
    }
    MSM_MARK(3);
    // 128-thread workgroups: the same single round of waves, placed and retired in finer units
    // (measured 1.051 -> 1.031 ms at 2^20; 64 threads the same, limiting residency by LDS slower)
    const u32 add_prev = ch > 0 ? 1u : 0u;  // later passes of the call add into the first pass's bucket sums
    // the fix-up by chunk boundaries when there are fewer chunks than buckets (the 2^19-bucket path; its
    // empty buckets were initialised by the two-pass sort), by buckets otherwise
    const bool by_chunks = two && nbt > T;
    u32* lastb = by_chunks ? (u32*)ctx->ws[WS_LASTB] : (u32*)nullptr;
    // wave priorities step down at these iterations of a chunk (k_accumulate), counted from its end: what is
    // left after the last step is what the SIMD's waves finish one after the other, so the steps sit late --
    // 24, 12 and 4 iterations before the end of a 52-entry chunk (swept with BN256_ACC_PRIO="a,b,c" on one box,
    // tools/ab_prio.sh: quarters 0.802-0.804 ms, 24,12,4 0.790-0.792, 12,6,2 0.790-0.794, 6,3,1 0.811-0.815;
    // without priorities 0.850).  Only for a call that has the chip to itself: with other calls of this context
    // in flight their kernels fill the SIMDs that this one's last waves leave, and young accumulation waves served
    // before another call's old ones cost throughput (tools/ab_prio_tp.sh: four calls in flight 885 -> 856 Mpts/s
    // with the steps, one call 1.265 -> 1.21 ms; priorities RISING with progress, so that a call's last waves leave
    // sooner, cost as much: 950 -> 912-946 by where the steps sit) -- then every wave runs at priority 0 from the
    // start, below the sort and the tail kernels (SMALL_PRIO, kernels.cuh).
    u32 prio_steps;
    {
      static const char* env = getenv("BN256_ACC_PRIO");
      static const char* env_always = getenv("BN256_ACC_PRIO_ALWAYS");  // experiments: 1 = also with calls in flight, 0 = never
      const int others = ctx->peers ? ctx->peers->load(std::memory_order_relaxed) - 1
                                    : ctx->lanes_inflight.load(std::memory_order_relaxed);
      const bool alone = env_always ? env_always[0] == '1' : others <= 0;
      unsigned a = (K * 15u) >> 5, bq = (K * 15u) >> 6, cq = K / 13u;
      if (env) (void)sscanf(env, "%u,%u,%u", &a, &bq, &cq);
      if (!alone) a = bq = cq = K;
      auto at = [&](unsigned back) { return (u32)std::min<u32>(255u, K > back ? K - back : 0u); };
      prio_steps = at(a) | (at(bq) << 8) | (at(cq) << 16);

    }
    const u32* chunk_hint = two && wide ? (const u32*)ctx->ws[WS_CHUNKB] : (const u32*)nullptr;
    if (b->no_inf)
      hipLaunchKernelGGL(k_accumulate<true>, dim3(nblocks(T, 128)), dim3(128), 0, s, (const aff_mem*)b->d, (const u32*)sval,
                         (const u32*)offsets, (u32)nbt, K, add_prev, buckets, partials, (const u32*)d_flags, lastb, prio_steps,
                         chunk_hint);
    else
      hipLaunchKernelGGL(k_accumulate<false>, dim3(nblocks(T, 128)), dim3(128), 0, s, (const aff_mem*)b->d, (const u32*)sval,
                         (const u32*)offsets, (u32)nbt, K, add_prev, buckets, partials, (const u32*)d_flags, lastb, prio_steps,
                         chunk_hint);
    MSM_MARK(4);
    if (by_chunks)
      hipLaunchKernelGGL(k_fix_boundary, dim3(nblocks(std::min<size_t>(T, 262144), 256)), dim3(256), 0, s,
                         (const u32*)offsets, (u32)nbt, K, (const u32*)lastb, buckets, (const xyzz_mem*)partials,
                         heavy_count, heavy_list, (const u32*)d_flags);
    else
      hipLaunchKernelGGL(k_bucket_fix, dim3(nblocks(nbt * 4, 64)), dim3(64), 0, s, (const u32*)offsets, (u32)nbt, K,
                         add_prev, buckets, (const xyzz_mem*)partials, heavy_count, heavy_list, (const u32*)d_flags);
    hipLaunchKernelGGL(k_heavy, dim3(256), dim3(256), 0, s, (const u32*)offsets, (u32)nbt, K, buckets,
                       (const xyzz_mem*)partials, heavy_count, (const u32*)heavy_list, (xyzz_mem*)ctx->ws[WS_HEAVYTMP]);
    MSM_MARK(5);
    HIPCHK(hipGetLastError());  // a failed launch is reported with its own pass, not after all of them
    if ((rc = upload_pass(cj + 1))) return rc;  // the next pass's scalars cross PCIe under this pass's kernels
    // per-pass phase times (the last pass is read after the reduction); not with host scalars: the wait
    // would keep the next pass's kernels from being enqueued under this pass
    if (timed && ch + 1 < nchunks && !hold_stream && !t.host_sc) {
      HIPCHK(hipStreamSynchronize(s));
      for (int i = 0; i < 5; i++) {
        float ms;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        ctx->last_ms[i] += ms;
        ctx->last_ms[6] += ms;
      }
    }
   }
  }
  if (nb <= FOLD_SMALL_MAX_NB) {  // every level and the packing in one launch (small calls: launch-bound)
    const u32 fold_threads = std::min<u32>(768u, std::max<u32>(64u, 2u * nb));  // nb / 2 quads at the widest level
    hipLaunchKernelGGL(k_fold_small, dim3(nsets), dim3(fold_threads), 0, s, sum_buckets, nsets, nb, q, ctx->d_hpack,
                       (const u32*)d_flags);
  } else {
  for (u32 l = 0; l < q; l++) {
    size_t threads = (size_t)nsets * (l + 1) * ((nb >> l) >> 1);
    const size_t threads_next = l + 1 < q ? (size_t)nsets * (l + 2) * ((nb >> (l + 1)) >> 1) : 0;
    if (threads_next >= ((size_t)1 << 17)) {  // this level and the next are both throughput work: one lane per four inputs
      const size_t lanes = (size_t)nsets * (l + 1) * ((nb >> l) >> 2);
      hipLaunchKernelGGL(k_fold_lane2, dim3(nblocks(std::min<size_t>(lanes, 131072), 256)), dim3(256), 0, s, sum_buckets,
                         nsets, nb, l);
      l++;
      continue;
    }
    if (threads >= ((size_t)1 << 17)) {  // a level of that many independent additions is throughput work: lane per addition
      // at most one full round of four waves per SIMD (1024 SIMDs x 64 lanes; the kernel is held to 128 VGPRs)
      const size_t launch = std::min<size_t>(threads, 262144);
      hipLaunchKernelGGL(k_fold_lane, dim3(nblocks(launch, 256)), dim3(256), 0, s, sum_buckets, nsets, nb, l);
    } else
      hipLaunchKernelGGL(k_fold, dim3(nblocks(threads * 4, 64)), dim3(64), 0, s, sum_buckets, nsets, nb, l);
  }
  hipLaunchKernelGGL(k_pack_results, dim3(nblocks(npack, 64)), dim3(64), 0, s, (const xyzz_mem*)sum_buckets, nsets, nb, q,
                     ctx->d_hpack, (const u32*)d_flags);
  }
  // results + flags are in the mapped pinned buffer when the stream has drained: no copy kernel behind the packing
  MSM_MARK(6);
  HIPCHK(hipGetLastError());
  pend->empty = false;
  pend->c = c;
  pend->q = q;
  pend->nsets = nsets;
  return BN256_OK;
#undef MSM_MARK
}

// Waits for the pipeline and runs the host tail; returns the sum as a host point.  MSM_RETRY (internal,
// never leaves the library): the caller enqueues the same call again.
#define MSM_RETRY 1
static int msm_finish(bn256_ctx* ctx, const msm_pending* pend, hostg1::Pt* result) {
  using namespace hostg1;
  *result = infinity();
  if (pend->empty) return BN256_OK;
  const int c = pend->c;
  const u32 q = pend->q, nsets = pend->nsets;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const u32* h_flags = ctx->h_pack + (size_t)nsets * (q + 1) * 32;  // appended by the packing kernel
  if (h_flags[0] & MSM_FLAG_OVERSIZED) {  // see msm_enqueue_srcs: repeat with the split-sort kernels
    ctx->force_split = 1;
    return MSM_RETRY;
  }
  if (ctx->used_split && !h_flags[1]) ctx->force_split = 0;  // no oversized group this time
  for (int i = 0; i < 6 && pend->timed; i++) {
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
    ctx->last_ms[i] += ms;
    ctx->last_ms[6] += ms;
  }

  // Host tail: sum_w 2^(c w) (T_w + sum_k 2^k G_{w,k}) as ONE Horner chain over bit positions
  // j = c w + k: (nsets-1) c + q - 1 doublings in all instead of a chain per set plus a chain over sets.
  Pt total = infinity();
  for (int j = ((int)nsets - 1) * c + (int)q - 1; j >= 0; j--) {
    total = dbl(total);
    const int w = j / c, k = j % c;
    const u32* base = ctx->h_pack + (size_t)w * (q + 1) * 32;
    if (k < (int)q) total = padd(total, from_device_words(base + (size_t)(k + 1) * 32));
    if (k == 0) total = padd(total, from_device_words(base));
  }
  *result = total;
  return BN256_OK;
}

static int msm_enqueue(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t stride, size_t n,
                       const uint8_t* d_scalars, size_t sc_stride, bool hold_stream, msm_pending* pend) {
  const msm_src one = {b, off, stride, n, d_scalars, sc_stride};
  return msm_enqueue_srcs(ctx, &one, 1, hold_stream, pend);
}

static int msm_core(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t stride, size_t n,
                    const uint8_t* d_scalars, hostg1::Pt* result, size_t sc_stride = 1,
                    const uint8_t* host_scalars = nullptr) {
  *result = hostg1::infinity();
  for (;;) {
    msm_pending pend;
    msm_src one = {b, off, stride, n, d_scalars, sc_stride};
    one.host_sc = host_scalars;
    int rc = msm_enqueue_srcs(ctx, &one, 1, false, &pend);
    if (rc) {
      if (host_scalars && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);  // no copy may outlive the caller's buffer
      return rc;
    }
    if ((rc = msm_finish(ctx, &pend, result)) != MSM_RETRY) return rc;
  }
}

static int range_check(const bn256_bases* b, size_t off, size_t n, size_t stride = 1) {
  if (b->nshard) return BN256_ENOTSUP;  // a sharded handle reaches the device only through the multi_* paths
  if (stride == 0) return BN256_EINVAL;
  if (off > b->n) return BN256_ELENGTH;
  if (n == 0) return BN256_OK;
  size_t avail = b->n - off;              // need (n-1)*stride + 1 <= avail
  if (avail == 0 || (n - 1) > (avail - 1) / stride) return BN256_ELENGTH;
  return BN256_OK;
}

extern "C" int bn256_msm_dev(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const void* d_scalars_be,
                             uint8_t out_xy_be[64]) {
  if (!ctx || !b || !out_xy_be || (!d_scalars_be && n)) return BN256_EINVAL;
  if (ctx->nshard || b->nshard) return multi_msm(ctx, b, off, n, (const uint8_t*)d_scalars_be, true, out_xy_be);
  int rc = range_check(b, off, n);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  hostg1::Pt r;
  rc = msm_core(ctx, b, off, 1, n, (const uint8_t*)d_scalars_be, &r);
  if (rc) return rc;
  hostg1::marshal(out_xy_be, r);
  return BN256_OK;
}

// bn256_msm_dev without hipMalloc / hipFree: the same pipeline (msm_core) with every workspace carved
// from this device's static scratch array instead of the context's hipMalloc blocks.
extern "C" int bn256_msm_dev_noalloc(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n,
                                     const void* d_scalars_be, uint8_t out_xy_be[64]) {
  if (!ctx || !b || !out_xy_be || (!d_scalars_be && n)) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  int rc = range_check(b, off, n);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->device < 0 || ctx->device >= 16 || !g_scratch_ptr[ctx->device]) return BN256_EINVAL;
  // swap the workspace table for an empty one that ws_reserve fills from the array; whatever the
  // outcome, the context's own blocks come back untouched
  struct swap_ws {
    bn256_ctx* c;
    void* ws[WS_COUNT];
    size_t cap[WS_COUNT];
    explicit swap_ws(bn256_ctx* c_) : c(c_) {
      memcpy(ws, c->ws, sizeof ws);
      memcpy(cap, c->cap, sizeof cap);
      memset(c->ws, 0, sizeof c->ws);
      memset(c->cap, 0, sizeof c->cap);
      c->fixed_ws = true;
      c->fixed_used = 0;
    }
    ~swap_ws() {
      memcpy(c->ws, ws, sizeof ws);
      memcpy(c->cap, cap, sizeof cap);
      c->fixed_ws = false;
    }
  } guard(ctx);
  hostg1::Pt r;
  rc = msm_core(ctx, b, off, 1, n, (const uint8_t*)d_scalars_be, &r);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);  // nothing of a failed call may still run on the shared array
    return rc;
  }
  hostg1::marshal(out_xy_be, r);
  return BN256_OK;
}

extern "C" int bn256_msm_submit(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const void* d_scalars_be,
                                int* ticket) {
  if (!ctx || !b || !ticket || (!d_scalars_be && n)) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  *ticket = -1;
  int rc = range_check(b, off, n);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  int l;
  if ((rc = lane_acquire(ctx, &l))) return rc;
  const uint8_t* sc = (const uint8_t*)d_scalars_be;
  rc = lane_start(ctx, l, false, [b, off, n, sc](bn256_ctx* lane, lane_ticket* t) -> int {
    hostg1::Pt r;
    int rc2 = msm_core(lane, b, off, 1, n, sc, &r);
    if (!rc2) hostg1::marshal(t->xy, r);
    return rc2;
  });
  if (rc) return rc;
  *ticket = l;
  return BN256_OK;
}

extern "C" int bn256_msm_wait(bn256_ctx* ctx, int ticket, uint8_t out_xy_be[64]) {
  if (!ctx || !out_xy_be || ticket < 0 || ticket >= BN256_ASYNC_LANES || !ctx->pending[ticket] || ctx->pending[ticket]->fold)
    return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  lane_join(ctx, ticket);
  lane_ticket* t = ctx->pending[ticket];
  ctx->pending[ticket] = nullptr;
  const int rc = t->rc;
  for (int i = 0; i < 7; i++) ctx->last_ms[i] = t->ms[i];
  if (rc)
    memcpy(g_hip_msg, t->msg, sizeof g_hip_msg);
  else
    memcpy(out_xy_be, t->xy, 64);
  delete t;
  return rc;
}

static int msm_host_scalars(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t stride, size_t n,
                            const uint8_t* scalars_be, hostg1::Pt* r) {
  int rc = range_check(b, off, n, stride);
  if (rc) return rc;
  if (n && (rc = ws_reserve(ctx, WS_SCALARS, n * 32))) return rc;
  // the upload is part of the pipeline (msm_src::host_sc): pass by pass on the copy stream
  return msm_core(ctx, b, off, stride, n, (const uint8_t*)ctx->ws[WS_SCALARS], r, 1, n ? scalars_be : nullptr);
}

extern "C" int bn256_msm(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const uint8_t* scalars_be,
                         uint8_t out_xy_be[64]) {
  if (!ctx || !b || !out_xy_be || (!scalars_be && n)) return BN256_EINVAL;
  if (ctx->nshard || b->nshard) return multi_msm(ctx, b, off, n, scalars_be, false, out_xy_be);
  HIPCHK(hipSetDevice(ctx->device));
  hostg1::Pt r;
  int rc = msm_host_scalars(ctx, b, off, 1, n, scalars_be, &r);
  if (rc) return rc;
  hostg1::marshal(out_xy_be, r);
  return BN256_OK;
}

// bn256_msm_multi's segments as ONE pipeline on ctx's stream when that is possible (*single = true):
// several small or table-less segments are gathered into one contiguous (points, scalars) pair (one
// sort, one bucket reduction, one host tail); a single segment runs over its handle.  Segments
// over large fixed-base tables keep their own pipelines (*single = false, nothing enqueued).
static int multi_enqueue(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* stride,
                         const size_t* n, const uint8_t* const* scalars_be, bool hold_stream, msm_pending* pend,
                         bool* single) {
  pend->empty = true;
  *single = true;
  size_t total_n = 0;
  bool big_table = false;
  int rc;
  for (int i = 0; i < k; i++) {
    if (!b[i] || (!scalars_be[i] && n[i])) return BN256_EINVAL;
    if ((rc = range_check(b[i], off[i], n[i], stride ? stride[i] : 1))) return rc;
    total_n += n[i];
    if (b[i]->levels > 1 && n[i] >= (1u << 15)) big_table = true;
  }
  if (total_n == 0) return BN256_OK;
  if (k == 1) {
    if ((rc = ws_reserve(ctx, WS_SCALARS, n[0] * 32))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->ws[WS_SCALARS], scalars_be[0], n[0] * 32, hipMemcpyHostToDevice, ctx->stream));
    if (hold_stream) HIPCHK(hipStreamSynchronize(ctx->stream));  // the caller's buffer is not ours after return
    return msm_enqueue(ctx, b[0], off[0], stride ? stride[0] : 1, n[0], (const uint8_t*)ctx->ws[WS_SCALARS], 1, hold_stream,
                       pend);
  }
  // every segment over a fixed-base table of one window width: the segments are the terms of one
  // pipeline (shared bucket set, one reduction, one host tail), nothing is gathered
  bool all_tables = true;
  int c0 = 0, live = 0;
  for (int i = 0; i < k; i++) {
    if (!n[i]) continue;
    live++;
    if (b[i]->levels <= 1 || (c0 && b[i]->c != c0)) all_tables = false;
    c0 = b[i]->c;
  }
  if (all_tables && live <= 16 && total_n < (1ull << 31)) {
    if ((rc = ws_reserve(ctx, WS_SCALARS, total_n * 32))) return rc;
    uint8_t* sc = (uint8_t*)ctx->ws[WS_SCALARS];
    msm_src srcs[16];
    int m = 0;
    size_t pos = 0;
    // the scalars are uploaded pass by pass beside the kernels (msm_src::host_sc).  The buffers must stay
    // valid until the pipeline has run: a blocking caller waits in msm_finish, a lane's job owns copies.
    for (int i = 0; i < k; i++) {
      if (!n[i]) continue;
      srcs[m] = msm_src{b[i], off[i], stride ? stride[i] : 1, n[i], sc + pos * 32, 1};
      srcs[m++].host_sc = scalars_be[i];
      pos += n[i];
    }
    rc = msm_enqueue_srcs(ctx, srcs, m, hold_stream, pend);
    if (rc && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    return rc;
  }
  if (big_table || total_n >= (1ull << 31)) {
    *single = false;
    return BN256_OK;
  }
  if ((rc = ws_reserve(ctx, WS_GATHER, total_n * sizeof(aff_mem))) || (rc = ws_reserve(ctx, WS_SCALARS, total_n * 32)))
    return rc;
  aff_mem* pts = (aff_mem*)ctx->ws[WS_GATHER];
  uint8_t* sc = (uint8_t*)ctx->ws[WS_SCALARS];
  size_t pos = 0;
  fr_arg none;
  memset(&none, 0, sizeof none);
  for (int i = 0; i < k; i++) {
    if (!n[i]) continue;
    HIPCHK(hipMemcpyAsync(sc + pos * 32, scalars_be[i], n[i] * 32, hipMemcpyHostToDevice, ctx->stream));
    pos += n[i];
  }
  if (hold_stream) HIPCHK(hipStreamSynchronize(ctx->stream));  // the caller's buffers are not ours after return
  pos = 0;
  {
    gather_list gl(ctx->stream);
    for (int i = 0; i < k; i++) {
      if (!n[i]) continue;
      gl.add(gather_desc{(const uint4*)(b[i]->d + off[i]), (uint4*)(pts + pos), nullptr, nullptr,
                         (u32)(stride ? stride[i] : 1), 1u, 0u, (u32)n[i], none});
      pos += n[i];
    }
    gl.flush();
  }
  bn256_bases tmp;  // read only while enqueueing
  tmp.d = pts;
  tmp.n = total_n;
  tmp.levels = 1;
  tmp.c = 0;
  tmp.top_scale = 1;
  tmp.no_inf = false;
  return msm_enqueue(ctx, &tmp, 0, 1, total_n, sc, 1, hold_stream, pend);
}

extern "C" int bn256_msm_multi_strided(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                                       const size_t* stride, const size_t* n, const uint8_t* const* scalars_be,
                                       uint8_t out_xy_be[64]) {
  if (!ctx || k < 0 || !out_xy_be || (k && (!b || !off || !n || !scalars_be))) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  for (;;) {
    msm_pending pend;
    bool single;
    int rc = multi_enqueue(ctx, k, b, off, stride, n, scalars_be, false, &pend, &single);
    if (rc) return rc;
    if (!single) break;
    hostg1::Pt r;
    if ((rc = msm_finish(ctx, &pend, &r)) == MSM_RETRY) continue;
    if (rc) return rc;
    hostg1::marshal(out_xy_be, r);
    return BN256_OK;
  }
  hostg1::Pt total = hostg1::infinity();
  float ms[7] = {0};
  for (int i = 0; i < k; i++) {
    hostg1::Pt r;
    int rc = msm_host_scalars(ctx, b[i], off[i], stride ? stride[i] : 1, n[i], scalars_be[i], &r);
    if (rc) return rc;
    total = hostg1::padd(total, r);
    for (int j = 0; j < 7; j++) ms[j] += ctx->last_ms[j];
  }
  for (int j = 0; j < 7; j++) ctx->last_ms[j] = ms[j];
  hostg1::marshal(out_xy_be, total);
  return BN256_OK;
}

extern "C" int bn256_msm_multi(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* n,
                               const uint8_t* const* scalars_be, uint8_t out_xy_be[64]) {
  return bn256_msm_multi_strided(ctx, k, b, off, nullptr, n, scalars_be, out_xy_be);
}

extern "C" int bn256_msm_multi_submit(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                                      const size_t* n, const uint8_t* const* scalars_be, int* ticket) {
  if (!ctx || k < 0 || !ticket || (k && (!b || !off || !n || !scalars_be))) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  *ticket = -1;
  HIPCHK(hipSetDevice(ctx->device));
  int l, rc;
  for (int i = 0; i < k; i++) {
    if (!b[i] || (!scalars_be[i] && n[i])) return BN256_EINVAL;
    if ((rc = range_check(b[i], off[i], n[i]))) return rc;
  }
  if ((rc = lane_acquire(ctx, &l))) return rc;
  // the caller's arrays and scalars are not ours after return: the call owns copies
  struct args_t {
    std::vector<const bn256_bases*> b;
    std::vector<size_t> off, n;
    std::vector<std::vector<uint8_t>> sc;
    std::vector<const uint8_t*> ptr;
  };
  auto a = std::make_shared<args_t>();
  for (int i = 0; i < k; i++) {
    a->b.push_back(b[i]);
    a->off.push_back(off[i]);
    a->n.push_back(n[i]);
    a->sc.emplace_back(scalars_be[i], scalars_be[i] + n[i] * 32);
  }
  for (auto& v : a->sc) a->ptr.push_back(v.data());
  rc = lane_start(ctx, l, false, [a, k](bn256_ctx* lane, lane_ticket* t) -> int {
    return bn256_msm_multi_strided(lane, k, a->b.data(), a->off.data(), nullptr, a->n.data(), a->ptr.data(), t->xy);
  });
  if (rc) return rc;
  *ticket = l;
  return BN256_OK;
}

// ---------------------------------------------------------------------------- element-wise
// (SCALAR_MUL_QUAD_MAX, above: up to that many points a per-point scalar multiplication runs one quad per point
// (latency), above it one lane per point (throughput): 2^14 quads are one wave per SIMD)
// generator folds: one quad per output up to here (a single latency chain), one lane per output above
static size_t fold_quad_max() {
  static const size_t v = [] {
    const char* e = getenv("BN256_FOLD_QUAD_MAX");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)24576;  // measured crossover: quad 1.0 ms at 2^14, 1.7 at 2^15; lane 1.34 ms flat to 2^16
  }();
  return v;
}
// k mod n for a 256-bit k (n > 2^255: one conditional subtraction), as five 64-bit words
static void scalar_mod_n(uint64_t k[5], const uint8_t be[32]) {
  static const uint64_t nw[4] = {0x1a2ef45b57ac7261ULL, 0x2e8d8e12f82b3924ULL, 0xaa6fecb86184dc21ULL,
                                 0x8fb501e34aa387f9ULL};
  hostg1::be_load(k, be);
  k[4] = 0;
  bool ge = true;
  for (int i = 3; i >= 0; i--)
    if (k[i] != nw[i]) { ge = k[i] > nw[i]; break; }
  if (ge) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)k[i] - nw[i] - (uint64_t)br;
      k[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
}
// Joint sparse form (Solinas) of (sa*ka, sb*kb) for magnitudes ka, kb (consumed) and signs sa, sb;
// see jsf_arg in kernels.cuh.
static void jsf_from_words(uint64_t k[2][5], bool neg_a, bool neg_b, jsf_arg* out) {
  memset(out, 0, sizeof *out);
  auto nonzero = [](const uint64_t* v) { return (v[0] | v[1] | v[2] | v[3] | v[4]) != 0; };
  const bool neg[2] = {neg_a, neg_b};
  u32 len = 0;
  while (nonzero(k[0]) || nonzero(k[1])) {
    int u[2] = {0, 0};
    for (int s = 0; s < 2; s++) {
      uint64_t lo = k[s][0], other = k[1 - s][0];
      if (lo & 1) {
        u[s] = (lo & 3) == 1 ? 1 : -1;
        if (((lo & 7) == 3 || (lo & 7) == 5) && (other & 3) == 2) u[s] = -u[s];
      }
    }
    for (int s = 0; s < 2; s++) {
      if (u[s] == 1) {
        k[s][0] -= 1;  // k odd: no borrow
      } else if (u[s] == -1) {
        for (int i = 0; i < 5; i++)
          if (++k[s][i] != 0) break;
      }
      for (int i = 0; i < 4; i++) k[s][i] = (k[s][i] >> 1) | (k[s][i + 1] << 63);
      k[s][4] >>= 1;
      if (neg[s]) u[s] = -u[s];
    }
    u32 nib = (u32)(u[0] == 0 ? 0 : (u[0] == 1 ? 1 : 3)) | ((u32)(u[1] == 0 ? 0 : (u[1] == 1 ? 1 : 3)) << 2);
    out->w[len >> 3] |= nib << ((len & 7) * 4);
    len++;
  }
  out->len = len;
}
// Joint sparse form of (a mod n, b mod n)
static void jsf_build(const uint8_t a_be[32], const uint8_t b_be[32], jsf_arg* out) {
  uint64_t k[2][5];
  scalar_mod_n(k[0], a_be);
  scalar_mod_n(k[1], b_be);
  jsf_from_words(k, false, false, out);
}

// GLV decomposition for the latency-bound folds.  BN curves carry the endomorphism
// phi(x, y) = (beta x, y) = lambda (x, y) (beta^3 = 1 in Fp, lambda^2 + lambda + 1 = 0 mod n;
// FE_BETA_LIMBS in fe29.cuh), so k = k1 + k2 lambda with |k1|, |k2| < 2^129 and
// k P = k1 P + phi(k2 P): two half-length chains that run side by side (k_fold_points_quad's
// blockIdx.y).  Babai rounding against the reduced basis (A1, B1), (-B1, B2) of the lattice
// {(x, y): x + y lambda = 0 mod n} (determinant n), with the quotients taken from 2^384 / n
// reciprocals: c1 = floor(k B2 / n), c2 = floor(k B1 / n) up to one unit, which only moves
// k1, k2 by one basis vector:  k1 = k - c1 A1 - c2 B1,  k2 = c2 B2 - c1 B1.
// (lambda = 0x196ac037f07e9f9eecf9176e4f8bfcd513be518b5264ac23d; B1 = 2u + 1.)
struct glv_half {
  uint64_t m[5];  // magnitude
  bool neg;
};
static void glv_split(const uint64_t k[5], glv_half* h1, glv_half* h2) {
  static const uint64_t A1[2] = {0x6808727cb7be300bULL, 0xbfcdfabe288a7c7bULL};
  static const uint64_t B1[1] = {0xb4ed5d35d8b10603ULL};
  static const uint64_t B2[2] = {0xb31b1546df0d2a08ULL, 0xbfcdfabe288a7c7aULL};
  static const uint64_t G1[5] = {0x60ee1bc8bdbf1528ULL, 0xe21d18f28a2a05c0ULL, 0xae13553c7c4a4445ULL,
                                 0x55ae596cffd56e3dULL, 0x0000000000000001ULL};
  static const uint64_t G2[4] = {0x369d86758c26a0b7ULL, 0x287e2fc7653eef7bULL, 0x424dd4af740d1660ULL,
                                 0x0000000000000001ULL};
  auto mul = [](uint64_t* r, int nr, const uint64_t* a, int na, const uint64_t* b, int nb) {
    for (int i = 0; i < nr; i++) r[i] = 0;
    for (int i = 0; i < na; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < nb && i + j < nr; j++) {
        c += (unsigned __int128)a[i] * b[j] + r[i + j];
        r[i + j] = (uint64_t)c;
        c >>= 64;
      }
      for (int t = i + nb; t < nr && c; t++) {
        c += r[t];
        r[t] = (uint64_t)c;
        c >>= 64;
      }
    }
  };
  auto sub5 = [](uint64_t* r, const uint64_t* a) {  // r -= a, five words, two's complement
    unsigned __int128 br = 0;
    for (int i = 0; i < 5; i++) {
      unsigned __int128 d = (unsigned __int128)r[i] - a[i] - (uint64_t)br;
      r[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  };
  auto finish = [](glv_half* h, const uint64_t* v) {  // two's complement -> sign and magnitude
    h->neg = (v[4] >> 63) != 0;
    unsigned __int128 c = h->neg ? 1 : 0;
    for (int i = 0; i < 5; i++) {
      c += h->neg ? ~v[i] : v[i];
      h->m[i] = (uint64_t)c;
      c >>= 64;
    }
  };
  uint64_t t[10], c1[3], c2[3], u[5], v[5];
  mul(t, 10, k, 4, G1, 5);
  c1[0] = t[6]; c1[1] = t[7]; c1[2] = t[8];
  mul(t, 10, k, 4, G2, 4);
  c2[0] = t[6]; c2[1] = t[7]; c2[2] = 0;
  // k1 = k - c1 A1 - c2 B1
  uint64_t k1[5] = {k[0], k[1], k[2], k[3], 0};
  mul(u, 5, c1, 3, A1, 2);
  sub5(k1, u);
  mul(u, 5, c2, 3, B1, 1);
  sub5(k1, u);
  // k2 = c2 B2 - c1 B1
  mul(v, 5, c2, 3, B2, 2);
  mul(u, 5, c1, 3, B1, 1);
  sub5(v, u);
  finish(h1, k1);
  finish(h2, v);
}
// out[0]: joint sparse form of (a1, b1); out[1]: of (a2, b2), where a = a1 + a2 lambda, b = b1 + b2 lambda
static void jsf_build_glv(const uint8_t a_be[32], const uint8_t b_be[32], jsf_arg out[2]) {
  uint64_t ka[5], kb[5];
  scalar_mod_n(ka, a_be);
  scalar_mod_n(kb, b_be);
  glv_half a[2], b[2];
  glv_split(ka, &a[0], &a[1]);
  glv_split(kb, &b[0], &b[1]);
  for (int h = 0; h < 2; h++) {
    uint64_t k[2][5];
    memcpy(k[0], a[h].m, sizeof k[0]);
    memcpy(k[1], b[h].m, sizeof k[1]);
    jsf_from_words(k, a[h].neg, b[h].neg, &out[h]);
  }
}

extern "C" int bn256_glv_split(const uint8_t k_be[32], uint8_t k1_be[32], int* k1_neg, uint8_t k2_be[32], int* k2_neg) {
  if (!k_be || !k1_be || !k1_neg || !k2_be || !k2_neg) return BN256_EINVAL;
  uint64_t k[5];
  scalar_mod_n(k, k_be);
  glv_half h1, h2;
  glv_split(k, &h1, &h2);
  if (h1.m[4] || h2.m[4]) return BN256_EINVAL;  // cannot happen: magnitudes are below 2^129
  hostg1::be_store(k1_be, h1.m);
  hostg1::be_store(k2_be, h2.m);
  *k1_neg = h1.neg;
  *k2_neg = h2.neg;
  return BN256_OK;
}

// xs[i] = s_lo * lo[i*stride] + s_hi * hi[i*stride] (XYZZ, in the context's scratch): up to 2^13
// outputs as two GLV half-chains of one quad per output plus k_glv_combine, up to fold_quad_max()
// one quad per output over the full joint sparse form, above that one lane per output.
static int fold_points(bn256_ctx* ctx, const aff_mem* lo, const aff_mem* hi, u32 stride, const uint8_t s_lo_be[32],
                       const uint8_t s_hi_be[32], u32 n_lo, u32 n_hi, xyzz_mem** xs_out) {
  const size_t n = std::max(n_lo, n_hi);
  // two chains of n quads each: one wave per SIMD up to 2^13 outputs (measured: 2^13 0.96 -> 0.67 ms,
  // 2^14 0.98 -> 1.08 ms)
  const bool glv = n <= ((size_t)1 << 13), quad = n <= fold_quad_max();
  xyzz_mem* xs = nullptr;
  int rc = xyzz_scratch(ctx, glv ? 2 * n : n, &xs);
  if (rc) return rc;
  if (glv) {
    jsf_arg jsf[2];
    jsf_build_glv(s_lo_be, s_hi_be, jsf);
    hipLaunchKernelGGL(k_fold_points_quad, dim3(nblocks(n * 4, 256), 2), dim3(256), 0, ctx->stream, lo, hi, stride, jsf[0],
                       jsf[1], xs, n_lo, n_hi);
    hipLaunchKernelGGL(k_glv_combine, dim3(nblocks(n * 4, 256)), dim3(256), 0, ctx->stream, xs, (u32)n);
  } else if (quad) {  // one full-length chain (grid.y = 1: the second argument is never read)
    jsf_arg jsf;
    jsf_build(s_lo_be, s_hi_be, &jsf);
    hipLaunchKernelGGL(k_fold_points_quad, dim3(nblocks(n * 4, 256), 1), dim3(256), 0, ctx->stream, lo, hi, stride, jsf,
                       jsf, xs, n_lo, n_hi);
  } else {  // both GLV halves on one doubling chain
    static const bool v1 = getenv("BN256_FOLD_XYZZ") && atoi(getenv("BN256_FOLD_XYZZ")) != 0;  // round 2's kernel (A/B runs)
    jsf_arg jsf[2];
    jsf_build_glv(s_lo_be, s_hi_be, jsf);
    if (v1)
      hipLaunchKernelGGL(k_fold_points_xyzz, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, lo, hi, stride, jsf[0], jsf[1],
                         xs, n_lo, n_hi);
    else
      hipLaunchKernelGGL(k_fold_points, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, lo, hi, stride, jsf[0], jsf[1], xs,
                         n_lo, n_hi);
  }
  *xs_out = xs;
  return BN256_OK;
}

// r[i] = s_i * in[i] (broadcast = 0) or s * in[i] (broadcast = 1) for i < n, normalised into the handle r;
// enqueued on ctx's stream and waited for.  `in` is device memory of ctx's device.
static int scalar_mul_into(bn256_ctx* ctx, const aff_mem* in, size_t n, const uint8_t* scalars_be, int broadcast,
                           bn256_bases* r) {
  if (!n) return BN256_OK;
  int rc;
  xyzz_mem* xs = nullptr;
  if (broadcast) {
    // one scalar for every point (vectorPointMulOnScalar): the uniform NAF chain of the fold kernel
    static const uint8_t zero[32] = {0};
    if ((rc = fold_points(ctx, in, in, 1u, scalars_be, zero, (u32)n, 0u, &xs))) return rc;
  } else {
    if ((rc = xyzz_scratch(ctx, n <= SCALAR_MUL_QUAD_MAX ? 2 * n : n, &xs))) return rc;
    size_t sb = n * 32;
    if ((rc = ws_reserve(ctx, WS_SCALARS, sb))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->ws[WS_SCALARS], scalars_be, sb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_scalar_mul(ctx, in, (const uint8_t*)ctx->ws[WS_SCALARS], xs, n, 0u, 0u))) return rc;
  }
  normalize_into(ctx, xs, r->d, n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return BN256_OK;
}

extern "C" int bn256_scalar_mul_batch(bn256_ctx* ctx, const bn256_bases* in, size_t off, size_t n,
                                      const uint8_t* scalars_be, int broadcast, bn256_bases** out) {
  if (!ctx || !in || !out || (!scalars_be && n)) return BN256_EINVAL;
  if (ctx->nshard) return multi_scalar_mul_batch(ctx, in, off, n, scalars_be, broadcast, out);
  *out = nullptr;
  int rc = range_check(in, off, n);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  bn256_bases* r = bases_alloc(n, 1, 0, &rc);
  if (!r) return rc;
  if ((rc = scalar_mul_into(ctx, (const aff_mem*)(in->d + off), n, scalars_be, broadcast, r))) {
    bn256_bases_free(ctx, r);
    return rc;
  }
  *out = r;
  return BN256_OK;
}

// r[i] = a[i] + b[i] for i < n (device memory of ctx's device), normalised into r; waited for
static int add_into(bn256_ctx* ctx, const aff_mem* a, const aff_mem* b, size_t n, bn256_bases* r) {
  if (!n) return BN256_OK;
  xyzz_mem* xs;
  int rc = xyzz_scratch(ctx, n, &xs);
  if (rc) return rc;
  hipLaunchKernelGGL(k_add_points, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, a, b, xs, (u32)n);
  normalize_into(ctx, xs, r->d, n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return BN256_OK;
}

extern "C" int bn256_add_batch(bn256_ctx* ctx, const bn256_bases* a, size_t aoff, const bn256_bases* b, size_t boff,
                               size_t n, bn256_bases** out) {
  if (!ctx || !a || !b || !out) return BN256_EINVAL;
  if (ctx->nshard) return multi_add_batch(ctx, a, aoff, b, boff, n, out);
  *out = nullptr;
  int rc = range_check(a, aoff, n);
  if (rc) return rc;
  if ((rc = range_check(b, boff, n))) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  bn256_bases* r = bases_alloc(n, 1, 0, &rc);
  if (!r) return rc;
  if ((rc = add_into(ctx, (const aff_mem*)(a->d + aoff), (const aff_mem*)(b->d + boff), n, r))) {
    bn256_bases_free(ctx, r);
    return rc;
  }
  *out = r;
  return BN256_OK;
}

extern "C" int bn256_com_batch(bn256_ctx* ctx, const uint8_t g_xy_be[64], const uint8_t h_xy_be[64],
                               const uint8_t* a_be, const uint8_t* b_be, size_t n, bn256_bases** out) {
  if (!ctx || !g_xy_be || !h_xy_be || !out || ((!a_be || !b_be) && n)) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  *out = nullptr;
  if (n >= (1ull << 29)) return BN256_ETOOLARGE;
  HIPCHK(hipSetDevice(ctx->device));
  uint8_t gh[128];
  memcpy(gh, g_xy_be, 64);
  memcpy(gh + 64, h_xy_be, 64);
  bn256_bases* pts = nullptr;
  int rc = bn256_bases_upload(ctx, gh, 2, 0, &pts);  // validates both points
  if (rc) return rc;
  bn256_bases* r = bases_alloc(n, 1, 0, &rc);
  if (!r) { bn256_bases_free(ctx, pts); return rc; }
  if (n) {
    xyzz_mem* xs = nullptr;
    if ((rc = xyzz_scratch(ctx, 2 * n, &xs)) || (rc = ws_reserve(ctx, WS_SCALARS, 2 * n * 32))) {
      bn256_bases_free(ctx, pts);
      bn256_bases_free(ctx, r);
      return rc;
    }
    uint8_t* ds = (uint8_t*)ctx->ws[WS_SCALARS];
    hipError_t e = hipMemcpyAsync(ds, a_be, n * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ds + n * 32, b_be, n * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_com_batch_quad, dim3(nblocks(n * 4, 256), 2), dim3(256), 0, ctx->stream, (const aff_mem*)pts->d,
                         (const uint8_t*)ds, (const uint8_t*)(ds + n * 32), xs, (u32)n);
      hipLaunchKernelGGL(k_glv_combine, dim3(nblocks(n * 4, 256)), dim3(256), 0, ctx->stream, xs, (u32)n);
      normalize_into(ctx, xs, r->d, n);
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) {
      ctx_note_hip(e, __LINE__);
      bn256_bases_free(ctx, pts);
      bn256_bases_free(ctx, r);
      return BN256_EHIP;
    }
  }
  bn256_bases_free(ctx, pts);
  *out = r;
  return BN256_OK;
}

// fold + normalisation into the handle r (max(n_lo, n_hi) points) on `on`'s stream and scratch; the
// caller synchronises
static int fold2_into(bn256_ctx* on, const aff_mem* lo, size_t n_lo, const aff_mem* hi, size_t n_hi, size_t stride,
                      const uint8_t s_lo_be[32], const uint8_t s_hi_be[32], bn256_bases* r) {
  size_t n = std::max(n_lo, n_hi);
  HIPCHK(hipSetDevice(on->device));
  if (n) {
    xyzz_mem* xs;
    int rc = fold_points(on, lo, hi, (u32)stride, s_lo_be, s_hi_be, (u32)n_lo, (u32)n_hi, &xs);
    if (rc) return rc;
    normalize_into(on, xs, r->d, n);
  }
  return BN256_OK;
}
static int fold2_enqueue(bn256_ctx* on, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                         size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                         bn256_bases** out) {
  *out = nullptr;
  int rc;
  if ((rc = range_check(lo, lo_off, n_lo, stride))) return rc;
  if ((rc = range_check(hi, hi_off, n_hi, stride))) return rc;
  HIPCHK(hipSetDevice(on->device));
  bn256_bases* r = bases_alloc(std::max(n_lo, n_hi), 1, 0, &rc);
  if (!r) return rc;
  if ((rc = fold2_into(on, (const aff_mem*)(lo->d + lo_off), n_lo, (const aff_mem*)(hi->d + hi_off), n_hi, stride, s_lo_be,
                       s_hi_be, r))) {
    bn256_bases_free(on, r);
    return rc;
  }
  *out = r;
  return BN256_OK;
}

extern "C" int bn256_fold2(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                           size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32],
                           const uint8_t s_hi_be[32], bn256_bases** out) {
  if (!ctx || !lo || !hi || !out || !s_lo_be || !s_hi_be || stride == 0) return BN256_EINVAL;
  if (ctx->nshard) return multi_fold2(ctx, lo, lo_off, n_lo, hi, hi_off, n_hi, stride, s_lo_be, s_hi_be, out);
  int rc = fold2_enqueue(ctx, lo, lo_off, n_lo, hi, hi_off, n_hi, stride, s_lo_be, s_hi_be, out);
  if (rc) return rc;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    bn256_bases_free(ctx, *out);
    *out = nullptr;
    return BN256_EHIP;
  }
  return BN256_OK;
}

extern "C" int bn256_fold2_submit(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo,
                                  const bn256_bases* hi, size_t hi_off, size_t n_hi, size_t stride,
                                  const uint8_t s_lo_be[32], const uint8_t s_hi_be[32], bn256_bases** out, int* ticket) {
  if (!ctx || !lo || !hi || !out || !ticket || !s_lo_be || !s_hi_be || stride == 0) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  *out = nullptr;
  *ticket = -1;
  HIPCHK(hipSetDevice(ctx->device));
  int l, rc;
  if ((rc = range_check(lo, lo_off, n_lo, stride)) || (rc = range_check(hi, hi_off, n_hi, stride))) return rc;
  if ((rc = lane_acquire(ctx, &l))) return rc;
  // the new handle exists at once (its memory comes from the caller's thread); the lane fills it
  bn256_bases* r = bases_alloc(std::max(n_lo, n_hi), 1, 0, &rc);
  if (!r) return rc;
  struct sc_t {
    uint8_t lo[32], hi[32];
  } sc;
  memcpy(sc.lo, s_lo_be, 32);
  memcpy(sc.hi, s_hi_be, 32);
  rc = lane_start(ctx, l, true, [=](bn256_ctx* lane, lane_ticket*) -> int {
    int rc2 = fold2_into(lane, (const aff_mem*)(lo->d + lo_off), n_lo, (const aff_mem*)(hi->d + hi_off), n_hi, stride, sc.lo,
                         sc.hi, r);
    if (rc2) return rc2;
    hipError_t e = hipStreamSynchronize(lane->stream);
    if (e != hipSuccess) {
      ctx_note_hip(e, __LINE__);
      return BN256_EHIP;
    }
    return BN256_OK;
  });
  if (rc) {
    bn256_bases_free(nullptr, r);
    return rc;
  }
  *out = r;
  *ticket = l;
  return BN256_OK;
}

extern "C" int bn256_ctx_drain(bn256_ctx* ctx) {
  if (!ctx) return BN256_EINVAL;
  if (ctx->nshard) {
    int rc = BN256_OK;
    for (int s = 0; s < ctx->nshard; s++) {
      int r = bn256_ctx_drain(ctx->shard[s]);
      if (r && !rc) rc = r;
    }
    return rc;
  }
  HIPCHK(hipSetDevice(ctx->device));
  hipError_t first = hipStreamSynchronize(ctx->stream);
  for (int l = 0; l < BN256_ASYNC_LANES; l++) {
    if (ctx->lane[l]) {
      lane_join(ctx, l);
      hipError_t e = hipStreamSynchronize(ctx->lane[l]->stream);
      if (e != hipSuccess && first == hipSuccess) first = e;
    }
    delete ctx->pending[l];  // the ticket is void from here on
    ctx->pending[l] = nullptr;
  }
  HIPCHK(first);
  return BN256_OK;
}

extern "C" int bn256_fold_wait(bn256_ctx* ctx, int ticket) {
  if (!ctx || ticket < 0 || ticket >= BN256_ASYNC_LANES || !ctx->pending[ticket] || !ctx->pending[ticket]->fold)
    return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  lane_join(ctx, ticket);
  lane_ticket* t = ctx->pending[ticket];
  ctx->pending[ticket] = nullptr;
  const int rc = t->rc;
  if (rc) memcpy(g_hip_msg, t->msg, sizeof g_hip_msg);
  delete t;
  return rc;
}

extern "C" int bn256_fold(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, const bn256_bases* hi, size_t hi_off,
                          size_t stride, size_t n, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                          bn256_bases** out) {
  return bn256_fold2(ctx, lo, lo_off, n, hi, hi_off, n, stride, s_lo_be, s_hi_be, out);
}

// ---------------------------------------------------------------------------- synthetic
extern "C" int bn256_random_scalars(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, void* d_out_be) {
  if (!ctx || (!d_out_be && n)) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  if (n >= (1ull << 32)) return BN256_ETOOLARGE;
  HIPCHK(hipSetDevice(ctx->device));
  if (n) {
    hipLaunchKernelGGL(k_synth_scalars, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, (uint8_t*)d_out_be,
                       (u64)seed, (u64)first, (u32)n, 0u);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return BN256_OK;
}

extern "C" int bn256_random_bases(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, int flags, bn256_bases** out) {
  if (!ctx || !out) return BN256_EINVAL;
  *out = nullptr;
  if (n >= (1ull << 31)) return BN256_ETOOLARGE;
  if (ctx->nshard) return multi_bases_make(ctx, nullptr, true, seed, first, n, flags, out);
  HIPCHK(hipSetDevice(ctx->device));
  bool precomp = (flags & BN256_PRECOMPUTE) != 0 && n > 0;
  int c = precomp ? pick_c(ctx, n, true) : 0;
  int levels = precomp ? windows_for(c) : 1;
  if (precomp && (unsigned long long)n * levels >= (1ull << 31)) return BN256_ETOOLARGE;
  int rc;
  bn256_bases* b = bases_alloc(n, levels, c, &rc);
  if (!b) return rc;
  if (n) {
    // G0 = (1, p - 2) marshalled, uploaded through the ordinary Unmarshal kernel
    uint8_t g0[64] = {0};
    g0[31] = 1;
    uint64_t y[4] = {hostg1::kP[0] - 2, hostg1::kP[1], hostg1::kP[2], hostg1::kP[3]};
    hostg1::be_store(g0 + 32, y);
    xyzz_mem* xs = nullptr;
    if ((rc = ws_reserve(ctx, WS_TMP, 64 + sizeof(aff_mem))) || (rc = ws_reserve(ctx, WS_SCALARS, n * 32)) ||
        (rc = xyzz_scratch(ctx, n <= SCALAR_MUL_QUAD_MAX ? 2 * n : n, &xs))) {
      bn256_bases_free(ctx, b);
      return rc;
    }
    uint8_t* tmp = (uint8_t*)ctx->ws[WS_TMP];
    aff_mem* d_g0 = (aff_mem*)(tmp + 64);
    hipError_t e = hipMemcpyAsync(tmp, g0, 64, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_err, 0, 4, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_points_from_be, dim3(1), dim3(256), 0, ctx->stream, (const uint8_t*)tmp, d_g0, 1u, ctx->d_err);
      hipLaunchKernelGGL(k_synth_scalars, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream,
                         (uint8_t*)ctx->ws[WS_SCALARS], (u64)seed, (u64)first, (u32)n, 1u);
      rc = launch_scalar_mul(ctx, (const aff_mem*)d_g0, (const uint8_t*)ctx->ws[WS_SCALARS], xs, n, 0u, 1u);
      if (rc) { bn256_bases_free(ctx, b); return rc; }
      normalize_into(ctx, xs, b->d, n);
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) { ctx_note_hip(e, __LINE__); bn256_bases_free(ctx, b); return BN256_EHIP; }
    b->no_inf = true;  // k_i * G0 with 0 < k_i < n
    if (precomp) {
      rc = bases_precompute(ctx, b);
      if (rc == BN256_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BN256_EHIP;
      if (rc) { bn256_bases_free(ctx, b); return rc; }
    }
  }
  *out = b;
  return BN256_OK;
}

// ---------------------------------------------------------------------------- Fr vectors
struct bn256_frvec {
  uint8_t* d;  // n x 32-byte big-endian scalars
  size_t n;
  size_t cap;  // bytes of the block behind d (pool)
  int dev;
  hipEvent_t ready;  // set by bn256_fr_dot_multi_dev under deferred waits: recorded behind the kernels that fill d
};

static bn256_frvec* frvec_alloc(size_t n, int* rc) {
  bn256_frvec* v = new (std::nothrow) bn256_frvec();
  if (!v) {
    *rc = BN256_ENOMEM;
    return nullptr;
  }
  v->d = nullptr;
  v->n = n;
  hipError_t e = pool::alloc((void**)&v->d, std::max<size_t>(n, 1) * 32, &v->cap, &v->dev);
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    delete v;
    *rc = e == hipErrorOutOfMemory ? BN256_ENOMEM : BN256_EHIP;
    return nullptr;
  }
  *rc = BN256_OK;
  return v;
}

extern "C" void bn256_fr_free(bn256_ctx* ctx, bn256_frvec* v) {
  if (!v) return;
  PRIMARY_SHARD(ctx);
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    wait_for_submitted(ctx);
  }
  if (v->ready) (void)hipEventDestroy(v->ready);
  pool::release(v->d, v->cap, v->dev);
  delete v;
}
extern "C" size_t bn256_fr_len(const bn256_frvec* v) { return v ? v->n : 0; }
extern "C" const void* bn256_fr_device_ptr(const bn256_frvec* v) { return v ? v->d : nullptr; }

extern "C" int bn256_fr_upload(bn256_ctx* ctx, const uint8_t* scalars_be, size_t n, bn256_frvec** out) {
  if (!ctx || !out || (!scalars_be && n)) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  *out = nullptr;
  if (n >= (1ull << 31)) return BN256_ETOOLARGE;
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  bn256_frvec* v = frvec_alloc(n, &rc);
  if (!v) return rc;
  if (n) {
    hipError_t e = hipMemcpyAsync(v->d, scalars_be, n * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx_note_hip(e, __LINE__); bn256_fr_free(ctx, v); return BN256_EHIP; }
  }
  *out = v;
  return BN256_OK;
}

extern "C" int bn256_fr_random(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, bn256_frvec** out) {
  if (!ctx || !out) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  *out = nullptr;
  if (n >= (1ull << 31)) return BN256_ETOOLARGE;
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  bn256_frvec* v = frvec_alloc(n, &rc);
  if (!v) return rc;
  rc = bn256_random_scalars(ctx, seed, first, n, v->d);
  if (rc) { bn256_fr_free(ctx, v); return rc; }
  *out = v;
  return BN256_OK;
}

extern "C" int bn256_fr_download(bn256_ctx* ctx, const bn256_frvec* v, size_t off, size_t n, uint8_t* scalars_be) {
  if (!ctx || !v || (!scalars_be && n)) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  if (off > v->n || n > v->n - off) return BN256_ELENGTH;
  HIPCHK(hipSetDevice(ctx->device));
  if (n && v->ready) {
    // results of bn256_fr_dot_multi_dev: wait for THEIR kernels only, not for whatever the context has enqueued
    // since (the next rounds' kernels), and copy outside the context's stream for the same reason
    HIPCHK(hipEventSynchronize(v->ready));
    HIPCHK(hipMemcpy(scalars_be, v->d + off * 32, n * 32, hipMemcpyDeviceToHost));
  } else if (n) {
    HIPCHK(hipMemcpyAsync(scalars_be, v->d + off * 32, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return BN256_OK;
}

// number of elements of the strided view v[off], v[off+stride], ... is at least n
static int fr_range_check(const bn256_frvec* v, size_t off, size_t n, size_t stride) {
  if (stride == 0) return BN256_EINVAL;
  if (off > v->n) return BN256_ELENGTH;
  if (n == 0) return BN256_OK;
  size_t avail = v->n - off;
  if (avail == 0 || (n - 1) > (avail - 1) / stride) return BN256_ELENGTH;
  return BN256_OK;
}
static fr_arg fr_arg_from_be(const uint8_t be[32]) {
  fr_arg a;
  for (int i = 0; i < 8; i++) {
    const uint8_t* p = be + 28 - 4 * i;
    a.w[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
  }
  return a;
}
// reduce an argument scalar below n (2^256 < 2n): the kernels assume canonical multipliers
static fr_arg fr_arg_canon(fr_arg a) {
  static const u32 nw[8] = {0x57ac7261u, 0x1a2ef45bu, 0xf82b3924u, 0x2e8d8e12u,
                            0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
  u32 t[8];
  uint64_t br = 0;
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)a.w[i] - nw[i] - br;
    t[i] = (u32)d;
    br = (d >> 32) & 1;
  }
  if (!br) memcpy(a.w, t, sizeof t);
  return a;
}

extern "C" int bn256_fr_fold_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off,
                                   const size_t* n_a, const bn256_frvec* const* b, const size_t* b_off, const size_t* n_b,
                                   const size_t* stride, const uint8_t* s_a_be, const uint8_t* s_b_be, bn256_frvec** out) {
  if (!ctx || k < 0 || (k && (!a || !a_off || !n_a || !b || !b_off || !n_b || !stride || !s_a_be || !s_b_be || !out)))
    return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  int rc;
  for (int i = 0; i < k; i++) {
    out[i] = nullptr;
    if (!a[i] || !b[i]) return BN256_EINVAL;
    if ((rc = fr_range_check(a[i], a_off[i], n_a[i], stride[i])) || (rc = fr_range_check(b[i], b_off[i], n_b[i], stride[i])))
      return rc;
  }
  HIPCHK(hipSetDevice(ctx->device));
  auto undo = [&]() {
    for (int i = 0; i < k; i++) {
      if (out[i]) bn256_fr_free(ctx, out[i]);
      out[i] = nullptr;
    }
  };
  for (int i0 = 0; i0 < k; i0 += FR_BATCH) {  // up to FR_BATCH folds per launch
    const int kb = std::min(k - i0, (int)FR_BATCH);
    fr_fold_batch batch;
    memset(&batch, 0, sizeof batch);
    u32 max_blk = 0;
    for (int j = 0; j < kb; j++) {
      const int i = i0 + j;
      const size_t n = std::max(n_a[i], n_b[i]);
      if (!(out[i] = frvec_alloc(n, &rc))) {
        undo();
        return rc;
      }
      fr_fold_desc& D = batch.d[j];
      D.a = (const uint8_t*)(a[i]->d + a_off[i] * 32);
      D.b = (const uint8_t*)(b[i]->d + b_off[i] * 32);
      D.out = out[i]->d;
      D.n_a = (u32)n_a[i];
      D.n_b = (u32)n_b[i];
      D.stride = (u32)stride[i];
      D.n = (u32)n;
      D.s_a = fr_arg_canon(fr_arg_from_be(s_a_be + 32 * i));
      D.s_b = fr_arg_canon(fr_arg_from_be(s_b_be + 32 * i));
      max_blk = std::max(max_blk, nblocks(n, 256));
    }
    if (max_blk) hipLaunchKernelGGL(k_fr_fold, dim3(max_blk, kb), dim3(256), 0, ctx->stream, batch);
  }
  hipError_t e = ctx->fr_defer ? hipGetLastError() : hipStreamSynchronize(ctx->stream);  // one wait for all k folds
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    undo();
    return BN256_EHIP;
  }
  return BN256_OK;
}

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

extern "C" int bn256_fr_tensor_step_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* w, int layout,
                                          const uint8_t* lo_be, const uint8_t* hi_be, bn256_frvec** out) {
  if (!ctx || k < 0 || (layout != 0 && layout != 1) || (k && (!w || !lo_be || !hi_be || !out))) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  int rc;
  for (int i = 0; i < k; i++) {
    out[i] = nullptr;
    if (w[i] && (!is_pow2(w[i]->n) || w[i]->n >= (1u << 30))) return BN256_EINVAL;
  }
  HIPCHK(hipSetDevice(ctx->device));
  for (int i0 = 0; i0 < k; i0 += FR_BATCH) {
    const int kb = std::min(k - i0, (int)FR_BATCH);
    fr_step_batch batch;
    memset(&batch, 0, sizeof batch);
    u32 max_blk = 0;
    for (int j = 0; j < kb; j++) {
      const int i = i0 + j;
      const size_t len = w[i] ? w[i]->n : 1;
      if (!(out[i] = frvec_alloc(2 * len, &rc))) {
        for (int q = 0; q < i; q++) {
          bn256_fr_free(ctx, out[q]);
          out[q] = nullptr;
        }
        return rc;
      }
      fr_step_desc& D = batch.d[j];
      D.w = (const uint8_t*)(w[i] ? w[i]->d : nullptr);
      D.out = out[i]->d;
      D.len = (u32)len;
      D.lo = fr_arg_canon(fr_arg_from_be(lo_be + 32 * i));
      D.hi = fr_arg_canon(fr_arg_from_be(hi_be + 32 * i));
      max_blk = std::max(max_blk, nblocks(2 * len, 256));
    }
    hipLaunchKernelGGL(k_fr_tensor_step, dim3(max_blk, kb), dim3(256), 0, ctx->stream, batch, (u32)layout);
  }
  HIPCHK(hipGetLastError());
  return BN256_OK;  // no wait: the only readers are later calls on this context, on the same stream
}

extern "C" int bn256_fr_tensor_expand_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* v, const size_t* n_out,
                                            const bn256_frvec* const* w, int layout, const int* sel, const int* swap,
                                            const uint8_t* s_clear_be, const uint8_t* s_set_be, bn256_frvec** out) {
  if (!ctx || k < 0 || (layout != 0 && layout != 1) ||
      (k && (!v || !n_out || !w || !sel || !swap || !s_clear_be || !s_set_be || !out)))
    return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  int rc;
  for (int i = 0; i < k; i++) {
    out[i] = nullptr;
    if (!v[i] || sel[i] < 0 || sel[i] > 30) return BN256_EINVAL;
    if (n_out[i] >= (1ull << 31)) return BN256_ETOOLARGE;
    if (w[i] && !is_pow2(w[i]->n)) return BN256_EINVAL;
    // half folds need whole blocks: n_out = n_t * 2^t with n_t a power of two
    if (layout == 1 && n_out[i] && (!is_pow2(n_out[i]) || (w[i] && w[i]->n > n_out[i]))) return BN256_EINVAL;
  }
  HIPCHK(hipSetDevice(ctx->device));
  auto undo = [&]() {
    for (int i = 0; i < k; i++) {
      if (out[i]) bn256_fr_free(ctx, out[i]);
      out[i] = nullptr;
    }
  };
  for (int i0 = 0; i0 < k; i0 += FR_BATCH) {
    const int kb = std::min(k - i0, (int)FR_BATCH);
    fr_expand_batch batch;
    memset(&batch, 0, sizeof batch);
    u32 max_blk = 0;
    for (int j = 0; j < kb; j++) {
      const int i = i0 + j;
      if (!(out[i] = frvec_alloc(n_out[i], &rc))) {
        undo();
        return rc;
      }
      u32 t = 0;
      while (w[i] && ((size_t)1 << t) < w[i]->n) t++;
      fr_expand_desc& D = batch.d[j];
      D.v = (const uint8_t*)v[i]->d;
      D.w = (const uint8_t*)(w[i] ? w[i]->d : nullptr);
      D.out = out[i]->d;
      D.n_v = (u32)v[i]->n;
      D.n_out = (u32)n_out[i];
      D.t = t;
      D.sel = (u32)sel[i];
      D.swap = (u32)(swap[i] != 0);
      D.s_clear = fr_arg_canon(fr_arg_from_be(s_clear_be + 32 * i));
      D.s_set = fr_arg_canon(fr_arg_from_be(s_set_be + 32 * i));
      max_blk = std::max(max_blk, nblocks(n_out[i], 256));
    }
    if (max_blk) hipLaunchKernelGGL(k_fr_tensor_expand, dim3(max_blk, kb), dim3(256), 0, ctx->stream, batch, (u32)layout);
  }
  hipError_t e = ctx->fr_defer ? hipGetLastError() : hipStreamSynchronize(ctx->stream);  // one wait for all k vectors
  if (e != hipSuccess) {
    ctx_note_hip(e, __LINE__);
    undo();
    return BN256_EHIP;
  }
  return BN256_OK;
}

extern "C" int bn256_fr_fold(bn256_ctx* ctx, const bn256_frvec* a, size_t a_off, size_t n_a, const bn256_frvec* b,
                             size_t b_off, size_t n_b, size_t stride, const uint8_t s_a_be[32],
                             const uint8_t s_b_be[32], bn256_frvec** out) {
  if (!ctx || !a || !b || !out || !s_a_be || !s_b_be) return BN256_EINVAL;
  return bn256_fr_fold_multi(ctx, 1, &a, &a_off, &n_a, &b, &b_off, &n_b, &stride, s_a_be, s_b_be, out);
}

// elements per thread: eight or more for throughput (a thread pays ~1.5 log2(start) products for its first weight);
// two for the small vectors of a proof's late rounds, where the product is one latency chain
static u32 fr_dot_per(size_t n) {
  return n <= 4096 ? 2u : (u32)std::max<size_t>(8, (n + 256 * 1024 - 1) / (256 * 1024));
}
// out_be: the k results go to the host (one wait).  out_dev (out_be == nullptr): they stay in HBM as 32-byte
// big-endian elements at out_dev, nothing is waited for when the context defers its Fr waits.
static int fr_dot_multi_core(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off,
                             const size_t* a_stride, const size_t* n_a, const bn256_frvec* const* b,
                             const size_t* b_off, const size_t* b_stride, const size_t* n_b,
                             const uint8_t* const* mu_be, uint8_t* out_be, uint8_t* out_dev) {
  if (!ctx || k < 0 || (k && (!a || !a_off || !a_stride || !n_a || !b || !b_off || !b_stride || !n_b || (!out_be && !out_dev))))
    return BN256_EINVAL;
  if (!k) return BN256_OK;
  int rc;
  size_t need = (size_t)k * 32 + 256;
  bool any = false;
  for (int i = 0; i < k; i++) {
    if (!a[i] || !b[i]) return BN256_EINVAL;
    if ((rc = fr_range_check(a[i], a_off[i], n_a[i], a_stride[i])) || (rc = fr_range_check(b[i], b_off[i], n_b[i], b_stride[i])))
      return rc;
    const size_t n = std::min(n_a[i], n_b[i]);  // the padded tail multiplies by zero
    if (!n) continue;
    any = true;
    const u32 per = fr_dot_per(n);
    need += (size_t)nblocks((n + per - 1) / per, 256) * sizeof(fr) + 256;
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (out_be) memset(out_be, 0, (size_t)k * 32);
  if (!any) {
    if (out_dev) HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)k * 32, ctx->stream));
    if (out_dev && !ctx->fr_defer) HIPCHK(hipStreamSynchronize(ctx->stream));
    return BN256_OK;
  }
  if ((rc = ws_reserve(ctx, WS_TMP, need))) return rc;
  // k results, then every product's partial sums.  The results go straight into the context's mapped pinned
  // buffer when they fit it (they do: 2048 products): no copy behind the summing kernel (see msm_enqueue_srcs)
  const bool direct = !out_dev && (size_t)k * 32 <= ctx->h_pack_cap;
  uint8_t* d_out = out_dev ? out_dev : direct ? (uint8_t*)ctx->d_hpack : (uint8_t*)ctx->ws[WS_TMP];
  uint8_t* cur = (uint8_t*)ctx->ws[WS_TMP] + (((size_t)k * 32 + 255) & ~(size_t)255);
  // products in launches of FR_DOT_BATCH; an empty product (nblk = 0) sums to zero
  for (int i0 = 0; i0 < k; i0 += FR_DOT_BATCH) {
    const int kb = std::min(k - i0, (int)FR_DOT_BATCH);
    fr_dot_batch batch;
    memset(&batch, 0, sizeof batch);
    u32 max_blk = 0;
    for (int j = 0; j < kb; j++) {
      const int i = i0 + j;
      const size_t n = std::min(n_a[i], n_b[i]);
      if (!n) continue;
      const u32 per = fr_dot_per(n);
      const u32 nthreads = (u32)((n + per - 1) / per);
      fr_dot_desc& D = batch.d[j];
      D.nblk = nblocks(nthreads, 256);
      D.partials = (fr*)cur;
      cur += ((size_t)D.nblk * sizeof(fr) + 255) & ~(size_t)255;
      D.a = (const uint8_t*)(a[i]->d + a_off[i] * 32);
      D.b = (const uint8_t*)(b[i]->d + b_off[i] * 32);
      D.a_stride = (u32)a_stride[i];
      D.n_a = (u32)n_a[i];
      D.b_stride = (u32)b_stride[i];
      D.n_b = (u32)n_b[i];
      D.n = (u32)n;
      D.per = per;
      const uint8_t* mb = mu_be ? mu_be[i] : nullptr;
      D.weighted = mb != nullptr;
      if (mb) D.mu = fr_arg_canon(fr_arg_from_be(mb));
      max_blk = std::max(max_blk, D.nblk);
    }
    if (max_blk) hipLaunchKernelGGL(k_fr_dot, dim3(max_blk, kb), dim3(256), 0, ctx->stream, batch);
    hipLaunchKernelGGL(k_fr_sum, dim3(kb), dim3(256), 0, ctx->stream, batch, d_out + 32 * (size_t)i0);
  }
  HIPCHK(hipGetLastError());
  if (out_dev) {
    if (!ctx->fr_defer) HIPCHK(hipStreamSynchronize(ctx->stream));
    return BN256_OK;
  }
  if (!direct) HIPCHK(hipMemcpyAsync(out_be, d_out, (size_t)k * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // one wait for all products
  if (direct) memcpy(out_be, ctx->h_pack, (size_t)k * 32);
  return BN256_OK;
}

extern "C" int bn256_fr_dot_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off,
                                  const size_t* a_stride, const size_t* n_a, const bn256_frvec* const* b,
                                  const size_t* b_off, const size_t* b_stride, const size_t* n_b,
                                  const uint8_t* const* mu_be, uint8_t* out_be) {
  if (!ctx || (k > 0 && !out_be)) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  return fr_dot_multi_core(ctx, k, a, a_off, a_stride, n_a, b, b_off, b_stride, n_b, mu_be, out_be, nullptr);
}

extern "C" int bn256_fr_dot_multi_dev(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off,
                                      const size_t* a_stride, const size_t* n_a, const bn256_frvec* const* b,
                                      const size_t* b_off, const size_t* b_stride, const size_t* n_b,
                                      const uint8_t* const* mu_be, bn256_frvec** out) {
  if (!ctx || !out || k < 1) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  *out = nullptr;
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  bn256_frvec* v = frvec_alloc((size_t)k, &rc);
  if (!v) return rc;
  rc = fr_dot_multi_core(ctx, k, a, a_off, a_stride, n_a, b, b_off, b_stride, n_b, mu_be, nullptr, (uint8_t*)v->d);
  if (!rc && ctx->fr_defer) {  // bn256_fr_download of this vector then waits for these kernels alone
    hipError_t e = hipEventCreateWithFlags(&v->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(v->ready, ctx->stream);
    if (e != hipSuccess) { ctx_note_hip(e, __LINE__); rc = BN256_EHIP; }
  }
  if (rc) { bn256_fr_free(ctx, v); return rc; }
  *out = v;
  return BN256_OK;
}

extern "C" int bn256_fr_dot(bn256_ctx* ctx, const bn256_frvec* a, size_t a_off, size_t a_stride, size_t n_a,
                            const bn256_frvec* b, size_t b_off, size_t b_stride, size_t n_b, const uint8_t* mu_be,
                            uint8_t out_be[32]) {
  if (!ctx || !a || !b || !out_be) return BN256_EINVAL;
  return bn256_fr_dot_multi(ctx, 1, &a, &a_off, &a_stride, &n_a, &b, &b_off, &b_stride, &n_b, &mu_be, out_be);
}

extern "C" int bn256_msm_fr(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t stride, size_t n,
                            const bn256_frvec* v, size_t v_off, size_t v_stride, uint8_t out_xy_be[64]) {
  if (!ctx || !b || !v || !out_xy_be) return BN256_EINVAL;
  int rc;
  if (ctx->nshard || b->nshard) {  // the Fr vector lives on the first device; contiguous views only
    if (stride != 1 || v_stride != 1) return BN256_ENOTSUP;
    if ((rc = fr_range_check(v, v_off, n, 1))) return rc;
    return multi_msm(ctx, b, off, n, v->d + v_off * 32, true, out_xy_be);
  }
  if ((rc = range_check(b, off, n, stride)) || (rc = fr_range_check(v, v_off, n, v_stride))) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  hostg1::Pt r;
  rc = msm_core(ctx, b, off, stride, n, v->d + v_off * 32, &r, v_stride);
  if (rc) return rc;
  hostg1::marshal(out_xy_be, r);
  return BN256_OK;
}

// gathers the segments into the context's workspaces and enqueues one table-less MSM over them
static int fr_multi_enqueue(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* stride,
                            const size_t* n, const bn256_frvec* const* v, const size_t* v_off, const size_t* v_stride,
                            const uint8_t* const* scale_be, bool hold_stream, msm_pending* pend) {
  pend->empty = true;
  size_t total = 0;
  int rc;
  for (int i = 0; i < k; i++) {
    if (!b[i] || !v[i]) return BN256_EINVAL;
    if ((rc = range_check(b[i], off[i], n[i], stride[i])) || (rc = fr_range_check(v[i], v_off[i], n[i], v_stride[i]))) return rc;
    total += n[i];
  }
  if (total == 0) return BN256_OK;
  if (total >= (1ull << 31)) return BN256_ETOOLARGE;
  HIPCHK(hipSetDevice(ctx->device));
  {
    // every segment over a fixed-base table of one window width, unscaled: the views are the terms of
    // one pipeline as they stand (the BP++ commitments <HVec, r||l> + <GVec, n> over resident
    // generators and scalars, go/bppp/docs_test.go:560-564,588-589,1002-1003) -- nothing is gathered
    bool all_tables = true;
    int c0 = 0, live = 0;
    for (int i = 0; i < k; i++) {
      if (!n[i]) continue;
      live++;
      if (b[i]->levels <= 1 || (c0 && b[i]->c != c0) || (scale_be && scale_be[i])) all_tables = false;
      c0 = b[i]->c;
    }
    if (all_tables && live <= 16) {
      msm_src srcs[16];
      int m = 0;
      for (int i = 0; i < k; i++)
        if (n[i]) srcs[m++] = msm_src{b[i], off[i], stride[i], n[i], v[i]->d + v_off[i] * 32, v_stride[i]};
      return msm_enqueue_srcs(ctx, srcs, m, hold_stream, pend);
    }
  }
  if ((rc = ws_reserve(ctx, WS_GATHER, total * sizeof(aff_mem))) || (rc = ws_reserve(ctx, WS_SCALARS, total * 32))) return rc;
  aff_mem* pts = (aff_mem*)ctx->ws[WS_GATHER];
  uint8_t* sc = (uint8_t*)ctx->ws[WS_SCALARS];
  size_t pos = 0;
  {
    gather_list gl(ctx->stream);
    for (int i = 0; i < k; i++) {
      if (!n[i]) continue;
      const uint8_t* s_be = scale_be ? scale_be[i] : nullptr;
      fr_arg scale;
      memset(&scale, 0, sizeof scale);
      if (s_be) scale = fr_arg_canon(fr_arg_from_be(s_be));
      gl.add(gather_desc{(const uint4*)(b[i]->d + off[i]), (uint4*)(pts + pos), (const uint8_t*)(v[i]->d + v_off[i] * 32),
                         sc + pos * 32, (u32)stride[i], (u32)v_stride[i], (u32)(s_be != nullptr), (u32)n[i], scale});
      pos += n[i];
    }
    gl.flush();
  }
  bn256_bases tmp;  // read only while enqueueing
  tmp.d = pts;
  tmp.n = total;
  tmp.levels = 1;
  tmp.c = 0;
  tmp.top_scale = 1;
  tmp.no_inf = false;
  return msm_enqueue(ctx, &tmp, 0, 1, total, sc, 1, hold_stream, pend);
}

extern "C" int bn256_msm_fr_multi(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                                  const size_t* stride, const size_t* n, const bn256_frvec* const* v,
                                  const size_t* v_off, const size_t* v_stride, const uint8_t* const* scale_be,
                                  uint8_t out_xy_be[64]) {
  if (!ctx || k < 0 || !out_xy_be || (k && (!b || !off || !stride || !n || !v || !v_off || !v_stride))) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  memset(out_xy_be, 0, 64);
  for (;;) {
    msm_pending pend;
    int rc = fr_multi_enqueue(ctx, k, b, off, stride, n, v, v_off, v_stride, scale_be, false, &pend);
    if (rc) return rc;
    hostg1::Pt r;
    if ((rc = msm_finish(ctx, &pend, &r)) == MSM_RETRY) continue;
    if (rc) return rc;
    hostg1::marshal(out_xy_be, r);
    return BN256_OK;
  }
}

extern "C" int bn256_msm_fr_multi_submit(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                                         const size_t* stride, const size_t* n, const bn256_frvec* const* v,
                                         const size_t* v_off, const size_t* v_stride, const uint8_t* const* scale_be,
                                         int* ticket) {
  if (!ctx || k < 0 || !ticket || (k && (!b || !off || !stride || !n || !v || !v_off || !v_stride))) return BN256_EINVAL;
  SINGLE_DEVICE_ONLY(ctx);
  *ticket = -1;
  HIPCHK(hipSetDevice(ctx->device));
  int l, rc;
  for (int i = 0; i < k; i++) {
    if (!b[i] || !v[i]) return BN256_EINVAL;
    if ((rc = range_check(b[i], off[i], n[i], stride[i])) || (rc = fr_range_check(v[i], v_off[i], n[i], v_stride[i]))) return rc;
  }
  if ((rc = lane_acquire(ctx, &l))) return rc;
  struct args_t {  // the caller's arrays are not ours after return
    std::vector<const bn256_bases*> b;
    std::vector<const bn256_frvec*> v;
    std::vector<size_t> off, stride, n, v_off, v_stride;
    std::vector<std::vector<uint8_t>> scale;
    std::vector<const uint8_t*> sptr;
  };
  auto a = std::make_shared<args_t>();
  for (int i = 0; i < k; i++) {
    a->b.push_back(b[i]);
    a->v.push_back(v[i]);
    a->off.push_back(off[i]);
    a->stride.push_back(stride[i]);
    a->n.push_back(n[i]);
    a->v_off.push_back(v_off[i]);
    a->v_stride.push_back(v_stride[i]);
    a->scale.emplace_back();
    if (scale_be && scale_be[i]) a->scale.back().assign(scale_be[i], scale_be[i] + 32);
  }
  for (auto& sv : a->scale) a->sptr.push_back(sv.empty() ? nullptr : sv.data());
  rc = lane_start(ctx, l, false, [a, k](bn256_ctx* lane, lane_ticket* t) -> int {
    return bn256_msm_fr_multi(lane, k, a->b.data(), a->off.data(), a->stride.data(), a->n.data(), a->v.data(),
                              a->v_off.data(), a->v_stride.data(), a->sptr.data(), t->xy);
  });
  if (rc) return rc;
  *ticket = l;
  return BN256_OK;
}

extern "C" int bn256_g1_sum(const uint8_t* xy_be, size_t k, uint8_t out_xy_be[64]) {
  if ((!xy_be && k) || !out_xy_be) return BN256_EINVAL;
  hostg1::Pt total = hostg1::infinity();
  for (size_t i = 0; i < k; i++) {
    hostg1::Pt p;
    int rc = hostg1::unmarshal(p, xy_be + 64 * i);
    if (rc) return rc;
    total = hostg1::padd(total, p);
  }
  hostg1::marshal(out_xy_be, total);
  return BN256_OK;
}

extern "C" int bn256_selftest(bn256_ctx* ctx) {
  if (!ctx) return BN256_EINVAL;
  PRIMARY_SHARD(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemsetAsync(ctx->d_err, 0, 4, ctx->stream));
  const u32 n = 1u << 16;
  hipLaunchKernelGGL(k_selftest, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, ctx->d_err, n);
  HIPCHK(hipMemcpyAsync(ctx->h_err, ctx->d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return (int)ctx->h_err[0];
}

// ---------------------------------------------------------------------------- multi-device contexts
// bn256_ctx_create_multi: the reference's caller is ONE goroutine in one process
// (go/bp/main.go:81-178), so a drop-in that is to use the whole node must shard inside the library:
// a handle made by such a context holds contiguous slices of the points, one per device of the list,
// an MSM runs every shard's pipeline side by side (one host thread per shard: its scalar upload,
// enqueue and host tail) and the <= 8 partial points are added on the host.  Nothing crosses between
// the devices but the caller's scalars on their way in (and 64 bytes per shard on the way out).
template <class F>
static int for_each_shard(int nshard, F fn) {
  int rcs[BN256_MAX_DEVICES] = {0};
  char msgs[BN256_MAX_DEVICES][sizeof g_hip_msg];
  std::thread th[BN256_MAX_DEVICES];
  for (int s = 1; s < nshard; s++)
    th[s] = std::thread([&, s] {
      g_hip_msg[0] = 0;
      rcs[s] = fn(s);
      memcpy(msgs[s], g_hip_msg, sizeof g_hip_msg);  // thread_local: carried back to the caller's thread
    });
  rcs[0] = fn(0);
  for (int s = 1; s < nshard; s++) th[s].join();
  for (int s = 0; s < nshard; s++)
    if (rcs[s]) {
      if (s) memcpy(g_hip_msg, msgs[s], sizeof g_hip_msg);
      return rcs[s];
    }
  return BN256_OK;
}

static int multi_bases_make(bn256_ctx* ctx, const uint8_t* xy_be, bool synth, uint64_t seed, size_t first, size_t n,
                            int flags, bn256_bases** out) {
  bn256_bases* b = new (std::nothrow) bn256_bases();
  if (!b) return BN256_ENOMEM;
  b->n = n;
  b->nshard = ctx->nshard;
  const size_t base = n / ctx->nshard, rem = n % ctx->nshard;  // the first n % nshard shards hold one more
  for (int s = 0; s <= ctx->nshard; s++) b->shard_off[s] = (size_t)s * base + std::min<size_t>((size_t)s, rem);
  int rc = for_each_shard(ctx->nshard, [&](int s) {
    const size_t lo = b->shard_off[s], m = b->shard_off[s + 1] - lo;
    return synth ? bn256_random_bases(ctx->shard[s], seed, first + lo, m, flags, &b->shard[s])
                 : bn256_bases_upload(ctx->shard[s], xy_be + 64 * lo, m, flags, &b->shard[s]);
  });
  if (rc) {
    multi_bases_free(ctx, b);
    return rc;
  }
  *out = b;
  return BN256_OK;
}

static void multi_bases_free(bn256_ctx* ctx, bn256_bases* b) {
  for (int s = 0; s < b->nshard; s++)
    bn256_bases_free(ctx && ctx->nshard == b->nshard ? ctx->shard[s] : nullptr, b->shard[s]);
  delete b;
}

static int multi_bases_download(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, uint8_t* xy_be) {
  if (!ctx->nshard || ctx->nshard != b->nshard) return BN256_EINVAL;
  for (int s = 0; s < b->nshard; s++) {
    const size_t lo = std::max(off, b->shard_off[s]), hi = std::min(off + n, b->shard_off[s + 1]);
    if (lo >= hi) continue;
    int rc = bn256_bases_download(ctx->shard[s], b->shard[s], lo - b->shard_off[s], hi - lo, xy_be + 64 * (lo - off));
    if (rc) return rc;
  }
  return BN256_OK;
}

// BN256_FORCE_PEER_COPY=1 (environment): every cross-shard operand takes the hipMemcpyPeerAsync
// path even when source and destination are the same device -- lets a one-GPU box (devices = {0, 0}) run
// the code an 8-GPU node runs (tests/test_gpu_multi_device_ctx.py)
static bool force_peer_copy() {  // read at every call: a test switches it on and off inside one process
  const char* e = getenv("BN256_FORCE_PEER_COPY");
  return e && atoi(e) != 0;
}

// scalars: host memory, or (on_device) memory of any one GPU -- a shard on another device copies its
// slice over the peer link first
static int multi_msm(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const uint8_t* scalars, bool on_device,
                     uint8_t out_xy_be[64]) {
  if (!ctx->nshard || ctx->nshard != b->nshard) return BN256_EINVAL;  // handle and context must be of one kind
  if (off > b->n || n > b->n - off) return BN256_ELENGTH;
  int src_dev = -1;
  if (on_device && n) {
    hipPointerAttribute_t attr;
    HIPCHK(hipPointerGetAttributes(&attr, scalars));
    src_dev = attr.device;
  }
  uint8_t parts[BN256_MAX_DEVICES][64];
  int rc = for_each_shard(ctx->nshard, [&](int s) -> int {
    memset(parts[s], 0, 64);
    const size_t lo = std::max(off, b->shard_off[s]), hi = std::min(off + n, b->shard_off[s + 1]);
    if (lo >= hi) return BN256_OK;
    bn256_ctx* c = ctx->shard[s];
    const size_t m = hi - lo, loc = lo - b->shard_off[s];
    const uint8_t* src = scalars + 32 * (lo - off);
    if (!on_device) return bn256_msm(c, b->shard[s], loc, m, src, parts[s]);
    if (src_dev == c->device && !force_peer_copy()) return bn256_msm_dev(c, b->shard[s], loc, m, src, parts[s]);
    HIPCHK(hipSetDevice(c->device));
    int r = ws_reserve(c, WS_SCALARS, m * 32);
    if (r) return r;
    HIPCHK(hipMemcpyPeerAsync(c->ws[WS_SCALARS], c->device, src, src_dev, m * 32, c->stream));
    return bn256_msm_dev(c, b->shard[s], loc, m, c->ws[WS_SCALARS], parts[s]);  // same stream: ordered after the copy
  });
  if (rc) return rc;
  return bn256_g1_sum(&parts[0][0], (size_t)ctx->nshard, out_xy_be);
}

// ---- element-wise entry points on a multi-device context ("the batched fold / scalar-mul
// kernels are element-wise and shard the same way with no collective").  The OUTPUT handle is sharded evenly
// (the slices bn256_bases_upload would make); shard s computes its slice on its own device with the
// single-device kernels.  An operand range that one shard on that device holds is read in place; anything
// else -- the IPA fold's lo / hi halves live on different GPUs (go/bp/main.go:355-356) -- is copied piece by
// piece over the peer link into a workspace of the computing shard (point-to-point hipMemcpyPeerAsync on the
// shard's own stream; no collective).  WNLA's even / odd fold (go/bppp/docs_test.go:1325-1326) and
// vectorPointMulOnScalar / hadamardPointMul over whole vectors stay local.
static bn256_bases* multi_out_alloc(bn256_ctx* ctx, size_t n) {
  bn256_bases* r = new (std::nothrow) bn256_bases();
  if (!r) return nullptr;
  r->n = n;
  r->levels = 1;
  r->top_scale = 1;
  r->nshard = ctx->nshard;
  const size_t base = n / ctx->nshard, rem = n % ctx->nshard;
  for (int s = 0; s <= ctx->nshard; s++) r->shard_off[s] = (size_t)s * base + std::min<size_t>((size_t)s, rem);
  return r;
}
// level 0 of [start, start + span) of a sharded handle as memory of shard context c's device
static int shard_view(bn256_ctx* c, const bn256_bases* b, size_t start, size_t span, int which, const aff_mem** out) {
  *out = nullptr;
  if (!span) return BN256_OK;
  if (!force_peer_copy())
    for (int t = 0; t < b->nshard; t++)
      if (b->shard[t]->dev == c->device && start >= b->shard_off[t] && start + span <= b->shard_off[t + 1]) {
        *out = b->shard[t]->d + (start - b->shard_off[t]);
        return BN256_OK;
      }
  int rc = ws_reserve(c, which, span * sizeof(aff_mem));
  if (rc) return rc;
  aff_mem* dst = (aff_mem*)c->ws[which];
  for (int t = 0; t < b->nshard; t++) {
    const size_t lo = std::max(start, b->shard_off[t]), hi = std::min(start + span, b->shard_off[t + 1]);
    if (lo >= hi) continue;
    HIPCHK(hipMemcpyPeerAsync(dst + (lo - start), c->device, b->shard[t]->d + (lo - b->shard_off[t]), b->shard[t]->dev,
                              (hi - lo) * sizeof(aff_mem), c->stream));
  }
  *out = dst;
  return BN256_OK;
}
static int multi_check(const bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, size_t stride) {
  if (!b->nshard || b->nshard != ctx->nshard) return BN256_EINVAL;  // handle and context must be of one kind
  if (stride == 0) return BN256_EINVAL;
  if (off > b->n) return BN256_ELENGTH;
  if (n == 0) return BN256_OK;
  const size_t avail = b->n - off;
  if (avail == 0 || (n - 1) > (avail - 1) / stride) return BN256_ELENGTH;
  return BN256_OK;
}
// runs fn(s, shard context, first output index, count, output handle of the shard) on every shard
template <class F>
static int multi_elementwise(bn256_ctx* ctx, size_t n, bn256_bases** out, F fn) {
  *out = nullptr;
  if (n >= (1ull << 31)) return BN256_ETOOLARGE;
  bn256_bases* r = multi_out_alloc(ctx, n);
  if (!r) return BN256_ENOMEM;
  int rc = for_each_shard(ctx->nshard, [&](int s) -> int {
    bn256_ctx* c = ctx->shard[s];
    HIPCHK(hipSetDevice(c->device));
    const size_t i0 = r->shard_off[s], m = r->shard_off[s + 1] - i0;
    int rc2;
    if (!(r->shard[s] = bases_alloc(m, 1, 0, &rc2))) return rc2;
    if (!m) return BN256_OK;
    rc2 = fn(s, c, i0, m, r->shard[s]);
    if (rc2) (void)hipStreamSynchronize(c->stream);  // nothing enqueued may outlive a failed call
    return rc2;
  });
  if (rc) {
    for (int s = 0; s < r->nshard; s++)
      if (r->shard[s]) bn256_bases_free(ctx->shard[s], r->shard[s]);
    delete r;
    return rc;
  }
  *out = r;
  return BN256_OK;
}

static int multi_scalar_mul_batch(bn256_ctx* ctx, const bn256_bases* in, size_t off, size_t n, const uint8_t* scalars_be,
                                  int broadcast, bn256_bases** out) {
  *out = nullptr;
  int rc = multi_check(ctx, in, off, n, 1);
  if (rc) return rc;
  return multi_elementwise(ctx, n, out, [&](int, bn256_ctx* c, size_t i0, size_t m, bn256_bases* r) -> int {
    const aff_mem* src;
    int rc2 = shard_view(c, in, off + i0, m, WS_PEER_A, &src);
    if (rc2) return rc2;
    return scalar_mul_into(c, src, m, broadcast ? scalars_be : scalars_be + 32 * i0, broadcast, r);
  });
}

static int multi_add_batch(bn256_ctx* ctx, const bn256_bases* a, size_t aoff, const bn256_bases* b, size_t boff, size_t n,
                           bn256_bases** out) {
  *out = nullptr;
  int rc;
  if ((rc = multi_check(ctx, a, aoff, n, 1)) || (rc = multi_check(ctx, b, boff, n, 1))) return rc;
  return multi_elementwise(ctx, n, out, [&](int, bn256_ctx* c, size_t i0, size_t m, bn256_bases* r) -> int {
    const aff_mem *pa, *pb;
    int rc2;
    if ((rc2 = shard_view(c, a, aoff + i0, m, WS_PEER_A, &pa)) || (rc2 = shard_view(c, b, boff + i0, m, WS_PEER_B, &pb)))
      return rc2;
    return add_into(c, pa, pb, m, r);
  });
}

static int multi_fold2(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                       size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                       bn256_bases** out) {
  *out = nullptr;
  int rc;
  if ((rc = multi_check(ctx, lo, lo_off, n_lo, stride)) || (rc = multi_check(ctx, hi, hi_off, n_hi, stride))) return rc;
  return multi_elementwise(ctx, std::max(n_lo, n_hi), out,
                           [&](int, bn256_ctx* c, size_t i0, size_t m, bn256_bases* r) -> int {
    // this slice's share of either side (a side that has run out counts as infinity: vectorPointsAdd's padding)
    const size_t m_lo = i0 < n_lo ? std::min(m, n_lo - i0) : 0, m_hi = i0 < n_hi ? std::min(m, n_hi - i0) : 0;
    const aff_mem *pl = nullptr, *ph = nullptr;
    int rc2;
    if ((rc2 = shard_view(c, lo, lo_off + i0 * stride, m_lo ? (m_lo - 1) * stride + 1 : 0, WS_PEER_A, &pl)) ||
        (rc2 = shard_view(c, hi, hi_off + i0 * stride, m_hi ? (m_hi - 1) * stride + 1 : 0, WS_PEER_B, &ph)))
      return rc2;
    if (!pl) pl = ph;  // never read (n = 0 on that side), but the kernels want a valid pointer
    if (!ph) ph = pl;
    if ((rc2 = fold2_into(c, pl, m_lo, ph, m_hi, stride, s_lo_be, s_hi_be, r))) return rc2;
    HIPCHK(hipStreamSynchronize(c->stream));
    return BN256_OK;
  });
}
