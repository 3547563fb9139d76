/*
 * libbn256msm -- C ABI of the MI355X-native multi-scalar-multiplication engine for the
 * BN256 (cloudflare/bn256) G1 commitments of olegfomenko/crypto's Bulletproofs (go/bp) and
 * Bulletproofs++ (go/bppp) provers.
 *
 * The reference has no FFI: its seam is the package-private helper set in go/bp/util.go
 * and go/bppp/util.go, and bn256.G1 can leave its package only through Marshal() /
 * Unmarshal().  That fixes the wire format used here:
 *   point  = 64 bytes, affine x || y, 32-byte big-endian each, infinity = 64 zero bytes
 *   scalar = 32 bytes big-endian (scalarTo32Byte, go/bp/util.go:209-220)
 *
 * Conventions: every entry point returns 0 or a negative BN256_E* code and never aborts
 * or throws.  Output buffers are caller-allocated; handles are library-owned and freed
 * explicitly; no caller pointer is retained after return (cgo rule).  A context is used
 * from one thread at a time; distinct contexts are independent.  A context is one GPU
 * (bn256_ctx_create: one process per GPU, partial sums of a sharded MSM are combined with
 * bn256_g1_sum) or 1..8 GPUs of the node behind the same calls (bn256_ctx_create_multi).
 */
#ifndef BN256MSM_H
#define BN256MSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bn256_ctx bn256_ctx;
typedef struct bn256_bases bn256_bases;

#define BN256_OK 0
#define BN256_EINVAL (-1)      /* null pointer / bad argument */
#define BN256_ELENGTH (-2)     /* range outside the handle ("invalid length for scalar mul") */
#define BN256_ECOORD (-3)      /* coordinate >= p (Unmarshal: "coordinate exceeds modulus") */
#define BN256_ENOTONCURVE (-4) /* Unmarshal: "malformed point" */
#define BN256_EHIP (-5)        /* HIP runtime failure */
#define BN256_ENOMEM (-6)
#define BN256_ETOOLARGE (-7)   /* n * windows exceeds 2^31 entries */
#define BN256_EBUSY (-8)       /* bn256_msm_submit: every lane holds a submitted call */
#define BN256_ENOTSUP (-10)    /* entry point not served by a multi-device context / sharded handle */
#define BN256_ESCRATCH (-9)    /* bn256_msm_dev_noalloc: the call does not fit the static scratch array */

/* bn256_bases_upload flags */
#define BN256_PRECOMPUTE 1 /* also store 2^(c*w) * P for every window w (fixed-base MSM) */

/* Opens GPU `device` (HIP ordinal).  Owns one stream and the scratch buffers. */
int bn256_ctx_create(int device, bn256_ctx** out);
void bn256_ctx_destroy(bn256_ctx* ctx);
/* One context over 1..BN256_MAX_DEVICES GPUs of the node, for the reference's caller as it is: one
 * goroutine in one process (go/bp/main.go:81-178) that cannot become a process per GPU.  A handle made
 * by such a context (bn256_bases_upload, bn256_random_bases) holds contiguous slices of the points,
 * slice s on devices[s] (the first n % ndev slices hold one point more; the same device may be listed
 * more than once); bn256_msm / bn256_msm_dev / bn256_msm_fr (contiguous views) run every slice's
 * pipeline side by side and add the <= 8 partial points on the host, so vectorPointScalarMul
 * (go/bp/util.go:165-175) uses the node through the same call.  Host scalars go to each device over
 * its own PCIe link; a device pointer may live on any one GPU (the other shards copy their slices
 * over the peer link).  The element-wise calls -- bn256_scalar_mul_batch, bn256_add_batch, bn256_fold,
 * bn256_fold2 (vectorPointMulOnScalar, the h1 loop, hadamardPointMul, the generator folds:
 * go/bp/util.go:177-195, go/bp/main.go:156-159,355-356, go/bppp/docs_test.go:1325-1326) -- return a
 * handle sharded evenly like an uploaded one; every shard computes its slice on its own device, reads
 * an operand range in place when one of its device's shards holds it and copies it over the peer link
 * otherwise (the IPA fold's lo / hi halves live on different GPUs; WNLA's even / odd fold and whole-vector
 * calls stay local): point-to-point copies, no collective.  Fr vectors, bn256_random_scalars and
 * bn256_selftest of such a context live on devices[0]; bn256_bases_free / _len / _download,
 * bn256_ctx_set_tuning and bn256_last_timing (slowest shard) work as usual; the remaining entry points
 * (multi-segment MSMs, the submit / wait pairs, bn256_com_batch, Fr kernels on strided views,
 * bn256_msm_dev_noalloc) return BN256_ENOTSUP.  BN256_FORCE_PEER_COPY=1 in the environment makes every
 * cross-shard operand take the peer-copy path even between shards of one device (test switch).  The
 * process-per-GPU form (bench.py, RCCL all_gather of the partials + bn256_g1_sum) needs nothing of this. */
#define BN256_MAX_DEVICES 8
int bn256_ctx_create_multi(const int* devices, int ndev, bn256_ctx** out);
/* The HIP stream (hipStream_t) every kernel of this context is launched on. */
void* bn256_ctx_stream(bn256_ctx* ctx);

/* Copies n affine points into HBM once (replaces the repeated traversal of []*bn256.G1
 * in vectorPointScalarMul, go/bp/util.go:165-175).  Validates like G1.Unmarshal:
 * coordinates < p and on the curve, or all-zero = infinity.  `xy_be` is host memory. */
int bn256_bases_upload(bn256_ctx* ctx, const uint8_t* xy_be, size_t n, int flags, bn256_bases** out);
void bn256_bases_free(bn256_ctx* ctx, bn256_bases* b);
size_t bn256_bases_len(const bn256_bases* b);

/* out = sum_{i<n} scalars[i] * bases[off+i]  -- vectorPointScalarMul (go/bp/util.go:165-175,
 * go/bppp/util.go:178-196).  n = 0 gives infinity (the BP++ variant's empty case); the
 * go/bp length panic and the BP++ pad/truncate rules live in the caller-side shim.
 * `scalars_be` is host memory (n * 32 bytes); the _dev variant takes a device pointer.  The upload is part of
 * the pipeline: over a fixed-base table, from 2^20 points on, the call runs as at least two passes, each pass's
 * scalars cross PCIe on a copy stream while the previous pass's kernels run (2^20 points, pinned host
 * memory: 1.98 -> 1.86 ms; pageable 2.41 -> 2.21 ms).  No copy is in flight when the call returns. */
int bn256_msm(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const uint8_t* scalars_be,
              uint8_t out_xy_be[64]);
int bn256_msm_dev(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const void* d_scalars_be,
                  uint8_t out_xy_be[64]);

/* Device inputs of every *_dev / *_submit / *_noalloc call: `d_scalars_be` must be 16-byte aligned (the
 * kernels read scalars with 128-bit loads), and the kernels run on the library's own non-blocking
 * streams (bn256_ctx_stream and one per lane), which do not wait for any other stream: when another
 * stream produced the scalars (torch's current stream, say), the caller synchronises that stream or
 * event before the call. */

/* bn256_msm_dev for callers that cannot call hipMalloc: tight serving loops, harnesses that forbid
 * allocation, code that must not touch the allocator between two points of a HIP graph capture (the
 * call itself is NOT capturable: it synchronises its stream and finishes on the host).  It has NO
 * counterpart in the reference (whose
 * vectorPointScalarMul, go/bp/util.go:165-175, allocates freely on the Go heap); same arguments, same
 * result bytes as bn256_msm_dev.  The call makes no hipMalloc / hipFree / hipHostMalloc and no other
 * allocating library call: the pipeline's workspaces (sort records, bucket offsets, bucket and partial
 * sums, packed results) are carved from a file-scope __device__ array of BN256_NOALLOC_SCRATCH_BYTES
 * that every GPU the library runs on holds until the process ends; its device address is resolved once
 * per device by bn256_ctx_create.  A call whose workspaces do not fit returns BN256_ESCRATCH before
 * anything is enqueued.  What has to fit is one PASS (a fixed-base call runs in passes of <= 9/8 * 2^20
 * points, a table-less one in passes of <= 2^20): the default 352 MiB hold a fixed-base pass of 2^20
 * points (278 MiB; 342 MiB with the split-sort workspace that skewed scalars need) and a table-less pass of
 * 2^20 (290 MiB), so calls of any length over the default window widths run; forced window widths whose
 * bucket sets outgrow the array (18 bits without a table: 409 MiB) are refused.  STANDING COST: the array is
 * part of the library's code object, so EVERY process that loads the library holds it on EVERY GPU it
 * touches, whether or not this entry point is ever called; a deployment that never calls it builds with
 * -DBN256_NOALLOC_SCRATCH_BYTES=256 (every call then returns BN256_ESCRATCH), one that needs larger
 * calls with a larger value.  All contexts on one GPU share that GPU's array and it keeps the last call's contents (the
 * pipeline initialises what it reads): calls that overlap on one GPU -- two contexts, two threads --
 * race; serialise them.  The other entry points are unaffected. */
#ifndef BN256_NOALLOC_SCRATCH_BYTES
#define BN256_NOALLOC_SCRATCH_BYTES ((size_t)352 << 20)
#endif
int bn256_msm_dev_noalloc(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const void* d_scalars_be,
                          uint8_t out_xy_be[64]);

/* bn256_msm_dev split in two, for callers with several independent commitments to make -- A and S
 * (go/bp/main.go:89,95), L and R of an inner-product level (go/bp/main.go:340-341), the WNLA
 * round's X and R (go/bppp/docs_test.go:1312-1320) -- none of which needs another's result.  The reference computes
 * them one after the other on its single goroutine; here submit hands the whole call (enqueueing the
 * pipeline, waiting for it, the host tail) to one of BN256_ASYNC_LANES internal lanes -- each a context of its own
 * (stream, workspaces) with a host thread of its own -- and returns at once; wait blocks until that
 * lane has finished and returns the same bytes bn256_msm_dev would.  The calls of the lanes overlap
 * on the GPU (+9 % points/s at 2^20 points) and their host work runs beside the caller's.  Argument
 * errors are reported by submit; errors of the call itself (BN256_ENOMEM, BN256_EHIP) by wait.  The
 * scalars must stay untouched, and the bases handle alive, until the matching wait; a submit while
 * every lane holds a call returns BN256_EBUSY; tickets are lane numbers, valid for one wait.  One caller
 * thread per context, as everywhere in this library.  Eight lanes (each is created at its first use): a WNLA
 * round has its two commitments and its two generator folds (go/bppp/docs_test.go:1312-1326) in flight
 * together, and a proof's small rounds -- latency chains of ~0.5 ms per commitment -- run four rounds deep
 * (the protocol's challenges come from the reader, not from X and R: docs_test.go:1294); with four lanes the
 * WNLA at |l| = 2^20 took 17.2-17.9 ms, with eight 16.6-17.1.  bench.py keeps four calls in flight. */
#define BN256_ASYNC_LANES 8
int bn256_msm_submit(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, const void* d_scalars_be,
                     int* ticket);
int bn256_msm_wait(bn256_ctx* ctx, int ticket, uint8_t out_xy_be[64]);

/* Sum of k segments (bases, off, n, scalars) -- vecCom / productCom / InnerProductH
 * (go/bp/util.go:15-27, :96-104) and the WNLA X / R commitments (go/bppp/docs_test.go:1312-1320).
 * Scalars are host pointers. */
int bn256_msm_multi(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* n,
                    const uint8_t* const* scalars_be, uint8_t out_xy_be[64]);

/* bn256_msm_multi split in two like bn256_msm_submit / bn256_msm_wait (same lanes and tickets,
 * BN256_EBUSY when all are taken; finish with bn256_msm_wait): L, R and P of an inner-product level
 * (go/bp/main.go:340-341,347) and A, S (go/bp/main.go:89,95) are independent commitments that the
 * reference computes one after the other.  The host scalars are copied before the call returns; the
 * bases handles must stay alive until the matching wait. */
int bn256_msm_multi_submit(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* n,
                           const uint8_t* const* scalars_be, int* ticket);

/* As bn256_msm_multi with a point stride per segment: segment j sums scalars[j][i] *
 * bases[j][off[j] + i*stride[j]].  stride 2 addresses the even / odd halves that the WNLA round
 * takes with reducePoints (go/bppp/util.go:213-226, go/bppp/docs_test.go:1297-1320). */
int bn256_msm_multi_strided(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                            const size_t* stride, const size_t* n, const uint8_t* const* scalars_be,
                            uint8_t out_xy_be[64]);

/* out_i = s_i * in[off+i] (broadcast = 0, go/bp/main.go:156-159) or s * in[off+i]
 * (broadcast = 1, vectorPointMulOnScalar, go/bp/util.go:177-183).  Result stays on device. */
int bn256_scalar_mul_batch(bn256_ctx* ctx, const bn256_bases* in, size_t off, size_t n,
                           const uint8_t* scalars_be, int broadcast, bn256_bases** out);

/* out_i = a[aoff+i] + b[boff+i] -- hadamardPointMul (go/bp/util.go:185-195). */
int bn256_add_batch(bn256_ctx* ctx, const bn256_bases* a, size_t aoff, const bn256_bases* b, size_t boff,
                    size_t n, bn256_bases** out);

/* out_i = s_lo * lo[lo_off + i*stride] + s_hi * hi[hi_off + i*stride]: the generator fold of the
 * inner-product argument (go/bp/main.go:355-356, stride 1 over halves) and of WNLA
 * (go/bppp/docs_test.go:1325-1326, stride 2 over even/odd). */
int bn256_fold(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, const bn256_bases* hi, size_t hi_off,
               size_t stride, size_t n, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
               bn256_bases** out);

/* out[i] = a_i * g + b_i * h for i < n: com() (go/bp/util.go:92-94) batched over independent
 * scalar pairs -- the k = 128 commitments X[i] = x0[i]*GenH + x1[i]*GenG of go/ve-ca/main.go:37-40
 * and the alpha0*GenH + alpha1*GenG part of its Verify (:102-109).  g, h: Marshal() bytes. */
int bn256_com_batch(bn256_ctx* ctx, const uint8_t g_xy_be[64], const uint8_t h_xy_be[64], const uint8_t* a_be,
                    const uint8_t* b_be, size_t n, bn256_bases** out);

/* Ragged form of bn256_fold: out has max(n_lo, n_hi) points and a missing side counts as
 * infinity -- vectorPointsAdd's padding (go/bppp/util.go:279-293) for odd-length WNLA rounds. */
int bn256_fold2(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                bn256_bases** out);

/* bn256_fold2 split in two: the generator folds of one level, g' = x^-1 G_lo + x G_hi and
 * h' = x H_lo + x^-1 H_hi (go/bp/main.go:355-356; go/bppp/docs_test.go:1325-1326), are independent
 * latency chains that the reference runs one after the other.  submit enqueues fold and
 * normalisation on a free lane (the lanes and tickets of bn256_msm_submit, BN256_EBUSY when all are
 * taken) and returns the new handle at once; its contents are defined only after bn256_fold_wait
 * on that ticket (bn256_msm_wait rejects a fold ticket and vice versa).  lo and hi must stay alive
 * until then. */
int bn256_fold2_submit(bn256_ctx* ctx, const bn256_bases* lo, size_t lo_off, size_t n_lo, const bn256_bases* hi,
                       size_t hi_off, size_t n_hi, size_t stride, const uint8_t s_lo_be[32], const uint8_t s_hi_be[32],
                       bn256_bases** out, int* ticket);
int bn256_fold_wait(bn256_ctx* ctx, int ticket);

/* Waits for everything this context has enqueued and voids every outstanding ticket (submitted MSMs
 * and folds; a handle returned by bn256_fold2_submit keeps its memory and is freed as usual).  For a
 * caller that lost a ticket on an error path -- an exception between submit and wait -- and must not
 * leave its lane busy: without this, BN256_ASYNC_LANES such losses make every submit return BN256_EBUSY. */
int bn256_ctx_drain(bn256_ctx* ctx);

/* Marshal() of n points of a handle into host memory. */
int bn256_bases_download(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t n, uint8_t* xy_be);

/* Deterministic synthetic generators / scalars for benchmarks, replacing points(n) /
 * values(n) (go/bp/util.go:51-63, :81-90) at sizes where n ScalarBaseMults on the CPU are
 * infeasible.  base_i = k_i * G0 with k_i = synth(seed, first + i) (0 -> 1); definition in
 * DESIGN.md.  bn256_random_scalars writes n * 32 big-endian bytes to a device pointer. */
int bn256_random_bases(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, int flags, bn256_bases** out);
int bn256_random_scalars(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, void* d_out_be);

/* ---- Fr vectors resident in HBM: the prover's scalar algebra ----------
 * A bn256_frvec is a []*big.Int reduced mod bn256.Order, stored as the reference's own scalar
 * encoding (32-byte big-endian, scalarTo32Byte go/bp/util.go:209-220), so the MSM kernels
 * consume it in place.  Views are (off, stride, n): element i is v[off + i * stride]. */
typedef struct bn256_frvec bn256_frvec;
int bn256_fr_upload(bn256_ctx* ctx, const uint8_t* scalars_be, size_t n, bn256_frvec** out);
int bn256_fr_random(bn256_ctx* ctx, uint64_t seed, size_t first, size_t n, bn256_frvec** out); /* = bn256_random_scalars */
int bn256_fr_download(bn256_ctx* ctx, const bn256_frvec* v, size_t off, size_t n, uint8_t* scalars_be);
size_t bn256_fr_len(const bn256_frvec* v);
const void* bn256_fr_device_ptr(const bn256_frvec* v);
void bn256_fr_free(bn256_ctx* ctx, bn256_frvec* v);

/* out[i] = s_a * a[a_off + i*stride] + s_b * b[b_off + i*stride] mod n for i < max(n_a, n_b); a
 * side that has run out reads 0.  Replaces vectorAdd(vectorMulOnScalar(..), vectorMulOnScalar(..)):
 * IPA a' = x a_lo + x^-1 a_hi (go/bp/main.go:357-358, stride 1), WNLA l' = l0 + y l1 and
 * n' = rho^-1 n0 + y n1 over reduceVector's even/odd halves (go/bppp/docs_test.go:1318-1319,
 * go/bppp/util.go:198-211,245-260, stride 2). */
int bn256_fr_fold(bn256_ctx* ctx, const bn256_frvec* a, size_t a_off, size_t n_a, const bn256_frvec* b,
                  size_t b_off, size_t n_b, size_t stride, const uint8_t s_a_be[32], const uint8_t s_b_be[32],
                  bn256_frvec** out);

/* out = sum_i a_i * b_i * w_i mod n over the shorter view (the other is zero-padded), with
 * w_i = 1 when mu_be is NULL -- vectorMul, go/bp/util.go:132-142, go/bppp/util.go:143-157 --
 * and w_i = mu^(i+1) otherwise -- weightVectorMul, go/bppp/util.go:159-176. */
int bn256_fr_dot(bn256_ctx* ctx, const bn256_frvec* a, size_t a_off, size_t a_stride, size_t n_a,
                 const bn256_frvec* b, size_t b_off, size_t b_stride, size_t n_b, const uint8_t* mu_be,
                 uint8_t out_be[32]);

/* k folds / k inner products as ONE call: the kernels are enqueued back to back and the host waits once
 * (and, for the products, copies k * 32 result bytes once).  A WNLA round needs five inner products
 * for v_x, v_r (go/bppp/docs_test.go:1297-1311) and three vector folds c', l', n' (:1327-1334); a wait
 * and a 32-byte copy per product was 0.2 ms of every round.  Arrays have k entries; s_a_be / s_b_be are
 * k * 32 bytes; mu_be is NULL or an array whose entries may be NULL (plain vectorMul); out_be is
 * k * 32 bytes.  Semantics per entry are those of bn256_fr_fold / bn256_fr_dot. */
int bn256_fr_fold_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off, const size_t* n_a,
                        const bn256_frvec* const* b, const size_t* b_off, const size_t* n_b, const size_t* stride,
                        const uint8_t* s_a_be, const uint8_t* s_b_be, bn256_frvec** out);
int bn256_fr_dot_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off, const size_t* a_stride,
                       const size_t* n_a, const bn256_frvec* const* b, const size_t* b_off, const size_t* b_stride,
                       const size_t* n_b, const uint8_t* const* mu_be, uint8_t* out_be);
/* The same k inner products with the results left in HBM: *out is a new vector of k elements.  Nothing crosses
 * PCIe and, while bn256_ctx_fr_defer_sync is on, nothing is waited for: v_x and v_r of a WNLA round
 * (go/bppp/docs_test.go:1297-1311) enter only X and R through v_x g and v_r g (:1312,1318), which the next
 * round does not need (its y comes from the reader, :1294), so a prover reads them with bn256_fr_download when
 * it collects the round's commitments instead of waiting for them inside the round (bn256_fr_download of *out
 * then waits for these k products alone, not for what the context has enqueued since).  k >= 1. */
int bn256_fr_dot_multi_dev(bn256_ctx* ctx, int k, const bn256_frvec* const* a, const size_t* a_off,
                           const size_t* a_stride, const size_t* n_a, const bn256_frvec* const* b, const size_t* b_off,
                           const size_t* b_stride, const size_t* n_b, const uint8_t* const* mu_be, bn256_frvec** out);

/* Scalar vectors for MSMs over generators that were NOT folded.  After t folding rounds G' = lo G_a (+) hi G_b
 * (WNLA: G'[i] = rho G[2i] + y G[2i+1], go/bppp/docs_test.go:1325-1326; inner-product argument:
 * G'[i] = x^-1 G[i] + x G[i + n/2], go/bp/main.go:355-356) the round-t generator is a fixed combination of the
 * frozen ones, G_t[I] = sum_j W(j) G_0[m(I, j)], W the tensor of the rounds' fold scalars, so every later
 * commitment <G_t, a> equals <G_0, W (x) a>: ONE MSM over the frozen vector in place of t generator folds --
 * each a latency chain of ~130 doublings per output, ~0.45 ms whatever the length, which is what the small
 * rounds of a proof spent their time on.  Same group elements, same transcript; the reference folds (on a CPU
 * it has no reason not to).  layout 0: even / odd folds, m = 2^t I + j; layout 1: half folds, m = I + n_t j
 * (n_out must then be a power of two).
 *   bn256_fr_tensor_step_multi    out_i = the weights W after one more round with scalars (lo_i, hi_i); w_i = NULL
 *                                 stands for the one-element vector [1].  Enqueued, not waited for: its only
 *                                 readers are later calls on this context.
 *   bn256_fr_tensor_expand_multi  out_i[m] = w_i[j] * s_i(I) * v_i[I ^ (swap_i << sel_i)], m < n_out_i, with
 *                                 s_i(I) = s_set_i if bit sel_i of I is set, else s_clear_i (k * 32 bytes each; a
 *                                 zero scalar blanks that half); swap_i = 1 pairs each half with the other
 *                                 half's scalars (<H0, l1> + <H1, l0>, docs_test.go:1312-1316; <G[n1:], a[:n1]>,
 *                                 go/bp/main.go:340); a source index past the end of v_i reads 0 (the BP++
 *                                 zero padding).  sel_i = 0 is the even / odd split, sel_i = log2(n_t) - 1 the
 *                                 lower / upper half.  One device call, one wait.
 * The vectors live on devices[0] of a multi-device context. */
int bn256_fr_tensor_step_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* w, int layout, const uint8_t* lo_be,
                               const uint8_t* hi_be, bn256_frvec** out);
int bn256_fr_tensor_expand_multi(bn256_ctx* ctx, int k, const bn256_frvec* const* v, const size_t* n_out,
                                 const bn256_frvec* const* w, int layout, const int* sel, const int* swap,
                                 const uint8_t* s_clear_be, const uint8_t* s_set_be, bn256_frvec** out);

/* Batching of small Fr calls.  A proof's small rounds are a handful of tiny kernels each (three vector folds, four
 * expanded vectors, five inner products: go/bppp/docs_test.go:1297-1334), and a wait per call (~25 us) was
 * most of such a round.  While on, the calls that return no host-visible value -- bn256_fr_fold(_multi),
 * bn256_fr_tensor_expand_multi, bn256_fr_dot_multi_dev -- return as soon as their kernels are enqueued; the
 * results are defined for every later call on this context at once (same stream) and for every call SUBMITTED to
 * a lane while the setting is on (bn256_msm_submit and its relatives, bn256_fold2_submit: the lane's stream waits
 * on the device for what this context had enqueued when the call was submitted); for everyone else after the
 * next call that does wait (bn256_fr_dot(_multi), bn256_fr_download, bn256_ctx_drain, any blocking MSM).  Off by
 * default. */
int bn256_ctx_fr_defer_sync(bn256_ctx* ctx, int on);

/* bn256_msm over views: sum_i v[v_off + i*v_stride] * b[off + i*stride], scalars never leave
 * the device (vectorPointScalarMul on reducePoints / reduceVector halves). */
int bn256_msm_fr(bn256_ctx* ctx, const bn256_bases* b, size_t off, size_t stride, size_t n, const bn256_frvec* v,
                 size_t v_off, size_t v_stride, uint8_t out_xy_be[64]);

/* Sum of k such view-MSMs as ONE device MSM (the segments are gathered into contiguous arrays
 * first): vecCom / productCom / InnerProductH (go/bp/util.go:15-27,96-104) and the WNLA X and R
 * commitments (go/bppp/docs_test.go:1296-1316) over resident vectors.  scale_be (array or NULL;
 * entries may be NULL) multiplies segment i's scalars: <rho * n1, G0> without a scaled copy. */
int bn256_msm_fr_multi(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off, const size_t* stride,
                       const size_t* n, const bn256_frvec* const* v, const size_t* v_off, const size_t* v_stride,
                       const uint8_t* const* scale_be, uint8_t out_xy_be[64]);

/* bn256_msm_fr_multi split in two, as bn256_msm_submit / bn256_msm_wait split bn256_msm_dev (same
 * lanes, same tickets, same BN256_EBUSY rule; finish with bn256_msm_wait): the WNLA round's X and R
 * commitments (go/bppp/docs_test.go:1312-1320) need neither each other's result nor the round's
 * inner products, so the prover mirror enqueues both, computes v_x and v_r on the context's own
 * stream meanwhile, and then waits.  The reference runs them one after the other on its single
 * goroutine.  Every handle named here must stay alive and unchanged until the matching wait. */
int bn256_msm_fr_multi_submit(bn256_ctx* ctx, int k, const bn256_bases* const* b, const size_t* off,
                              const size_t* stride, const size_t* n, const bn256_frvec* const* v, const size_t* v_off,
                              const size_t* v_stride, const uint8_t* const* scale_be, int* ticket);

/* Host-side sum of k marshalled points (combination of per-GPU partial MSMs). */
int bn256_g1_sum(const uint8_t* xy_be, size_t k, uint8_t out_xy_be[64]);

/* Window width in bits for a given n (0 = library default heuristic) and chunk length of the
 * bucket-accumulation kernel (0 = default).  Tuning knobs; results never depend on them. */
int bn256_ctx_set_tuning(bn256_ctx* ctx, int window_bits, int chunk);

/* Per-phase GPU time (ms, HIP events on the context stream) of the last bn256_msm* call:
 * [0] recode+count  [1] scan  [2] scatter  [3] bucket accumulation  [4] partial fix-up
 * [5] bucket reduction  [6] total device time.  Returns the number of values written.  Zeros unless the
 * instrumentation was switched on with bn256_ctx_set_phase_timing(ctx, 1) (or BN256_PHASE_TIMING=1 in the
 * environment; =0 forces it off): the seven event records cost 6-8 us of stream time each -- a third of a
 * 64-point call, 1-2.5 % of a 2^20-point one -- and make a call of several passes wait for each pass. */
int bn256_ctx_set_phase_timing(bn256_ctx* ctx, int on);
int bn256_last_timing(bn256_ctx* ctx, float* ms, int cap);

/* Field-arithmetic self test on the device (fe_mul inline-asm path against its portable
 * restatement and algebraic identities).  Returns 0 when every lane agrees. */
int bn256_selftest(bn256_ctx* ctx);

/* Host-side GLV split used by the generator folds (no counterpart in the reference: cloudflare's
 * curvePoint.Mul does its own lattice decomposition internally): k mod n = k1 + k2 * lambda (mod n)
 * with |k1|, |k2| < 2^129, lambda the eigenvalue of (x, y) -> (beta x, y).  Magnitudes as 32-byte
 * big-endian, signs as 0 / 1.  Pure host arithmetic (no GPU needed); exported for the tests. */
int bn256_glv_split(const uint8_t k_be[32], uint8_t k1_be[32], int* k1_neg, uint8_t k2_be[32], int* k2_neg);

const char* bn256_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* BN256MSM_H */
