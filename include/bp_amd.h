/*
 * libbpamd -- host-side mirror of olegfomenko/crypto's go/bp (Bulletproofs range proof +
 * inner-product argument) on top of libbn256msm.  The Go toolchain is absent from the build
 * image, so the host layer above the C ABI is C++ (crypto_amd/csrc/bp_host.cpp) with the
 * reference's function names inside; these C entry points exist so the tests can drive it.
 *
 * Randomness is injected (the reference draws from crypto/rand, go/bp/util.go:51-63,81-90):
 * `seed` starts a SplitMix64 byte stream that stands in for rand.Reader, consumed in the
 * reference's order, so transcripts are reproducible and comparable
 * byte-for-byte with the oracle's.
 */
#ifndef BP_AMD_H
#define BP_AMD_H
#include <stddef.h>
#include <stdint.h>
#include "bn256msm.h"
#ifdef __cplusplus
extern "C" {
#endif

#define BP_EFAIL (-20)   /* exception text in bp_last_error() */
#define BP_EVERIFY (-21) /* Verify returned an error; the reference's message in bp_last_error() */

/* ---- Product entry points -------------------------------------------------------------------------
 * The reference draws its generators and every blinding value from crypto/rand (go/bp/util.go:51-63,
 * :81-90) and keeps no discrete logarithm.  These three do the same: randomness comes from the
 * operating system's CSPRNG (getrandom(2)), generators are the caller's (resident handles), nothing is
 * derived from a seed.
 *   bp_setup   NewBulletProofPublic(n), go/bp/main.go:27-34: G[n], H[n] as new handles (the caller frees
 *              them with bn256_bases_free), U, g, h as Marshal() bytes; the k of k*G0 are discarded.
 *   bp_prove   (*BulletProofPublic).Prove(v, prv), go/bp/main.go:81-178; same wire format as below.
 *   bp_verify  (*BulletProofPublic).Verify, go/bp/main.go:181-249: 0 = accepted, BP_EVERIFY = rejected.
 * Proofs are randomised, so their test is prove -> verify (here and by the oracle's restatement of
 * Verify), not byte equality. */
int bp_setup(bn256_ctx* ctx, int n, bn256_bases** G, bn256_bases** H, uint8_t U[64], uint8_t g[64], uint8_t h[64]);
int bp_prove(bn256_ctx* ctx, int n, const bn256_bases* G, const bn256_bases* H, const uint8_t U[64], const uint8_t g[64],
             const uint8_t h[64], const uint8_t v_be[32], const uint8_t prv_be[32], uint8_t* out, size_t cap,
             size_t* out_len);
int bp_verify(bn256_ctx* ctx, int n, const bn256_bases* G, const bn256_bases* H, const uint8_t U[64], const uint8_t g[64],
              const uint8_t h[64], const uint8_t* proof, size_t len);

/* ---- TEST HARNESS ONLY: the *_det entry points -----------------------------------------------------
 * A 64-bit SplitMix64 seed stands in for rand.Reader so that transcripts can be compared byte for byte
 * with the oracle's.  The seed generates the generators as k_i*G0 (known discrete logarithms: not
 * sound) and every blinding value (a 2^64 search space: not hiding), and the verifier is handed the
 * same seed.  Never a product path; use bp_setup / bp_prove / bp_verify above.
 *
 * NewBulletProofPublic(n) from the seeded reader, prv := next rand.Int, then
 * (*BulletProofPublic).Prove(v, prv) -- go/bp/main.go:27-34, :81-178, go/bp/main_test.go:16-36.
 * Wire format of the proof (the reference defines none; BulletProof struct go/bp/main.go:59-78):
 *   V, A, S, T1, T2 (64 B each) | tx, taux, nu (32 B BE each) | u32 BE k | L[k], R[k] (64 B each) | a, b (32 B) */
int bp_prove_det(bn256_ctx* ctx, int n, const uint8_t v_be[32], uint64_t seed, uint8_t* out, size_t cap, size_t* out_len);

/* (*BulletProofPublic).Verify -- go/bp/main.go:181-249; public parameters regenerated from the seed.
 * 0 = accepted, BP_EVERIFY = rejected (reference error string in bp_last_error()). */
int bp_verify_det(bn256_ctx* ctx, int n, uint64_t seed, const uint8_t* proof, size_t len);

/* InnerArgumentPublic.Prove + Verify round trip -- go/bp/main.go:263-377 (TestInnerArgumentProof shape). */
int bp_inner_product_roundtrip(bn256_ctx* ctx, int n, uint64_t seed, const uint8_t* a_be, const uint8_t* b_be, int tamper);

/* TestWNLA (go/bppp/docs_test.go:1242-1267) + wnla (:1269-1338) with the reader injected.
 * out: C | u32 BE rounds | (X, R) per round | u32 |l| | l.. | u32 |n| | n..  (opened vectors).
 * 0 = the argument's final check passed, BP_EVERIFY = "Failed to verify!". */
int bp_wnla_det(bn256_ctx* ctx, uint64_t seed, int nl, int nn, const uint8_t* l_be, const uint8_t* n_be,
                uint8_t* out, size_t cap, size_t* out_len);

/* A circuit for bp_ac_det: ACPublic / AcPrivate of go/bppp/docs_test.go:12-41 flattened.  Scalars are
 * 32-byte big-endian, matrices row-major.  The PartitionF (typ, index) -> *int is tabulated: f_lo[j],
 * f_ll[j], f_lr[j] (j < Nv; typ 1, 2, 3) and f_no[j] (j < Nm; typ 4) hold the index into wo, or -1. */
typedef struct {
  int Nm, Nl, Nv, Nw, No, K;
  int Fl, Fm;
  const uint8_t* Wm; /* Nm x Nw */
  const uint8_t* Wl; /* Nl x Nw */
  const uint8_t* Am; /* Nm */
  const uint8_t* Al; /* Nl */
  const uint8_t* v;  /* K x Nv */
  const uint8_t* sv; /* K */
  const uint8_t* wl; /* Nm */
  const uint8_t* wr; /* Nm */
  const uint8_t* wo; /* No */
  const int32_t* f_lo;
  const int32_t* f_ll;
  const int32_t* f_lr;
  const int32_t* f_no;
} bp_ac_circuit;

/* ArithmeticCircuitProtocol -- go/bppp/docs_test.go:496-1240 (Com :514-519, CommitOL :522-567, CommitR
 * :569-591, Cs :1002-1003, PT :1077-1078, C(t) :1173-1177, its three bytes.Equal checks, then wnla
 * :1239) with the reader injected: pre_draws values, G, GVec[Nm], HVec[9+Nv], post_draws values, then
 * the protocol's own draws in the reference's order.  Every vector commitment is a device MSM.
 * out: V[K] | Co Cl Cr V_ Cs PT CT (64 B each) | 3 check bytes (V_, C(t), Eq. 45) | vT (32 B) |
 *      u32 BE |l(T)|, l(T).. | u32 |n(T)|, n(T).. | u32 |c(T)|, c(T).. |
 *      u32 rounds | (X, R) per round | u32 |l|, l.. | u32 |n|, n..   (the WNLA part)
 * 0 = the final WNLA check passed, BP_EVERIFY = "Failed to verify!". */
int bp_ac_det(bn256_ctx* ctx, uint64_t seed, int pre_draws, int post_draws, const bp_ac_circuit* c, uint8_t* out,
              size_t cap, size_t* out_len);

/* matrixMulOnVector(w, Wl), w = wl || wr || wo (go/bppp/util.go:311-319): the "Wl*w" vector the reference's
 * circuit tests print (go/bppp/docs_test.go:326) and record for TestAC3 (:365).  Host arithmetic only (usable
 * without a GPU); only Nm, Nl, No, Nw, Wl, wl, wr, wo of the circuit are read.  out: Nl x 32 bytes BE. */
int bp_ac_wlw(const bp_ac_circuit* c, uint8_t* out);

/* BASELINE.json configs[3] shape: the same argument over synthetic commitment vectors of length
 * nl / nn (e.g. 2^20); reports wall ms for set-up and for the rounds, and the round count. */
int bp_wnla_synthetic(bn256_ctx* ctx, uint64_t seed, size_t nl, size_t nn, double* ms_setup, double* ms_run,
                      int* rounds);

/* ---- go/schnorr-bn256 -------------------------------------------------------
 * Scalars are 32-byte big-endian; S is the reference's UNREDUCED big.Int r + hash * prv
 * (schnorr.go:34,52), 64-byte big-endian; m is the *big.Int message (Msg's result or any scalar).
 *   bp_schnorr_msg          Msg(bytes...), schnorr.go:26-28: Keccak256(data) mod Order.
 *   bp_schnorr_multisig     MultiSigSchnorr, schnorr.go:32-40 (host arithmetic only).
 *   bp_schnorr_sign_det     SignSchnorr, schnorr.go:44-59, with r from the seeded reader (R(G),
 *                           schnorr.go:17-24; r * G runs on the device).  Test harness, like *_det above.
 *   bp_schnorr_verify_batch VerifySchnorr, schnorr.go:64-79, over n signatures at once: the device
 *                           computes hash_i * PublicKey_i + R_i and S_i * G, ok[i] = bytes.Equal. */
int bp_schnorr_msg(const uint8_t* data, size_t len, uint8_t m_be[32]);
int bp_schnorr_multisig(const uint8_t prv_be[32], const uint8_t r_be[32], const uint8_t pub_common[64],
                        const uint8_t r_common[64], const uint8_t m_be[32], uint8_t S_be[64]);
int bp_schnorr_sign_det(bn256_ctx* ctx, uint64_t seed, const uint8_t prv_be[32], const uint8_t pub[64], const uint8_t G[64],
                        const uint8_t m_be[32], uint8_t R_out[64], uint8_t S_be[64]);
int bp_schnorr_verify_batch(bn256_ctx* ctx, size_t n, const uint8_t* R, const uint8_t* S_be, const uint8_t* pub,
                            const uint8_t G[64], const uint8_t* m_be, uint8_t* ok);

/* ---- go/ve-ca: verifiable encryption by cut-and-choose, k = 128 ---------------
 * Proof (go/ve-ca/main.go:10-16) on the wire (the reference defines no format): E[4][k] | Alpha[2][k] |
 * P[2][k] | Share[2][k], 32-byte big-endian each, then X[k] as Marshal() bytes.
 *   bp_veca_encrypt_det  Encrypt, main.go:18-92, reader injected; X[i] = x0[i]*GenH + x1[i]*GenG is one
 *                        bn256_com_batch.  Test harness (seeded).
 *   bp_veca_verify       Verify, main.go:94-133: the k sigma answers alpha0*GenH + alpha1*GenG + rho[b]*C
 *                        are one bn256_com_batch + bn256_add_batch.  0 = true; BP_EVERIFY = false with the
 *                        message the reference prints ("invalid sigma answer" / "invalid e0 answer" /
 *                        "invalid e1 answer") in bp_last_error().
 *   bp_veca_decrypt      Decrypt, main.go:135-174 (host only). */
#define BP_VECA_K 128
#define BP_VECA_PROOF_BYTES (10 * BP_VECA_K * 32 + BP_VECA_K * 64)
int bp_veca_encrypt_det(bn256_ctx* ctx, uint64_t seed, const uint8_t v_be[32], const uint8_t r_be[32],
                        const uint8_t s_r_be[32], const uint8_t g_be[32], const uint8_t GenG[64], const uint8_t GenH[64],
                        uint8_t* proof_out, uint8_t C_out[64]);
int bp_veca_verify(bn256_ctx* ctx, const uint8_t* proof, const uint8_t s_r_be[32], const uint8_t C[64],
                   const uint8_t GenG[64], const uint8_t GenH[64]);
int bp_veca_decrypt(const uint8_t u_be[32], const uint8_t* proof, const uint8_t C[64], const uint8_t GenG[64],
                    const uint8_t GenH[64], uint8_t v_out[32], uint8_t r_out[32]);

/* hash()'s digest function: legacy Keccak-256 -- go/bp/util.go:232. */
int bp_keccak256(const uint8_t* msg, size_t len, uint8_t out[32]);

const char* bp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
