/*
 * oracle/kzg_oracle.h - C API of the CPU ORACLE (a restatement of succinctlabs/kzg-rs
 * v0.2.8's verification path).  TEST INFRASTRUCTURE ONLY: used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / timed CPU
 * baseline.  The product path (kzg_rs_amd/) never links, loads or calls it.
 *
 * Return codes mirror the reference's Result<bool, KzgError> (src/enums.rs:6-18):
 *   ORACLE_OK                  -> Ok(*ok)
 *   ORACLE_BADARGS             -> Err(KzgError::BadArgs)            (undecodable / non-canonical input)
 *   ORACLE_INVALID_LENGTH      -> Err(KzgError::InvalidBytesLength)
 *   ORACLE_ERROR               -> Err(KzgError::InternalError)
 */
#ifndef KZG_ORACLE_H
#define KZG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_OK = 0, ORACLE_BADARGS = 1, ORACLE_ERROR = 2, ORACLE_INVALID_LENGTH = 4 };

typedef struct oracle_settings oracle_settings;

/* Parse the trusted-setup text format of src/trusted_setup.txt (build.rs:23-87):
 * "4096\n65\n" + 4096 G1 (Lagrange) hex lines + 65 G2 (monomial) hex lines.
 * load_g1 != 0 additionally decompresses and bit-reversal-permutes the G1 points
 * (build.rs:79,89-105); verification itself only needs roots_of_unity and g2_points[1]. */
oracle_settings *oracle_settings_load_txt(const char *txt, size_t len, int load_g1);
/* Custom settings from [tau]G2 (96-byte compressed): the reference supports custom
 * settings through EnvKzgSettings::Custom (src/trusted_setup.rs:52-57). */
oracle_settings *oracle_settings_from_tau_g2(const uint8_t tau_g2[96]);
void oracle_settings_free(oracle_settings *s);
/* big-endian canonical bytes of roots_of_unity[i] (bit-reversed order, build.rs:131-144) */
void oracle_settings_root(const oracle_settings *s, size_t i, uint8_t out_be[32]);
/* compressed g1_points[i] (after bit-reversal) - only when loaded with load_g1 */
int oracle_settings_g1(const oracle_settings *s, size_t i, uint8_t out[48]);
void oracle_settings_g2(const oracle_settings *s, int i /*0 or 1*/, uint8_t out[96]);

/* src/kzg_proof.rs:353-397 */
int oracle_verify_kzg_proof(int *ok, const uint8_t c[48], const uint8_t z[32], const uint8_t y[32],
                            const uint8_t proof[48], const oracle_settings *s);
/* src/kzg_proof.rs:446-470 */
int oracle_verify_blob_kzg_proof(int *ok, const uint8_t *blob, const uint8_t c[48], const uint8_t proof[48],
                                 const oracle_settings *s);
/* src/kzg_proof.rs:472-525 (n==0 / n==1 shortcuts included; the Vec-length-mismatch branches
 * :491-501 live in the caller because a single n cannot express them).
 * nthreads > 1 splits the per-blob loop (:261-273) across threads for the "all cores" CPU
 * baseline; the reference itself is single-threaded. be_transcript != 0 switches the z/y
 * serialisation inside the r transcript to c-kzg-4844's big-endian (quirk Q1). */
int oracle_verify_blob_kzg_proof_batch(int *ok, const uint8_t *blobs, const uint8_t *commitments,
                                       const uint8_t *proofs, size_t n, const oracle_settings *s, int nthreads,
                                       int be_transcript);
/* Same, also returning intermediates: zs/ys (n x 32 B big-endian), r (32 B BE), A and B
 * (48-byte compressed: A = sum r^i pi_i, B = sum r^i (C_i - y_i G) + sum r^i z_i pi_i). */
int oracle_verify_blob_kzg_proof_batch_ex(int *ok, const uint8_t *blobs, const uint8_t *commitments,
                                          const uint8_t *proofs, size_t n, const oracle_settings *s,
                                          int nthreads, int be_transcript, uint8_t *zs, uint8_t *ys,
                                          uint8_t r_be[32], uint8_t A[48], uint8_t B[48]);

/* src/kzg_proof.rs:399-444 (verify_kzg_proof_batch) over byte inputs */
int oracle_verify_kzg_proof_batch(int *ok, const uint8_t *commitments, const uint8_t *zs_be, const uint8_t *ys_be,
                                  const uint8_t *proofs, size_t n, const oracle_settings *s, int be_transcript);

/* src/kzg_proof.rs:46-72: z (big-endian canonical) from blob + compressed commitment bytes */
int oracle_compute_challenge(uint8_t z_be[32], const uint8_t *blob, const uint8_t commitment[48]);
/* src/kzg_proof.rs:94-133 (blob parsed per src/dtypes.rs:48-57; z given as BE bytes reduced mod r,
 * like scalar_from_bytes_unchecked) */
int oracle_evaluate_polynomial_in_evaluation_form(uint8_t y_be[32], const uint8_t *blob, const uint8_t z_be[32],
                                                  const oracle_settings *s);
/* src/kzg_proof.rs:291-348: r (BE canonical) */
int oracle_compute_r(uint8_t r_be[32], const uint8_t *commitments, const uint8_t *zs_be, const uint8_t *ys_be,
                     const uint8_t *proofs, size_t n, int be_transcript);

/* primitives exposed for parity tests of individual kernels */
int oracle_g1_decompress(uint8_t xy_be[96], int *is_inf, const uint8_t in[48]); /* 0 ok / ORACLE_BADARGS */
int oracle_g1_msm(uint8_t out[48], const uint8_t *points48, const uint8_t *scalars_be, size_t n);
int oracle_g1_mul(uint8_t out[48], const uint8_t point[48], const uint8_t scalar_be[32]);
int oracle_g2_mul(uint8_t out[96], const uint8_t point[96], const uint8_t scalar_be[32]);
int oracle_g1_add(uint8_t out[48], const uint8_t a[48], const uint8_t b[48]);
/* e(a1,a2) == e(b1,b2)  (src/pairings.rs:5-9); points compressed, subgroup unchecked */
int oracle_pairings_verify(int *ok, const uint8_t a1[48], const uint8_t a2[96], const uint8_t b1[48],
                           const uint8_t b2[96]);
void oracle_sha256(uint8_t out[32], const uint8_t *data, size_t len);
void oracle_fr_mul(uint8_t out_be[32], const uint8_t a_be[32], const uint8_t b_be[32]);
void oracle_fr_inv(uint8_t out_be[32], const uint8_t a_be[32]);
/* Montgomery / tower constants, for the anchors of SURVEY.md 10.3 */
void oracle_constants(uint8_t fr_R[32], uint8_t fr_R2[32], uint64_t *fr_inv, uint8_t fp_R[48], uint8_t fp_R2[48],
                      uint64_t *fp_inv);

/* bench_threads.c: the CPU baseline under T pthreads, each making independent single-threaded calls for `seconds` -
 * kind 0: oracle_verify_kzg_proof per tuple, kind 1: oracle_verify_blob_kzg_proof_batch of per_call blobs.  out: [0] calls,
 * [1] wall seconds, [2] calls that did not return Ok(true), [3] aggregate calls/s (sum of the threads' own rates), [4] / [5] the
 * slowest / fastest thread's calls/s. */
int oracle_bench_threads(double out[6], int kind, size_t threads, double seconds, const uint8_t *blobs, const uint8_t *c,
                         const uint8_t *z, const uint8_t *y, const uint8_t *p, size_t n_items, size_t per_call,
                         const oracle_settings *s);

#ifdef __cplusplus
}
#endif
#endif
