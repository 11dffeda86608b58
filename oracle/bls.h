/*
 * oracle/bls.h - BLS12-381 arithmetic for the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by
 * the product path (kzg_rs_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * The reference (succinctlabs/kzg-rs v0.2.8) takes all field/curve/pairing arithmetic
 * from the un-vendored crate sp1_bls12_381 =0.8.0-sp1-6.0.0 (Cargo.toml:13-17), which is
 * absent from /root/reference.  This file restates the PUBLISHED BLS12-381 definitions
 * (p, r, x = -0xd201000000010000, E: y^2 = x^3 + 4, E': y^2 = x^3 + 4(u+1),
 * Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(u+1)), Fp12 = Fp6[w]/(w^2-v)); only group
 * elements, canonical scalars and booleans are observable at the reference's call
 * sites, so internal representation is free.
 *
 * Parity pin: checked against all 175 c-kzg-4844 mainnet vectors + the two scalar KATs
 * held by the reference's tests (tests/test_oracle_vectors.py), see DESIGN.md.
 *
 * Representation: 64-bit little-endian limbs, Montgomery form (R = 2^256 for Fr,
 * 2^384 for Fp).  Every derived constant (R, R^2, -m^-1 mod 2^64, Frobenius
 * coefficients, generators' coordinates) is COMPUTED at bls_init() from p, r and the
 * standard compressed generator encodings, not remembered.
 */
#ifndef ORACLE_BLS_H
#define ORACLE_BLS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } fr_t;
typedef struct { uint64_t l[6]; } fp_t;
typedef struct { fp_t c0, c1; } fp2_t;       /* c0 + c1 u          */
typedef struct { fp2_t c0, c1, c2; } fp6_t;  /* c0 + c1 v + c2 v^2 */
typedef struct { fp6_t c0, c1; } fp12_t;     /* c0 + c1 w          */

typedef struct { fp_t x, y; int inf; } g1a_t;       /* affine           */
typedef struct { fp_t x, y, z; } g1_t;              /* Jacobian, z==0 <=> infinity */
typedef struct { fp2_t x, y; int inf; } g2a_t;
typedef struct { fp2_t x, y, z; } g2_t;             /* Jacobian         */

void bls_init(void); /* idempotent; must be called once before anything else */

/* ---- Fr ---- */
extern const uint64_t FR_MOD[4];
void fr_zero(fr_t *r);
void fr_one(fr_t *r);
void fr_from_u64(fr_t *r, uint64_t v);
int  fr_from_be_canonical(fr_t *r, const uint8_t b[32]); /* 0 ok, -1 if >= r */
void fr_from_be_reduce(fr_t *r, const uint8_t b[32]);    /* big-endian integer mod r */
void fr_from_raw_reduce(fr_t *r, const uint64_t le[4]);  /* Scalar::from_raw */
void fr_to_be(uint8_t b[32], const fr_t *a);
void fr_to_le(uint8_t b[32], const fr_t *a);
void fr_to_raw(uint64_t le[4], const fr_t *a);           /* canonical integer limbs */
void fr_add(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *r, const fr_t *a);
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sqr(fr_t *r, const fr_t *a);
void fr_pow(fr_t *r, const fr_t *a, const uint64_t *e, int nlimbs);
void fr_inv(fr_t *r, const fr_t *a); /* a != 0 */
int  fr_eq(const fr_t *a, const fr_t *b);
int  fr_is_zero(const fr_t *a);

/* ---- Fp ---- */
extern const uint64_t FP_MOD[6];
void fp_zero(fp_t *r);
void fp_one(fp_t *r);
int  fp_from_be_canonical(fp_t *r, const uint8_t b[48]); /* -1 if >= p */
void fp_to_be(uint8_t b[48], const fp_t *a);
void fp_to_raw(uint64_t le[6], const fp_t *a);
void fp_from_raw(fp_t *r, const uint64_t le[6]);         /* value < p */
void fp_add(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *r, const fp_t *a);
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sqr(fp_t *r, const fp_t *a);
void fp_pow(fp_t *r, const fp_t *a, const uint64_t *e, int nlimbs);
void fp_inv(fp_t *r, const fp_t *a);
int  fp_sqrt(fp_t *r, const fp_t *a); /* 0 ok, -1 if a is not a square */
int  fp_eq(const fp_t *a, const fp_t *b);
int  fp_is_zero(const fp_t *a);
int  fp_is_lex_largest(const fp_t *a); /* a > (p-1)/2 */

/* ---- towers ---- */
void fp2_zero(fp2_t *r);
void fp2_one(fp2_t *r);
void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_neg(fp2_t *r, const fp2_t *a);
void fp2_conj(fp2_t *r, const fp2_t *a);
void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_sqr(fp2_t *r, const fp2_t *a);
void fp2_mul_fp(fp2_t *r, const fp2_t *a, const fp_t *k);
void fp2_mul_xi(fp2_t *r, const fp2_t *a); /* times (1 + u) */
void fp2_inv(fp2_t *r, const fp2_t *a);
int  fp2_sqrt(fp2_t *r, const fp2_t *a);
int  fp2_eq(const fp2_t *a, const fp2_t *b);
int  fp2_is_zero(const fp2_t *a);
void fp2_pow(fp2_t *r, const fp2_t *a, const uint64_t *e, int nlimbs);

void fp6_mul(fp6_t *r, const fp6_t *a, const fp6_t *b);
void fp6_inv(fp6_t *r, const fp6_t *a);

void fp12_one(fp12_t *r);
void fp12_mul(fp12_t *r, const fp12_t *a, const fp12_t *b);
void fp12_sqr(fp12_t *r, const fp12_t *a);
void fp12_conj(fp12_t *r, const fp12_t *a);
void fp12_inv(fp12_t *r, const fp12_t *a);
void fp12_frobenius(fp12_t *r, const fp12_t *a); /* a^p */
int  fp12_eq(const fp12_t *a, const fp12_t *b);
int  fp12_is_one(const fp12_t *a);

/* ---- G1 ---- */
extern g1a_t G1_GENERATOR;
void g1_set_inf(g1_t *r);
int  g1_is_inf(const g1_t *a);
void g1_from_affine(g1_t *r, const g1a_t *a);
void g1_to_affine(g1a_t *r, const g1_t *a);
void g1_dbl(g1_t *r, const g1_t *a);
void g1_add(g1_t *r, const g1_t *a, const g1_t *b);
void g1_add_affine(g1_t *r, const g1_t *a, const g1a_t *b);
void g1_neg(g1_t *r, const g1_t *a);
void g1a_neg(g1a_t *r, const g1a_t *a);
void g1_mul(g1_t *r, const g1_t *a, const fr_t *k);           /* double-and-add over the canonical scalar */
void g1_mul_raw(g1_t *r, const g1_t *a, const uint64_t *e, int nlimbs);
int  g1a_is_on_curve(const g1a_t *a);
int  g1a_in_subgroup(const g1a_t *a);                          /* [r]P == O */
int  g1_decompress(g1a_t *r, const uint8_t b[48], int check_subgroup); /* 0 ok, -1 rejected */
void g1_compress(uint8_t b[48], const g1a_t *a);
void g1_msm(g1_t *r, const g1a_t *pts, const fr_t *scalars, size_t n); /* Pippenger */

/* ---- G2 ---- */
extern g2a_t G2_GENERATOR;
void g2_set_inf(g2_t *r);
void g2_from_affine(g2_t *r, const g2a_t *a);
void g2_to_affine(g2a_t *r, const g2_t *a);
void g2_dbl(g2_t *r, const g2_t *a);
void g2_add(g2_t *r, const g2_t *a, const g2_t *b);
void g2_neg(g2_t *r, const g2_t *a);
void g2_mul(g2_t *r, const g2_t *a, const fr_t *k);
int  g2a_is_on_curve(const g2a_t *a);
int  g2_decompress(g2a_t *r, const uint8_t b[96]); /* unchecked subgroup, like build.rs:73 */
void g2_compress(uint8_t b[96], const g2a_t *a);

/* ---- pairing ---- */
void miller_loop2(fp12_t *f, const g1a_t *p1, const g2a_t *q1, const g1a_t *p2, const g2a_t *q2);
void final_exponentiation(fp12_t *r, const fp12_t *f);
int  pairings_verify(const g1a_t *a1, const g2a_t *a2, const g1a_t *b1, const g2a_t *b2);

/* ---- sha256 ---- */
void sha256(uint8_t out[32], const uint8_t *data, size_t len);

#ifdef __cplusplus
}
#endif
#endif
