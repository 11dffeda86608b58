/*
 * oracle/fields.c - Fr (4x64) and Fp (6x64) Montgomery arithmetic for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle/bls.h).
 *
 * Semantics restated from the reference's call sites:
 *   Scalar::from_bytes (LE canonical, reject >= r)  <- src/kzg_proof.rs:36 (after BE->LE reversal :28-34)
 *   Scalar::from_raw   (value mod r)                <- src/kzg_proof.rs:90, build.rs:139
 *   Scalar::to_bytes   (little-endian canonical)    <- src/kzg_proof.rs:321,326
 *   Scalar add/sub/mul/invert/pow                   <- src/kzg_proof.rs:105-131,172-198
 */
#include "bls.h"
#include <string.h>

typedef unsigned __int128 u128;

const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                            0x73eda753299d7d48ULL};
const uint64_t FP_MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                            0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};

static uint64_t FR_INV, FP_INV;      /* -m^-1 mod 2^64 */
static fr_t FR_R, FR_R2;             /* R mod r, R^2 mod r */
static fp_t FP_R, FP_R2;
static uint64_t FR_M2[4];            /* r - 2 */
static uint64_t FP_M2[6];            /* p - 2 */
static uint64_t FP_SQRT_E[6];        /* (p + 1) / 4 */
static uint64_t FP_HALF[6];          /* (p - 1) / 2 */

/* ---------------------------------------------------------------- generic n-limb helpers */

static inline int ge_n(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}

static inline uint64_t add_n(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}

static inline uint64_t sub_n(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

static inline void mod_add_n(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6];
    uint64_t c = add_n(t, a, b, n);
    if (c || ge_n(t, m, n)) sub_n(t, t, m, n);
    memcpy(r, t, 8 * n);
}

static inline void mod_sub_n(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6];
    if (sub_n(t, a, b, n)) add_n(t, t, m, n);
    memcpy(r, t, 8 * n);
}

/* CIOS Montgomery multiplication: r = a * b * 2^(-64 n) mod m.  Requires a*b < m * 2^(64n). */
static inline void mont_mul_n(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m,
                              uint64_t inv, int n) {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[n];
        t[n] = (uint64_t)c;
        t[n + 1] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * inv;
        c = (u128)q * m[0] + t[0];
        c >>= 64;
        for (int j = 1; j < n; j++) {
            c += (u128)q * m[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[n];
        t[n - 1] = (uint64_t)c;
        t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    if (t[n] || ge_n(t, m, n)) sub_n(t, t, m, n);
    memcpy(r, t, 8 * n);
}

static uint64_t neg_inv64(uint64_t m0) {
    uint64_t x = 1; /* Newton: x <- x (2 - m0 x) */
    for (int i = 0; i < 6; i++) x *= 2 - m0 * x;
    return (uint64_t)0 - x;
}

/* 2^k mod m by repeated doubling, starting from 1 */
static void pow2_mod(uint64_t *r, int k, const uint64_t *m, int n) {
    uint64_t t[6] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < k; i++) mod_add_n(t, t, t, m, n);
    memcpy(r, t, 8 * n);
}

/* ---------------------------------------------------------------- Fr */

void fr_zero(fr_t *r) { memset(r, 0, sizeof *r); }
void fr_one(fr_t *r) { *r = FR_R; }
void fr_add(fr_t *r, const fr_t *a, const fr_t *b) { mod_add_n(r->l, a->l, b->l, FR_MOD, 4); }
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) { mod_sub_n(r->l, a->l, b->l, FR_MOD, 4); }
void fr_neg(fr_t *r, const fr_t *a) {
    fr_t z;
    fr_zero(&z);
    fr_sub(r, &z, a);
}
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b) { mont_mul_n(r->l, a->l, b->l, FR_MOD, FR_INV, 4); }
void fr_sqr(fr_t *r, const fr_t *a) { fr_mul(r, a, a); }
int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

void fr_from_raw_reduce(fr_t *r, const uint64_t le[4]) {
    /* any 256-bit value: v * R^2 * R^-1 = v R mod r (v * R2 < 2^256 * r holds) */
    fr_t t;
    memcpy(t.l, le, 32);
    fr_mul(r, &t, &FR_R2);
}

void fr_from_u64(fr_t *r, uint64_t v) {
    uint64_t le[4] = {v, 0, 0, 0};
    fr_from_raw_reduce(r, le);
}

static void be_to_limbs(uint64_t *le, const uint8_t *b, int n) {
    for (int i = 0; i < n; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (n - 1 - i) + k];
        le[i] = w;
    }
}

static void limbs_to_be(uint8_t *b, const uint64_t *le, int n) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) b[8 * (n - 1 - i) + k] = (uint8_t)(le[i] >> (56 - 8 * k));
}

int fr_from_be_canonical(fr_t *r, const uint8_t b[32]) {
    uint64_t le[4];
    be_to_limbs(le, b, 4);
    if (ge_n(le, FR_MOD, 4)) return -1;
    fr_from_raw_reduce(r, le);
    return 0;
}

void fr_from_be_reduce(fr_t *r, const uint8_t b[32]) {
    uint64_t le[4];
    be_to_limbs(le, b, 4);
    fr_from_raw_reduce(r, le);
}

void fr_to_raw(uint64_t le[4], const fr_t *a) {
    fr_t one = {{1, 0, 0, 0}}, t;
    fr_mul(&t, a, &one);
    memcpy(le, t.l, 32);
}

void fr_to_be(uint8_t b[32], const fr_t *a) {
    uint64_t le[4];
    fr_to_raw(le, a);
    limbs_to_be(b, le, 4);
}

void fr_to_le(uint8_t b[32], const fr_t *a) {
    uint64_t le[4];
    fr_to_raw(le, a);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(le[i] >> (8 * k));
}

void fr_pow(fr_t *r, const fr_t *a, const uint64_t *e, int nlimbs) {
    fr_t acc = FR_R, base = *a;
    for (int i = 0; i < 64 * nlimbs; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &base);
        fr_sqr(&base, &base);
    }
    *r = acc;
}

void fr_inv(fr_t *r, const fr_t *a) { fr_pow(r, a, FR_M2, 4); }

/* ---------------------------------------------------------------- Fp */

void fp_zero(fp_t *r) { memset(r, 0, sizeof *r); }
void fp_one(fp_t *r) { *r = FP_R; }
void fp_add(fp_t *r, const fp_t *a, const fp_t *b) { mod_add_n(r->l, a->l, b->l, FP_MOD, 6); }
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b) { mod_sub_n(r->l, a->l, b->l, FP_MOD, 6); }
void fp_neg(fp_t *r, const fp_t *a) {
    fp_t z;
    fp_zero(&z);
    fp_sub(r, &z, a);
}
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b) { mont_mul_n(r->l, a->l, b->l, FP_MOD, FP_INV, 6); }
void fp_sqr(fp_t *r, const fp_t *a) { fp_mul(r, a, a); }
int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fp_is_zero(const fp_t *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0;
}

void fp_from_raw(fp_t *r, const uint64_t le[6]) {
    fp_t t;
    memcpy(t.l, le, 48);
    fp_mul(r, &t, &FP_R2);
}

int fp_from_be_canonical(fp_t *r, const uint8_t b[48]) {
    uint64_t le[6];
    be_to_limbs(le, b, 6);
    if (ge_n(le, FP_MOD, 6)) return -1;
    fp_from_raw(r, le);
    return 0;
}

void fp_to_raw(uint64_t le[6], const fp_t *a) {
    fp_t one = {{1, 0, 0, 0, 0, 0}}, t;
    fp_mul(&t, a, &one);
    memcpy(le, t.l, 48);
}

void fp_to_be(uint8_t b[48], const fp_t *a) {
    uint64_t le[6];
    fp_to_raw(le, a);
    limbs_to_be(b, le, 6);
}

void fp_pow(fp_t *r, const fp_t *a, const uint64_t *e, int nlimbs) {
    fp_t acc = FP_R, base = *a;
    for (int i = 0; i < 64 * nlimbs; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
        fp_sqr(&base, &base);
    }
    *r = acc;
}

void fp_inv(fp_t *r, const fp_t *a) { fp_pow(r, a, FP_M2, 6); }

int fp_sqrt(fp_t *r, const fp_t *a) {
    /* p = 3 mod 4: candidate a^((p+1)/4) */
    fp_t c, c2;
    fp_pow(&c, a, FP_SQRT_E, 6);
    fp_sqr(&c2, &c);
    if (!fp_eq(&c2, a)) return -1;
    *r = c;
    return 0;
}

int fp_is_lex_largest(const fp_t *a) {
    uint64_t le[6];
    fp_to_raw(le, a);
    /* a > (p-1)/2 */
    for (int i = 5; i >= 0; i--) {
        if (le[i] > FP_HALF[i]) return 1;
        if (le[i] < FP_HALF[i]) return 0;
    }
    return 0;
}

/* ---------------------------------------------------------------- init */

void fields_init(void) {
    FR_INV = neg_inv64(FR_MOD[0]);
    FP_INV = neg_inv64(FP_MOD[0]);
    pow2_mod(FR_R.l, 256, FR_MOD, 4);
    pow2_mod(FR_R2.l, 512, FR_MOD, 4);
    pow2_mod(FP_R.l, 384, FP_MOD, 6);
    pow2_mod(FP_R2.l, 768, FP_MOD, 6);
    uint64_t two4[4] = {2, 0, 0, 0}, two6[6] = {2, 0, 0, 0, 0, 0}, one6[6] = {1, 0, 0, 0, 0, 0};
    sub_n(FR_M2, FR_MOD, two4, 4);
    sub_n(FP_M2, FP_MOD, two6, 6);
    /* (p+1)/4 : p+1 does not overflow 384 bits */
    uint64_t t[6];
    add_n(t, FP_MOD, one6, 6);
    for (int i = 0; i < 6; i++) FP_SQRT_E[i] = (t[i] >> 2) | (i < 5 ? t[i + 1] << 62 : 0);
    sub_n(t, FP_MOD, one6, 6);
    for (int i = 0; i < 6; i++) FP_HALF[i] = (t[i] >> 1) | (i < 5 ? t[i + 1] << 63 : 0);
}

void fields_constants(fr_t *r, fr_t *r2, uint64_t *ri, fp_t *p, fp_t *p2, uint64_t *pi) {
    *r = FR_R;
    *r2 = FR_R2;
    *ri = FR_INV;
    *p = FP_R;
    *p2 = FP_R2;
    *pi = FP_INV;
}
