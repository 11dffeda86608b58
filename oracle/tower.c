/*
 * oracle/tower.c - Fp2 / Fp6 / Fp12 for the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 * Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - xi), xi = 1 + u, Fp12 = Fp6[w]/(w^2 - v).
 * Used only under the reference's pairing call site (src/pairings.rs:5-9) and the G2
 * arithmetic at src/kzg_proof.rs:210-211,385-386.
 */
#include "bls.h"
#include <string.h>

/* ---------------------------------------------------------------- Fp2 */

void fp2_zero(fp2_t *r) { memset(r, 0, sizeof *r); }
void fp2_one(fp2_t *r) {
    fp_one(&r->c0);
    fp_zero(&r->c1);
}
void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_add(&r->c0, &a->c0, &b->c0);
    fp_add(&r->c1, &a->c1, &b->c1);
}
void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_sub(&r->c0, &a->c0, &b->c0);
    fp_sub(&r->c1, &a->c1, &b->c1);
}
void fp2_neg(fp2_t *r, const fp2_t *a) {
    fp_neg(&r->c0, &a->c0);
    fp_neg(&r->c1, &a->c1);
}
void fp2_conj(fp2_t *r, const fp2_t *a) {
    r->c0 = a->c0;
    fp_neg(&r->c1, &a->c1);
}
void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    /* Karatsuba: (a0 b0 - a1 b1) + ((a0+a1)(b0+b1) - a0b0 - a1b1) u */
    fp_t t0, t1, s0, s1, m;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1);
    fp_add(&s1, &b->c0, &b->c1);
    fp_mul(&m, &s0, &s1);
    fp_sub(&r->c0, &t0, &t1);
    fp_sub(&m, &m, &t0);
    fp_sub(&r->c1, &m, &t1);
}
void fp2_sqr(fp2_t *r, const fp2_t *a) {
    /* (a0+a1)(a0-a1) + 2 a0 a1 u */
    fp_t s, d, m;
    fp_add(&s, &a->c0, &a->c1);
    fp_sub(&d, &a->c0, &a->c1);
    fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&r->c0, &s, &d);
    fp_add(&r->c1, &m, &m);
}
void fp2_mul_fp(fp2_t *r, const fp2_t *a, const fp_t *k) {
    fp_mul(&r->c0, &a->c0, k);
    fp_mul(&r->c1, &a->c1, k);
}
void fp2_mul_xi(fp2_t *r, const fp2_t *a) {
    /* (a0 + a1 u)(1 + u) = (a0 - a1) + (a0 + a1) u */
    fp_t t0, t1;
    fp_sub(&t0, &a->c0, &a->c1);
    fp_add(&t1, &a->c0, &a->c1);
    r->c0 = t0;
    r->c1 = t1;
}
void fp2_inv(fp2_t *r, const fp2_t *a) {
    fp_t n, t;
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n);
    fp_mul(&t, &a->c1, &n);
    fp_neg(&r->c1, &t);
}
int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }

void fp2_pow(fp2_t *r, const fp2_t *a, const uint64_t *e, int nlimbs) {
    fp2_t acc, base = *a;
    fp2_one(&acc);
    for (int i = 0; i < 64 * nlimbs; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, &base);
        fp2_sqr(&base, &base);
    }
    *r = acc;
}

int fp2_sqrt(fp2_t *r, const fp2_t *a) {
    /* a = (x + y u)^2  =>  x^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2,  y = a1 / (2x) */
    if (fp2_is_zero(a)) {
        fp2_zero(r);
        return 0;
    }
    fp_t n, t, s, two, inv2, x2, x, y;
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    if (fp_sqrt(&s, &n)) return -1;
    fp_one(&two);
    fp_add(&two, &two, &two);
    fp_inv(&inv2, &two);
    for (int k = 0; k < 2; k++) {
        if (k == 0) fp_add(&x2, &a->c0, &s);
        else fp_sub(&x2, &a->c0, &s);
        fp_mul(&x2, &x2, &inv2);
        if (fp_sqrt(&x, &x2) || fp_is_zero(&x)) continue;
        fp_add(&t, &x, &x);
        fp_inv(&t, &t);
        fp_mul(&y, &a->c1, &t);
        fp2_t c = {x, y}, c2;
        fp2_sqr(&c2, &c);
        if (fp2_eq(&c2, a)) {
            *r = c;
            return 0;
        }
    }
    return -1;
}

/* ---------------------------------------------------------------- Fp6 */

static void fp6_add(fp6_t *r, const fp6_t *a, const fp6_t *b) {
    fp2_add(&r->c0, &a->c0, &b->c0);
    fp2_add(&r->c1, &a->c1, &b->c1);
    fp2_add(&r->c2, &a->c2, &b->c2);
}
static void fp6_sub(fp6_t *r, const fp6_t *a, const fp6_t *b) {
    fp2_sub(&r->c0, &a->c0, &b->c0);
    fp2_sub(&r->c1, &a->c1, &b->c1);
    fp2_sub(&r->c2, &a->c2, &b->c2);
}
static void fp6_neg(fp6_t *r, const fp6_t *a) {
    fp2_neg(&r->c0, &a->c0);
    fp2_neg(&r->c1, &a->c1);
    fp2_neg(&r->c2, &a->c2);
}
static void fp6_mul_v(fp6_t *r, const fp6_t *a) {
    /* (c0 + c1 v + c2 v^2) v = xi c2 + c0 v + c1 v^2 */
    fp2_t t;
    fp2_mul_xi(&t, &a->c2);
    r->c2 = a->c1;
    r->c1 = a->c0;
    r->c0 = t;
}
void fp6_mul(fp6_t *r, const fp6_t *a, const fp6_t *b) {
    fp2_t v0, v1, v2, s, t, u, c0, c1, c2;
    fp2_mul(&v0, &a->c0, &b->c0);
    fp2_mul(&v1, &a->c1, &b->c1);
    fp2_mul(&v2, &a->c2, &b->c2);
    /* c0 = v0 + xi((a1+a2)(b1+b2) - v1 - v2) */
    fp2_add(&s, &a->c1, &a->c2);
    fp2_add(&t, &b->c1, &b->c2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v1);
    fp2_sub(&u, &u, &v2);
    fp2_mul_xi(&u, &u);
    fp2_add(&c0, &u, &v0);
    /* c1 = (a0+a1)(b0+b1) - v0 - v1 + xi v2 */
    fp2_add(&s, &a->c0, &a->c1);
    fp2_add(&t, &b->c0, &b->c1);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v1);
    fp2_mul_xi(&t, &v2);
    fp2_add(&c1, &u, &t);
    /* c2 = (a0+a2)(b0+b2) - v0 - v2 + v1 */
    fp2_add(&s, &a->c0, &a->c2);
    fp2_add(&t, &b->c0, &b->c2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v2);
    fp2_add(&c2, &u, &v1);
    r->c0 = c0;
    r->c1 = c1;
    r->c2 = c2;
}
void fp6_inv(fp6_t *r, const fp6_t *a) {
    fp2_t t0, t1, t2, m, d;
    /* t0 = c0^2 - xi c1 c2 ; t1 = xi c2^2 - c0 c1 ; t2 = c1^2 - c0 c2 */
    fp2_sqr(&t0, &a->c0);
    fp2_mul(&m, &a->c1, &a->c2);
    fp2_mul_xi(&m, &m);
    fp2_sub(&t0, &t0, &m);
    fp2_sqr(&t1, &a->c2);
    fp2_mul_xi(&t1, &t1);
    fp2_mul(&m, &a->c0, &a->c1);
    fp2_sub(&t1, &t1, &m);
    fp2_sqr(&t2, &a->c1);
    fp2_mul(&m, &a->c0, &a->c2);
    fp2_sub(&t2, &t2, &m);
    /* d = c0 t0 + xi (c2 t1 + c1 t2) */
    fp2_mul(&d, &a->c2, &t1);
    fp2_mul(&m, &a->c1, &t2);
    fp2_add(&d, &d, &m);
    fp2_mul_xi(&d, &d);
    fp2_mul(&m, &a->c0, &t0);
    fp2_add(&d, &d, &m);
    fp2_inv(&d, &d);
    fp2_mul(&r->c0, &t0, &d);
    fp2_mul(&r->c1, &t1, &d);
    fp2_mul(&r->c2, &t2, &d);
}

/* ---------------------------------------------------------------- Fp12 */

static fp2_t FROB_GAMMA[6]; /* xi^(k (p-1)/6), k = 0..5, for w^k */

void fp12_one(fp12_t *r) {
    memset(r, 0, sizeof *r);
    fp_one(&r->c0.c0.c0);
}
void fp12_mul(fp12_t *r, const fp12_t *a, const fp12_t *b) {
    fp6_t t0, t1, s0, s1, m, v;
    fp6_mul(&t0, &a->c0, &b->c0);
    fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s0, &a->c0, &a->c1);
    fp6_add(&s1, &b->c0, &b->c1);
    fp6_mul(&m, &s0, &s1);
    fp6_sub(&m, &m, &t0);
    fp6_sub(&m, &m, &t1);
    fp6_mul_v(&v, &t1);
    fp6_add(&r->c0, &t0, &v);
    r->c1 = m;
}
void fp12_sqr(fp12_t *r, const fp12_t *a) {
    /* c0 = (a0 + a1)(a0 + v a1) - a0a1 - v a0a1 ; c1 = 2 a0a1 */
    fp6_t ab, s, t, va, m;
    fp6_mul(&ab, &a->c0, &a->c1);
    fp6_add(&s, &a->c0, &a->c1);
    fp6_mul_v(&va, &a->c1);
    fp6_add(&t, &a->c0, &va);
    fp6_mul(&m, &s, &t);
    fp6_sub(&m, &m, &ab);
    fp6_mul_v(&va, &ab);
    fp6_sub(&r->c0, &m, &va);
    fp6_add(&r->c1, &ab, &ab);
}
void fp12_conj(fp12_t *r, const fp12_t *a) {
    r->c0 = a->c0;
    fp6_neg(&r->c1, &a->c1);
}
void fp12_inv(fp12_t *r, const fp12_t *a) {
    /* 1/(a0 + a1 w) = (a0 - a1 w) / (a0^2 - v a1^2) */
    fp6_t t0, t1, d;
    fp6_mul(&t0, &a->c0, &a->c0);
    fp6_mul(&t1, &a->c1, &a->c1);
    fp6_mul_v(&t1, &t1);
    fp6_sub(&d, &t0, &t1);
    fp6_inv(&d, &d);
    fp6_mul(&r->c0, &a->c0, &d);
    fp6_mul(&t0, &a->c1, &d);
    fp6_neg(&r->c1, &t0);
}
void fp12_frobenius(fp12_t *r, const fp12_t *a) {
    /* sum c_k w^k -> sum conj(c_k) gamma_k w^k ; tower (c_i . c_j) is w^(2j + i) */
    fp2_t t;
    const fp2_t *src[6] = {&a->c0.c0, &a->c1.c0, &a->c0.c1, &a->c1.c1, &a->c0.c2, &a->c1.c2};
    fp2_t *dst[6] = {&r->c0.c0, &r->c1.c0, &r->c0.c1, &r->c1.c1, &r->c0.c2, &r->c1.c2};
    for (int k = 0; k < 6; k++) {
        fp2_conj(&t, src[k]);
        fp2_mul(dst[k], &t, &FROB_GAMMA[k]);
    }
}
int fp12_eq(const fp12_t *a, const fp12_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fp12_is_one(const fp12_t *a) {
    fp12_t one;
    fp12_one(&one);
    return fp12_eq(a, &one);
}

void tower_init(void) {
    /* e = (p - 1) / 6 by long division of the limbs */
    uint64_t e[6], t[6];
    memcpy(t, FP_MOD, 48);
    t[0] -= 1;
    unsigned __int128 rem = 0;
    for (int i = 5; i >= 0; i--) {
        unsigned __int128 cur = (rem << 64) | t[i];
        e[i] = (uint64_t)(cur / 6);
        rem = cur % 6;
    }
    fp2_t xi, g;
    fp_one(&xi.c0);
    fp_one(&xi.c1);
    fp2_pow(&g, &xi, e, 6);
    fp2_one(&FROB_GAMMA[0]);
    for (int k = 1; k < 6; k++) fp2_mul(&FROB_GAMMA[k], &FROB_GAMMA[k - 1], &g);
}
