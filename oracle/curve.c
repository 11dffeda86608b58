/*
 * oracle/curve.c - G1 / G2 group law, (de)compression, Pippenger MSM for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle/bls.h).
 *
 * Restates the semantics the reference obtains from sp1_bls12_381 (absent) at:
 *   G1Affine::from_compressed       src/kzg_proof.rs:18   (flags, x < p, sqrt, sign, subgroup)
 *   G1Affine::to_compressed         src/kzg_proof.rs:61,316,331
 *   G1/G2 add, sub, scalar mul      src/kzg_proof.rs:210-214,385-389,423-424,433
 *   G1Projective::msm_variable_base src/kzg_proof.rs:419,429,430
 *   G2Affine::from_compressed_unchecked  build.rs:73
 * Encoding: ZCash/IETF BLS12-381 compressed points.
 */
#include "bls.h"
#include <stdlib.h>
#include <string.h>

g1a_t G1_GENERATOR;
g2a_t G2_GENERATOR;
static fp_t FP_B1;  /* 4 */
static fp2_t FP2_B2; /* 4 (1 + u) */

/* ---------------------------------------------------------------- generic Jacobian group law */

#define CURVE_IMPL(G, GA, F, f)                                                                      \
    void G##_set_inf(G##_t *r) {                                                                     \
        f##_one(&r->x);                                                                              \
        f##_one(&r->y);                                                                              \
        f##_zero(&r->z);                                                                             \
    }                                                                                                \
    int G##_is_inf_(const G##_t *a) { return f##_is_zero(&a->z); }                                   \
    void G##_from_affine(G##_t *r, const GA##_t *a) {                                                \
        if (a->inf) {                                                                                \
            G##_set_inf(r);                                                                          \
            return;                                                                                  \
        }                                                                                            \
        r->x = a->x;                                                                                 \
        r->y = a->y;                                                                                 \
        f##_one(&r->z);                                                                              \
    }                                                                                                \
    void G##_to_affine(GA##_t *r, const G##_t *a) {                                                  \
        if (f##_is_zero(&a->z)) {                                                                    \
            memset(r, 0, sizeof *r);                                                                 \
            r->inf = 1;                                                                              \
            return;                                                                                  \
        }                                                                                            \
        F zi, zi2, zi3;                                                                              \
        f##_inv(&zi, &a->z);                                                                         \
        f##_sqr(&zi2, &zi);                                                                          \
        f##_mul(&zi3, &zi2, &zi);                                                                    \
        f##_mul(&r->x, &a->x, &zi2);                                                                 \
        f##_mul(&r->y, &a->y, &zi3);                                                                 \
        r->inf = 0;                                                                                  \
    }                                                                                                \
    void G##_dbl(G##_t *r, const G##_t *p) {                                                         \
        F A, B, C, D, E, Fq, t, X3, Y3, Z3;                                                          \
        f##_sqr(&A, &p->x);                                                                          \
        f##_sqr(&B, &p->y);                                                                          \
        f##_sqr(&C, &B);                                                                             \
        f##_add(&t, &p->x, &B);                                                                      \
        f##_sqr(&t, &t);                                                                             \
        f##_sub(&t, &t, &A);                                                                         \
        f##_sub(&t, &t, &C);                                                                         \
        f##_add(&D, &t, &t);                                                                         \
        f##_add(&E, &A, &A);                                                                         \
        f##_add(&E, &E, &A);                                                                         \
        f##_sqr(&Fq, &E);                                                                            \
        f##_sub(&X3, &Fq, &D);                                                                       \
        f##_sub(&X3, &X3, &D);                                                                       \
        f##_sub(&t, &D, &X3);                                                                        \
        f##_mul(&Y3, &E, &t);                                                                        \
        f##_add(&C, &C, &C);                                                                         \
        f##_add(&C, &C, &C);                                                                         \
        f##_add(&C, &C, &C);                                                                         \
        f##_sub(&Y3, &Y3, &C);                                                                       \
        f##_mul(&Z3, &p->y, &p->z);                                                                  \
        f##_add(&Z3, &Z3, &Z3);                                                                      \
        r->x = X3;                                                                                   \
        r->y = Y3;                                                                                   \
        r->z = Z3;                                                                                   \
    }                                                                                                \
    void G##_add(G##_t *r, const G##_t *p, const G##_t *q) {                                         \
        if (f##_is_zero(&p->z)) {                                                                    \
            *r = *q;                                                                                 \
            return;                                                                                  \
        }                                                                                            \
        if (f##_is_zero(&q->z)) {                                                                    \
            *r = *p;                                                                                 \
            return;                                                                                  \
        }                                                                                            \
        F Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;                              \
        f##_sqr(&Z1Z1, &p->z);                                                                       \
        f##_sqr(&Z2Z2, &q->z);                                                                       \
        f##_mul(&U1, &p->x, &Z2Z2);                                                                  \
        f##_mul(&U2, &q->x, &Z1Z1);                                                                  \
        f##_mul(&S1, &p->y, &q->z);                                                                  \
        f##_mul(&S1, &S1, &Z2Z2);                                                                    \
        f##_mul(&S2, &q->y, &p->z);                                                                  \
        f##_mul(&S2, &S2, &Z1Z1);                                                                    \
        if (f##_eq(&U1, &U2)) {                                                                      \
            if (f##_eq(&S1, &S2)) G##_dbl(r, p);                                                     \
            else G##_set_inf(r);                                                                     \
            return;                                                                                  \
        }                                                                                            \
        f##_sub(&H, &U2, &U1);                                                                       \
        f##_sub(&Rr, &S2, &S1);                                                                      \
        f##_sqr(&HH, &H);                                                                            \
        f##_mul(&HHH, &H, &HH);                                                                      \
        f##_mul(&V, &U1, &HH);                                                                       \
        f##_sqr(&X3, &Rr);                                                                           \
        f##_sub(&X3, &X3, &HHH);                                                                     \
        f##_sub(&X3, &X3, &V);                                                                       \
        f##_sub(&X3, &X3, &V);                                                                       \
        f##_sub(&t, &V, &X3);                                                                        \
        f##_mul(&Y3, &Rr, &t);                                                                       \
        f##_mul(&t, &S1, &HHH);                                                                      \
        f##_sub(&Y3, &Y3, &t);                                                                       \
        f##_mul(&Z3, &p->z, &q->z);                                                                  \
        f##_mul(&Z3, &Z3, &H);                                                                       \
        r->x = X3;                                                                                   \
        r->y = Y3;                                                                                   \
        r->z = Z3;                                                                                   \
    }                                                                                                \
    void G##_add_affine(G##_t *r, const G##_t *p, const GA##_t *q) {                                 \
        G##_t qq;                                                                                    \
        G##_from_affine(&qq, q);                                                                     \
        G##_add(r, p, &qq);                                                                          \
    }                                                                                                \
    void G##_neg(G##_t *r, const G##_t *a) {                                                         \
        r->x = a->x;                                                                                 \
        f##_neg(&r->y, &a->y);                                                                       \
        r->z = a->z;                                                                                 \
    }                                                                                                \
    void G##_mul_raw(G##_t *r, const G##_t *a, const uint64_t *e, int nlimbs) {                      \
        G##_t acc;                                                                                   \
        G##_set_inf(&acc);                                                                           \
        for (int i = 64 * nlimbs - 1; i >= 0; i--) {                                                 \
            G##_dbl(&acc, &acc);                                                                     \
            if ((e[i / 64] >> (i % 64)) & 1) G##_add(&acc, &acc, a);                                 \
        }                                                                                            \
        *r = acc;                                                                                    \
    }                                                                                                \
    void G##_mul(G##_t *r, const G##_t *a, const fr_t *k) {                                          \
        uint64_t e[4];                                                                               \
        fr_to_raw(e, k);                                                                             \
        G##_mul_raw(r, a, e, 4);                                                                     \
    }

CURVE_IMPL(g1, g1a, fp_t, fp)
CURVE_IMPL(g2, g2a, fp2_t, fp2)

int g1_is_inf(const g1_t *a) { return fp_is_zero(&a->z); }

void g1a_neg(g1a_t *r, const g1a_t *a) {
    *r = *a;
    if (!a->inf) fp_neg(&r->y, &a->y);
}

int g1a_is_on_curve(const g1a_t *a) {
    if (a->inf) return 1;
    fp_t l, rr;
    fp_sqr(&l, &a->y);
    fp_sqr(&rr, &a->x);
    fp_mul(&rr, &rr, &a->x);
    fp_add(&rr, &rr, &FP_B1);
    return fp_eq(&l, &rr);
}

int g2a_is_on_curve(const g2a_t *a) {
    if (a->inf) return 1;
    fp2_t l, rr;
    fp2_sqr(&l, &a->y);
    fp2_sqr(&rr, &a->x);
    fp2_mul(&rr, &rr, &a->x);
    fp2_add(&rr, &rr, &FP2_B2);
    return fp2_eq(&l, &rr);
}

int g1a_in_subgroup(const g1a_t *a) {
    /* the definition: [r]P == O  (reproduces every vector outcome) */
    g1_t p, q;
    g1_from_affine(&p, a);
    g1_mul_raw(&q, &p, FR_MOD, 4);
    return g1_is_inf(&q);
}

/* ---------------------------------------------------------------- compression */

int g1_decompress(g1a_t *r, const uint8_t b[48], int check_subgroup) {
    int c_flag = (b[0] >> 7) & 1, i_flag = (b[0] >> 6) & 1, s_flag = (b[0] >> 5) & 1;
    uint8_t xb[48];
    memcpy(xb, b, 48);
    xb[0] &= 0x1f;
    if (!c_flag) return -1;
    if (i_flag) {
        if (s_flag) return -1;
        for (int i = 0; i < 48; i++)
            if (xb[i]) return -1;
        memset(r, 0, sizeof *r);
        r->inf = 1;
        return 0;
    }
    fp_t x, y2, y;
    if (fp_from_be_canonical(&x, xb)) return -1;
    fp_sqr(&y2, &x);
    fp_mul(&y2, &y2, &x);
    fp_add(&y2, &y2, &FP_B1);
    if (fp_sqrt(&y, &y2)) return -1;
    if (fp_is_lex_largest(&y) != s_flag) fp_neg(&y, &y);
    r->x = x;
    r->y = y;
    r->inf = 0;
    if (check_subgroup && !g1a_in_subgroup(r)) return -1;
    return 0;
}

void g1_compress(uint8_t b[48], const g1a_t *a) {
    if (a->inf) {
        memset(b, 0, 48);
        b[0] = 0xc0;
        return;
    }
    fp_to_be(b, &a->x);
    b[0] |= 0x80;
    if (fp_is_lex_largest(&a->y)) b[0] |= 0x20;
}

static int fp2_is_lex_largest(const fp2_t *y) {
    if (!fp_is_zero(&y->c1)) return fp_is_lex_largest(&y->c1);
    return fp_is_lex_largest(&y->c0);
}

int g2_decompress(g2a_t *r, const uint8_t b[96]) {
    int c_flag = (b[0] >> 7) & 1, i_flag = (b[0] >> 6) & 1, s_flag = (b[0] >> 5) & 1;
    uint8_t xb[96];
    memcpy(xb, b, 96);
    xb[0] &= 0x1f;
    if (!c_flag) return -1;
    if (i_flag) {
        if (s_flag) return -1;
        for (int i = 0; i < 96; i++)
            if (xb[i]) return -1;
        memset(r, 0, sizeof *r);
        r->inf = 1;
        return 0;
    }
    fp2_t x, y2, y;
    if (fp_from_be_canonical(&x.c1, xb)) return -1;      /* x.c1 first */
    if (fp_from_be_canonical(&x.c0, xb + 48)) return -1; /* then x.c0  */
    fp2_sqr(&y2, &x);
    fp2_mul(&y2, &y2, &x);
    fp2_add(&y2, &y2, &FP2_B2);
    if (fp2_sqrt(&y, &y2)) return -1;
    if (fp2_is_lex_largest(&y) != s_flag) fp2_neg(&y, &y);
    r->x = x;
    r->y = y;
    r->inf = 0;
    return 0;
}

void g2_compress(uint8_t b[96], const g2a_t *a) {
    if (a->inf) {
        memset(b, 0, 96);
        b[0] = 0xc0;
        return;
    }
    fp_to_be(b, &a->x.c1);
    fp_to_be(b + 48, &a->x.c0);
    b[0] |= 0x80;
    if (fp2_is_lex_largest(&a->y)) b[0] |= 0x20;
}

/* ---------------------------------------------------------------- Pippenger MSM (bucket method) */

void g1_msm(g1_t *r, const g1a_t *pts, const fr_t *scalars, size_t n) {
    g1_t acc;
    g1_set_inf(&acc);
    if (n == 0) {
        *r = acc;
        return;
    }
    int c = 1;
    while ((1ULL << (c + 3)) < n && c < 16) c++; /* ~ log2(n) - 3 */
    if (n < 8) c = 2;
    size_t nb = ((size_t)1 << c) - 1;
    g1_t *buckets = (g1_t *)malloc(nb * sizeof(g1_t));
    uint64_t (*raw)[4] = (uint64_t (*)[4])malloc(n * 32);
    for (size_t i = 0; i < n; i++) fr_to_raw(raw[i], &scalars[i]);
    int windows = (255 + c - 1) / c;
    for (int w = windows - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) g1_dbl(&acc, &acc);
        for (size_t b = 0; b < nb; b++) g1_set_inf(&buckets[b]);
        int lo = w * c;
        for (size_t i = 0; i < n; i++) {
            if (pts[i].inf) continue;
            uint64_t d = raw[i][lo / 64] >> (lo % 64);
            if (lo % 64 + c > 64 && lo / 64 + 1 < 4) d |= raw[i][lo / 64 + 1] << (64 - lo % 64);
            d &= nb;
            if (d) g1_add_affine(&buckets[d - 1], &buckets[d - 1], &pts[i]);
        }
        g1_t run, sum;
        g1_set_inf(&run);
        g1_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) {
            g1_add(&run, &run, &buckets[b]);
            g1_add(&sum, &sum, &run);
        }
        g1_add(&acc, &acc, &sum);
    }
    free(buckets);
    free(raw);
    *r = acc;
}

/* ---------------------------------------------------------------- init */

static uint8_t hexval(char ch) { return ch <= '9' ? ch - '0' : ch - 'a' + 10; }
static void unhex(uint8_t *out, const char *s, int n) {
    for (int i = 0; i < n; i++) out[i] = (uint8_t)(hexval(s[2 * i]) << 4 | hexval(s[2 * i + 1]));
}

void curve_init(void) {
    fp_t one;
    fp_one(&one);
    fp_add(&FP_B1, &one, &one);
    fp_add(&FP_B1, &FP_B1, &FP_B1);
    FP2_B2.c0 = FP_B1;
    FP2_B2.c1 = FP_B1;
    /* standard generators, given by their compressed encodings (the G2 one is
     * also line 4099 of the trusted setup, checked by tests) */
    uint8_t b[96];
    unhex(b, "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 48);
    g1_decompress(&G1_GENERATOR, b, 0);
    unhex(b,
          "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
          "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8",
          96);
    g2_decompress(&G2_GENERATOR, b);
}
