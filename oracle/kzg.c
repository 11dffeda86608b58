/*
 * oracle/kzg.c - the KZG verification protocol, restated step for step from
 * /root/reference/src/kzg_proof.rs (file:line cited on each function).
 * TEST INFRASTRUCTURE ONLY (see oracle/kzg_oracle.h).
 */
#include "kzg_oracle.h"
#include "bls.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define N_FE 4096
#define BLOB_BYTES 131072

void fields_init(void);
void tower_init(void);
void curve_init(void);
void pairing_init(void);
void fields_constants(fr_t *r, fr_t *r2, uint64_t *ri, fp_t *p, fp_t *p2, uint64_t *pi);

static pthread_once_t once = PTHREAD_ONCE_INIT;
static void init_all(void) {
    fields_init();
    tower_init();
    curve_init();
    pairing_init();
    uint8_t d[32];
    sha256(d, (const uint8_t *)"", 0);
}
void bls_init(void) { pthread_once(&once, init_all); }

struct oracle_settings {
    fr_t roots[N_FE]; /* bit-reversed order: build.rs:131-144 */
    g2a_t g2[2];      /* g2_points[0] (generator), g2_points[1] = [tau]G2 */
    g1a_t *g1;        /* optional, bit-reversal permuted: build.rs:79 */
};

/* src/consts.rs:90-95: SCALE2_ROOT_OF_UNITY[12], little-endian limbs */
static const uint64_t OMEGA_4096[4] = {0xe206da11a5d36306ULL, 0x0ad1347b378fbf96ULL, 0xfc3e8acfe0f8245fULL,
                                       0x564c0a11a0f704f4ULL};

static unsigned bitrev12(unsigned i) {
    unsigned r = 0;
    for (int k = 0; k < 12; k++) r |= ((i >> k) & 1) << (11 - k);
    return r;
}

/* build.rs:131-170 */
static int compute_roots(fr_t *roots) {
    fr_t w, cur, one;
    fr_from_raw_reduce(&w, OMEGA_4096);
    fr_one(&one);
    cur = one;
    for (unsigned i = 0; i < N_FE; i++) {
        roots[bitrev12(i)] = cur;
        fr_mul(&cur, &cur, &w);
    }
    return fr_eq(&cur, &one) ? 0 : -1; /* expand_root_of_unity's "last element must be 1" */
}

static int hexnib(int ch) {
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
    if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
    return -1;
}

static const char *next_line(const char *p, const char *end, const char **ls, size_t *ll) {
    if (p >= end) return NULL;
    const char *q = p;
    while (q < end && *q != '\n') q++;
    *ls = p;
    *ll = (size_t)(q - p);
    if (*ll && p[*ll - 1] == '\r') (*ll)--;
    return q < end ? q + 1 : end;
}

static int parse_hex_line(uint8_t *out, size_t nbytes, const char *s, size_t len) {
    if (len != 2 * nbytes) return -1;
    for (size_t i = 0; i < nbytes; i++) {
        int a = hexnib(s[2 * i]), b = hexnib(s[2 * i + 1]);
        if (a < 0 || b < 0) return -1;
        out[i] = (uint8_t)(a << 4 | b);
    }
    return 0;
}

oracle_settings *oracle_settings_load_txt(const char *txt, size_t len, int load_g1) {
    bls_init();
    const char *p = txt, *end = txt + len, *ls;
    size_t ll;
    if (!(p = next_line(p, end, &ls, &ll))) return NULL;
    long n1 = strtol(ls, NULL, 10);
    if (!(p = next_line(p, end, &ls, &ll))) return NULL;
    long n2 = strtol(ls, NULL, 10);
    if (n1 != N_FE || n2 < 2) return NULL;
    oracle_settings *s = (oracle_settings *)calloc(1, sizeof *s);
    if (compute_roots(s->roots)) goto fail;
    if (load_g1) s->g1 = (g1a_t *)calloc(N_FE, sizeof(g1a_t));
    for (long i = 0; i < n1; i++) {
        if (!(p = next_line(p, end, &ls, &ll))) goto fail;
        if (load_g1) {
            uint8_t b[48];
            if (parse_hex_line(b, 48, ls, ll)) goto fail;
            if (g1_decompress(&s->g1[bitrev12((unsigned)i)], b, 0)) goto fail; /* unchecked: build.rs:68 */
        }
    }
    for (long i = 0; i < 2; i++) {
        uint8_t b[96];
        if (!(p = next_line(p, end, &ls, &ll))) goto fail;
        if (parse_hex_line(b, 96, ls, ll)) goto fail;
        if (g2_decompress(&s->g2[i], b)) goto fail;
    }
    return s;
fail:
    oracle_settings_free(s);
    return NULL;
}

oracle_settings *oracle_settings_from_tau_g2(const uint8_t tau_g2[96]) {
    bls_init();
    oracle_settings *s = (oracle_settings *)calloc(1, sizeof *s);
    if (compute_roots(s->roots) || g2_decompress(&s->g2[1], tau_g2)) {
        oracle_settings_free(s);
        return NULL;
    }
    s->g2[0] = G2_GENERATOR;
    return s;
}

void oracle_settings_free(oracle_settings *s) {
    if (!s) return;
    free(s->g1);
    free(s);
}

void oracle_settings_root(const oracle_settings *s, size_t i, uint8_t out_be[32]) { fr_to_be(out_be, &s->roots[i]); }
int oracle_settings_g1(const oracle_settings *s, size_t i, uint8_t out[48]) {
    if (!s->g1 || i >= N_FE) return ORACLE_BADARGS;
    g1_compress(out, &s->g1[i]);
    return ORACLE_OK;
}
void oracle_settings_g2(const oracle_settings *s, int i, uint8_t out[96]) { g2_compress(out, &s->g2[i & 1]); }

/* ---------------------------------------------------------------- protocol pieces */

/* src/kzg_proof.rs:17-25 */
static int safe_g1_affine_from_bytes(g1a_t *r, const uint8_t b[48]) {
    return g1_decompress(r, b, 1) ? ORACLE_BADARGS : ORACLE_OK;
}

/* src/kzg_proof.rs:27-43 */
static int safe_scalar_affine_from_bytes(fr_t *r, const uint8_t b[32]) {
    return fr_from_be_canonical(r, b) ? ORACLE_BADARGS : ORACLE_OK;
}

/* src/dtypes.rs:48-57 */
static int blob_as_polynomial(fr_t *poly, const uint8_t *blob) {
    for (int i = 0; i < N_FE; i++)
        if (safe_scalar_affine_from_bytes(&poly[i], blob + 32 * i)) return ORACLE_BADARGS;
    return ORACLE_OK;
}

/* Per-thread scratch of the per-blob loop.  Rounds 1-4 malloc'ed and freed the big temporaries (the 131 KB hash input,
 * the 256 KB inversion arrays, the 128 KB parsed polynomial) for every blob: above glibc's mmap threshold that is an mmap + munmap + ~100 page faults per
 * blob, and T threads of one process doing it serialise on the process's address-space lock - the reason the 64-thread CPU
 * baseline ran at 104 blobs/s per thread against 489 alone.  One allocation per thread, freed when the thread
 * ends. */
typedef struct {
    uint8_t hash_in[16 + 16 + BLOB_BYTES + 48];
    fr_t inv[2 * N_FE];
    fr_t poly[N_FE];
} scratch_t;
static pthread_key_t scratch_key;
static pthread_once_t scratch_once = PTHREAD_ONCE_INIT;
static void scratch_make_key(void) { pthread_key_create(&scratch_key, free); }
static scratch_t *scratch(void) {
    pthread_once(&scratch_once, scratch_make_key);
    scratch_t *p = (scratch_t *)pthread_getspecific(scratch_key);
    if (!p) {
        p = (scratch_t *)malloc(sizeof *p);
        pthread_setspecific(scratch_key, p);
    }
    return p;
}

/* src/kzg_proof.rs:46-72 and :74-91 (digest as a big-endian integer, reduced mod r by from_raw) */
static void compute_challenge(fr_t *z, const uint8_t *blob, const uint8_t commitment[48]) {
    uint8_t *t = scratch()->hash_in;
    memcpy(t, "FSBLOBVERIFY_V1_", 16);
    memset(t + 16, 0, 16);
    t[16 + 14] = (uint8_t)(N_FE >> 8); /* u64_be(0) || u64_be(4096) */
    t[16 + 15] = (uint8_t)(N_FE & 0xff);
    memcpy(t + 32, blob, BLOB_BYTES);
    memcpy(t + 32 + BLOB_BYTES, commitment, 48);
    uint8_t d[32];
    sha256(d, t, 32 + BLOB_BYTES + 48);
    fr_from_be_reduce(z, d);
}

/* src/kzg_proof.rs:155-201 */
static int batch_inversion(fr_t *out, const fr_t *a, size_t len) {
    fr_t acc;
    fr_one(&acc);
    for (size_t i = 0; i < len; i++) {
        out[i] = acc;
        fr_mul(&acc, &acc, &a[i]);
    }
    if (fr_is_zero(&acc)) return ORACLE_BADARGS;
    fr_inv(&acc, &acc);
    for (size_t i = len; i-- > 0;) {
        fr_mul(&out[i], &out[i], &acc);
        fr_mul(&acc, &acc, &a[i]);
    }
    return ORACLE_OK;
}

/* src/kzg_proof.rs:94-133 */
static int evaluate_polynomial_in_evaluation_form(fr_t *y, const fr_t *poly, const fr_t *x, const oracle_settings *s) {
    fr_t *inv_in = scratch()->inv, *inv = inv_in + N_FE;
    int rc = ORACLE_OK;
    for (int i = 0; i < N_FE; i++) {
        if (fr_eq(x, &s->roots[i])) {
            *y = poly[i];
            goto done;
        }
        fr_sub(&inv_in[i], x, &s->roots[i]);
    }
    if ((rc = batch_inversion(inv, inv_in, N_FE))) goto done;
    fr_t out, t;
    fr_zero(&out);
    for (int i = 0; i < N_FE; i++) {
        fr_mul(&t, &inv[i], &s->roots[i]);
        fr_mul(&t, &t, &poly[i]);
        fr_add(&out, &out, &t);
    }
    fr_from_u64(&t, N_FE);
    fr_inv(&t, &t);
    fr_mul(&out, &out, &t);
    const uint64_t e[4] = {N_FE, 0, 0, 0};
    fr_pow(&t, x, e, 4);
    fr_t one;
    fr_one(&one);
    fr_sub(&t, &t, &one);
    fr_mul(y, &out, &t);
done:
    return rc;
}

/* src/kzg_proof.rs:203-223 and the inlined copy at :385-396 */
static int verify_kzg_proof_impl(const g1a_t *commitment, const fr_t *z, const fr_t *y, const g1a_t *proof,
                                 const oracle_settings *s) {
    g2_t g2gen, xz, tau;
    g2a_t x_minus_z;
    g2_from_affine(&g2gen, &G2_GENERATOR);
    g2_mul(&xz, &g2gen, z);
    g2_neg(&xz, &xz);
    g2_from_affine(&tau, &s->g2[1]);
    g2_add(&xz, &tau, &xz);
    g2_to_affine(&x_minus_z, &xz);
    g1_t g1gen, yg, c;
    g1a_t p_minus_y;
    g1_from_affine(&g1gen, &G1_GENERATOR);
    g1_mul(&yg, &g1gen, y);
    g1_neg(&yg, &yg);
    g1_from_affine(&c, commitment);
    g1_add(&c, &c, &yg);
    g1_to_affine(&p_minus_y, &c);
    return pairings_verify(&p_minus_y, &G2_GENERATOR, proof, &x_minus_z);
}

int oracle_verify_kzg_proof(int *ok, const uint8_t cb[48], const uint8_t zb[32], const uint8_t yb[32],
                            const uint8_t pb[48], const oracle_settings *s) {
    bls_init();
    fr_t z, y;
    g1a_t c, p;
    int rc;
    if ((rc = safe_scalar_affine_from_bytes(&z, zb))) return rc;
    if ((rc = safe_scalar_affine_from_bytes(&y, yb))) return rc;
    if ((rc = safe_g1_affine_from_bytes(&c, cb))) return rc;
    if ((rc = safe_g1_affine_from_bytes(&p, pb))) return rc;
    *ok = verify_kzg_proof_impl(&c, &z, &y, &p, s);
    return ORACLE_OK;
}

int oracle_verify_blob_kzg_proof(int *ok, const uint8_t *blob, const uint8_t cb[48], const uint8_t pb[48],
                                 const oracle_settings *s) {
    bls_init();
    g1a_t c, p;
    fr_t z, y;
    int rc;
    if ((rc = safe_g1_affine_from_bytes(&c, cb))) return rc;
    fr_t *poly = scratch()->poly;
    if ((rc = blob_as_polynomial(poly, blob))) goto done;
    if ((rc = safe_g1_affine_from_bytes(&p, pb))) goto done;
    compute_challenge(&z, blob, cb); /* to_compressed(from_compressed(b)) == b for accepted b */
    if ((rc = evaluate_polynomial_in_evaluation_form(&y, poly, &z, s))) goto done;
    *ok = verify_kzg_proof_impl(&c, &z, &y, &p, s);
done:
    return rc;
}

/* src/kzg_proof.rs:291-348 (+ :279-289 by the caller) */
static void compute_r(fr_t *r, const uint8_t *commitments, const fr_t *zs, const fr_t *ys, const uint8_t *proofs,
                      size_t n, int be) {
    size_t sz = 32 + n * 160;
    uint8_t *t = (uint8_t *)malloc(sz);
    memcpy(t, "RCKZGBATCH___V1_", 16);
    memset(t + 16, 0, 16);
    t[16 + 6] = (uint8_t)(N_FE >> 8);
    t[16 + 7] = (uint8_t)(N_FE & 0xff);
    for (int k = 0; k < 8; k++) t[24 + k] = (uint8_t)((uint64_t)n >> (56 - 8 * k));
    uint8_t *o = t + 32;
    for (size_t i = 0; i < n; i++, o += 160) {
        memcpy(o, commitments + 48 * i, 48);
        if (be) {
            fr_to_be(o + 48, &zs[i]);
            fr_to_be(o + 80, &ys[i]);
        } else {
            fr_to_le(o + 48, &zs[i]); /* Scalar::to_bytes() is little-endian: :321,:326 (quirk Q1) */
            fr_to_le(o + 80, &ys[i]);
        }
        memcpy(o + 112, proofs + 48 * i, 48);
    }
    uint8_t d[32];
    sha256(d, t, sz);
    free(t);
    fr_from_be_reduce(r, d);
}

int oracle_compute_r(uint8_t r_be[32], const uint8_t *commitments, const uint8_t *zs_be, const uint8_t *ys_be,
                     const uint8_t *proofs, size_t n, int be_transcript) {
    bls_init();
    fr_t *zs = (fr_t *)malloc(2 * n * sizeof(fr_t) + 1), *ys = zs + n, r;
    for (size_t i = 0; i < n; i++) {
        fr_from_be_reduce(&zs[i], zs_be + 32 * i);
        fr_from_be_reduce(&ys[i], ys_be + 32 * i);
    }
    compute_r(&r, commitments, zs, ys, proofs, n, be_transcript);
    fr_to_be(r_be, &r);
    free(zs);
    return ORACLE_OK;
}

/* src/kzg_proof.rs:251-277, one slice of the loop */
typedef struct {
    const uint8_t *blobs, *commitments;
    size_t lo, hi;
    fr_t *zs, *ys;
    const oracle_settings *s;
    int rc;
} slice_t;

static void *slice_run(void *arg) {
    slice_t *sl = (slice_t *)arg;
    fr_t *poly = scratch()->poly;
    sl->rc = ORACLE_OK;
    for (size_t i = sl->lo; i < sl->hi; i++) {
        const uint8_t *blob = sl->blobs + (size_t)BLOB_BYTES * i;
        if ((sl->rc = blob_as_polynomial(poly, blob))) break;
        compute_challenge(&sl->zs[i], blob, sl->commitments + 48 * i);
        if ((sl->rc = evaluate_polynomial_in_evaluation_form(&sl->ys[i], poly, &sl->zs[i], sl->s))) break;
    }
    return NULL;
}

/* src/kzg_proof.rs:399-444 */
static int verify_kzg_proof_batch(int *ok, const g1a_t *commitments, const uint8_t *cbytes, const fr_t *zs,
                                  const fr_t *ys, const g1a_t *proofs, const uint8_t *pbytes, size_t n,
                                  const oracle_settings *s, int be, uint8_t r_be[32], uint8_t A48[48], uint8_t B48[48]) {
    fr_t r, *r_powers = (fr_t *)malloc((2 * n + 1) * sizeof(fr_t)), *r_times_z = r_powers + n;
    g1a_t *c_minus_y = (g1a_t *)malloc((n + 1) * sizeof(g1a_t));
    compute_r(&r, cbytes, zs, ys, pbytes, n, be);
    /* compute_powers :279-289 */
    if (n) fr_one(&r_powers[0]); /* :281-283: an empty batch has no powers */
    for (size_t i = 1; i < n; i++) fr_mul(&r_powers[i], &r_powers[i - 1], &r);
    g1_t proof_lincomb, proof_z_lincomb, c_minus_y_lincomb, rhs, gen, t, c;
    g1_msm(&proof_lincomb, proofs, r_powers, n); /* :419 */
    g1_from_affine(&gen, &G1_GENERATOR);
    for (size_t i = 0; i < n; i++) { /* :422-426 */
        g1_mul(&t, &gen, &ys[i]);
        g1_neg(&t, &t);
        g1_from_affine(&c, &commitments[i]);
        g1_add(&c, &c, &t);
        g1_to_affine(&c_minus_y[i], &c);
        fr_mul(&r_times_z[i], &r_powers[i], &zs[i]);
    }
    g1_msm(&proof_z_lincomb, proofs, r_times_z, n);     /* :429 */
    g1_msm(&c_minus_y_lincomb, c_minus_y, r_powers, n); /* :430 */
    g1_add(&rhs, &c_minus_y_lincomb, &proof_z_lincomb);  /* :433 */
    g1a_t a1, b1;
    g1_to_affine(&a1, &proof_lincomb);
    g1_to_affine(&b1, &rhs);
    *ok = pairings_verify(&a1, &s->g2[1], &b1, &G2_GENERATOR); /* :436-441 */
    if (r_be) fr_to_be(r_be, &r);
    if (A48) g1_compress(A48, &a1);
    if (B48) g1_compress(B48, &b1);
    free(r_powers);
    free(c_minus_y);
    return ORACLE_OK;
}

int oracle_verify_blob_kzg_proof_batch_ex(int *ok, const uint8_t *blobs, const uint8_t *commitments,
                                          const uint8_t *proofs, size_t n, const oracle_settings *s, int nthreads,
                                          int be, uint8_t *zs_out, uint8_t *ys_out, uint8_t r_be[32],
                                          uint8_t A48[48], uint8_t B48[48]) {
    bls_init();
    if (n == 0) { /* :478-480 */
        *ok = 1;
        return ORACLE_OK;
    }
    if (n == 1 && !zs_out) /* :482-489 */
        return oracle_verify_blob_kzg_proof(ok, blobs, commitments, proofs, s);
    int rc = ORACLE_OK;
    g1a_t *cs = (g1a_t *)malloc(2 * n * sizeof(g1a_t)), *ps = cs + n;
    fr_t *zs = (fr_t *)malloc(2 * n * sizeof(fr_t)), *ys = zs + n;
    for (size_t i = 0; i < n && !rc; i++) rc = safe_g1_affine_from_bytes(&cs[i], commitments + 48 * i); /* :503 */
    for (size_t i = 0; i < n && !rc; i++) rc = safe_g1_affine_from_bytes(&ps[i], proofs + 48 * i);      /* :508 */
    /* validate_batched_input :225-249 is implied by from_compressed (quirk Q5) */
    if (!rc) {
        if (nthreads < 1) nthreads = 1;
        if ((size_t)nthreads > n) nthreads = (int)n;
        slice_t *sl = (slice_t *)calloc((size_t)nthreads, sizeof(slice_t));
        pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
        for (int k = 0; k < nthreads; k++) {
            sl[k] = (slice_t){blobs, commitments, n * (size_t)k / nthreads, n * (size_t)(k + 1) / nthreads, zs, ys, s, 0};
            if (nthreads == 1) slice_run(&sl[k]);
            else pthread_create(&th[k], NULL, slice_run, &sl[k]);
        }
        for (int k = 0; k < nthreads; k++) {
            if (nthreads > 1) pthread_join(th[k], NULL);
            if (sl[k].rc && !rc) rc = sl[k].rc;
        }
        free(sl);
        free(th);
    }
    if (!rc) {
        if (zs_out)
            for (size_t i = 0; i < n; i++) {
                fr_to_be(zs_out + 32 * i, &zs[i]);
                fr_to_be(ys_out + 32 * i, &ys[i]);
            }
        rc = verify_kzg_proof_batch(ok, cs, commitments, zs, ys, ps, proofs, n, s, be, r_be, A48, B48);
    }
    free(cs);
    free(zs);
    return rc;
}

int oracle_verify_blob_kzg_proof_batch(int *ok, const uint8_t *blobs, const uint8_t *commitments,
                                       const uint8_t *proofs, size_t n, const oracle_settings *s, int nthreads,
                                       int be) {
    return oracle_verify_blob_kzg_proof_batch_ex(ok, blobs, commitments, proofs, n, s, nthreads, be, NULL, NULL, NULL,
                                                 NULL, NULL);
}

/* src/kzg_proof.rs:399-444 with byte inputs: commitments/proofs compressed (decoded WITH the subgroup check, as the
 * typed &[G1Affine] arguments of the reference can only hold valid points), zs/ys big-endian canonical. */
int oracle_verify_kzg_proof_batch(int *ok, const uint8_t *commitments, const uint8_t *zs_be, const uint8_t *ys_be,
                                  const uint8_t *proofs, size_t n, const oracle_settings *s, int be) {
    bls_init();
    int rc = ORACLE_OK;
    g1a_t *cs = (g1a_t *)malloc((2 * n + 1) * sizeof(g1a_t)), *ps = cs + n;
    fr_t *zs = (fr_t *)malloc((2 * n + 1) * sizeof(fr_t)), *ys = zs + n;
    for (size_t i = 0; i < n && !rc; i++) {
        if (fr_from_be_canonical(&zs[i], zs_be + 32 * i) || fr_from_be_canonical(&ys[i], ys_be + 32 * i)) rc = ORACLE_BADARGS;
        else if (g1_decompress(&cs[i], commitments + 48 * i, 1) || g1_decompress(&ps[i], proofs + 48 * i, 1)) rc = ORACLE_BADARGS;
    }
    if (!rc) rc = verify_kzg_proof_batch(ok, cs, commitments, zs, ys, ps, proofs, n, s, be, NULL, NULL, NULL);
    free(cs);
    free(zs);
    return rc;
}

int oracle_compute_challenge(uint8_t z_be[32], const uint8_t *blob, const uint8_t commitment[48]) {
    bls_init();
    fr_t z;
    compute_challenge(&z, blob, commitment);
    fr_to_be(z_be, &z);
    return ORACLE_OK;
}

int oracle_evaluate_polynomial_in_evaluation_form(uint8_t y_be[32], const uint8_t *blob, const uint8_t z_be[32],
                                                  const oracle_settings *s) {
    bls_init();
    fr_t z, y, *poly = scratch()->poly;
    int rc = blob_as_polynomial(poly, blob);
    if (!rc) {
        fr_from_be_reduce(&z, z_be);
        rc = evaluate_polynomial_in_evaluation_form(&y, poly, &z, s);
    }
    if (!rc) fr_to_be(y_be, &y);
    return rc;
}

/* ---------------------------------------------------------------- primitives for kernel parity tests */

int oracle_g1_decompress(uint8_t xy_be[96], int *is_inf, const uint8_t in[48]) {
    bls_init();
    g1a_t p;
    if (g1_decompress(&p, in, 1)) return ORACLE_BADARGS;
    *is_inf = p.inf;
    memset(xy_be, 0, 96);
    if (!p.inf) {
        fp_to_be(xy_be, &p.x);
        fp_to_be(xy_be + 48, &p.y);
    }
    return ORACLE_OK;
}

int oracle_g1_msm(uint8_t out[48], const uint8_t *points48, const uint8_t *scalars_be, size_t n) {
    bls_init();
    g1a_t *pts = (g1a_t *)malloc((n + 1) * sizeof(g1a_t));
    fr_t *sc = (fr_t *)malloc((n + 1) * sizeof(fr_t));
    int rc = ORACLE_OK;
    for (size_t i = 0; i < n && !rc; i++) {
        if (g1_decompress(&pts[i], points48 + 48 * i, 0)) rc = ORACLE_BADARGS;
        fr_from_be_reduce(&sc[i], scalars_be + 32 * i);
    }
    if (!rc) {
        g1_t r;
        g1a_t ra;
        g1_msm(&r, pts, sc, n);
        g1_to_affine(&ra, &r);
        g1_compress(out, &ra);
    }
    free(pts);
    free(sc);
    return rc;
}

int oracle_g1_mul(uint8_t out[48], const uint8_t point[48], const uint8_t scalar_be[32]) {
    bls_init();
    g1a_t p, ra;
    g1_t pj, r;
    fr_t k;
    if (g1_decompress(&p, point, 0)) return ORACLE_BADARGS;
    fr_from_be_reduce(&k, scalar_be);
    g1_from_affine(&pj, &p);
    g1_mul(&r, &pj, &k);
    g1_to_affine(&ra, &r);
    g1_compress(out, &ra);
    return ORACLE_OK;
}

int oracle_g1_add(uint8_t out[48], const uint8_t a[48], const uint8_t b[48]) {
    bls_init();
    g1a_t pa, pb, ra;
    g1_t ja, jb;
    if (g1_decompress(&pa, a, 0) || g1_decompress(&pb, b, 0)) return ORACLE_BADARGS;
    g1_from_affine(&ja, &pa);
    g1_from_affine(&jb, &pb);
    g1_add(&ja, &ja, &jb);
    g1_to_affine(&ra, &ja);
    g1_compress(out, &ra);
    return ORACLE_OK;
}

int oracle_g2_mul(uint8_t out[96], const uint8_t point[96], const uint8_t scalar_be[32]) {
    bls_init();
    g2a_t p, ra;
    g2_t pj, r;
    fr_t k;
    if (g2_decompress(&p, point)) return ORACLE_BADARGS;
    fr_from_be_reduce(&k, scalar_be);
    g2_from_affine(&pj, &p);
    g2_mul(&r, &pj, &k);
    g2_to_affine(&ra, &r);
    g2_compress(out, &ra);
    return ORACLE_OK;
}

int oracle_pairings_verify(int *ok, const uint8_t a1[48], const uint8_t a2[96], const uint8_t b1[48],
                           const uint8_t b2[96]) {
    bls_init();
    g1a_t pa, pb;
    g2a_t qa, qb;
    if (g1_decompress(&pa, a1, 0) || g1_decompress(&pb, b1, 0) || g2_decompress(&qa, a2) || g2_decompress(&qb, b2))
        return ORACLE_BADARGS;
    *ok = pairings_verify(&pa, &qa, &pb, &qb);
    return ORACLE_OK;
}

void oracle_sha256(uint8_t out[32], const uint8_t *data, size_t len) {
    bls_init();
    sha256(out, data, len);
}

void oracle_fr_mul(uint8_t out_be[32], const uint8_t a_be[32], const uint8_t b_be[32]) {
    bls_init();
    fr_t a, b;
    fr_from_be_reduce(&a, a_be);
    fr_from_be_reduce(&b, b_be);
    fr_mul(&a, &a, &b);
    fr_to_be(out_be, &a);
}

void oracle_fr_inv(uint8_t out_be[32], const uint8_t a_be[32]) {
    bls_init();
    fr_t a;
    fr_from_be_reduce(&a, a_be);
    fr_inv(&a, &a);
    fr_to_be(out_be, &a);
}

void oracle_constants(uint8_t fr_R[32], uint8_t fr_R2[32], uint64_t *fr_inv_, uint8_t fp_R[48], uint8_t fp_R2[48],
                      uint64_t *fp_inv_) {
    bls_init();
    fr_t r, r2;
    fp_t p, p2;
    fields_constants(&r, &r2, fr_inv_, &p, &p2, fp_inv_);
    /* raw limbs (these ARE the plain integers R mod m, R^2 mod m), big-endian */
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) {
            fr_R[8 * (3 - i) + k] = (uint8_t)(r.l[i] >> (56 - 8 * k));
            fr_R2[8 * (3 - i) + k] = (uint8_t)(r2.l[i] >> (56 - 8 * k));
        }
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 8; k++) {
            fp_R[8 * (5 - i) + k] = (uint8_t)(p.l[i] >> (56 - 8 * k));
            fp_R2[8 * (5 - i) + k] = (uint8_t)(p2.l[i] >> (56 - 8 * k));
        }
}
