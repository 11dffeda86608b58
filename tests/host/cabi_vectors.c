/* A C11 caller of libkzg_rs_amd.so: proves that include/kzg_rs_amd.h compiles as C (gcc -std=c11 -Wall -Werror), that
 * its prototypes bind to the library's definitions, and - on a GPU box - runs all 175 c-kzg-4844 vectors the reference
 * ships (tests/golden, re-encoded into a flat file by tests/test_cabi_c_program.py) through the three reference-shaped
 * entry points with the strict convention null <=> error.  The caller-side checks that need the Vec lengths (from_slice
 * length errors src/dtypes.rs:20-25, the early returns and length mismatches of src/kzg_proof.rs:478-501) are done here
 * as the shim would do them.
 *
 *   cabi_vectors <trusted_setup.txt> <vectors.bin>      run the vectors (needs a gfx950 device)
 *   cabi_vectors <trusted_setup.txt> --no-gpu           no device: every entry point must fail cleanly with KZG_ERROR
 *
 * flat file: u32 magic 'KZGV', u32 count, then per case: u8 kind (1 verify_kzg_proof, 2 verify_blob_kzg_proof,
 * 3 verify_blob_kzg_proof_batch), i8 expect (1 true, 0 false, -1 error), fields as u32 length + bytes:
 *   kind 1: commitment z y proof      kind 2: blob commitment proof
 *   kind 3: u32 n_blobs + blobs, u32 n_commitments + commitments, u32 n_proofs + proofs */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kzg_rs_amd.h"

typedef struct {
    const uint8_t *p;
    uint32_t len;
} field;

static const uint8_t *cur, *end;
static int take(void *out, size_t n) {
    if ((size_t)(end - cur) < n) return -1;
    memcpy(out, cur, n);
    cur += n;
    return 0;
}
static int take_field(field *f) {
    if (take(&f->len, 4) || (size_t)(end - cur) < f->len) return -1;
    f->p = cur;
    cur += f->len;
    return 0;
}

static uint8_t *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) {
        free(buf);
        buf = NULL;
    }
    fclose(f);
    if (buf) *len = (size_t)n;
    return buf;
}

/* 1 true, 0 false, -1 error.  (The call and the read of *ok are two statements at every use: as two arguments of one
 * call their order of evaluation would be unspecified.) */
static int outcome(KzgRet rc, bool ok) { return rc == KZG_OK ? (ok ? 1 : 0) : -1; }

static int run_batch(uint32_t nb, const field *blobs, uint32_t nc, const field *cs, uint32_t np, const field *ps, const KzgSettings *s) {
    /* Bytes48::from_slice / Blob::from_slice of every element come first in the reference's test harness */
    for (uint32_t i = 0; i < nb; i++)
        if (blobs[i].len != KZG_BYTES_PER_BLOB) return -1;
    for (uint32_t i = 0; i < nc; i++)
        if (cs[i].len != KZG_BYTES_PER_COMMITMENT) return -1;
    for (uint32_t i = 0; i < np; i++)
        if (ps[i].len != KZG_BYTES_PER_PROOF) return -1;
    bool ok = false;
    if (nb == 0) return 1;                      /* src/kzg_proof.rs:478-480 */
    if (nb == 1) {                              /* :482-489 */
        if (nc < 1 || np < 1) return -1;        /* the reference panics on [0] */
        KzgRet rc1 = kzg_verify_blob_kzg_proof(&ok, blobs[0].p, cs[0].p, ps[0].p, s);
        return outcome(rc1, ok);
    }
    if (nb != nc || nb != np) return -1;        /* :491-501 */
    uint8_t *b = (uint8_t *)malloc((size_t)nb * KZG_BYTES_PER_BLOB), *c = (uint8_t *)malloc((size_t)nb * 48), *p = (uint8_t *)malloc((size_t)nb * 48);
    if (!b || !c || !p) return -2;
    for (uint32_t i = 0; i < nb; i++) {
        memcpy(b + (size_t)i * KZG_BYTES_PER_BLOB, blobs[i].p, KZG_BYTES_PER_BLOB);
        memcpy(c + 48 * (size_t)i, cs[i].p, 48);
        memcpy(p + 48 * (size_t)i, ps[i].p, 48);
    }
    KzgRet rcb = kzg_verify_blob_kzg_proof_batch(&ok, b, c, p, nb, s);
    int r = outcome(rcb, ok);
    free(b);
    free(c);
    free(p);
    return r;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s trusted_setup.txt vectors.bin | --no-gpu\n", argv[0]);
        return 2;
    }
    size_t tlen = 0, vlen = 0;
    uint8_t *txt = read_file(argv[1], &tlen);
    if (!txt) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    KzgSettings *s = NULL;
    KzgRet rc = kzg_settings_load_trusted_setup(&s, (const char *)txt, tlen);
    if (strcmp(argv[2], "--no-gpu") == 0) {
        /* no CPU fallback: a clean KZG_ERROR with a message, nothing allocated, and the handle-free entry points still answer */
        bool ok = true;
        uint8_t z48[48] = {0xc0}, z32[32] = {0};
        if (rc != KZG_ERROR || s != NULL || !kzg_last_error()[0]) {
            fprintf(stderr, "expected KZG_ERROR without a device, got rc %d (%s)\n", (int)rc, kzg_last_error());
            return 1;
        }
        if (kzg_verify_kzg_proof(&ok, z48, z32, z32, z48, NULL) != KZG_BADARGS) return 1; /* null handle */
        if (kzg_verify_blob_kzg_proof_batch(&ok, NULL, NULL, NULL, 0, NULL) != KZG_BADARGS) return 1;
        uint8_t rec[160] = {0}, r[32];
        if (kzg_batch_challenges(r, rec, 0, 1, 1) != KZG_OK) return 1; /* pure host code */
        kzg_settings_free(NULL);
        printf("no-gpu ok: %s\n", kzg_last_error());
        return 0;
    }
    if (rc != KZG_OK) {
        fprintf(stderr, "kzg_settings_load_trusted_setup: rc %d (%s)\n", (int)rc, kzg_last_error());
        return 1;
    }
    uint8_t *vec = read_file(argv[2], &vlen);
    if (!vec) {
        fprintf(stderr, "cannot read %s\n", argv[2]);
        return 2;
    }
    cur = vec;
    end = vec + vlen;
    uint32_t magic = 0, count = 0;
    if (take(&magic, 4) || take(&count, 4) || magic != 0x56475a4bu) {
        fprintf(stderr, "bad vector file\n");
        return 2;
    }
    unsigned pass = 0, by_kind[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < count; i++) {
        uint8_t kind = 0;
        int8_t expect = 0;
        int got = -3;
        if (take(&kind, 1) || take(&expect, 1)) return 2;
        if (kind == 1) {
            field c, z, y, p;
            if (take_field(&c) || take_field(&z) || take_field(&y) || take_field(&p)) return 2;
            bool ok = false;
            if (c.len != 48 || z.len != 32 || y.len != 32 || p.len != 48) got = -1; /* from_slice: InvalidBytesLength */
            else {
                KzgRet r1 = kzg_verify_kzg_proof(&ok, c.p, z.p, y.p, p.p, s);
                got = outcome(r1, ok);
            }
        } else if (kind == 2) {
            field b, c, p;
            if (take_field(&b) || take_field(&c) || take_field(&p)) return 2;
            bool ok = false;
            if (b.len != KZG_BYTES_PER_BLOB || c.len != 48 || p.len != 48) got = -1;
            else {
                KzgRet r2 = kzg_verify_blob_kzg_proof(&ok, b.p, c.p, p.p, s);
                got = outcome(r2, ok);
            }
        } else if (kind == 3) {
            uint32_t n[3];
            field *f[3];
            for (int k = 0; k < 3; k++) {
                if (take(&n[k], 4)) return 2;
                f[k] = (field *)calloc(n[k] ? n[k] : 1, sizeof(field));
                for (uint32_t j = 0; j < n[k]; j++)
                    if (take_field(&f[k][j])) return 2;
            }
            got = run_batch(n[0], f[0], n[1], f[1], n[2], f[2], s);
            for (int k = 0; k < 3; k++) free(f[k]);
        } else {
            return 2;
        }
        if (got == expect) {
            pass++;
            by_kind[kind]++;
        } else {
            fprintf(stderr, "case %u (kind %u): expected %d, got %d (%s)\n", i, kind, expect, got, got == -1 ? kzg_last_error() : "");
        }
    }
    kzg_settings_free(s);
    printf("%u vectors: %u ok (verify_kzg_proof %u, verify_blob_kzg_proof %u, verify_blob_kzg_proof_batch %u)\n", count, pass, by_kind[1],
           by_kind[2], by_kind[3]);
    free(vec);
    free(txt);
    return pass == count ? 0 : 1;
}
