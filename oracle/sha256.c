/*
 * oracle/sha256.c - FIPS 180-4 SHA-256 for the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 * Stands in for the reference's sha2::Sha256::digest (src/kzg_proof.rs:70,344).
 * Round constants are derived at first use from the cube roots of the first 64 primes
 * (their published definition) so no table is remembered.
 */
#include "bls.h"
#include <string.h>

static uint32_t K[64];
static uint32_t H0[8];
static int sha_ready;

/* floor(frac(root) * 2^32) for the k-th root of prime q, by integer Newton on 96-bit ints */
static uint32_t frac_root(unsigned q, int k) {
    /* find floor(q^(1/k) * 2^32) via binary search on x: x^k <= q * 2^(32k) */
    unsigned __int128 lo = 0, hi = (unsigned __int128)1 << 40;
    while (hi - lo > 1) {
        unsigned __int128 mid = (lo + hi) / 2;
        /* compare mid^k with q << (32k) using long doubles is unsafe; do exact 256-bit via splitting */
        /* mid < 2^40, k <= 3  => mid^3 < 2^120 fits in 128 bits; q * 2^96 < 2^105 fits */
        unsigned __int128 pw = mid;
        for (int i = 1; i < k; i++) pw *= mid;
        unsigned __int128 target = (unsigned __int128)q << (32 * k);
        if (pw <= target) lo = mid;
        else hi = mid;
    }
    return (uint32_t)lo;
}

static void sha_init_tables(void) {
    unsigned primes[64], np = 0;
    for (unsigned c = 2; np < 64; c++) {
        int is_p = 1;
        for (unsigned d = 2; d * d <= c; d++)
            if (c % d == 0) {
                is_p = 0;
                break;
            }
        if (is_p) primes[np++] = c;
    }
    for (int i = 0; i < 64; i++) K[i] = frac_root(primes[i], 3);
    for (int i = 0; i < 8; i++) H0[i] = frac_root(primes[i], 2);
    sha_ready = 1;
}

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void compress(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

void sha256(uint8_t out[32], const uint8_t *data, size_t len) {
    if (!sha_ready) sha_init_tables();
    uint32_t st[8];
    memcpy(st, H0, sizeof st);
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) compress(st, data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    compress(st, tail);
    if (tl == 128) compress(st, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(st[i] >> 24);
        out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8);
        out[4 * i + 3] = (uint8_t)st[i];
    }
}
