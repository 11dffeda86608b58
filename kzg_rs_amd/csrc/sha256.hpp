// sha256.hpp - FIPS 180-4 SHA-256 compression for gfx950, one hash state per lane.
// Replaces the reference's sha2::Sha256::digest at src/kzg_proof.rs:70 (per-blob Fiat-Shamir
// challenge).  The per-blob transcript is a 2050-block serial chain, so the only parallelism
// is across blobs (one lane each); the 16-word message window stays in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kzg {

static constexpr uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct Sha256State {
    uint32_t h[8];
};

__device__ __forceinline__ void sha256_init(Sha256State& s) {
    s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
    s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
}

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950 has a 3-input arbitrary boolean op (v_bitop3_b32); hipcc finds it only partly, so spell it out:
// truth-table byte indexed by (a << 2 | b << 1 | c).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// one 64-byte block; w[16] = message words (already big-endian decoded); destroyed.
// ~14 VALU instructions per round + ~10 per scheduled word: 3 v_alignbit + 1 v_bitop3 per Sigma/sigma,
// one v_bitop3 each for Ch and Maj, v_add3_u32 for the sums.
__device__ __forceinline__ void sha256_compress(Sha256State& s, uint32_t (&w)[16]) {
    uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            wi = (w[i & 15] + s0) + (w[(i - 7) & 15] + s1);
            w[i & 15] = wi;
        }
        uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        uint32_t t1 = (h + S1 + sha_ch(e, f, g)) + (SHA_K[i] + wi);
        uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        uint32_t t2 = S0 + sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}

// ---- split form: the message schedule (+K) and the 64 rounds as separate functions, for the
// producer / consumer challenge kernel (fr_kernels.hpp: k_blob_challenge_split)
// kw[t] = W[t] + K[t], t = 0..63, from the 16 message words
__device__ __forceinline__ void sha256_schedule_kw(uint32_t (&kw)[64], uint32_t (&w)[16]) {
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            wi = (w[i & 15] + s0) + (w[(i - 7) & 15] + s1);
            w[i & 15] = wi;
        }
        kw[i] = wi + SHA_K[i];
    }
}
// four rounds t..t+3 with kw = (K+W)[t..t+3]
__device__ __forceinline__ void sha256_rounds4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& e, uint32_t& f,
                                               uint32_t& g, uint32_t& h, const uint4& kw) {
    const uint32_t k4[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        uint32_t t1 = (h + S1 + sha_ch(e, f, g)) + k4[i];
        uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        uint32_t t2 = S0 + sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
}

// ---- two lanes per hash state (the latency form of the challenge kernel, fr_kernels.hpp: k_blob_challenge_split2).
// A wavefront issues at most one instruction every ~4.3 cycles whatever it is, so ONE blob's serial chain is as fast as its
// instruction count per block allows: one lane of a pair carries (a, b, c) and d, the other (e, f, g, h), and one uniform
// instruction stream serves both - 10 instructions per round instead of 14.5.  The pair sits four lanes apart (lanes 0-3 of
// every eight: a-chains of four blobs; lanes 4-7: their e-chains), because a DPP bank mask enables groups of four lanes:
//   x0..x2 : a, b, c | e, f, g         hk : 0 | h + kw      dd : d | -
//   Sigma  : three v_alignbit with the rotation amounts in a VGPR (2, 13, 22 | 6, 11, 25) + one xor3
//   F      : Maj(a, b, c) = Ch(~(a ^ b), b, c) | Ch(e, f, g): u = bitop3(x0, x1, m) (m = ~0 | 0), F = bitop3(u, x1, x2, Ch)
//   w      : Sigma + F + hk  =  T2 | T1
//   hk'    : e-lanes only: x2 + kw'   (v_add_u32_dpp with the identity permutation and bank_mask:0xa: the bank mask is a
//            free execution mask - the a-lanes keep their zero)
//   x0'    : e-lanes: w + dd[lane - 4] = T1 + d     (v_add_u32_dpp row_shr:4 bank_mask:0xa)
//            a-lanes: w + w[lane + 4]  = T2 + T1    (v_add_u32_dpp row_shl:4 bank_mask:0x5)
// The DPP read of w comes two instructions after the add3 that wrote it (the two e-lane additions sit in between): no
// wait states, no select.
struct Sha2LaneConsts {
    uint32_t n1, n2, n3;  // rotation amounts of this lane's Sigma
    uint32_t m;           // ~0 on the a-chain lanes, 0 on the e-chain lanes
};
// One round as text: operands %0..%4 = the ring of state registers (x0, x1, x2, dd, next x0 - it turns by one per round),
// %5 = hk (x3 + kw of THIS round, prepared during the previous one; updated in place on the e-lanes only - the a-lanes
// keep the zero it was initialised with), %6..%8 = temporaries, %9..%12 = n1, n2, n3, m, KW = the NEXT round's message word.
// Rounds are chained inside ONE asm statement, ten at a time: the order of a round's last three instructions is the DPP
// hazard protection (the compiler does not look inside; the two masked writes of the next x0 are kept apart as well:
// back to back they cost a twelfth slot, tools/microbench/issuebench.hip), and the compiler pads every consumer of an asm result with a
// wait state - also between two asm statements - which would put the eleventh instruction back into every round.
#define KZG_SHA2L_ROUND(X0, X1, X2, DD, NX, KW)                                                       \
    "v_alignbit_b32 %6, " X0 ", " X0 ", %9\n\t"                                                       \
    "v_alignbit_b32 %7, " X0 ", " X0 ", %10\n\t"                                                      \
    "v_alignbit_b32 %8, " X0 ", " X0 ", %11\n\t"                                                      \
    "v_bitop3_b32 %6, %6, %7, %8 bitop3:0x96\n\t"                 /* Sigma */                        \
    "v_bitop3_b32 %7, " X0 ", " X1 ", %12 bitop3:0xd2\n\t"        /* m ? ~(x0 ^ x1) : x0 */          \
    "v_bitop3_b32 %7, %7, " X1 ", " X2 " bitop3:0xca\n\t"         /* Ch(u, x1, x2) */                \
    "v_add3_u32 %6, %6, %7, %5\n\t"                               /* w = Sigma + F + hk */           \
    "v_add_u32_dpp " NX ", " DD ", %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"           /* e: d[lane - 4] + w */ \
    "v_add_u32_dpp %5, " X2 ", " KW " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\t" /* e: hk' = x2 + kw' */ \
    "v_add_u32_dpp " NX ", %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"               /* a: w[lane + 4] + w */
#define KZG_SHA2L_ROUNDS5(K0, K1, K2, K3, K4)                                                                                  \
    KZG_SHA2L_ROUND("%0", "%1", "%2", "%3", "%4", K0) KZG_SHA2L_ROUND("%4", "%0", "%1", "%2", "%3", K1)                       \
    KZG_SHA2L_ROUND("%3", "%4", "%0", "%1", "%2", K2) KZG_SHA2L_ROUND("%2", "%3", "%4", "%0", "%1", K3)                       \
    KZG_SHA2L_ROUND("%1", "%2", "%3", "%4", "%0", K4)
#define KZG_SHA2L_OPERANDS "+v"(r.x0), "+v"(r.x1), "+v"(r.x2), "+v"(r.dd), "+v"(r.nx), "+v"(r.hk), "=&v"(t0), "=&v"(t1), "=&v"(t2)
struct Sha2LaneRing {
    uint32_t x0, x1, x2, dd, nx, hk;
};
// ten rounds: k[0..9] = the message words of the rounds AFTER each of them; the ring is back in place afterwards
__device__ __forceinline__ void sha256_rounds10_2lane(Sha2LaneRing& r, const uint32_t* k, const Sha2LaneConsts& c) {
    uint32_t t0, t1, t2;
    asm volatile(KZG_SHA2L_ROUNDS5("%13", "%14", "%15", "%16", "%17") KZG_SHA2L_ROUNDS5("%18", "%19", "%20", "%21", "%22")
                 : KZG_SHA2L_OPERANDS
                 : "v"(c.n1), "v"(c.n2), "v"(c.n3), "v"(c.m), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]),
                   "v"(k[8]), "v"(k[9]));
}
// the last four rounds of a block (k[3] = 0): afterwards x0..x2, dd are renamed back into place
__device__ __forceinline__ void sha256_rounds4_2lane(Sha2LaneRing& r, const uint32_t* k, const Sha2LaneConsts& c) {
    uint32_t t0, t1, t2;
    asm volatile(KZG_SHA2L_ROUND("%0", "%1", "%2", "%3", "%4", "%13") KZG_SHA2L_ROUND("%4", "%0", "%1", "%2", "%3", "%14")
                 KZG_SHA2L_ROUND("%3", "%4", "%0", "%1", "%2", "%15") KZG_SHA2L_ROUND("%2", "%3", "%4", "%0", "%1", "%16")
                 : KZG_SHA2L_OPERANDS
                 : "v"(c.n1), "v"(c.n2), "v"(c.n3), "v"(c.m), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
    const uint32_t a = r.x1, b = r.x2, cc = r.dd, d = r.nx;  // the ring turned by four
    r.x0 = a;
    r.x1 = b;
    r.x2 = cc;
    r.dd = d;
}

}  // namespace kzg
