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
// instruction count per block allows: the even lane of a pair carries (a, b, c) and d, the odd lane (e, f, g, h), and one
// uniform instruction stream serves both - 11 instructions per round instead of 14.5:
//   x0..x2 : a, b, c | e, f, g         x3 : 0 | h          dd : d | -
//   Sigma  : three v_alignbit with the rotation amounts in a VGPR (2, 13, 22 | 6, 11, 25) + one xor3
//   F      : Maj(a, b, c) = Ch(~(a ^ b), b, c) | Ch(e, f, g): u = bitop3(x0, x1, m) (m = ~0 | 0), F = bitop3(u, x1, x2, Ch)
//   w      : Sigma + F + hk,  hk = x3 + kw  =  T2 | T1   (kw reads as zero on the even lanes; hk is prepared a round ahead)
//   x0'    : w + swap(y),  y = dd | w  (one v_cndmask; a DPP bank mask selects groups of four lanes, not odd lanes),
//            swap = quad_perm [1,0,3,2]:  T2 + T1 | T1 + d
//   x3'    : x2 & ~m  (0 | g),  dd' = x2
struct Sha2LaneConsts {
    uint32_t n1, n2, n3;  // rotation amounts of this lane's Sigma
    uint32_t m;           // ~0 on the even (a-chain) lanes, 0 on the odd (e-chain) lanes
};
// hk = x3 + kw of THIS round (prepared during the previous one); kw_next = the next round's message word.  The two
// instructions that prepare the next round sit between the select that writes y and the DPP add that reads it across
// lanes (a VALU write followed by a DPP read needs two wait states: an s_nop otherwise, one issue slot per round).
__device__ __forceinline__ void sha256_round_2lane(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& hk, uint32_t& dd, uint32_t kw_next,
                                                   const Sha2LaneConsts& c) {
    const uint32_t S = xor3(__builtin_amdgcn_alignbit(x0, x0, c.n1), __builtin_amdgcn_alignbit(x0, x0, c.n2), __builtin_amdgcn_alignbit(x0, x0, c.n3));
    const uint32_t u = __builtin_amdgcn_bitop3_b32(x0, x1, c.m, 0xD2);  // m ? ~(x0 ^ x1) : x0
    const uint32_t F = sha_ch(u, x1, x2);
    const uint32_t w = S + F + hk;
    const uint32_t y = c.m ? dd : w;  // even lanes: d, odd lanes: T1
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t x3n = x2 & ~c.m;   // next round's h (odd lanes) / 0 (even lanes)
    uint32_t hkn = x3n + kw_next;
    asm volatile("" : "+v"(hkn));     // (keeps the addition here: reassociation would sink it into the next round's sum)
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0xB1, 0xF, 0xF, true);  // the pair's other lane
    const uint32_t nx = w + t;
    dd = x2;
    hk = hkn;
    x2 = x1;
    x1 = x0;
    x0 = nx;
}
// four rounds; kwz = this quad's message words (zero on the even lanes), kw_after = the first word of the next quad
__device__ __forceinline__ void sha256_rounds4_2lane(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& hk, uint32_t& dd, const uint4& kwz,
                                                     uint32_t kw_after, const Sha2LaneConsts& c) {
    sha256_round_2lane(x0, x1, x2, hk, dd, kwz.y, c);
    sha256_round_2lane(x0, x1, x2, hk, dd, kwz.z, c);
    sha256_round_2lane(x0, x1, x2, hk, dd, kwz.w, c);
    sha256_round_2lane(x0, x1, x2, hk, dd, kw_after, c);
}

}  // namespace kzg
