// kzg_capi.hip - libkzg_rs_amd.so: the C ABI of include/kzg_rs_amd.h over the gfx950 kernels.
//
// Host side of the reference's verification flow (src/kzg_proof.rs:353-525).  The host does no
// field or curve arithmetic: it parses bytes, launches kernels, hashes the (n * 160 + 32)-byte
// batch transcript (src/kzg_proof.rs:291-348) with SHA-256 - a single serial chain that a host
// core finishes in a fraction of the time one GPU lane would need - and reduces the 256-bit
// digest below r by at most two subtractions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_rs_amd.h"
#include "fr_kernels.hpp"
#include "g1.hpp"
#include "msm.hpp"
#include "slp.hpp"

using namespace kzg;

// ---------------------------------------------------------------- embedded SLP programs
#if !defined(__HIP_DEVICE_COMPILE__)
// KZG_DATA_DIR is a string literal supplied by the build (kzg_rs_amd/build.py)
__asm__(".section .rodata\n"
        ".balign 16\n.global kzg_slp_prep_begin\nkzg_slp_prep_begin:\n.incbin \"" KZG_DATA_DIR "/slp_prep.bin\"\n"
        ".global kzg_slp_prep_end\nkzg_slp_prep_end:\n"
        ".balign 16\n.global kzg_slp_verify_begin\nkzg_slp_verify_begin:\n.incbin \"" KZG_DATA_DIR "/slp_verify.bin\"\n"
        ".global kzg_slp_verify_end\nkzg_slp_verify_end:\n"
        ".text\n");
#endif
extern "C" const unsigned char kzg_slp_prep_begin[], kzg_slp_prep_end[], kzg_slp_verify_begin[], kzg_slp_verify_end[];

// ---------------------------------------------------------------- small kernels
// points [0, n0) come from bytes0, [n0, n) from bytes1 (commitments then proofs in one launch)
__global__ __launch_bounds__(64) void k_g1_decode(const uint8_t* __restrict__ bytes0, const uint8_t* __restrict__ bytes1, int n0,
                                                  G1Aff* __restrict__ out, uint32_t* __restrict__ flag, int n, int check_subgroup) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = i < n0 ? bytes0 + (size_t)i * 48 : bytes1 + (size_t)(i - n0) * 48;
    G1Aff a;
    uint32_t st = g1_decompress(a, src, check_subgroup != 0);
    out[i] = a;
    flag[i] = st;
}

__global__ void k_g2_decompress(const uint8_t* __restrict__ bytes, Fp* __restrict__ out4, uint32_t* __restrict__ flag) {
    if (threadIdx.x || blockIdx.x) return;
    G2Aff q;
    uint32_t st = g2_decompress(q, bytes);
    out4[0] = q.x.c0;
    out4[1] = q.x.c1;
    out4[2] = q.y.c0;
    out4[3] = q.y.c1;
    *flag = st;
}

// n G2 points, one per workgroup (trusted-setup load: build.rs:73, from_compressed_unchecked)
__global__ void k_g2_decompress_n(const uint8_t* __restrict__ bytes, Fp* __restrict__ out4, uint32_t* __restrict__ flag) {
    if (threadIdx.x) return;
    const int i = blockIdx.x;
    G2Aff q;
    q.x.c0 = q.x.c1 = q.y.c0 = q.y.c1 = FpF::zero();
    uint32_t st = g2_decompress(q, bytes + 96 * (size_t)i);
    out4[4 * i] = q.x.c0;
    out4[4 * i + 1] = q.x.c1;
    out4[4 * i + 2] = q.y.c0;
    out4[4 * i + 3] = q.y.c1;
    flag[i] = st;
}

// affine points -> 48 compressed bytes (flag != 0: the identity encoding)
__global__ void k_aff_compress(const G1Aff* __restrict__ pts, const uint32_t* __restrict__ flag, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_compress(out + 48 * (size_t)i, pts[i], flag[i] != 0);
}

// blob bytes -> MSM scalars: element i of blob b (32 big-endian bytes) as plain little-endian limbs; status[b] |= 1
// when an element is >= r (src/dtypes.rs:48-57)
__global__ void k_blob_scalars(const uint8_t* __restrict__ blobs, Fr* __restrict__ scalars, uint32_t* __restrict__ status, int total) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint4* src = reinterpret_cast<const uint4*>(blobs) + 2 * (size_t)t;
    Fr v = fr_from_be_words(src[0], src[1]);
    if (FrF::geq_mod(v)) atomicOr(&status[t / FE_PER_BLOB], 1u);
    scalars[t] = v;
}

// term tables of n_out independent MSMs over the same 4096 points: term t of output b = (point t, scalar b * 4096 + t)
__global__ void k_commit_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int total) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    term_point[t] = t % FE_PER_BLOB;
    term_scalar[t] = t;
}

// Jacobian -> compressed, one point per thread
__global__ void k_jac_compress_n(const G1Jac* __restrict__ p, uint8_t* __restrict__ out, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    G1Aff a;
    bool finite = g1_to_affine(a, p[i]);
    g1_compress(out + 48 * (size_t)i, a, !finite);
}

__global__ void k_g2_generator(Fp* __restrict__ out4) {
    if (threadIdx.x || blockIdx.x) return;
    out4[0] = fp_const(consts::G2_GEN_X0_MONT);
    out4[1] = fp_const(consts::G2_GEN_X1_MONT);
    out4[2] = fp_const(consts::G2_GEN_Y0_MONT);
    out4[3] = fp_const(consts::G2_GEN_Y1_MONT);
}

// re-compress an affine G2 point (x.c1 || x.c0 with flags) - settings round-trip check
__global__ void k_g2_compress(const Fp* __restrict__ in4, uint8_t* __restrict__ out96) {
    if (threadIdx.x || blockIdx.x) return;
    Fp x0 = FpF::from_mont(in4[0]), x1 = FpF::from_mont(in4[1]);
    FpF::to_be_bytes(out96, x1);
    FpF::to_be_bytes(out96 + 48, x0);
    out96[0] |= 0x80;
    bool largest = FpF::is_zero(in4[3]) ? fp_is_lex_largest(in4[2]) : fp_is_lex_largest(in4[3]);
    if (largest) out96[0] |= 0x20;
}

// term tables of the batch equation (msm.hpp) for a launch group of B batches of n blobs (T = B n):
// points: C of all batches [0, T), pi of all batches [T, 2T), generator at 2T; scalars of batch b at b(2n+1).
// blockIdx.y = batch.  Tables are [2B][max_terms], row 2b = output A, row 2b+1 = output B.
__global__ void k_batch_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int n, int T, int max_terms) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, bt = blockIdx.y;
    uint32_t* tpA = term_point + (size_t)(2 * bt) * max_terms;
    uint32_t* tsA = term_scalar + (size_t)(2 * bt) * max_terms;
    uint32_t* tpB = tpA + max_terms;
    uint32_t* tsB = tsA + max_terms;
    const uint32_t sb = (uint32_t)bt * (2 * n + 1);
    if (t < n) {
        tpA[t] = T + bt * n + t;  // (pi_t, a_t) -> A
        tsA[t] = sb + t;
        tpB[t] = T + bt * n + t;  // (pi_t, b_t) -> B
        tsB[t] = sb + n + t;
        tpB[n + t] = bt * n + t;  // (C_t, a_t) -> B
        tsB[n + t] = sb + t;
    }
    if (t == 0) {
        tpB[2 * n] = 2 * T;       // (G, g) -> B
        tsB[2 * n] = sb + 2 * n;
    }
}

// the G1 generator as point `idx` (the -(sum r^i y_i) G term of the batch equation)
__global__ void k_set_generator(G1Aff* __restrict__ points, uint32_t* __restrict__ pflag, int idx) {
    if (threadIdx.x || blockIdx.x) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    points[idx] = g;
    pflag[idx] = 0;
}

// transcript records on the device: out[i] = C_i (48) | z_i (32, LE) | y_i (32, LE) | pi_i (48) as 40 little words
__global__ void k_pack_records(const uint32_t* __restrict__ c, const uint32_t* __restrict__ p, const uint32_t* __restrict__ z,
                               const uint32_t* __restrict__ y, uint32_t* __restrict__ out, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T * 40) return;
    int i = t / 40, k = t % 40;
    out[t] = k < 12 ? c[12 * i + k] : k < 20 ? z[8 * i + k - 12] : k < 28 ? y[8 * i + k - 20] : p[12 * i + k - 28];
}

// the generator and its precomputed multiples (gen_mult[4], made once per settings) as point `idx` of a workspace
__global__ void k_set_generator_multiples(G1Aff* __restrict__ points, uint32_t* __restrict__ pflag, G1Jac* __restrict__ mult,
                                          const G1Jac* __restrict__ gen_mult, int idx, int stride, int chunks) {
    if (threadIdx.x || blockIdx.x) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    points[idx] = g;
    pflag[idx] = 0;
    for (int j = 0; j < chunks; j++) mult[(size_t)j * stride + idx] = gen_mult[j];
}

// plain msm: output 0 over terms (point t, scalar t)
__global__ void k_plain_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        term_point[t] = t;
        term_scalar[t] = t;
    }
}

// n==1 / verify_kzg_proof scalars: a_0 = 1, b_0 = z, g = -y   (r^0 = 1, so no transcript hash is needed); block = batch
__global__ void k_single_scalars(const Fr* __restrict__ z, const Fr* __restrict__ y, Fr* __restrict__ scalars) {
    if (threadIdx.x) return;
    const int bt = blockIdx.x;
    Fr one = FrF::zero();
    one.l[0] = 1;
    scalars[3 * bt] = one;
    scalars[3 * bt + 1] = z[bt];
    scalars[3 * bt + 2] = FrF::from_mont(FrF::neg(FrF::to_mont(y[bt])));
}

// MSM results (Jacobian) -> SLP inputs; the identity is canonicalised to (0, 1, 0)
__global__ void k_jac_to_slp(const G1Jac* __restrict__ ab, Fp* __restrict__ slp_in) {
    int o = threadIdx.x;
    if (o >= 2) return;
    ab += 2 * blockIdx.x;       // block = batch
    slp_in += 6 * blockIdx.x;
    G1Jac p = ab[o];
    if (g1_is_identity(p)) p = g1_identity();
    slp_in[3 * o] = p.x;
    slp_in[3 * o + 1] = p.y;
    slp_in[3 * o + 2] = p.z;
}

// sum `world` partial (A_k, B_k) pairs (multi-GPU fold; src/kzg_proof.rs:433 generalised)
// partials: [world][B][2]; block = batch
__global__ void k_fold_partials(const G1Jac* __restrict__ partials, int world, int B, G1Jac* __restrict__ ab) {
    int o = threadIdx.x;
    if (o >= 2) return;
    const int bt = blockIdx.x;
    G1Jac acc = partials[2 * bt + o];
    for (int k = 1; k < world; k++) acc = g1_add(acc, partials[(size_t)2 * (k * B + bt) + o]);
    ab[2 * bt + o] = acc;
}

// Jacobian -> 48-byte compressed
__global__ void k_jac_compress(const G1Jac* __restrict__ p, uint8_t* __restrict__ out, int count) {
    int i = threadIdx.x;
    if (i >= count || blockIdx.x) return;
    G1Aff a;
    bool finite = g1_to_affine(a, p[i]);
    g1_compress(out + 48 * i, a, !finite);
}

// affine (decoded) -> Jacobian inputs of the pairing program (kzg_pairing_check)
__global__ void k_aff_to_slp(const G1Aff* __restrict__ pts, const uint32_t* __restrict__ flag, Fp* __restrict__ slp_in) {
    int o = threadIdx.x;
    if (o >= 2 || blockIdx.x) return;
    G1Jac p = flag[o] == G1_INFINITY ? g1_identity() : g1_from_affine(pts[o]);
    slp_in[3 * o] = p.x;
    slp_in[3 * o + 1] = p.y;
    slp_in[3 * o + 2] = p.z;
}

// affine -> x || y big-endian (plain)
__global__ void k_aff_to_bytes(const G1Aff* __restrict__ pts, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FpF::to_be_bytes(out + 96 * (size_t)i, FpF::from_mont(pts[i].x));
    FpF::to_be_bytes(out + 96 * (size_t)i + 48, FpF::from_mont(pts[i].y));
}

// out[i] = compress(scalars[i] * G1 generator)  - prover-side helper used to build synthetic
// (commitment, proof) pairs under a known-tau test setup; not on the verification path.
__global__ __launch_bounds__(64) void k_g1_mul_generator(const Fr* __restrict__ scalars, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    Fr k = scalars[i];
    G1Jac acc = g1_identity();
    for (int b = 254; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((k.l[b >> 5] >> (b & 31)) & 1) acc = g1_add_affine(acc, g);
    }
    G1Aff a;
    bool finite = g1_to_affine(a, acc);
    g1_compress(out + 48 * (size_t)i, a, !finite);
}

// ---------------------------------------------------------------- host helpers
static thread_local std::string g_err;
static KzgRet fail(KzgRet rc, const std::string& msg) {
    g_err = msg;
    return rc;
}
extern "C" const char* kzg_last_error(void) { return g_err.c_str(); }

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (void)hipGetLastError();                                                                       \
            return fail(KZG_ERROR, std::string("HIP: ") + hipGetErrorString(e_) + " at " #expr);           \
        }                                                                                                  \
    } while (0)

// event timing that never leaves a sticky HIP error behind (an event may not have been recorded on this path)
static void elapsed(float* out, hipEvent_t a, hipEvent_t b) {
    if (hipEventElapsedTime(out, a, b) != hipSuccess) {
        (void)hipGetLastError();
        *out = 0.f;
    }
}

// SHA-256 (FIPS 180-4) for the batch transcript - host code, independent of the device kernel
namespace hostsha {
alignas(16) static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void block(uint32_t st[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace hostsha
#include <immintrin.h>
namespace hostsha {
// x86 SHA extensions (runtime-detected): the batch transcript is one serial chain, so per-block latency is
// what matters; sha256rnds2 does it at ~1.5 GB/s per core.
__attribute__((target("sha,sse4.1,ssse3"))) static void blocks_ni(uint32_t st[8], const uint8_t* data, size_t nblocks) {
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i TMP = _mm_loadu_si128((const __m128i*)&st[0]);
    __m128i STATE1 = _mm_loadu_si128((const __m128i*)&st[4]);
    TMP = _mm_shuffle_epi32(TMP, 0xB1);
    STATE1 = _mm_shuffle_epi32(STATE1, 0x1B);
    __m128i STATE0 = _mm_alignr_epi8(TMP, STATE1, 8);
    STATE1 = _mm_blend_epi16(STATE1, TMP, 0xF0);
    while (nblocks--) {
        const __m128i ABEF = STATE0, CDGH = STATE1;
        __m128i M[4];
        for (int g = 0; g < 16; g++) {
            if (g < 4) M[g] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * g)), MASK);
            __m128i msg = _mm_add_epi32(M[g & 3], _mm_loadu_si128((const __m128i*)&K[4 * g]));
            STATE1 = _mm_sha256rnds2_epu32(STATE1, STATE0, msg);
            if (g >= 3 && g < 15) {
                __m128i t = _mm_alignr_epi8(M[g & 3], M[(g - 1) & 3], 4);
                M[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(M[(g + 1) & 3], t), M[g & 3]);
            }
            msg = _mm_shuffle_epi32(msg, 0x0E);
            STATE0 = _mm_sha256rnds2_epu32(STATE0, STATE1, msg);
            if (g >= 1 && g < 13) M[(g - 1) & 3] = _mm_sha256msg1_epu32(M[(g - 1) & 3], M[g & 3]);
        }
        STATE0 = _mm_add_epi32(STATE0, ABEF);
        STATE1 = _mm_add_epi32(STATE1, CDGH);
        data += 64;
    }
    TMP = _mm_shuffle_epi32(STATE0, 0x1B);
    STATE1 = _mm_shuffle_epi32(STATE1, 0xB1);
    STATE0 = _mm_blend_epi16(TMP, STATE1, 0xF0);
    STATE1 = _mm_alignr_epi8(STATE1, TMP, 8);
    _mm_storeu_si128((__m128i*)&st[0], STATE0);
    _mm_storeu_si128((__m128i*)&st[4], STATE1);
}
static bool have_ni() {
    static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return v;
}
#else
static bool have_ni() { return false; }
static void blocks_ni(uint32_t*, const uint8_t*, size_t) {}
#endif
static void digest(uint8_t out[32], const uint8_t* data, size_t len) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    if (have_ni()) blocks_ni(st, data, full);
    else
        for (size_t i = 0; i < full; i++) block(st, data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    block(st, tail);
    if (tl == 128) block(st, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}
}  // namespace hostsha

// r (big-endian) and helpers on 32-byte big-endian integers
static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                 0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
static bool be_geq_r(const uint8_t v[32]) { return memcmp(v, R_BE, 32) >= 0; }
static void be_sub_r(uint8_t v[32]) {
    int borrow = 0;
    for (int i = 31; i >= 0; i--) {
        int d = (int)v[i] - R_BE[i] - borrow;
        borrow = d < 0;
        v[i] = (uint8_t)(d + (borrow << 8));
    }
}
static void reverse32(uint8_t* dst, const uint8_t* src) {
    for (int i = 0; i < 32; i++) dst[i] = src[31 - i];
}

// ---------------------------------------------------------------- settings
struct DevProgram {
    SlpProgram p{};
    void* blob = nullptr;  // device copy of the whole program
};

constexpr size_t MAX_WORLD = 64;
constexpr unsigned MSM_MAX_SLICES = 32;
constexpr size_t LATENCY_MAX_BLOBS = 4096;  // launches up to this size: CU-split stream pair + the latency MSM layout
struct Workspace {
    size_t cap_n = 0;       // batch capacity
    size_t cap_b = 0;       // batches-per-group capacity
    size_t pending_n = 0, pending_b = 0, finish_b = 0;  // group currently in flight on this handle
    int chunks = MSM_CHUNKS;                              // MSM layout of the group in flight (msm.hpp)
    size_t off_r = 0, off_part = 0, off_out = 0, off_parts = 0;  // pinned-buffer layout
    size_t cap_stage = 0;   // staged host-input capacity (blobs)
    Fr *d_z = nullptr, *d_y = nullptr, *d_scalars = nullptr, *d_partial = nullptr, *d_r = nullptr;
    uint32_t *d_status = nullptr, *d_pflag = nullptr, *d_term_point = nullptr, *d_term_scalar = nullptr, *d_sorted = nullptr;
    G1Aff* d_points = nullptr;
    G1Jac *d_window = nullptr, *d_window_sl = nullptr, *d_ab = nullptr, *d_mult = nullptr, *d_parts = nullptr;
    Fp *d_slp_in = nullptr, *d_slp_out = nullptr;
    uint8_t *d_stage_blobs = nullptr, *d_stage_cp = nullptr, *d_bytes = nullptr, *d_records = nullptr;
    // pinned host mirrors
    uint8_t* h_buf = nullptr;
    size_t h_cap = 0;
};

struct KzgSettings {
    int device = 0;
    Fr *d_M = nullptr, *d_DM = nullptr;            // roots of unity, 8x32 Montgomery (R, R^2 scalings)
    Fr29Mem *d_M29 = nullptr, *d_DM29 = nullptr;   // the same in radix 2^29 (fr29.hpp), what k_blob_evaluate reads
    Fp* d_tau4 = nullptr;   // [tau]G2 affine (x.c0 x.c1 y.c0 y.c1), Montgomery
    Fp* d_prep = nullptr;   // prepared lines: [tau]G2 then generator (2 * 408 Fp)
    G1Jac* d_gen_mult = nullptr;  // the generator's MSM tables: [0, 4) the default layout, [4, 20) the latency layout (msm.hpp)
    // full trusted setup (kzg_settings_load_trusted_setup only; not needed by verification):
    G1Aff* d_g1 = nullptr;            // g1_points, bit-reversal permuted (build.rs:79,89-105), 4096 entries
    uint32_t* d_g1_flag = nullptr;    // 0 finite / 1 identity (unchecked decode, build.rs:68)
    G1Jac* d_g1_mult = nullptr;       // their MSM multiples (msm.hpp), valid iff g1_in_subgroup
    bool g1_in_subgroup = false;      // every G1 point lies in the r-torsion (what the GLV multiples need)
    Fp* d_g2 = nullptr;               // g2_points (monomial), n_g2 x 4 Fp
    size_t n_g2 = 0;
    uint8_t g1_first[2][48] = {};     // g1_points[0], [1] of the FILE order, for the monomial-form check (build.rs:107-129)
    DevProgram prep, verify;
    // s1 / s2: the two streams the current launch uses (challenge chain | point decode).  They point at the plain pair,
    // or - for a small launch (a single batch) - at a pair confined to disjoint halves of the CUs: the 16 two-wave
    // workgroups of the challenge chain and the 32 decode waves otherwise land on the same first CUs of every XCD and,
    // run to run, share SIMDs (the chain then takes 4.9 ms instead of 3.5 ms).  Measured: one 1 024-blob batch 9.1 ms on
    // the split pair, 10.1-11.5 ms on the plain pair; KZG_CU_MASK=0 disables the split pair.
    mutable hipStream_t s1 = nullptr, s2 = nullptr;
    hipStream_t s_plain[2] = {nullptr, nullptr};
    mutable hipStream_t s_half[2] = {nullptr, nullptr};
    mutable bool s_half_tried = false;
    hipEvent_t ev[12] = {};
    mutable std::mutex mu;
    mutable Workspace ws;
    mutable float timings[8] = {};
};

static KzgRet upload_program(DevProgram& dp, const unsigned char* begin, const unsigned char* end) {
    size_t len = (size_t)(end - begin);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(begin);
    if (len < 64 || w[0] != SLP_MAGIC) return fail(KZG_ERROR, "embedded SLP program is corrupt");
    HIPCHK(hipMalloc(&dp.blob, len));
    HIPCHK(hipMemcpy(dp.blob, begin, len, hipMemcpyHostToDevice));
    SlpProgram& p = dp.p;
    p.lanes = w[1]; p.n_slots = w[2]; p.n_steps = w[3]; p.n_const = w[4]; p.n_in = w[5]; p.n_set = w[6]; p.n_out = w[7];
    const uint32_t* d = reinterpret_cast<const uint32_t*>(dp.blob);
    size_t off = 16;
    p.consts = reinterpret_cast<const Fp*>(d + off);
    off += (size_t)12 * p.n_const;
    p.out_slots = d + off;
    off += p.n_out;
    p.kinds = d + off;
    off += p.n_steps;
    p.desc = reinterpret_cast<const uint2*>(d + off);
    if ((off + (size_t)2 * p.lanes * p.n_steps) * 4 != len) return fail(KZG_ERROR, "embedded SLP program has the wrong size");
    return KZG_OK;
}

static KzgRet run_program(const DevProgram& dp, const Fp* d_in, const Fp* d_set, Fp* d_out, int instances, hipStream_t st) {
    size_t lds = (size_t)dp.p.n_slots * 48 + (size_t)2 * SLP_GROUP * dp.p.lanes * sizeof(uint2);  // slots | descriptor ring
    if (dp.p.lanes == 64) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slp_run<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_slp_run<false>, dim3(instances), dim3(64), lds, st, dp.p, d_in, d_set, d_out);
    } else {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slp_run<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_slp_run<true>, dim3(instances), dim3(dp.p.lanes), lds, st, dp.p, d_in, d_set, d_out);
    }
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

static KzgRet settings_build(KzgSettings* s, const uint8_t tau_g2[96]);
// a handle from g2_points[1]; on any failure everything allocated so far is released
static KzgRet settings_common(KzgSettings** out, const uint8_t tau_g2[96]) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(KZG_ERROR, "no HIP device: this library has no CPU fallback");
    }
    KzgSettings* s = new KzgSettings();
    KzgRet rc = settings_build(s, tau_g2);
    if (rc != KZG_OK) {
        const std::string msg = g_err;  // kzg_settings_free may run HIP calls; keep the first error
        kzg_settings_free(s);
        g_err = msg;
        return rc;
    }
    *out = s;
    return KZG_OK;
}
static KzgRet settings_build(KzgSettings* s, const uint8_t tau_g2[96]) {
    HIPCHK(hipGetDevice(&s->device));
    HIPCHK(hipStreamCreateWithFlags(&s->s_plain[0], hipStreamNonBlocking));
    s->s1 = s->s_plain[0];
    // KZG_SINGLE_STREAM=1 (profiling aid): run the point-decode chain on the same stream as the challenge chain, so
    // per-dispatch PMC counters are not polluted by a concurrent kernel
    if (getenv("KZG_SINGLE_STREAM") && getenv("KZG_SINGLE_STREAM")[0] == '1') s->s2 = s->s1;
    else {
        HIPCHK(hipStreamCreateWithFlags(&s->s_plain[1], hipStreamNonBlocking));
        s->s2 = s->s_plain[1];
    }
    for (auto& e : s->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipMalloc(&s->d_M, sizeof(Fr) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_DM, sizeof(Fr) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_M29, sizeof(Fr29Mem) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_DM29, sizeof(Fr29Mem) * FE_PER_BLOB));
    hipLaunchKernelGGL(k_roots_tables, dim3(FE_PER_BLOB / 64), dim3(64), 0, s->s1, s->d_M, s->d_DM);
    hipLaunchKernelGGL(k_roots_tables29, dim3(FE_PER_BLOB / 64), dim3(64), 0, s->s1, s->d_M, s->d_M29, s->d_DM29);
    HIPCHK(hipGetLastError());
    KzgRet rc;
    if ((rc = upload_program(s->prep, kzg_slp_prep_begin, kzg_slp_prep_end)) != KZG_OK) return rc;
    if ((rc = upload_program(s->verify, kzg_slp_verify_begin, kzg_slp_verify_end)) != KZG_OK) return rc;
    // decompress [tau]G2 on the device, then prepare the lines of [tau]G2 and of the generator
    uint8_t* d_bytes;
    uint32_t* d_flag;
    Fp* d_q;  // 2 instances x 4 Fp
    HIPCHK(hipMalloc(&d_bytes, 96));
    HIPCHK(hipMalloc(&d_flag, 4));
    HIPCHK(hipMalloc(&d_q, sizeof(Fp) * 8));
    HIPCHK(hipMalloc(&s->d_tau4, sizeof(Fp) * 4));
    HIPCHK(hipMalloc(&s->d_prep, sizeof(Fp) * 2 * s->prep.p.n_out));
    HIPCHK(hipMemcpyAsync(d_bytes, tau_g2, 96, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g2_decompress, dim3(1), dim3(64), 0, s->s1, d_bytes, d_q, d_flag);
    hipLaunchKernelGGL(k_g2_generator, dim3(1), dim3(64), 0, s->s1, d_q + 4);
    HIPCHK(hipGetLastError());
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(s->d_tau4, d_q, sizeof(Fp) * 4, hipMemcpyDeviceToDevice, s->s1));
    if ((rc = run_program(s->prep, d_q, nullptr, s->d_prep, 2, s->s1)) != KZG_OK) return rc;
    {  // multiples of the generator (msm.hpp): the same for every batch
        G1Aff* d_g;
        uint32_t* d_gf;
        HIPCHK(hipMalloc(&d_g, sizeof(G1Aff)));
        HIPCHK(hipMalloc(&d_gf, 4));
        HIPCHK(hipMalloc(&s->d_gen_mult, sizeof(G1Jac) * (MSM_CHUNKS + MSM_CHUNKS_LATENCY)));
        hipLaunchKernelGGL(k_set_generator, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, 0);
        hipLaunchKernelGGL(k_g1_multiples, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, s->d_gen_mult, 1, 1, MSM_CHUNKS);
        hipLaunchKernelGGL(k_g1_multiples, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, s->d_gen_mult + MSM_CHUNKS, 1, 1, MSM_CHUNKS_LATENCY);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->s1));
        HIPCHK(hipFree(d_g));
        HIPCHK(hipFree(d_gf));
    }
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d_bytes));
    HIPCHK(hipFree(d_flag));
    HIPCHK(hipFree(d_q));
    if (flag != G1_OK) return fail(KZG_BAD_SETUP, "g2_points[1] is not a valid (finite) compressed G2 point");
    if (s->verify.p.n_set != 2 * s->prep.p.n_out || s->verify.p.n_in != 6 || s->prep.p.n_in != 4)
        return fail(KZG_ERROR, "embedded SLP programs do not fit together");
    return KZG_OK;
}

static int hexnib(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

extern "C" KzgRet kzg_settings_load_trusted_setup(KzgSettings** out, const char* txt, size_t len) {
    if (!out || !txt) return fail(KZG_BADARGS, "null argument");
    // line-oriented parse of build.rs:23-56
    std::vector<std::pair<const char*, size_t>> lines;
    const char *p = txt, *end = txt + len;
    while (p < end) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!q) q = end;
        size_t l = (size_t)(q - p);
        if (l && p[l - 1] == '\r') l--;
        lines.emplace_back(p, l);
        p = q + 1;
    }
    if (lines.size() < 2) return fail(KZG_BAD_SETUP, "trusted setup: missing header lines");
    long n1 = strtol(std::string(lines[0].first, lines[0].second).c_str(), nullptr, 10);
    long n2 = strtol(std::string(lines[1].first, lines[1].second).c_str(), nullptr, 10);
    if (n1 != FE_PER_BLOB) return fail(KZG_BAD_SETUP, "trusted setup: expected 4096 G1 points");
    if (n2 < 2 || (long)lines.size() < 2 + n1 + n2) return fail(KZG_BAD_SETUP, "trusted setup: truncated file");
    // hex -> bytes for every point line (hex_to_bytes, build.rs:15-21: KzgError::InvalidHexFormat)
    auto unhex = [&](uint8_t* dst, const std::pair<const char*, size_t>& ln, size_t nbytes) {
        if (ln.second != 2 * nbytes) return false;
        for (size_t i = 0; i < nbytes; i++) {
            int a = hexnib(ln.first[2 * i]), b = hexnib(ln.first[2 * i + 1]);
            if (a < 0 || b < 0) return false;
            dst[i] = (uint8_t)(a << 4 | b);
        }
        return true;
    };
    std::vector<uint8_t> g1b(48 * (size_t)n1), g2b(96 * (size_t)n2);
    uint8_t first[2][48];
    for (long i = 0; i < n1; i++) {
        // stored bit-reversal permuted (build.rs:79,89-105): file line i -> slot brp(i)
        uint8_t tmp[48];
        if (!unhex(tmp, lines[2 + i], 48)) return fail(KZG_BAD_SETUP, "trusted setup: bad G1 line");
        if (i < 2) memcpy(first[i], tmp, 48);
        uint32_t r = 0;
        for (int k = 0; k < 12; k++) r |= ((uint32_t)(i >> k) & 1u) << (11 - k);
        memcpy(g1b.data() + 48 * (size_t)r, tmp, 48);
    }
    for (long i = 0; i < n2; i++)
        if (!unhex(g2b.data() + 96 * (size_t)i, lines[2 + n1 + i], 96)) return fail(KZG_BAD_SETUP, "trusted setup: bad G2 line");
    KzgRet rc = settings_common(out, g2b.data() + 96);
    if (rc != KZG_OK) return rc;
    KzgSettings* s = *out;
    *out = nullptr;
    auto bail = [&](KzgRet code, const char* msg) {
        kzg_settings_free(s);
        return fail(code, msg);
    };
    memcpy(s->g1_first, first, sizeof first);
    // G1 Lagrange points: unchecked decode (build.rs:66-70) for the table, and the decode + subgroup test + multiples
    // pass of the MSM (msm.hpp) so that commitments can be computed against them
    uint8_t* d_bytes;
    uint32_t *d_flag2, *d_gflag;
    G1Aff* d_tmp;
    const int N = (int)n1;
    HIPCHK(hipMalloc(&d_bytes, std::max(g1b.size(), g2b.size())));
    HIPCHK(hipMalloc(&d_flag2, 4 * (size_t)N));
    HIPCHK(hipMalloc(&d_tmp, sizeof(G1Aff) * (size_t)N));
    HIPCHK(hipMalloc(&s->d_g1, sizeof(G1Aff) * (size_t)N));
    HIPCHK(hipMalloc(&s->d_g1_flag, 4 * (size_t)N));
    HIPCHK(hipMalloc(&s->d_g1_mult, sizeof(G1Jac) * MSM_CHUNKS * (size_t)N));
    HIPCHK(hipMemcpyAsync(d_bytes, g1b.data(), g1b.size(), hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s->s1, d_bytes, d_bytes, N, s->d_g1, s->d_g1_flag, N, 0);
    hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s->s1, d_bytes, d_bytes, N, d_tmp,
                       d_flag2, s->d_g1_mult, N, N);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> f1((size_t)N), f2((size_t)N);
    HIPCHK(hipMemcpyAsync(f1.data(), s->d_g1_flag, 4 * (size_t)N, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(f2.data(), d_flag2, 4 * (size_t)N, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    s->g1_in_subgroup = true;
    for (int i = 0; i < N; i++) {
        if (f1[i] == G1_INVALID) return bail(KZG_BAD_SETUP, "load_trusted_setup Invalid g1 bytes");
        if (f2[i] == G1_INVALID) s->g1_in_subgroup = false;
    }
    // G2 monomial points: all of them decoded (build.rs:72-75); verification itself reads only [1]
    s->n_g2 = (size_t)n2;
    HIPCHK(hipMalloc(&s->d_g2, sizeof(Fp) * 4 * (size_t)n2));
    HIPCHK(hipMalloc(&d_gflag, 4 * (size_t)n2));
    HIPCHK(hipMemcpyAsync(d_bytes, g2b.data(), g2b.size(), hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g2_decompress_n, dim3((unsigned)n2), dim3(64), 0, s->s1, d_bytes, s->d_g2, d_gflag);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> fg((size_t)n2);
    HIPCHK(hipMemcpyAsync(fg.data(), d_gflag, 4 * (size_t)n2, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d_bytes));
    HIPCHK(hipFree(d_flag2));
    HIPCHK(hipFree(d_gflag));
    HIPCHK(hipFree(d_tmp));
    for (long i = 0; i < n2; i++)
        if (fg[(size_t)i] == G1_INVALID) return bail(KZG_BAD_SETUP, "load_trusted_setup Invalid g2 bytes");
    *out = s;
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_from_tau_g2(KzgSettings** out, const uint8_t tau_g2[96]) {
    if (!out || !tau_g2) return fail(KZG_BADARGS, "null argument");
    return settings_common(out, tau_g2);
}

static void ws_free(Workspace& w) {
    void* ptrs[] = {w.d_z, w.d_y, w.d_scalars, w.d_partial, w.d_r, w.d_status, w.d_pflag, w.d_term_point, w.d_term_scalar,
                    w.d_sorted, w.d_points, w.d_window, w.d_window_sl, w.d_ab, w.d_mult, w.d_parts, w.d_slp_in, w.d_slp_out, w.d_stage_blobs, w.d_stage_cp, w.d_bytes,
                    w.d_records};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (w.h_buf) (void)hipHostFree(w.h_buf);
    w = Workspace();
}

extern "C" void kzg_settings_free(KzgSettings* s) {
    if (!s) return;
    ws_free(s->ws);
    void* ptrs[] = {s->d_g1, s->d_g1_flag, s->d_g1_mult, s->d_g2, s->d_M, s->d_DM, s->d_M29, s->d_DM29, s->d_tau4, s->d_prep, s->d_gen_mult, s->prep.blob, s->verify.blob};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {s->s_plain[0], s->s_plain[1], s->s_half[0], s->s_half[1]})
        if (st) (void)hipStreamDestroy(st);
    delete s;
}

// The evaluation kernel over T blobs on stream s1 (radix-2^29 form; KZG_EVALUATE_KERNEL=32 selects the 8x32 form,
// kept for A/B measurement and as a cross-check).
static void launch_evaluate(const KzgSettings* s, const void* d_blobs, const Fr* d_z, Fr* d_y, uint32_t* d_status, size_t T) {
    static const bool use32 = [] {
        const char* e = getenv("KZG_EVALUATE_KERNEL");
        return e && strcmp(e, "32") == 0;
    }();
    if (use32)
        hipLaunchKernelGGL(k_blob_evaluate32, dim3((unsigned)T), dim3(64), 0, s->s1, (const uint8_t*)d_blobs, d_z, s->d_M, s->d_DM, d_y, d_status);
    else
        hipLaunchKernelGGL(k_blob_evaluate, dim3((unsigned)T), dim3(64), 0, s->s1, (const uint8_t*)d_blobs, d_z, s->d_M29, s->d_DM29, d_y, d_status);
}

// The challenge kernel over T blobs on stream s1: the producer/consumer form (half the serial chain, lowest latency)
// while every pair of waves can have a CU to itself, the one-lane-per-blob form (highest throughput) beyond that.
// KZG_CHALLENGE_KERNEL = lane | split forces one of them (A/B measurement, cross-check in the tests).
static KzgRet launch_challenge(const KzgSettings* s, const void* d_blobs, const void* d_commitments, Fr* d_z, size_t T) {
    static const int forced = [] {
        const char* e = getenv("KZG_CHALLENGE_KERNEL");
        return !e ? 0 : strcmp(e, "lane") == 0 ? 1 : strcmp(e, "split") == 0 ? 2 : 0;
    }();
    const uint8_t *bl = (const uint8_t*)d_blobs, *cm = (const uint8_t*)d_commitments;
    const bool lane = forced ? forced == 1 : T > 64 * 256;
    if (lane)
        hipLaunchKernelGGL(k_blob_challenge, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s->s1, bl, cm, d_z, (int)T);
    else
        hipLaunchKernelGGL(k_blob_challenge_split, dim3((unsigned)((T + 63) / 64)), dim3(128), 0, s->s1, bl, cm, d_z, (int)T);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// Workspace for a launch group of B batches with T blobs in total (B = 1 for the single-call entry points).
static KzgRet ws_reserve(const KzgSettings* s, size_t T, size_t B, bool stage) {
    Workspace& w = s->ws;
    if (T > w.cap_n || B > w.cap_b) {
        size_t keep_stage = w.cap_stage;
        uint8_t *sb = w.d_stage_blobs, *sc = w.d_stage_cp;
        w.d_stage_blobs = nullptr;
        w.d_stage_cp = nullptr;
        size_t capT = T > w.cap_n ? T : w.cap_n, capB = B > w.cap_b ? B : w.cap_b;
        ws_free(w);
        w.d_stage_blobs = sb;
        w.d_stage_cp = sc;
        w.cap_stage = keep_stage;
        if (capT < 16) capT = 16;
        const size_t np = 2 * capT + 1;            // points: C's, pi's, generator
        const size_t nsc = 2 * capT + capB;        // scalars: (2n+1) per batch
        const size_t nterm = 4 * capT + 2 * capB;  // term table rows: [2B][2n+1]
        HIPCHK(hipMalloc(&w.d_z, sizeof(Fr) * capT));
        HIPCHK(hipMalloc(&w.d_y, sizeof(Fr) * capT));
        HIPCHK(hipMalloc(&w.d_scalars, sizeof(Fr) * nsc));
        HIPCHK(hipMalloc(&w.d_partial, sizeof(Fr) * ((capT + 255) / 256 + capB)));
        HIPCHK(hipMalloc(&w.d_r, sizeof(Fr) * capB));
        HIPCHK(hipMalloc(&w.d_status, 4 * capT));
        HIPCHK(hipMalloc(&w.d_pflag, 4 * np));
        HIPCHK(hipMalloc(&w.d_term_point, 4 * nterm));
        HIPCHK(hipMalloc(&w.d_term_scalar, 4 * nterm));
        HIPCHK(hipMalloc(&w.d_sorted, 4 * MSM_WINDOWS * nterm));
        HIPCHK(hipMalloc(&w.d_points, sizeof(G1Aff) * np));
        HIPCHK(hipMalloc(&w.d_window, sizeof(G1Jac) * 2 * MSM_WINDOWS * capB));
        HIPCHK(hipMalloc(&w.d_window_sl, sizeof(G1Jac) * 2 * MSM_WINDOWS * MSM_MAX_SLICES * 4));  // sliced launches have <= 4 batches
        HIPCHK(hipMalloc(&w.d_mult, sizeof(G1Jac) * std::max((size_t)MSM_CHUNKS * np, (size_t)MSM_CHUNKS_LATENCY * std::min(np, (size_t)(2 * LATENCY_MAX_BLOBS + 1)))));
        HIPCHK(hipMalloc(&w.d_ab, sizeof(G1Jac) * 2 * capB));
        HIPCHK(hipMalloc(&w.d_parts, sizeof(G1Jac) * 2 * capB * MAX_WORLD));
        HIPCHK(hipMalloc(&w.d_slp_in, sizeof(Fp) * 6 * capB));
        HIPCHK(hipMalloc(&w.d_slp_out, sizeof(Fp) * 6 * capB));
        HIPCHK(hipMalloc(&w.d_bytes, 96 * np));
        HIPCHK(hipMalloc(&w.d_records, 160 * capT));
        w.off_r = 256 * capT + 4096;                 // pinned layout: [per-blob area | r | own partials | out | gathered partials]
        w.off_part = w.off_r + 32 * capB;
        w.off_out = w.off_part + 288 * capB;
        w.off_parts = w.off_out + 288 * capB;
        w.h_cap = w.off_parts + 288 * capB * MAX_WORLD;
        HIPCHK(hipHostMalloc(&w.h_buf, w.h_cap));
        w.cap_n = capT;
        w.cap_b = capB;
    }
    if (stage && T > w.cap_stage) {
        if (w.d_stage_blobs) (void)hipFree(w.d_stage_blobs);
        if (w.d_stage_cp) (void)hipFree(w.d_stage_cp);
        w.d_stage_blobs = w.d_stage_cp = nullptr;
        size_t cap = T < 4 ? 4 : T;
        HIPCHK(hipMalloc(&w.d_stage_blobs, (size_t)BLOB_BYTES * cap));
        HIPCHK(hipMalloc(&w.d_stage_cp, 96 * cap));
        w.cap_stage = cap;
    }
    return KZG_OK;
}

// (window, chunk) blocks of the MSM: separate while the launch has few batches (latency), merged per window once the
// batch dimension alone fills the chip (msm.hpp MsmDesc::chunks_per_block); KZG_MSM_CPB = 1 | 2 | 4 overrides.
static int msm_chunks_per_block(size_t B) {
    static const int forced = [] {
        const char* e = getenv("KZG_MSM_CPB");
        int v = e ? atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4) ? v : 0;
    }();
    if (forced) return forced;
    return B >= 32 ? 4 : B >= 16 ? 2 : 1;
}

// ---------------------------------------------------------------- the tail: MSM + pairing
// Group of B batches of n blobs (T = B n).  scalars of batch b at b(2n+1): a [0,n), b [n,2n), g at 2n;
// points: C [0,T), pi [T,2T), G at 2T, multiples with stride 2T+1.  Leaves (A, B) of batch b in ws.d_ab[2b..].
static KzgRet run_msm(const KzgSettings* s, size_t n, size_t B) {
    Workspace& w = s->ws;
    const int T = (int)(n * B), mt = (int)(2 * n + 1);
    hipLaunchKernelGGL(k_batch_terms, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, s->s1, w.d_term_point,
                       w.d_term_scalar, (int)n, T, mt);
    MsmDesc d{};
    d.mult = w.d_mult;
    d.pflag = w.d_pflag;
    d.scalars = w.d_scalars;
    d.term_point = w.d_term_point;
    d.term_scalar = w.d_term_scalar;
    d.sorted = w.d_sorted;
    d.window_sums = w.d_window;
    d.nterms[0] = (int)n;
    d.nterms[1] = (int)(2 * n + 1);
    d.max_terms = mt;
    d.stride = 2 * T + 1;
    d.chunks = w.chunks;
    d.chunks_per_block = w.chunks == MSM_CHUNKS ? msm_chunks_per_block(B) : 1;
    const unsigned slots = d.chunks / d.chunks_per_block, W = MSM_WINDOWS / d.chunks;
    // one large batch: slice the terms of an output over several workgroups until the launch has ~1000 of them
    // (each slice keeps >= 1024 terms of the smaller output)
    unsigned S = 1;
    while (S < MSM_MAX_SLICES && W * slots * 2 * B * S < 768 && n / (2 * S) >= 1024) S *= 2;
    d.slices = (int)S;
    d.window_sums = S > 1 ? w.d_window_sl : w.d_window;
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    const int nsc = (int)(B * (2 * n + 1));
    hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, s->s1, w.d_scalars, nsc);
    hipLaunchKernelGGL(k_msm_window, dim3(W, slots, (unsigned)(2 * B * S)), dim3(256), 0, s->s1, d);
    if (S > 1)
        hipLaunchKernelGGL(k_msm_fold_slices, dim3((unsigned)(2 * B * slots * W)), dim3(64), 0, s->s1, w.d_window_sl, w.d_window, (int)S, (int)W);
    hipLaunchKernelGGL(k_msm_combine, dim3((unsigned)(2 * B)), dim3(64), 0, s->s1, w.d_window, w.d_ab, (int)slots, (int)W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    return KZG_OK;
}

// decode 2T points (all C then all pi) into ws.d_points / d_pflag together with their 2^(64j) multiples, generator
// (precomputed multiples) as point 2T - all on stream s2, beside the SHA-256 chain
static KzgRet launch_decode(const KzgSettings* s, const void* d_commitments, const void* d_proofs, size_t T) {
    Workspace& w = s->ws;
    const int np = (int)(2 * T + 1);
    unsigned blocks = (unsigned)((2 * T + 63) / 64);
    static const bool no_latency_layout = getenv("KZG_MSM_LATENCY_LAYOUT") && getenv("KZG_MSM_LATENCY_LAYOUT")[0] == '0';
    w.chunks = (T <= LATENCY_MAX_BLOBS && !no_latency_layout) ? MSM_CHUNKS_LATENCY : MSM_CHUNKS;
    const uint8_t *c = (const uint8_t*)d_commitments, *p = (const uint8_t*)d_proofs;
    if (w.chunks == MSM_CHUNKS_LATENCY)
        hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS_LATENCY>, dim3(blocks), dim3(64), 0, s->s2, c, p, (int)T, w.d_points, w.d_pflag,
                           w.d_mult, (int)(2 * T), np);
    else
        hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS>, dim3(blocks), dim3(64), 0, s->s2, c, p, (int)T, w.d_points, w.d_pflag, w.d_mult,
                           (int)(2 * T), np);
    HIPCHK(hipEventRecord(s->ev[10], s->s2));
    hipLaunchKernelGGL(k_set_generator_multiples, dim3(1), dim3(64), 0, s->s2, w.d_points, w.d_pflag, w.d_mult,
                       s->d_gen_mult + (w.chunks == MSM_CHUNKS ? 0 : MSM_CHUNKS), (int)(2 * T), np, w.chunks);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// ---------------------------------------------------------------- the three phases, each as launch + wait
// A handle is a small state machine: phase1_launch -> phase1_wait -> phase2_launch -> phase2_wait ->
// finish_launch -> finish_wait.  Launch halves only enqueue work on the handle's streams (plus the host
// transcript hashes in phase 2); wait halves block on this handle's stream only.  A call processes a launch
// GROUP of B independent batches of n blobs each (own transcript, r, MSM and pairing instance per batch): at
// n = 1024 every phase is a latency-bound serial chain that uses a sliver of the chip, so the batch dimension
// inside the kernels is what fills the machine.

// The stream pair of a launch of T blobs (KzgSettings::s1/s2): the split pair, made on first use, for a small launch.
// Every earlier launch of the handle has been waited for by then, so switching pairs is safe.
static void select_streams(const KzgSettings* s, size_t T) {
    if (!s->s_plain[1]) return;  // KZG_SINGLE_STREAM
    const bool small = T <= LATENCY_MAX_BLOBS;
    if (small && !s->s_half_tried) {
        s->s_half_tried = true;
        const char* e = getenv("KZG_CU_MASK");
        hipDeviceProp_t prop;
        if (!(e && e[0] == '0') && hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount >= 64) {
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> lo(words, 0), hi(words, 0);
            for (int i = 0; i < ncu; i++) ((i < ncu / 2) ? lo : hi)[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&s->s_half[0], words, lo.data()) != hipSuccess ||
                hipExtStreamCreateWithCUMask(&s->s_half[1], words, hi.data()) != hipSuccess) {
                (void)hipGetLastError();
                if (s->s_half[0]) (void)hipStreamDestroy(s->s_half[0]);
                s->s_half[0] = s->s_half[1] = nullptr;
            }
        }
    }
    const bool use_half = small && s->s_half[0];
    s->s1 = use_half ? s->s_half[0] : s->s_plain[0];
    s->s2 = use_half ? s->s_half[1] : s->s_plain[1];
}

// Phase 1 (no communication): point decode + multiples || (challenge -> evaluate) for all T = B n blobs.
static KzgRet phase1_launch_locked(const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n, size_t B,
                                   const KzgSettings* s) {
    Workspace& w = s->ws;
    const size_t T = n * B;
    KzgRet rc;
    select_streams(s, T);
    HIPCHK(hipEventRecord(s->ev[0], s->s1));
    HIPCHK(hipStreamWaitEvent(s->s2, s->ev[0], 0));
    HIPCHK(hipEventRecord(s->ev[5], s->s2));
    if ((rc = launch_decode(s, d_commitments, d_proofs, T)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[6], s->s2));
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4 * T, s->s1));
    if ((rc = launch_challenge(s, d_blobs, d_commitments, w.d_z, T)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[7], s->s1));
    launch_evaluate(s, d_blobs, w.d_z, w.d_y, w.d_status, T);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[8], s->s1));
    HIPCHK(hipStreamWaitEvent(s->s1, s->ev[6], 0));
    HIPCHK(hipEventRecord(s->ev[1], s->s1));
    // the transcript records, packed on the device; pinned host mirror: [records 160 T | status 4 T | point flags 8 T]
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((40 * T + 255) / 256)), dim3(256), 0, s->s1, (const uint32_t*)d_commitments,
                       (const uint32_t*)d_proofs, (const uint32_t*)w.d_z, (const uint32_t*)w.d_y, (uint32_t*)w.d_records, (int)T);
    HIPCHK(hipGetLastError());
    uint8_t* h = w.h_buf;
    uint32_t* h_status = reinterpret_cast<uint32_t*>(h + 160 * T);
    uint32_t* h_pflag = h_status + T;
    HIPCHK(hipMemcpyAsync(h, w.d_records, 160 * T, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(h_status, w.d_status, 4 * T, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(h_pflag, w.d_pflag, 8 * T, hipMemcpyDeviceToHost, s->s1));
    w.pending_n = n;
    w.pending_b = B;
    return KZG_OK;
}

// records_out (optional): [B][n] x 160 bytes  C(48) || z(32, LE) || y(32, LE) || pi(48) - exactly the per-blob slices
// of the batch transcripts of src/kzg_proof.rs:314-334.  The handle keeps them (pinned host + device) for phase 2.  bad_out (optional, B bytes): 1 where a batch holds an invalid input
// (then the call still returns KZG_OK); without bad_out any invalid input makes the whole call return KZG_BADARGS.
static KzgRet phase1_wait_locked(uint8_t* records_out, uint8_t* bad_out, const KzgSettings* s) {
    Workspace& w = s->ws;
    const size_t n = w.pending_n, B = w.pending_b, T = n * B;
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[1], s->ev[0], s->ev[1]);
    elapsed(&s->timings[4], s->ev[7], s->ev[8]);
    elapsed(&s->timings[5], s->ev[0], s->ev[7]);
    elapsed(&s->timings[6], s->ev[5], s->ev[10]);
    elapsed(&s->timings[7], s->ev[10], s->ev[6]);
    uint8_t* h = w.h_buf;
    uint32_t* h_status = reinterpret_cast<uint32_t*>(h + 160 * T);
    uint32_t* h_pflag = h_status + T;
    // error order of the reference: commitments (:503), proofs (:508), then blobs (:263); all map to BadArgs
    bool any_bad = false;
    for (size_t b = 0; b < B; b++) {
        bool bad = false;
        for (size_t i = b * n; i < (b + 1) * n; i++) bad |= h_pflag[i] == G1_INVALID || h_pflag[T + i] == G1_INVALID || h_status[i] != 0;
        if (bad_out) bad_out[b] = bad;
        any_bad |= bad;
    }
    if (any_bad && !bad_out) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) src/kzg_proof.rs:19-23,38-40
    if (records_out) memcpy(records_out, h, 160 * T);  // the device limb arrays ARE Scalar::to_bytes() (little-endian), :321,:326
    return KZG_OK;
}

// Phase 2: per batch b, r_b from its FULL transcript, this shard's scalars r_b^(offset+i) and its partial sums (A, B)_b.
// Requires phase 1 of the same group on this handle.  The records come in one of three layouts:
//   all_records != NULL, world == 0 : [B][n_total]          every batch's records in global blob order
//   all_records != NULL, world  > 0 : [world][B][n]         as an all-gather of equal shards leaves them (n_total = world n)
//   all_records == NULL             : the handle's own records of phase 1 (single rank: n_total = n)
static KzgRet phase2_launch_locked(const uint8_t* all_records, size_t n_total, size_t offset, const KzgSettings* s, size_t world = 0) {
    Workspace& w = s->ws;
    const size_t n = w.pending_n, B = w.pending_b;
    if (!all_records) {
        if (n_total != n || offset != 0) return fail(KZG_BADARGS, "local phase 2 needs n_total == n_local");
        all_records = w.h_buf;
        world = 0;
    }
    if (world && n_total != world * n) return fail(KZG_BADARGS, "gathered phase 2 needs equal shards");
    if (n_total == 1) {
        // verify_blob_kzg_proof path (:482-489): r^0 = 1, no batch challenge
        hipLaunchKernelGGL(k_single_scalars, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_z, w.d_y, w.d_scalars);
    } else {
        // compute_r_powers :291-348, once per batch.  The transcripts are hashed on the host (SHA-NI): one serial chain
        // of 160 n_total + 32 bytes per batch - hopeless on a GPU lane, ~2 GB/s on a CPU core - and the batches of a
        // launch group are independent, so they are spread over a few host threads (KZG_HOST_THREADS, default 16).
        auto digest_range = [&](size_t b0, size_t b1) {
            std::vector<uint8_t> t(32 + 160 * n_total);
            memcpy(t.data(), "RCKZGBATCH___V1_", 16);
            memset(t.data() + 16, 0, 16);
            t[22] = (uint8_t)(FE_PER_BLOB >> 8);
            t[23] = (uint8_t)(FE_PER_BLOB & 0xff);
            for (int k = 0; k < 8; k++) t[24 + k] = (uint8_t)((uint64_t)n_total >> (56 - 8 * k));
            for (size_t b = b0; b < b1; b++) {
                if (world == 0) memcpy(t.data() + 32, all_records + 160 * n_total * b, 160 * n_total);
                else
                    for (size_t k = 0; k < world; k++) memcpy(t.data() + 32 + 160 * n * k, all_records + 160 * n * (k * B + b), 160 * n);
                uint8_t dg[32];
                hostsha::digest(dg, t.data(), t.size());
                while (be_geq_r(dg)) be_sub_r(dg);  // digest mod r: at most two subtractions (2^256 < 3r)
                reverse32(w.h_buf + w.off_r + 32 * b, dg);  // pinned staging for the async H2D copy
            }
        };
        static const size_t host_threads = [] {
            const char* e = getenv("KZG_HOST_THREADS");
            long v = e ? atol(e) : 16;
            return (size_t)(v < 1 ? 1 : v > 64 ? 64 : v);
        }();
        const size_t nthr = std::min(host_threads, std::min(B, (B * 160 * n_total) / (512 * 1024) + 1));
        if (nthr <= 1) {
            digest_range(0, B);
        } else {
            std::vector<std::thread> pool;
            for (size_t k = 1; k < nthr; k++) pool.emplace_back(digest_range, B * k / nthr, B * (k + 1) / nthr);
            digest_range(0, B / nthr);
            for (auto& th : pool) th.join();
        }
        HIPCHK(hipMemcpyAsync(w.d_r, w.h_buf + w.off_r, 32 * B, hipMemcpyHostToDevice, s->s1));
        unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_batch_scalars, dim3(blocks, (unsigned)B), dim3(256), 0, s->s1, w.d_r, w.d_z, w.d_y, w.d_scalars,
                           w.d_partial, (int)n, (unsigned long long)offset);
        hipLaunchKernelGGL(k_finish_g, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_partial, (int)blocks, w.d_scalars, (int)n);
    }
    HIPCHK(hipGetLastError());
    KzgRet rc = run_msm(s, n, B);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipMemcpyAsync(w.h_buf + w.off_part, w.d_ab, 288 * B, hipMemcpyDeviceToHost, s->s1));
    return KZG_OK;
}

static KzgRet phase2_wait_locked(uint8_t* partial_out /* B x 288 */, const KzgSettings* s) {
    Workspace& w = s->ws;
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    if (partial_out) memcpy(partial_out, w.h_buf + w.off_part, 288 * w.pending_b);
    return KZG_OK;
}

// Finish: fold `world` partial sets ([world][B] x 288 B), or take the handle's own (A, B)_b when partials == nullptr,
// and run one pairing instance per batch.
static KzgRet finish_launch_locked(const uint8_t* partials, size_t world, size_t B, const KzgSettings* s) {
    Workspace& w = s->ws;
    if (partials) {
        if (world > MAX_WORLD) return fail(KZG_BADARGS, "world size above 64");
        memcpy(w.h_buf + w.off_parts, partials, 288 * world * B);
        HIPCHK(hipMemcpyAsync(w.d_parts, w.h_buf + w.off_parts, 288 * world * B, hipMemcpyHostToDevice, s->s1));
        hipLaunchKernelGGL(k_fold_partials, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_parts, (int)world, (int)B, w.d_ab);
    }
    hipLaunchKernelGGL(k_jac_to_slp, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_ab, w.d_slp_in);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[4], s->s1));
    KzgRet rc = run_program(s->verify, w.d_slp_in, s->d_prep, w.d_slp_out, (int)B, s->s1);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[9], s->s1));
    HIPCHK(hipMemcpyAsync(w.h_buf + w.off_out, w.d_slp_out, sizeof(Fp) * 6 * B, hipMemcpyDeviceToHost, s->s1));
    w.finish_b = B;
    return KZG_OK;
}

static KzgRet finish_wait_locked(bool* ok /* B */, const KzgSettings* s) {
    Workspace& w = s->ws;
    HIPCHK(hipStreamSynchronize(s->s1));
    const uint32_t* h = reinterpret_cast<const uint32_t*>(w.h_buf + w.off_out);
    for (size_t b = 0; b < w.finish_b; b++) {
        uint32_t any = 0;
        for (int i = 0; i < 72; i++) any |= h[72 * b + i];
        ok[b] = any == 0;
    }
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    elapsed(&s->timings[3], s->ev[4], s->ev[9]);
    elapsed(&s->timings[0], s->ev[0], s->ev[9]);
    return KZG_OK;
}

static KzgRet batch_device_locked(bool* ok, const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n,
                                  const KzgSettings* s) {
    KzgRet rc;
    if ((rc = phase1_launch_locked(d_blobs, d_commitments, d_proofs, n, 1, s)) != KZG_OK) return rc;
    if ((rc = phase1_wait_locked(nullptr, nullptr, s)) != KZG_OK) return rc;
    if ((rc = phase2_launch_locked(nullptr, n, 0, s)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, 1, s)) != KZG_OK) return rc;  // same stream: no host round trip needed
    return finish_wait_locked(ok, s);
}

// ---- multi-GPU / grouped / pipelined entry points (include/kzg_rs_amd.h) ----
#define KZG_ENTER(cond)                                          \
    if (!(cond)) return fail(KZG_BADARGS, "bad argument");       \
    std::lock_guard<std::mutex> lk(s->mu);                       \
    HIPCHK(hipSetDevice(s->device));

extern "C" KzgRet kzg_shard_phase1_launch(const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n_local,
                                          size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && d_blobs && d_commitments && d_proofs && n_local && n_batches);
    KzgRet rc = ws_reserve(s, n_local * n_batches, n_batches, false);
    if (rc != KZG_OK) return rc;
    return phase1_launch_locked(d_blobs, d_commitments, d_proofs, n_local, n_batches, s);
}
extern "C" KzgRet kzg_shard_phase1_wait(uint8_t* records_out, uint8_t* bad_out, const KzgSettings* s) {
    KZG_ENTER(s && s->ws.pending_n);
    return phase1_wait_locked(records_out, bad_out, s);
}
extern "C" KzgRet kzg_shard_phase2_launch(const uint8_t* all_records, size_t n_total, size_t offset, const KzgSettings* s) {
    KZG_ENTER(s && s->ws.pending_n && offset + s->ws.pending_n <= n_total);
    return phase2_launch_locked(all_records, n_total, offset, s);
}
extern "C" KzgRet kzg_shard_phase2_launch_gathered(const uint8_t* gathered, size_t world, size_t rank, const KzgSettings* s) {
    KZG_ENTER(s && gathered && s->ws.pending_n && world && rank < world);
    return phase2_launch_locked(gathered, world * s->ws.pending_n, rank * s->ws.pending_n, s, world);
}
extern "C" KzgRet kzg_shard_records_device(void* d_records_out, const KzgSettings* s) {
    KZG_ENTER(s && d_records_out && s->ws.pending_n);
    HIPCHK(hipMemcpyAsync(d_records_out, s->ws.d_records, 160 * s->ws.pending_n * s->ws.pending_b, hipMemcpyDeviceToDevice, s->s1));
    return KZG_OK;
}
extern "C" KzgRet kzg_shard_phase2_wait(uint8_t* partial_out, const KzgSettings* s) {
    KZG_ENTER(s && partial_out);
    return phase2_wait_locked(partial_out, s);
}
extern "C" KzgRet kzg_shard_finish_launch(const uint8_t* partials, size_t world, size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && n_batches && (partials ? world > 0 : true));  // partials == NULL: pair this handle's own sums (single rank)
    KzgRet rc = ws_reserve(s, 2 * n_batches, n_batches, false);
    if (rc != KZG_OK) return rc;
    return finish_launch_locked(partials, world, n_batches, s);
}
extern "C" KzgRet kzg_shard_finish_wait(bool* ok, const KzgSettings* s) {
    KZG_ENTER(s && ok);
    return finish_wait_locked(ok, s);
}
// blocking single-batch forms
extern "C" KzgRet kzg_shard_phase1(uint8_t* records_out, const void* d_blobs, const void* d_commitments, const void* d_proofs,
                                   size_t n_local, const KzgSettings* s) {
    KzgRet rc = kzg_shard_phase1_launch(d_blobs, d_commitments, d_proofs, n_local, 1, s);
    return rc != KZG_OK ? rc : kzg_shard_phase1_wait(records_out, nullptr, s);
}
extern "C" KzgRet kzg_shard_phase2(uint8_t partial_out[288], const uint8_t* all_records, size_t n_total, size_t offset,
                                   size_t n_local, const KzgSettings* s) {
    if (s && n_local != s->ws.pending_n) return fail(KZG_BADARGS, "kzg_shard_phase2 without a matching kzg_shard_phase1");
    KzgRet rc = kzg_shard_phase2_launch(all_records, n_total, offset, s);
    return rc != KZG_OK ? rc : kzg_shard_phase2_wait(partial_out, s);
}
extern "C" KzgRet kzg_shard_finish(bool* ok, const uint8_t* partials, size_t world, const KzgSettings* s) {
    KzgRet rc = kzg_shard_finish_launch(partials, world, 1, s);
    return rc != KZG_OK ? rc : kzg_shard_finish_wait(ok, s);
}

// B independent batches of n blobs each in ONE launch group: blobs / commitments / proofs are contiguous device
// arrays of B*n entries, batch b = entries [b n, (b+1) n); ok_out[b] and (optional) err_out[b] per batch.
extern "C" KzgRet kzg_verify_blob_kzg_proof_batches_device(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments,
                                                           const void* d_proofs, size_t n, size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && ok_out && d_blobs && d_commitments && d_proofs && n && n_batches);
    KzgRet rc = ws_reserve(s, n * n_batches, n_batches, false);
    if (rc != KZG_OK) return rc;
    if ((rc = phase1_launch_locked(d_blobs, d_commitments, d_proofs, n, n_batches, s)) != KZG_OK) return rc;
    if ((rc = phase1_wait_locked(nullptr, err_out, s)) != KZG_OK) return rc;
    if ((rc = phase2_launch_locked(nullptr, n, 0, s)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, n_batches, s)) != KZG_OK) return rc;
    if ((rc = finish_wait_locked(ok_out, s)) != KZG_OK) return rc;
    if (err_out)
        for (size_t b = 0; b < n_batches; b++)
            if (err_out[b]) ok_out[b] = false;
    return KZG_OK;
}

extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_device(bool* ok, const void* d_blobs, const void* d_commitments,
                                                         const void* d_proofs, size_t n, const KzgSettings* s) {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {  // src/kzg_proof.rs:478-480
        *ok = true;
        return KZG_OK;
    }
    if (!d_blobs || !d_commitments || !d_proofs) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    KzgRet rc = ws_reserve(s, n, 1, false);
    if (rc != KZG_OK) return rc;
    return batch_device_locked(ok, d_blobs, d_commitments, d_proofs, n, s);
}

extern "C" KzgRet kzg_verify_blob_kzg_proof_batch(bool* ok, const uint8_t* blobs, const uint8_t* commitments,
                                                  const uint8_t* proofs, size_t n, const KzgSettings* s) {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {
        *ok = true;
        return KZG_OK;
    }
    if (!blobs || !commitments || !proofs) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    KzgRet rc = ws_reserve(s, n, 1, true);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    select_streams(s, n);  // before the staging copies: they must be on the stream the kernels of this launch run on
    HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blobs, (size_t)BLOB_BYTES * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_stage_cp, commitments, 48 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_stage_cp + 48 * n, proofs, 48 * n, hipMemcpyHostToDevice, s->s1));
    return batch_device_locked(ok, w.d_stage_blobs, w.d_stage_cp, w.d_stage_cp + 48 * n, n, s);
}

extern "C" KzgRet kzg_verify_blob_kzg_proof(bool* ok, const uint8_t* blob, const uint8_t commitment[48], const uint8_t proof[48],
                                            const KzgSettings* s) {
    // src/kzg_proof.rs:446-470; the batch entry's n == 1 branch is this very function (:482-489)
    return kzg_verify_blob_kzg_proof_batch(ok, blob, commitment, proof, 1, s);
}

extern "C" KzgRet kzg_verify_kzg_proof_batch(bool* ok, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                             const uint8_t* proofs, size_t n, const KzgSettings* s);
extern "C" KzgRet kzg_verify_kzg_proof(bool* ok, const uint8_t commitment[48], const uint8_t z[32], const uint8_t y[32],
                                       const uint8_t proof[48], const KzgSettings* s) {
    // src/kzg_proof.rs:353-397: the same equation as the batch form with the single scalar r^0 = 1:
    // e(pi, [tau]G2) == e(C - [y]G + [z]pi, G2)  <=>  e(pi, [tau - z]G2) == e(C - [y]G, G2)
    if (!commitment || !z || !y || !proof) return fail(KZG_BADARGS, "null argument");
    return kzg_verify_kzg_proof_batch(ok, commitment, z, y, proof, 1, s);
}

// KzgProof::verify_kzg_proof_batch (src/kzg_proof.rs:399-444) over byte inputs: n (commitment, z, y, proof) tuples checked
// with ONE random linear combination and ONE pairing.  Same pipeline as the blob batch minus challenge + evaluation.
extern "C" KzgRet kzg_verify_kzg_proof_batch(bool* ok, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                             const uint8_t* proofs, size_t n, const KzgSettings* s) {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {  // compute_r_powers on an empty batch: both MSMs are the identity, e(O, .) == e(O, .)
        *ok = true;
        return KZG_OK;
    }
    if (!commitments || !zs || !ys || !proofs) return fail(KZG_BADARGS, "null argument");
    for (size_t i = 0; i < n; i++)
        if (be_geq_r(zs + 32 * i) || be_geq_r(ys + 32 * i)) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, n);
    KzgRet rc = ws_reserve(s, n, 1, true);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    HIPCHK(hipEventRecord(s->ev[0], s->s1));
    // z, y: big-endian -> the device's little-endian limb arrays (= the transcript's encoding)
    std::vector<uint8_t> records(160 * n);
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = records.data() + 160 * i;
        memcpy(o, commitments + 48 * i, 48);
        reverse32(o + 48, zs + 32 * i);
        reverse32(o + 80, ys + 32 * i);
        memcpy(o + 112, proofs + 48 * i, 48);
        memcpy(w.h_buf + 32 * i, o + 48, 32);
        memcpy(w.h_buf + 32 * n + 32 * i, o + 80, 32);
    }
    HIPCHK(hipMemcpyAsync(w.d_z, w.h_buf, 32 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_y, w.h_buf + 32 * n, 32 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_stage_cp, commitments, 48 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_stage_cp + 48 * n, proofs, 48 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    if ((rc = launch_decode(s, w.d_stage_cp, w.d_stage_cp + 48 * n, n)) != KZG_OK) return rc;
    uint32_t* h_pflag = reinterpret_cast<uint32_t*>(w.h_buf + 64 * n);
    HIPCHK(hipMemcpyAsync(h_pflag, w.d_pflag, 8 * n, hipMemcpyDeviceToHost, s->s2));
    HIPCHK(hipStreamSynchronize(s->s2));
    for (size_t i = 0; i < 2 * n; i++)
        if (h_pflag[i] == G1_INVALID) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    w.pending_n = n;
    w.pending_b = 1;
    // n == 1: r^0 = 1 whatever the transcript hashes to, which is phase 2's n_total == 1 branch (scalars 1, z, -y)
    if ((rc = phase2_launch_locked(records.data(), n, 0, s)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, 1, s)) != KZG_OK) return rc;
    return finish_wait_locked(ok, s);
}

// ---------------------------------------------------------------- pieces
extern "C" KzgRet kzg_compute_challenges(uint8_t* z_out, const uint8_t* blobs, const uint8_t* commitments, size_t n,
                                         const KzgSettings* s) {
    if (!s || !z_out || !blobs || !commitments) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, true);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    select_streams(s, n);  // before the staging copies: they must be on the stream the kernels of this launch run on
    HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blobs, (size_t)BLOB_BYTES * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_stage_cp, commitments, 48 * n, hipMemcpyHostToDevice, s->s1));
    if ((rc = launch_challenge(s, w.d_stage_blobs, w.d_stage_cp, w.d_z, n)) != KZG_OK) return rc;
    HIPCHK(hipMemcpyAsync(w.h_buf, w.d_z, 32 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    for (size_t i = 0; i < n; i++) reverse32(z_out + 32 * i, w.h_buf + 32 * i);
    return KZG_OK;
}

static KzgRet evaluate_device_locked(void* d_y, const void* d_blobs, const void* d_z, size_t n, const KzgSettings* s,
                                     bool* any_bad) {
    Workspace& w = s->ws;
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4 * n, s->s1));
    HIPCHK(hipEventRecord(s->ev[7], s->s1));
    launch_evaluate(s, d_blobs, (const Fr*)d_z, (Fr*)d_y, w.d_status, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[8], s->s1));
    uint32_t* h_status = reinterpret_cast<uint32_t*>(w.h_buf + 64 * n);
    HIPCHK(hipMemcpyAsync(h_status, w.d_status, 4 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[4], s->ev[7], s->ev[8]);
    *any_bad = false;
    for (size_t i = 0; i < n; i++) *any_bad |= h_status[i] != 0;
    return KZG_OK;
}

extern "C" KzgRet kzg_evaluate_polynomials_device(void* d_y, const void* d_blobs, const void* d_z, size_t n, const KzgSettings* s) {
    if (!s || !d_y || !d_blobs || !d_z) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, false);
    if (rc != KZG_OK) return rc;
    bool bad = false;
    if ((rc = evaluate_device_locked(d_y, d_blobs, d_z, n, s, &bad)) != KZG_OK) return rc;
    return bad ? fail(KZG_BADARGS, "Failed to parse G1Affine from bytes") : KZG_OK;
}

extern "C" KzgRet kzg_evaluate_polynomials(uint8_t* ys_out, const uint8_t* blobs, const uint8_t* zs, size_t n, const KzgSettings* s) {
    if (!s || !ys_out || !blobs || !zs) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, true);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    for (size_t i = 0; i < n; i++) reverse32(w.h_buf + 32 * i, zs + 32 * i);
    HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blobs, (size_t)BLOB_BYTES * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_z, w.h_buf, 32 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    bool bad = false;
    if ((rc = evaluate_device_locked(w.d_y, w.d_stage_blobs, w.d_z, n, s, &bad)) != KZG_OK) return rc;
    if (bad) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    HIPCHK(hipMemcpy(w.h_buf, w.d_y, 32 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) reverse32(ys_out + 32 * i, w.h_buf + 32 * i);
    return KZG_OK;
}

extern "C" KzgRet kzg_g1_decompress(uint8_t* status_out, uint8_t* xy_out, const uint8_t* points48, size_t n, const KzgSettings* s) {
    if (!s || !status_out || !points48) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, (n + 1) / 2 + 1, 1, false);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    HIPCHK(hipMemcpyAsync(w.d_bytes, points48, 48 * n, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, (int)n, w.d_points, w.d_pflag, (int)n, 1);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> st(n);
    HIPCHK(hipMemcpyAsync(st.data(), w.d_pflag, 4 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    for (size_t i = 0; i < n; i++) status_out[i] = (uint8_t)st[i];
    if (xy_out) {
        uint8_t* d_xy;
        HIPCHK(hipMalloc(&d_xy, 96 * n));
        hipLaunchKernelGGL(k_aff_to_bytes, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_points, d_xy, (int)n);
        HIPCHK(hipMemcpyAsync(xy_out, d_xy, 96 * n, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        HIPCHK(hipFree(d_xy));
    }
    return KZG_OK;
}

extern "C" KzgRet kzg_g1_msm(uint8_t out[48], const uint8_t* points48, const uint8_t* scalars, size_t n, const KzgSettings* s) {
    if (!s || !out || (n && (!points48 || !scalars))) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, (n + 1) / 2 + 1, 1, false);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    int mt = (int)(n ? n : 1);
    // scalars: big-endian, reduced mod r on the host (at most two subtractions), little-endian limbs on the device
    std::vector<uint8_t> le(32 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        uint8_t t[32];
        memcpy(t, scalars + 32 * i, 32);
        while (be_geq_r(t)) be_sub_r(t);
        reverse32(le.data() + 32 * i, t);
    }
    if (n) {
        HIPCHK(hipMemcpyAsync(w.d_bytes, points48, 48 * n, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemcpyAsync(w.d_scalars, le.data(), 32 * n, hipMemcpyHostToDevice, s->s1));
        hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, (int)n, w.d_points, w.d_pflag, (int)n, 1);
        hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->s1, w.d_scalars, (int)n);
        hipLaunchKernelGGL(k_plain_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->s1, w.d_term_point, w.d_term_scalar, (int)n);
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> st(n);
        HIPCHK(hipMemcpyAsync(st.data(), w.d_pflag, 4 * n, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        for (size_t i = 0; i < n; i++)
            if (st[i] == G1_INVALID) return fail(KZG_BADARGS, "invalid G1 point");
    }
    if (n) hipLaunchKernelGGL(k_g1_multiples, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_points, w.d_pflag, w.d_mult, (int)n, mt, MSM_CHUNKS);
    MsmDesc d{};
    d.mult = w.d_mult;
    d.pflag = w.d_pflag;
    d.scalars = w.d_scalars;
    d.term_point = w.d_term_point;
    d.term_scalar = w.d_term_scalar;
    d.sorted = w.d_sorted;
    d.window_sums = w.d_window;
    d.nterms[0] = (int)n;
    d.nterms[1] = 0;
    d.max_terms = mt;
    d.stride = mt;
    unsigned S = 1;  // slice a large MSM over more workgroups (msm.hpp MsmDesc::slices)
    while (S < MSM_MAX_SLICES && 8 * MSM_CHUNKS * S < 768 && n / (2 * S) >= 1024) S *= 2;
    d.slices = (int)S;
    d.window_sums = S > 1 ? w.d_window_sl : w.d_window;
    d.chunks = MSM_CHUNKS;
    d.chunks_per_block = 1;
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    hipLaunchKernelGGL(k_msm_window, dim3(8, MSM_CHUNKS, S), dim3(256), 0, s->s1, d);
    if (S > 1) hipLaunchKernelGGL(k_msm_fold_slices, dim3(MSM_CHUNKS * 8), dim3(64), 0, s->s1, w.d_window_sl, w.d_window, (int)S, 8);
    hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, s->s1, w.d_window, w.d_ab, MSM_CHUNKS, 8);
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    hipLaunchKernelGGL(k_jac_compress, dim3(1), dim3(64), 0, s->s1, w.d_ab, w.d_bytes, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, w.d_bytes, 48, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    return KZG_OK;
}

extern "C" KzgRet kzg_g1_mul_generator(uint8_t* out48, const uint8_t* scalars, size_t n, const KzgSettings* s) {
    if (!s || (n && (!out48 || !scalars))) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    std::vector<uint8_t> le(32 * n);
    for (size_t i = 0; i < n; i++) {
        uint8_t t[32];
        memcpy(t, scalars + 32 * i, 32);
        while (be_geq_r(t)) be_sub_r(t);
        reverse32(le.data() + 32 * i, t);
    }
    Fr* d_s;
    uint8_t* d_o;
    HIPCHK(hipMalloc(&d_s, 32 * n));
    HIPCHK(hipMalloc(&d_o, 48 * n));
    HIPCHK(hipMemcpyAsync(d_s, le.data(), 32 * n, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_mul_generator, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, d_s, d_o, (int)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out48, d_o, 48 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d_s));
    HIPCHK(hipFree(d_o));
    return KZG_OK;
}

extern "C" KzgRet kzg_pairing_check(bool* ok, const uint8_t a[48], const uint8_t b[48], const KzgSettings* s) {
    if (!ok || !a || !b || !s) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, 2, 1, false);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    HIPCHK(hipMemcpyAsync(w.d_bytes, a, 48, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_bytes + 48, b, 48, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3(1), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, 2, w.d_points, w.d_pflag, 2, 0);
    hipLaunchKernelGGL(k_aff_to_slp, dim3(1), dim3(64), 0, s->s1, w.d_points, w.d_pflag, w.d_slp_in);
    HIPCHK(hipGetLastError());
    uint32_t* h = reinterpret_cast<uint32_t*>(w.h_buf);
    HIPCHK(hipMemcpyAsync(h, w.d_pflag, 8, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    if ((rc = run_program(s->verify, w.d_slp_in, s->d_prep, w.d_slp_out, 1, s->s1)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[4], s->s1));
    HIPCHK(hipMemcpyAsync(h + 2, w.d_slp_out, sizeof(Fp) * 6, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[3], s->ev[3], s->ev[4]);
    if (h[0] == G1_INVALID || h[1] == G1_INVALID) return fail(KZG_BADARGS, "invalid G1 point");
    uint32_t any = 0;
    for (int i = 0; i < 72; i++) any |= h[2 + i];
    *ok = any == 0;
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_root_of_unity(const KzgSettings* s, size_t i, uint8_t out[32]) {
    if (!s || !out || i >= FE_PER_BLOB) return fail(KZG_BADARGS, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    // the table holds w*R mod r; strip the Montgomery factor with one host-side REDC (test/diagnostic path only)
    Fr m;
    HIPCHK(hipMemcpy(&m, s->d_M + i, sizeof(Fr), hipMemcpyDeviceToHost));
    // host Montgomery reduction of one element: t = m * R^-1 mod r with 64-bit arithmetic
    const uint32_t* MOD = consts::FR_MOD;
    uint32_t t[9] = {0};
    for (int k = 0; k < 8; k++) t[k] = m.l[k];
    for (int k = 0; k < 8; k++) {
        uint32_t q = t[0] * FR_INV32;
        uint64_t c = ((uint64_t)q * MOD[0] + t[0]) >> 32;
        for (int j = 1; j < 8; j++) {
            uint64_t x = (uint64_t)q * MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c;
        t[7] = (uint32_t)x;
        t[8] = (uint32_t)(x >> 32);
    }
    uint8_t be[32];
    for (int k = 0; k < 8; k++) {
        be[4 * (7 - k)] = (uint8_t)(t[k] >> 24); be[4 * (7 - k) + 1] = (uint8_t)(t[k] >> 16);
        be[4 * (7 - k) + 2] = (uint8_t)(t[k] >> 8); be[4 * (7 - k) + 3] = (uint8_t)t[k];
    }
    if (t[8] || be_geq_r(be)) be_sub_r(be);
    memcpy(out, be, 32);
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_tau_g2(const KzgSettings* s, uint8_t out[96]) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    uint8_t* d;
    HIPCHK(hipMalloc(&d, 96));
    hipLaunchKernelGGL(k_g2_compress, dim3(1), dim3(64), 0, s->s1, s->d_tau4, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 96, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d));
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_g1_point(const KzgSettings* s, size_t i, uint8_t out[48]) {
    if (!s || !out || i >= FE_PER_BLOB) return fail(KZG_BADARGS, "bad argument");
    if (!s->d_g1) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    uint8_t* d;
    HIPCHK(hipMalloc(&d, 48));
    hipLaunchKernelGGL(k_aff_compress, dim3(1), dim3(64), 0, s->s1, s->d_g1 + i, s->d_g1_flag + i, d, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 48, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d));
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_g2_point(const KzgSettings* s, size_t i, uint8_t out[96]) {
    if (!s || !out) return fail(KZG_BADARGS, "bad argument");
    if (!s->d_g2 || i >= s->n_g2) return fail(KZG_BADARGS, "no such G2 point in these settings");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    uint8_t* d;
    HIPCHK(hipMalloc(&d, 96));
    hipLaunchKernelGGL(k_g2_compress, dim3(1), dim3(64), 0, s->s1, s->d_g2 + 4 * i, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 96, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipFree(d));
    return KZG_OK;
}

extern "C" KzgRet kzg_pairing_check(bool* ok, const uint8_t a[48], const uint8_t b[48], const KzgSettings* s);
// is_trusted_setup_in_lagrange_form (build.rs:107-129; its result is discarded by the reference's loader):
// e(g1[1], g2[0]) == e(g1[0], g2[1]) on the points in FILE order - true for a monomial-form G1 section, false for the
// Lagrange-form file the crate ships.
extern "C" KzgRet kzg_settings_is_monomial_form(bool* ok, const KzgSettings* s) {
    if (!s || !ok) return fail(KZG_BADARGS, "null argument");
    if (!s->d_g1) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    return kzg_pairing_check(ok, s->g1_first[0], s->g1_first[1], s);  // e(g1[0], [tau]G2) == e(g1[1], G2)
}

// ---------------------------------------------------------------- prover side (not in the reference)
// c-kzg-4844's blob_to_kzg_commitment / compute_kzg_proof / compute_blob_kzg_proof: 4096-term MSMs over the settings'
// Lagrange points, PROVER_CHUNK blobs per launch of the MSM kernels.
constexpr size_t PROVER_CHUNK = 64;
struct ProverBufs {
    uint8_t *d_blobs = nullptr, *d_out = nullptr, *d_cm = nullptr;
    Fr *d_sc = nullptr, *d_z = nullptr, *d_y = nullptr;
    uint32_t *d_tp = nullptr, *d_ts = nullptr, *d_sorted = nullptr, *d_status = nullptr, *d_cflag = nullptr;
    G1Jac *d_win = nullptr, *d_res = nullptr;
    G1Aff* d_cpts = nullptr;
    ~ProverBufs() {
        void* ptrs[] = {d_blobs, d_out, d_cm, d_sc, d_z, d_y, d_tp, d_ts, d_sorted, d_status, d_cflag, d_win, d_res, d_cpts};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
    }
    KzgRet alloc() {
        const size_t NT = (size_t)FE_PER_BLOB, CH = PROVER_CHUNK;
        HIPCHK(hipMalloc(&d_blobs, (size_t)BLOB_BYTES * CH));
        HIPCHK(hipMalloc(&d_sc, sizeof(Fr) * NT * CH));
        HIPCHK(hipMalloc(&d_tp, 4 * NT * CH));
        HIPCHK(hipMalloc(&d_ts, 4 * NT * CH));
        HIPCHK(hipMalloc(&d_sorted, 4 * NT * CH * MSM_WINDOWS));
        HIPCHK(hipMalloc(&d_status, 4 * CH));
        HIPCHK(hipMalloc(&d_win, sizeof(G1Jac) * MSM_WINDOWS * CH));
        HIPCHK(hipMalloc(&d_res, sizeof(G1Jac) * CH));
        HIPCHK(hipMalloc(&d_out, 48 * CH));
        HIPCHK(hipMalloc(&d_z, sizeof(Fr) * CH));
        HIPCHK(hipMalloc(&d_y, sizeof(Fr) * CH));
        HIPCHK(hipMalloc(&d_cm, 48 * CH));
        HIPCHK(hipMalloc(&d_cflag, 4 * CH));
        HIPCHK(hipMalloc(&d_cpts, sizeof(G1Aff) * CH));
        return KZG_OK;
    }
};
static KzgRet prover_ready(const KzgSettings* s) {
    if (!s->d_g1_mult) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    if (!s->g1_in_subgroup) return fail(KZG_BAD_SETUP, "a G1 setup point is outside the r-torsion subgroup");
    return KZG_OK;
}
// m MSMs: out[b] = compress(sum_i sc[b][i] * g1_points[i]); sc = plain canonical scalars (destroyed: GLV split in place)
static KzgRet setup_msm(const KzgSettings* s, ProverBufs& b, size_t m) {
    const size_t NT = (size_t)FE_PER_BLOB;
    const int total = (int)(m * NT);
    hipLaunchKernelGGL(k_commit_terms, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_tp, b.d_ts, total);
    hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_sc, total);
    MsmDesc d{};
    d.mult = s->d_g1_mult;
    d.pflag = s->d_g1_flag;
    d.scalars = b.d_sc;
    d.term_point = b.d_tp;
    d.term_scalar = b.d_ts;
    d.sorted = b.d_sorted;
    d.window_sums = b.d_win;
    d.nterms[0] = d.nterms[1] = (int)NT;
    d.max_terms = (int)NT;
    d.stride = (int)NT;
    d.slices = 1;
    d.chunks = MSM_CHUNKS;
    d.chunks_per_block = m >= 16 ? 4 : 1;
    const unsigned slots = MSM_CHUNKS / d.chunks_per_block;
    hipLaunchKernelGGL(k_msm_window, dim3(8, slots, (unsigned)m), dim3(256), 0, s->s1, d);
    hipLaunchKernelGGL(k_msm_combine, dim3((unsigned)m), dim3(64), 0, s->s1, b.d_win, b.d_res, (int)slots, 8);
    hipLaunchKernelGGL(k_jac_compress_n, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s->s1, b.d_res, b.d_out, (int)m);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// C_b = sum_i blob_b[i] * g1_points[i].  blobs: n * 131072 bytes, host memory; out: n * 48 bytes.
extern "C" KzgRet kzg_blob_to_kzg_commitment(uint8_t* out48, const uint8_t* blobs, size_t n, const KzgSettings* s) {
    if (!s || (n && (!out48 || !blobs))) return fail(KZG_BADARGS, "null argument");
    KzgRet rc = prover_ready(s);
    if (rc != KZG_OK || n == 0) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    ProverBufs b;
    if ((rc = b.alloc()) != KZG_OK) return rc;
    for (size_t lo = 0; lo < n; lo += PROVER_CHUNK) {
        const size_t m = std::min(PROVER_CHUNK, n - lo);
        const int total = (int)(m * FE_PER_BLOB);
        HIPCHK(hipMemcpyAsync(b.d_blobs, blobs + (size_t)BLOB_BYTES * lo, (size_t)BLOB_BYTES * m, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemsetAsync(b.d_status, 0, 4 * m, s->s1));
        hipLaunchKernelGGL(k_blob_scalars, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_blobs, b.d_sc, b.d_status, total);
        if ((rc = setup_msm(s, b, m)) != KZG_OK) return rc;
        std::vector<uint32_t> st(m);
        HIPCHK(hipMemcpyAsync(out48 + 48 * lo, b.d_out, 48 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(st.data(), b.d_status, 4 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        for (size_t i = 0; i < m; i++)
            if (st[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) Blob::as_polynomial, src/dtypes.rs:48-57
    }
    return KZG_OK;
}

// Shared body of compute_kzg_proof (zs given) and compute_blob_kzg_proof (z = the Fiat-Shamir challenge of (blob,
// commitment), src/kzg_proof.rs:46-72): y = p(z), pi = sum_i q_i g1_points[i] with q the quotient (fr_kernels.hpp).
static KzgRet compute_proofs(uint8_t* proofs48, uint8_t* ys32, const uint8_t* blobs, const uint8_t* zs, const uint8_t* commitments,
                             size_t n, const KzgSettings* s) {
    KzgRet rc = prover_ready(s);
    if (rc != KZG_OK || n == 0) return rc;
    if (zs)
        for (size_t i = 0; i < n; i++)
            if (be_geq_r(zs + 32 * i)) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) :36-41
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);
    ProverBufs b;
    if ((rc = b.alloc()) != KZG_OK) return rc;
    std::vector<uint8_t> le(32 * PROVER_CHUNK);
    for (size_t lo = 0; lo < n; lo += PROVER_CHUNK) {
        const size_t m = std::min(PROVER_CHUNK, n - lo);
        HIPCHK(hipMemcpyAsync(b.d_blobs, blobs + (size_t)BLOB_BYTES * lo, (size_t)BLOB_BYTES * m, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemsetAsync(b.d_status, 0, 4 * m, s->s1));
        if (zs) {
            for (size_t i = 0; i < m; i++) reverse32(le.data() + 32 * i, zs + 32 * (lo + i));
            HIPCHK(hipMemcpyAsync(b.d_z, le.data(), 32 * m, hipMemcpyHostToDevice, s->s1));
            HIPCHK(hipStreamSynchronize(s->s1));  // `le` is reused by the next chunk
        } else {
            HIPCHK(hipMemcpyAsync(b.d_cm, commitments + 48 * lo, 48 * m, hipMemcpyHostToDevice, s->s1));
            hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s->s1, b.d_cm, b.d_cm, (int)m, b.d_cpts, b.d_cflag, (int)m, 1);
            if ((rc = launch_challenge(s, b.d_blobs, b.d_cm, b.d_z, m)) != KZG_OK) return rc;
        }
        launch_evaluate(s, b.d_blobs, b.d_z, b.d_y, b.d_status, m);
        hipLaunchKernelGGL(k_blob_quotient, dim3((unsigned)m), dim3(64), 0, s->s1, b.d_blobs, b.d_z, b.d_y, s->d_M, b.d_sc, b.d_status);
        HIPCHK(hipGetLastError());
        if ((rc = setup_msm(s, b, m)) != KZG_OK) return rc;
        std::vector<uint32_t> st(m), cf(m, 0);
        std::vector<uint8_t> yl(32 * m);
        HIPCHK(hipMemcpyAsync(proofs48 + 48 * lo, b.d_out, 48 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(st.data(), b.d_status, 4 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(yl.data(), b.d_y, 32 * m, hipMemcpyDeviceToHost, s->s1));
        if (!zs) HIPCHK(hipMemcpyAsync(cf.data(), b.d_cflag, 4 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        for (size_t i = 0; i < m; i++) {
            if (cf[i] == G1_INVALID || st[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
            if (ys32) reverse32(ys32 + 32 * (lo + i), yl.data() + 32 * i);
        }
    }
    return KZG_OK;
}
extern "C" KzgRet kzg_compute_kzg_proof(uint8_t* proofs48, uint8_t* ys32, const uint8_t* blobs, const uint8_t* zs, size_t n,
                                        const KzgSettings* s) {
    if (!s || (n && (!proofs48 || !ys32 || !blobs || !zs))) return fail(KZG_BADARGS, "null argument");
    return compute_proofs(proofs48, ys32, blobs, zs, nullptr, n, s);
}
extern "C" KzgRet kzg_compute_blob_kzg_proof(uint8_t* proofs48, const uint8_t* blobs, const uint8_t* commitments, size_t n,
                                             const KzgSettings* s) {
    if (!s || (n && (!proofs48 || !blobs || !commitments))) return fail(KZG_BADARGS, "null argument");
    return compute_proofs(proofs48, nullptr, blobs, nullptr, commitments, n, s);
}

// diagnostic / test hook: the host-side SHA-256 used for the batch transcript (force_portable skips SHA-NI)
extern "C" int kzg_debug_host_sha256(uint8_t out[32], const uint8_t* data, size_t len, int force_portable) {
    if (force_portable) {
        uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        if (len % 64) return -1;
        for (size_t i = 0; i < len / 64; i++) hostsha::block(st, data + 64 * i);
        for (int i = 0; i < 8; i++) { out[4*i] = (uint8_t)(st[i] >> 24); out[4*i+1] = (uint8_t)(st[i] >> 16); out[4*i+2] = (uint8_t)(st[i] >> 8); out[4*i+3] = (uint8_t)st[i]; }
        return 0;
    }
    hostsha::digest(out, data, len);
    return hostsha::have_ni() ? 1 : 0;
}

// diagnostic: in-kernel shader clock (MHz) = delta s_memtime / delta s_memrealtime * 100 MHz
__global__ void k_clock_probe(unsigned long long* out, int spin) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0 + (x == 12345u);
    }
}

// diagnostic (tools/ only): run the VERIFY program on `instances` copies of zero inputs, `reps` times;
// returns the average kernel time and the in-kernel shader clock seen by a 1-block probe launched alone.
extern "C" KzgRet kzg_debug_slp_bench(float* ms_out, float* mhz_out, int instances, int reps, const KzgSettings* s) {
    if (!s || !ms_out || instances < 1) return fail(KZG_BADARGS, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    Fp *d_in, *d_out;
    unsigned long long* d_clk;
    HIPCHK(hipMalloc(&d_in, sizeof(Fp) * 6 * instances));
    HIPCHK(hipMalloc(&d_out, sizeof(Fp) * 6 * instances));
    HIPCHK(hipMalloc(&d_clk, 16));
    HIPCHK(hipMemset(d_in, 0, sizeof(Fp) * 6 * instances));
    KzgRet rc = run_program(s->verify, d_in, s->d_prep, d_out, instances, s->s1);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    for (int i = 0; i < reps; i++)
        if ((rc = run_program(s->verify, d_in, s->d_prep, d_out, instances, s->s1)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    HIPCHK(hipEventSynchronize(s->ev[3]));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[2], s->ev[3]));
    *ms_out = ms / reps;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s->s1, d_clk, 200000);
    unsigned long long h[2];
    HIPCHK(hipMemcpyAsync(h, d_clk, 16, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    if (mhz_out) *mhz_out = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_clk);
    return KZG_OK;
}

extern "C" KzgRet kzg_last_timings(const KzgSettings* s, float out_ms[8]) {
    if (!s || !out_ms) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_ms, s->timings, sizeof(float) * 8);
    return KZG_OK;
}
