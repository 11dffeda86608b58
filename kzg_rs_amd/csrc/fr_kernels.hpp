// fr_kernels.hpp - the per-blob Fr hot path on gfx950:
//   k_roots_tables      roots of unity (reference build.rs:131-170) in two Montgomery scalings
//   k_blob_challenge    Fiat-Shamir challenge z (reference src/kzg_proof.rs:46-91)
//   k_blob_evaluate     y = p(z), p given in evaluation form (reference src/kzg_proof.rs:94-133
//                       + batch_inversion :155-201 + Blob::as_polynomial src/dtypes.rs:48-57)
//
// Evaluation algorithm (inversion-free restatement of the barycentric formula).
// The reference computes  y = (z^4096 - 1)/4096 * sum_i p_i w_i / (z - w_i)  with one batch
// inversion.  Because prod_i (z - w_i) = z^4096 - 1 exactly, putting the 4096 fractions over
// their common denominator cancels the (z^4096 - 1) factor and leaves
//        y = (1/4096) * N,   N = sum_i p_i w_i prod_{j != i} (z - w_j),
// an identity of polynomials in z - so it is also correct at z = w_i (the reference's early
// return :109-111) and needs no inversion and no special case.  With  w_i/(z - w_i) = z/(z - w_i) - 1
// the w_i factor leaves the sum:
//        N = z * N0 - (z^4096 - 1) * S,   N0 = sum_i p_i prod_{j != i} (z - w_j),   S = sum_i p_i.
// N0 is built by a binary tree over the bit-reversed index order: a block of 2^L consecutive
// indices k*2^L.. is a coset whose denominator is  D_{L,k} = z^(2^L) - roots[k]  (the first
// 2^(12-L) roots of the SAME table, and roots[2k+1] = -roots[2k]), so
//        N0_{L+1,k} = Z_L (N0_{L,2k} + N0_{L,2k+1}) + roots[2k] (N0_{L,2k} - N0_{L,2k+1}),  Z_L = z^(2^L),
// from the leaves N0_{0,i} = p_i: 2 multiplications per merge, 2*4095 + 12 (powers of z) + 5 = 8 207
// Fr multiplications per blob instead of ~20 480 + an inversion + a 256-step pow; the result is the
// same field element, bit for bit.
//
// Mapping: ONE WAVEFRONT PER BLOB.  Lane t owns the 64 consecutive elements 64t..64t+63
// (2 KiB of the blob), folds them depth-first with a 6-entry LDS stack (levels 1..6), then the
// last 6 levels run across lanes with ds_bpermute shuffles.  No barriers beyond the one that
// publishes the powers of z.
#pragma once
#include "field.hpp"
#include "fr29.hpp"
#include "sha256.hpp"

namespace kzg {

#ifndef KZG_AB_VARIANTS
#define KZG_AB_VARIANTS 0
#endif
constexpr int FE_PER_BLOB = 4096;
constexpr int BLOB_BYTES = 131072;

__device__ __forceinline__ uint32_t bitrev12(uint32_t i) { return __brev(i) >> 20; }

// M[j]  = roots[j] * R      (Montgomery form),  roots[j] = omega^bitrev12(j)
// DM[j] = roots[j] * R^2    ("double Montgomery": mul(DM[j], plain x) = roots[j]*x in Montgomery form)
__global__ void k_roots_tables(Fr* __restrict__ M, Fr* __restrict__ DM) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= FE_PER_BLOB) return;
    uint32_t e = bitrev12(j);
    Fr w, acc = FrF::one();
#pragma unroll
    for (int i = 0; i < 8; i++) w.l[i] = consts::FR_OMEGA_MONT[i];
    for (int b = 11; b >= 0; b--) {
        acc = FrF::sqr(acc);
        if ((e >> b) & 1) acc = FrF::mul(acc, w);
    }
    M[j] = acc;
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = consts::FR_R2[i];
    DM[j] = FrF::mul(acc, r2);
}

// 32 big-endian bytes (two uint4 as loaded from memory) -> 8 little-endian limbs
__device__ __forceinline__ Fr fr_from_be_words(const uint4& hi, const uint4& lo) {
    Fr r;
    r.l[7] = __builtin_bswap32(hi.x); r.l[6] = __builtin_bswap32(hi.y);
    r.l[5] = __builtin_bswap32(hi.z); r.l[4] = __builtin_bswap32(hi.w);
    r.l[3] = __builtin_bswap32(lo.x); r.l[2] = __builtin_bswap32(lo.y);
    r.l[1] = __builtin_bswap32(lo.z); r.l[0] = __builtin_bswap32(lo.w);
    return r;
}
__device__ __forceinline__ void fr_to_be_words(uint4& hi, uint4& lo, const Fr& a) {
    hi = make_uint4(__builtin_bswap32(a.l[7]), __builtin_bswap32(a.l[6]), __builtin_bswap32(a.l[5]), __builtin_bswap32(a.l[4]));
    lo = make_uint4(__builtin_bswap32(a.l[3]), __builtin_bswap32(a.l[2]), __builtin_bswap32(a.l[1]), __builtin_bswap32(a.l[0]));
}

__device__ __forceinline__ void bswap4(uint32_t* w, const uint4& q) {
    w[0] = __builtin_bswap32(q.x); w[1] = __builtin_bswap32(q.y); w[2] = __builtin_bswap32(q.z); w[3] = __builtin_bswap32(q.w);
}

// ---------------------------------------------------------------- challenge (one lane per blob)
// transcript = "FSBLOBVERIFY_V1_" || u64_be(0) || u64_be(4096) || blob || commitment   (131 152 B)
// z = int_be(sha256(transcript)) mod r        (src/kzg_proof.rs:46-91)
// Output: z as canonical little-endian limbs (plain integer, NOT Montgomery).
//
// Every lane streams a different blob, so one load instruction touches 64 different cache lines.  The transcript is a
// 32-byte header followed by the blob, so its 64-byte blocks straddle the blob's 128-byte lines: line m = blob bytes
// [128 m, 128 m + 128) feeds block 2m = (32 bytes carried over | L0 L1), block 2m + 1 = (L2 .. L5) and carries (L6 L7)
// on.  A lane fetches a WHOLE line with eight back-to-back 16-byte loads, one line ahead of the two compressions that
// consume it: fetched 16 bytes at a time, a line is re-read from L2 up to eight times once the CU's waves outrun the
// 32 KiB L1.  No LDS, no barriers: this form runs within a few % of the pure-register SHA-256 issue rate
// (tools/microbench/shabench.hip) as soon as every SIMD has a wave, which the producer/consumer form below cannot
// reach; that one exists for the LATENCY of a single batch.
// 256-thread workgroups on purpose: the hardware deals the four waves of a workgroup round-robin over the CU's four
// SIMDs, while single-wave workgroups are placed with no regard for balance - the same kernel launched as 1024
// workgroups of 64 threads ran 1.8x slower (10.3 ms against 5.6 ms for 65 536 blobs) with two waves sharing a SIMD
// on some CUs and idle SIMDs on others.
// ktime (optional): { max over waves of ~start, max of end } in ticks of the 100 MHz s_memrealtime counter - the kernel's
// own execution interval, which a HIP-event pair around the launch cannot give while other launch groups share the chip
// (the events also time the wait for free CUs); { sum of shader cycles (s_memtime), sum of reference ticks } over the waves in
// ktime[2..3]: the shader clock the kernel ran at while the other kernels of the pipeline shared the chip.  Zeroed by the host
// before the launch.
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_blob_challenge_t(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments,
                                                                Fr* __restrict__ z_out, int n, unsigned long long* __restrict__ ktime) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool stamp = ktime && (threadIdx.x & 63) == 0;
    unsigned long long clk0 = 0, ref0 = 0;
    if (stamp) {
        ref0 = __builtin_amdgcn_s_memrealtime();
        clk0 = __builtin_amdgcn_s_memtime();
        atomicMax(&ktime[0], ~ref0);
    }
    const uint4* blob = reinterpret_cast<const uint4*>(blobs + (size_t)i * BLOB_BYTES);
    const uint4* cm = reinterpret_cast<const uint4*>(commitments + (size_t)i * 48);
    Sha256State st;
    sha256_init(st);
    uint4 L[8], c0, c1;
#pragma unroll
    for (int j = 0; j < 8; j++) L[j] = blob[j];
#pragma unroll 1
    for (int m = 0; m < 1024; m++) {
        uint32_t w[16];
        if (m == 0) {
            w[0] = 0x4653424c; w[1] = 0x4f425645; w[2] = 0x52494659; w[3] = 0x5f56315f;  // "FSBLOBVERIFY_V1_"
            w[4] = 0; w[5] = 0; w[6] = 0; w[7] = FE_PER_BLOB;                            // u64_be(0) | u64_be(4096)
        } else {
            bswap4(w, c0);
            bswap4(w + 4, c1);
        }
        bswap4(w + 8, L[0]);
        bswap4(w + 12, L[1]);
        const uint4 n2 = L[2], n3 = L[3], n4 = L[4], n5 = L[5];
        c0 = L[6];
        c1 = L[7];
        {  // the next line, in flight during the two compressions below (the last iteration re-reads its own line: no
           // branch in the loop, so the compiler's wait counters stay exact - a two-way merge here made every iteration
           // wait for three of the loads it had just issued)
            const uint4* p = blob + 8 * (m + 1 < 1024 ? m + 1 : 1023);
#pragma unroll
            for (int j = 0; j < 8; j++) L[j] = p[j];
            // all eight loads of the line issue HERE: left alone, the scheduler sinks three of them ~3 000 cycles into the
            // first compression, by when the line may have left the L2 again (HBM reads were 1.20x the blob bytes)
            __builtin_amdgcn_sched_barrier(0x7);  // ALU instructions may cross, memory instructions may not
        }
        sha256_compress(st, w);  // block 2m
        bswap4(w, n2); bswap4(w + 4, n3); bswap4(w + 8, n4); bswap4(w + 12, n5);
        sha256_compress(st, w);  // block 2m + 1
    }
    L[0] = cm[0];
    L[1] = cm[1];
    L[2] = cm[2];
    {
        uint32_t w[16];
        bswap4(w, c0); bswap4(w + 4, c1); bswap4(w + 8, L[0]); bswap4(w + 12, L[1]);  // block 2048: blob tail | commitment[0..32)
        sha256_compress(st, w);
        bswap4(w, L[2]);  // block 2049: commitment[32..48) + 0x80 pad + bit length (131152 * 8)
        w[4] = 0x80000000u;
#pragma unroll
        for (int k = 5; k < 15; k++) w[k] = 0;
        w[15] = 131152u * 8u;
        sha256_compress(st, w);
    }
    // digest as big-endian integer, reduced mod r: to_mont reduces (d*R2*R^-1 = dR mod r), from_mont strips R
    Fr d;
#pragma unroll
    for (int k = 0; k < 8; k++) d.l[k] = st.h[7 - k];
    z_out[i] = FrF::from_mont(FrF::to_mont(d));
    if (stamp) {  // ktime[2] / ktime[3]: shader cycles / reference ticks summed over the waves = the clock the SIMDs really ran at
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), ref1 = __builtin_amdgcn_s_memrealtime();
        atomicMax(&ktime[1], ref1);
        atomicAdd(&ktime[2], clk1 - clk0);
        atomicAdd(&ktime[3], ref1 - ref0);
    }
}

// ---------------------------------------------------------------- challenge, producer / consumer form
// Same result as k_blob_challenge.  The 2050-block SHA-256 chain of a blob is serial, and a wave that is alone
// on its SIMD issues only one instruction every ~4-5 cycles, so what matters is the INSTRUCTION COUNT ON THE
// CHAIN.  A PAIR of waves serves 64 blobs: the producer loads the next 64-byte block, expands the message
// schedule and adds the round constants (~580 instructions), and hands the 64 (K+W) words to the consumer through
// a double-buffered LDS tile; the consumer only runs the 64 rounds (~930 instructions per block instead of
// ~1400).  One barrier per block.
//
// One pair per workgroup: the two waves land on different SIMDs of a CU (a workgroup's waves are dealt round-robin).
// Used while every pair can have a CU to itself; larger launches use k_blob_challenge, which has the higher throughput.
constexpr int CHALLENGE_TILE_U4 = 2 * 16 * 64;  // [buffer][round quad][lane] uint4 = 32 KiB, conflict-free b128 rows
__global__ __launch_bounds__(128) void k_blob_challenge_split(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments,
                                                              Fr* __restrict__ z_out, int n) {
    __shared__ uint4 tile[CHALLENGE_TILE_U4];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0 consumer, 1 producer
    int i = blockIdx.x * 64 + lane;
    const bool live = i < n;
    if (!live) i = n - 1;  // redundant work keeps the barriers uniform
    constexpr int NBLK = 2050;

    // Both roles run NBLK + 1 barriers; each keeps its own loop so the register allocation of one role does not carry
    // the other's live state.
    if (role == 1) {
        const uint4* blob = reinterpret_cast<const uint4*>(blobs + (size_t)i * BLOB_BYTES);
        const uint4* cm = reinterpret_cast<const uint4*>(commitments + (size_t)i * 48);
        // expand the 16 message words of transcript block b, hand (W + K)[0..64) to the consumer
        auto emit = [=](uint32_t (&w)[16], int b) {
            uint32_t kw[64];
            sha256_schedule_kw(kw, w);
            uint4* dst = tile + (b & 1) * (16 * 64) + lane;
#pragma unroll
            for (int t = 0; t < 16; t++) dst[t * 64] = make_uint4(kw[4 * t], kw[4 * t + 1], kw[4 * t + 2], kw[4 * t + 3]);
            __syncthreads();  // block b is ready; the consumer is done with block b - 1, whose buffer block b + 1 reuses
        };
        // The transcript is a 32-byte header followed by the blob, so its 64-byte blocks straddle the blob's 128-byte
        // cache lines: line m = blob bytes [128 m, 128 m + 128) feeds block 2m = (32 bytes carried over | L0 L1),
        // block 2m + 1 = (L2 .. L5) and carries (L6 L7) on.  A lane fetches a WHOLE line with eight back-to-back
        // 16-byte loads and then lives on it for two blocks: every lane of the wave streams a different blob, so a
        // load instruction touches 64 different lines, and fetching a line 16 bytes at a time over two blocks re-reads
        // it from L2 up to eight times once the CU's producers outrun the 32 KiB L1.
        uint4 L[8], c0, c1;
#pragma unroll
        for (int j = 0; j < 8; j++) L[j] = blob[j];
#pragma unroll 1
        for (int m = 0; m < 1024; m++) {
            uint32_t w[16];
            if (m == 0) {
                w[0] = 0x4653424c; w[1] = 0x4f425645; w[2] = 0x52494659; w[3] = 0x5f56315f;  // "FSBLOBVERIFY_V1_"
                w[4] = 0; w[5] = 0; w[6] = 0; w[7] = FE_PER_BLOB;                            // u64_be(0) | u64_be(4096)
            } else {
                bswap4(w, c0);
                bswap4(w + 4, c1);
            }
            bswap4(w + 8, L[0]);
            bswap4(w + 12, L[1]);
            emit(w, 2 * m);
            bswap4(w, L[2]); bswap4(w + 4, L[3]); bswap4(w + 8, L[4]); bswap4(w + 12, L[5]);
            c0 = L[6];
            c1 = L[7];
            if (m + 1 < 1024) {  // the next line, in flight while block 2m + 1 is expanded
                const uint4* p = blob + 8 * (m + 1);
#pragma unroll
                for (int j = 0; j < 8; j++) L[j] = p[j];
            } else {
                L[0] = cm[0]; L[1] = cm[1]; L[2] = cm[2];
            }
            emit(w, 2 * m + 1);
        }
        {
            uint32_t w[16];
            bswap4(w, c0); bswap4(w + 4, c1); bswap4(w + 8, L[0]); bswap4(w + 12, L[1]);  // blob tail | commitment[0..32)
            emit(w, 2048);
            bswap4(w, L[2]);  // commitment[32..48) + 0x80 pad + bit length
            w[4] = 0x80000000u;
#pragma unroll
            for (int k = 5; k < 15; k++) w[k] = 0;
            w[15] = 131152u * 8u;
            emit(w, 2049);
        }
        __syncthreads();
    } else {
        Sha256State st;
        sha256_init(st);
        __syncthreads();
#pragma unroll 1
        for (int b = 0; b < NBLK; b++) {
            uint32_t a = st.h[0], bb = st.h[1], c = st.h[2], d = st.h[3], e = st.h[4], f = st.h[5], g = st.h[6], h = st.h[7];
            // issue all 16 LDS reads of the block at once (they pipeline), then run the rounds out of registers:
            // a just-in-time read every 4 rounds costs one LDS round trip per quad on the serial chain
            const uint4* src = tile + (b & 1) * (16 * 64) + lane;
            uint4 kw[16];
#pragma unroll
            for (int t = 0; t < 16; t++) kw[t] = src[t * 64];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                asm volatile("" : "+v"(kw[t].x), "+v"(kw[t].y), "+v"(kw[t].z), "+v"(kw[t].w));  // keep the reads hoisted
                sha256_rounds4(a, bb, c, d, e, f, g, h, kw[t]);
            }
            st.h[0] += a; st.h[1] += bb; st.h[2] += c; st.h[3] += d; st.h[4] += e; st.h[5] += f; st.h[6] += g; st.h[7] += h;
            __syncthreads();
        }
        if (live) {
            Fr dgst;
#pragma unroll
            for (int k = 0; k < 8; k++) dgst.l[k] = st.h[7 - k];
            z_out[i] = FrF::from_mont(FrF::to_mont(dgst));
        }
    }
}

// The same with TWO LANES PER BLOB on the consumer side (sha256.hpp: sha256_round_2lane): a workgroup of three wavefronts
// serves 64 blobs - wave 2 is the producer (one lane per blob, as above), waves 0 and 1 run the rounds for 32 blobs each,
// one lane of a pair on the a-chain, the lane four further on the e-chain.  ~690 instructions per block on the serial
// chain instead of ~930.  The (K + W) tile is read by both lanes of a pair at the same address; the even lanes read a row of
// zeros instead (their sum takes no message word).
//
// SEG = true is the same chain in SEGMENTS, for a blob that is still arriving (capi_verify.hpp: a host Vec<Blob> crosses PCIe
// in slices ACROSS the blobs - SHA-256 consumes a blob front to back, so lines [m0, m1) of every blob can be hashed while the
// next slice is on the link): the launch covers the 128-byte lines [m0, m1) of every blob - blocks 2 m0 .. 2 m1 - 1, and
// the two tail blocks when m1 = 1024.  What travels between two launches is the 32-byte midstate per blob (`mid`, word k of
// blob i at mid[8 i + k]); the 32 blob bytes a block carries over from the line before are re-read from the slice that has
// already landed.  SEG = false compiles to the one-launch kernel (m0 = 0, m1 = 1024, no midstate).
template <bool SEG>
__global__ __launch_bounds__(192) void k_blob_challenge_split2_t(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments,
                                                                 Fr* __restrict__ z_out, int n, int m0_arg, int m1_arg, uint32_t* __restrict__ mid) {
    __shared__ uint4 tile[CHALLENGE_TILE_U4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0, 1 consumers; 2 producer
    const int M0 = SEG ? m0_arg : 0, M1 = SEG ? m1_arg : 1024;
    const bool last_seg = !SEG || M1 == 1024;
    const int NBLK = 2 * (M1 - M0) + (last_seg ? 2 : 0);
    if (wave == 2) {
        int i = blockIdx.x * 64 + lane;
        if (i >= n) i = n - 1;  // redundant work keeps the barriers uniform
        const uint4* blob = reinterpret_cast<const uint4*>(blobs + (size_t)i * BLOB_BYTES);
        const uint4* cm = reinterpret_cast<const uint4*>(commitments + (size_t)i * 48);
        // expand the 16 message words of transcript block b, hand (W + K)[0..64) to the consumer
        auto emit = [=](uint32_t (&w)[16], int b) {
            uint32_t kw[64];
            sha256_schedule_kw(kw, w);
            uint4* dst = tile + (b & 1) * (16 * 64) + lane;
#pragma unroll
            for (int t = 0; t < 16; t++) dst[t * 64] = make_uint4(kw[4 * t], kw[4 * t + 1], kw[4 * t + 2], kw[4 * t + 3]);
            __syncthreads();  // block b is ready; the consumer is done with block b - 1, whose buffer block b + 1 reuses
        };
        // The transcript is a 32-byte header followed by the blob, so its 64-byte blocks straddle the blob's 128-byte
        // cache lines: line m = blob bytes [128 m, 128 m + 128) feeds block 2m = (32 bytes carried over | L0 L1),
        // block 2m + 1 = (L2 .. L5) and carries (L6 L7) on.  A lane fetches a WHOLE line with eight back-to-back
        // 16-byte loads and then lives on it for two blocks: every lane of the wave streams a different blob, so a
        // load instruction touches 64 different lines, and fetching a line 16 bytes at a time over two blocks re-reads
        // it from L2 up to eight times once the CU's producers outrun the 32 KiB L1.
        uint4 L[8], c0, c1;
#pragma unroll
        for (int j = 0; j < 8; j++) L[j] = blob[8 * M0 + j];
        if (SEG && M0 > 0) {  // the last 32 bytes of line M0 - 1
            c0 = blob[8 * M0 - 2];
            c1 = blob[8 * M0 - 1];
        }
#pragma unroll 1
        for (int m = M0; m < M1; m++) {
            uint32_t w[16];
            if (m == 0) {
                w[0] = 0x4653424c; w[1] = 0x4f425645; w[2] = 0x52494659; w[3] = 0x5f56315f;  // "FSBLOBVERIFY_V1_"
                w[4] = 0; w[5] = 0; w[6] = 0; w[7] = FE_PER_BLOB;                            // u64_be(0) | u64_be(4096)
            } else {
                bswap4(w, c0);
                bswap4(w + 4, c1);
            }
            bswap4(w + 8, L[0]);
            bswap4(w + 12, L[1]);
            emit(w, 2 * (m - M0));
            bswap4(w, L[2]); bswap4(w + 4, L[3]); bswap4(w + 8, L[4]); bswap4(w + 12, L[5]);
            c0 = L[6];
            c1 = L[7];
            if (m + 1 < M1) {  // the next line, in flight while block 2m + 1 is expanded (never beyond the segment: it may not have landed)
                const uint4* p = blob + 8 * (m + 1);
#pragma unroll
                for (int j = 0; j < 8; j++) L[j] = p[j];
            } else if (last_seg) {
                L[0] = cm[0]; L[1] = cm[1]; L[2] = cm[2];
            }
            emit(w, 2 * (m - M0) + 1);
        }
        if (last_seg) {
            uint32_t w[16];
            bswap4(w, c0); bswap4(w + 4, c1); bswap4(w + 8, L[0]); bswap4(w + 12, L[1]);  // blob tail | commitment[0..32)
            emit(w, 2 * (M1 - M0));
            bswap4(w, L[2]);  // commitment[32..48) + 0x80 pad + bit length
            w[4] = 0x80000000u;
#pragma unroll
            for (int k = 5; k < 15; k++) w[k] = 0;
            w[15] = 131152u * 8u;
            emit(w, 2 * (M1 - M0) + 1);
        }
        __syncthreads();
    } else {
        // lanes 0-3 of every eight: the a-chains of four blobs, lanes 4-7: their e-chains (DPP bank masks work on groups of four)
        const int blob_in_block = 32 * wave + 4 * (lane >> 3) + (lane & 3);
        int i = blockIdx.x * 64 + blob_in_block;
        const bool live = i < n;
        const bool a_lane = (lane & 4) == 0;
        Sha2LaneConsts c;
        c.n1 = a_lane ? 2u : 6u;
        c.n2 = a_lane ? 13u : 11u;
        c.n3 = a_lane ? 22u : 25u;
        c.m = a_lane ? 0xffffffffu : 0u;
        // this lane's half of the state: h0..h3 (a-chain) or h4..h7 (e-chain)
        uint32_t H0 = a_lane ? 0x6a09e667u : 0x510e527fu, H1 = a_lane ? 0xbb67ae85u : 0x9b05688cu, H2 = a_lane ? 0x3c6ef372u : 0x1f83d9abu,
                 H3 = a_lane ? 0xa54ff53au : 0x5be0cd19u;
        uint32_t* my_mid = SEG ? mid + 8 * (size_t)(live ? i : n - 1) + (a_lane ? 0 : 4) : nullptr;
        if (SEG && M0 > 0) {
            const uint4 q = *reinterpret_cast<const uint4*>(my_mid);
            H0 = q.x; H1 = q.y; H2 = q.z; H3 = q.w;
        }
        __syncthreads();
#pragma unroll 1
        for (int b = 0; b < NBLK; b++) {
            // all 16 LDS reads of the block at once (they pipeline); the a-lanes read their e-lanes' words too and never
            // use them (the additions of kw are masked to the e-lanes)
            const uint4* src = tile + (b & 1) * (16 * 64) + blob_in_block;
            constexpr int stride = 64;
            uint32_t kw[65];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint4 q = src[t * stride];
                kw[4 * t] = q.x; kw[4 * t + 1] = q.y; kw[4 * t + 2] = q.z; kw[4 * t + 3] = q.w;
            }
            kw[64] = 0;
            Sha2LaneRing r;
            r.x0 = H0; r.x1 = H1; r.x2 = H2; r.dd = H3;
            r.hk = a_lane ? 0u : H3 + kw[0];  // x3 + kw of round 0
            asm volatile("" : "=v"(r.nx));  // (defined, whatever it holds)
#pragma unroll
            for (int t = 0; t < 6; t++) sha256_rounds10_2lane(r, kw + 10 * t + 1, c);
            sha256_rounds4_2lane(r, kw + 61, c);
            // after the last round hk = x3 + 0: h on the e-lanes, 0 on the a-lanes
            H0 += r.x0;
            H1 += r.x1;
            H2 += r.x2;
            H3 += a_lane ? r.dd : r.hk;  // d sits in dd on the a-lanes, h in hk on the e-lanes
            __syncthreads();
        }
        if (SEG && !last_seg) {  // the midstate of the chain so far: this lane's four words
            if (live) *reinterpret_cast<uint4*>(my_mid) = make_uint4(H0, H1, H2, H3);
            return;
        }
        // the e-lane's words to the a-lane (row_shl:4: lane i reads lane i + 4), which reduces the digest and writes z
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H0, 0x104, 0xF, 0xF, true), e1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H1, 0x104, 0xF, 0xF, true);
        const uint32_t e2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H2, 0x104, 0xF, 0xF, true), e3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H3, 0x104, 0xF, 0xF, true);
        if (live && a_lane) {
            const uint32_t hh[8] = {H0, H1, H2, H3, e0, e1, e2, e3};
            Fr dgst;
#pragma unroll
            for (int k = 0; k < 8; k++) dgst.l[k] = hh[7 - k];
            z_out[i] = FrF::from_mont(FrF::to_mont(dgst));
        }
    }
}

// ---------------------------------------------------------------- evaluation (one wavefront per blob)
__device__ __forceinline__ Fr fr_shfl_xor(const Fr& a, int mask) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_xor((int)a.l[i], mask, 64);
    return r;
}

#if KZG_AB_VARIANTS  // (the 8x32 form of the evaluation: A/B build only, option evaluate_kernel=32)

// z_in: plain little-endian limbs (any value < 2^256; reduced mod r here, like scalar_from_bytes_unchecked)
// y_out: plain little-endian canonical limbs.  status[b] |= 1 when a blob element is >= r
// (src/kzg_proof.rs:36-41 -> KzgError::BadArgs).
__global__ __launch_bounds__(64, 4) void k_blob_evaluate32(const uint8_t* __restrict__ blobs, const Fr* __restrict__ z_in,
                                                      const Fr* __restrict__ M, const Fr* __restrict__ DM,
                                                      Fr* __restrict__ y_out, uint32_t* __restrict__ status) {
    const int blob_idx = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ Fr Z[14];              // Z[L] = z^(2^L), Montgomery; Z[13] = z R^2 (takes plain operands)
    __shared__ uint4 stack[6][2][64]; // levels 1..6, two 16-byte halves, lane-major: conflict-free b128
    if (lane == 0) {
        Fr z = FrF::to_mont(z_in[blob_idx]);
        Fr r2;
#pragma unroll
        for (int i = 0; i < 8; i++) r2.l[i] = consts::FR_R2[i];
        Z[0] = z;
        Z[13] = FrF::mul(z, r2);
        for (int l = 1; l <= 12; l++) {
            z = FrF::sqr(z);
            Z[l] = z;
        }
    }
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(blobs + (size_t)blob_idx * BLOB_BYTES) + (size_t)lane * 128;
    const Fr zd = Z[13];
    bool bad = false;
    Fr n, psum = FrF::zero();  // psum: this lane's share of S = sum_i p_i (plain)
    for (int q = 0; q < 32; q++) {
        uint4 a_hi = src[4 * q], a_lo = src[4 * q + 1], b_hi = src[4 * q + 2], b_lo = src[4 * q + 3];
        Fr pa = fr_from_be_words(a_hi, a_lo), pb = fr_from_be_words(b_hi, b_lo);
        bad |= FrF::geq_mod(pa) | FrF::geq_mod(pb);
        Fr u = FrF::sub(pa, pb), s = FrF::add(pa, pb);
        int k = 32 * lane + q;  // level-1 node index
        psum = FrF::add(psum, s);
        n = FrF::add(FrF::mul(zd, s), FrF::mul(DM[2 * k], u));  // z s + roots[2k] u, Montgomery
        int level = 1;
        for (int qq = q; qq & 1; qq >>= 1) {
            Fr na;
            uint4 h0 = stack[level - 1][0][lane], h1 = stack[level - 1][1][lane];
            na.l[0] = h0.x; na.l[1] = h0.y; na.l[2] = h0.z; na.l[3] = h0.w;
            na.l[4] = h1.x; na.l[5] = h1.y; na.l[6] = h1.z; na.l[7] = h1.w;
            Fr sum = FrF::add(na, n), dif = FrF::sub(na, n);
            n = FrF::add(FrF::mul(Z[level], sum), FrF::mul(M[k & ~1], dif));
            k >>= 1;
            level++;
        }
        if (level <= 6 && q != 31) {
            stack[level - 1][0][lane] = make_uint4(n.l[0], n.l[1], n.l[2], n.l[3]);
            stack[level - 1][1][lane] = make_uint4(n.l[4], n.l[5], n.l[6], n.l[7]);
        }
    }
    // n = N_{6,lane}; fold across lanes
    for (int L = 6; L < 12; L++) {
        int sh = L - 6;
        Fr other = fr_shfl_xor(n, 1 << sh);
        int j = lane >> sh;  // node index at level L
        bool left = (j & 1) == 0;
        Fr na, nb;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            na.l[i] = left ? n.l[i] : other.l[i];
            nb.l[i] = left ? other.l[i] : n.l[i];
        }
        Fr sum = FrF::add(na, nb), dif = FrF::sub(na, nb);
        n = FrF::add(FrF::mul(Z[L], sum), FrF::mul(M[j & ~1], dif));
    }
    for (int sh = 1; sh < 64; sh <<= 1) psum = FrF::add(psum, fr_shfl_xor(psum, sh));
    unsigned long long any_bad = __ballot(bad);
    if (lane == 0) {
        Fr inv;
#pragma unroll
        for (int i = 0; i < 8; i++) inv.l[i] = consts::FR_INV4096_PLAIN[i];
        // N = z N0 - (z^4096 - 1) S;  mul(Z[13], S) = z S R, so (z^4096 - 1) S R = Z_12 S R - S R with S R = to_mont(S)
        Fr sm = FrF::to_mont(psum);
        Fr nn = FrF::sub(FrF::mul(n, Z[0]), FrF::sub(FrF::mul(Z[12], sm), sm));
        y_out[blob_idx] = FrF::mul(nn, inv);  // (N R)(1/4096) R^-1 = N/4096, plain
        if (any_bad) atomicOr(&status[blob_idx], 1u);
    }
}

#endif  // KZG_AB_VARIANTS

// ---------------------------------------------------------------- evaluation in radix 2^29 (fr29.hpp)
// The same tree as k_blob_evaluate32 with every field element in 9 x 29-bit limbs: products accumulate a whole
// column in one 64-bit register without carry instructions, additions are limb-wise and nothing is reduced inside
// the tree (value and limb bounds: fr29.hpp), and the two products of a merge share one Montgomery reduction.
// ~0.45x the VALU cycles of the 8x32 form.
struct alignas(16) Fr29Mem {  // table entry: 9 limbs padded to 48 bytes (two b128 loads + one b32)
    uint32_t l[12];
};
__device__ __forceinline__ Fr29 fr29_load(const Fr29Mem* p) {
    const uint4 a = *reinterpret_cast<const uint4*>(p->l), b = *reinterpret_cast<const uint4*>(p->l + 4);
    Fr29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = p->l[8];
    return r;
}
__device__ __forceinline__ Fr29 fr29_shfl_xor(const Fr29& a, int mask) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor((int)a.l[i], mask, 64);
    return r;
}
// M29[j] = roots[j] R' , DM29[j] = roots[j] R'^2  (R' = 2^261; values below 2r, limbs below 2^29), from the 8x32 table M
__global__ void k_roots_tables29(const Fr* __restrict__ M, Fr29Mem* __restrict__ M29, Fr29Mem* __restrict__ DM29) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= FE_PER_BLOB) return;
    Fr w = FrF::from_mont(M[j]);
    const Fr29 r2 = fr29_const(c29::FR29_R2);
    Fr29 m = fr29_mul(fr29_from_words(w.l), r2), dm = fr29_mul(m, r2);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        M29[j].l[i] = i < 9 ? m.l[i] : 0u;
        DM29[j].l[i] = i < 9 ? dm.l[i] : 0u;
    }
}

// The table entries in the order k_blob_evaluate reads them: slot-major, lane-minor, split into two b128 planes and a
// b32 plane so that every table load of a wavefront is one contiguous 1 KiB (or 256 B) run.  (Read straight from M29 /
// DM29 the 64 lanes of a load hit 64 different cache lines 3 KiB apart: 1.7 MB of L2->L1 fills per blob for 160 KB
// of entries.)  Slots:  q (0..31)            leaf pair q of every lane          DM29[2 (32 lane + q)]
//                       32 + 32 - (32 >> (L-1)) + (q >> L)   in-lane merge at level L = 1..5   M29[((32 lane + q) >> (L-1)) & ~1]
//                       63 + (L - 6)         cross-lane merge at level L = 6..11  M29[(lane >> (L-6)) & ~1]
constexpr int EVAL_SLOTS = 69;
struct EvalTables {
    const uint4 *a, *b;   // limbs 0..3, 4..7
    const uint32_t* c;    // limb 8
};
__global__ void k_eval_tables(const Fr29Mem* __restrict__ M29, const Fr29Mem* __restrict__ DM29, uint4* __restrict__ ta,
                              uint4* __restrict__ tb, uint32_t* __restrict__ tc) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    const Fr29Mem* e;
    if (slot < 32) {
        e = DM29 + 2 * (32 * lane + slot);
    } else if (slot < 63) {
        int L = 1, rel = slot - 32;
        while (rel >= (32 >> L)) { rel -= 32 >> L; L++; }
        const int q = (rel << L) | ((1 << L) - 1);
        e = M29 + (((32 * lane + q) >> (L - 1)) & ~1);
    } else {
        e = M29 + ((lane >> (slot - 63)) & ~1);
    }
    ta[slot * 64 + lane] = make_uint4(e->l[0], e->l[1], e->l[2], e->l[3]);
    tb[slot * 64 + lane] = make_uint4(e->l[4], e->l[5], e->l[6], e->l[7]);
    tc[slot * 64 + lane] = e->l[8];
}
__device__ __forceinline__ Fr29 eval_table_load(const EvalTables& t, int slot, int lane) {
    const uint4 a = t.a[slot * 64 + lane], b = t.b[slot * 64 + lane];
    Fr29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = t.c[slot * 64 + lane];
    return r;
}

// The same entry through buffer descriptors (wave-uniform 128-bit resources in SGPRs): the per-lane part of the address is ONE
// 32-bit register that never changes (lane * 16, lane * 4), the slot travels in the scalar offset - no 64-bit address
// pairs in VGPRs and no vector address arithmetic in the loop.
struct EvalRsrc {
    __amdgpu_buffer_rsrc_t a, b, c;
};
__device__ __forceinline__ Fr29 eval_table_load(const EvalRsrc& t, int slot, uint32_t lane16, uint32_t lane4) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(t.a, lane16, slot * 1024, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(t.b, lane16, slot * 1024, 0);
    Fr29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = __builtin_amdgcn_raw_buffer_load_b32(t.c, lane4, slot * 256, 0);
    return r;
}

// Evaluation = k_eval_powers -> k_blob_evaluate -> k_eval_finish.  The serial parts of a blob's evaluation (13 squarings
// of z before the tree, 4 products after it) would run on one lane of the blob's wavefront at the price of 64: they are
// taken out into one-lane-per-blob kernels, with 576 bytes of scratch per blob in between:
//     [ Z[0..12] = z^(2^L) R' ; Z[13] = z R'^2 (for plain operands) ; tail: N0 , S R' ]   16 x 9 words
// z_in: plain little-endian limbs (any value < 2^256; reduced mod r here, like scalar_from_bytes_unchecked)
// y_out: plain little-endian canonical limbs.  status[b] |= 1 when a blob element is >= r
// (src/kzg_proof.rs:36-41 -> KzgError::BadArgs).
constexpr int EVAL_SCRATCH_WORDS = 16 * 9;
__global__ __launch_bounds__(64) void k_eval_powers(const Fr* __restrict__ z_in, uint32_t* __restrict__ scratch, int T) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= T) return;
    const Fr zin = z_in[b];
    const Fr29 r2 = fr29_const(c29::FR29_R2);
    Fr29* Z = reinterpret_cast<Fr29*>(scratch + (size_t)b * EVAL_SCRATCH_WORDS);
    Fr29 z = fr29_mul(fr29_from_words(zin.l), r2);  // any z < 2^256 < 2.3 r
    Z[0] = z;
    Z[13] = fr29_mul(z, r2);
    for (int l = 1; l <= 12; l++) {
        z = fr29_mul(z, z);
        Z[l] = z;
    }
}
// tail: N0 (the tree's root) and S R' (the sum of the blob's elements), both below 3r with limbs below 2^29
__global__ __launch_bounds__(64) void k_eval_finish(const uint32_t* __restrict__ scratch, Fr* __restrict__ y_out, int T) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= T) return;
    const Fr29* Z = reinterpret_cast<const Fr29*>(scratch + (size_t)b * EVAL_SCRATCH_WORDS);
    const Fr29 n = Z[14], sm = Z[15];
    // N = z N0 - (z^4096 - 1) S = z N0 + S - z^4096 S
    const Fr29 x = fr29_mul(sm, Z[12]);
    const Fr29 nw = fr29_sub_biased(fr29_add(fr29_mul(n, Z[0]), sm), x);
    const Fr29 y29 = fr29_mul(nw, fr29_const(c29::FR29_INV4096_PLAIN));  // (N R')(1/4096) R'^-1 = N/4096, below 2r
    Fr y;
    fr29_to_words(y.l, y29);
    y_out[b] = FrF::reduce_once(y);
}

// One wavefront per blob, four blobs per workgroup (single-wave workgroups are not spread evenly over the SIMDs).
// Register budget of two waves per SIMD (192 VGPRs used): with 168 the allocator reuses the prefetch registers and the
// wait counters turn the prefetches into stalls - three waves per SIMD were 2 % slower than two with working prefetches.
constexpr int EVAL_BLOBS_PER_BLOCK = 4;
constexpr int EVAL_SPREAD_LDS = 96 * 1024;  // dynamic LDS nobody touches: with it a CU holds ONE workgroup (launch_evaluate)
// SPLIT = true (round 4, the product's form): three wavefronts per SIMD - 166 VGPRs, no scratch - by (i) buffer descriptors
// instead of 64-bit address pairs, (ii) a line that is live once instead of twice (see the loop).  SPLIT = false is round 3's
// kernel (192 VGPRs, two wavefronts per SIMD), kept in the A/B build (option evaluate_kernel=r3) for measurement.
template <bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 3 : 2) void k_blob_evaluate_t(const uint8_t* __restrict__ blobs, const EvalTables tab,
                                                       uint32_t* __restrict__ scratch, uint32_t* __restrict__ status, int T,
                                                       unsigned long long* __restrict__ ktime) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blob_idx = blockIdx.x * EVAL_BLOBS_PER_BLOCK + wave;
    const bool active = blob_idx < T;
    __shared__ Fr29 Zs[EVAL_BLOBS_PER_BLOCK][14];              // Z[L] = z^(2^L) R'; Z[13] = z R'^2 (takes plain operands)
    __shared__ uint4 stack_as[EVAL_BLOBS_PER_BLOCK][4][64], stack_bs[EVAL_BLOBS_PER_BLOCK][4][64];  // levels 2..5, limbs 0..3 / 4..7, lane-major: conflict-free b128
    __shared__ uint32_t stack_cs[EVAL_BLOBS_PER_BLOCK][4][64];                                      // limb 8
    Fr29* Z = Zs[wave];
    uint4 (*stack_a)[64] = stack_as[wave], (*stack_b)[64] = stack_bs[wave];
    uint32_t (*stack_c)[64] = stack_cs[wave];
    if (!active) return;  // no workgroup barrier below: a wavefront only ever touches its own part of the LDS
    kstamp_in(ktime);
    // (the wavefront's blob index is uniform, but derived from threadIdx: readfirstlane makes that provable - the blob's
    // addresses then live in scalar registers, and a buffer operation is not wrapped in a waterfall loop)
    const int blob_u = SPLIT ? __builtin_amdgcn_readfirstlane(blob_idx) : blob_idx;
    uint32_t* const my = scratch + (size_t)blob_u * EVAL_SCRATCH_WORDS;
    for (int i = lane; i < 14 * 9; i += 64) reinterpret_cast<uint32_t*>(Z)[i] = my[i];
    const uint4* src = reinterpret_cast<const uint4*>(blobs + (size_t)blob_idx * BLOB_BYTES) + (size_t)lane * 128;
    const Fr29 zd = Z[13];
    bool bad = false;
    Fr29 n, psum;  // psum: this lane's share of S = sum_i p_i (plain integer, < 64 r)
#pragma unroll
    for (int i = 0; i < 9; i++) psum.l[i] = 0;
    // One iteration = one 128-byte line of the blob = four elements = two leaf pairs and their level-1 merge, all in
    // registers (a lane that fetched the two halves of a line in different iterations had them evicted in between:
    // 1.6x the blob in HBM reads).  Software pipelining: the next line, the two leaf entries of the next iteration and
    // the root entry of the next merge are requested one product ahead of their use (three waves per SIMD do not hide an
    // L2 / HBM round trip by themselves).
    uint4 cur[8];
    if constexpr (SPLIT) {
    const __amdgpu_buffer_rsrc_t rs_blob = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(blobs) + (size_t)blob_u * BLOB_BYTES, 0, BLOB_BYTES, 0x00020000);
    EvalRsrc rs;
    rs.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(tab.a), 0, EVAL_SLOTS * 64 * 16, 0x00020000);
    rs.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(tab.b), 0, EVAL_SLOTS * 64 * 16, 0x00020000);
    rs.c = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tab.c), 0, EVAL_SLOTS * 64 * 4, 0x00020000);
    const uint32_t lane2k = (uint32_t)lane * 2048u, lane16 = (uint32_t)lane * 16u, lane4 = (uint32_t)lane * 4u;
    auto blob_load = [&](int byte_off) {  // 16 bytes of this lane's 2 KiB of the blob; byte_off is uniform (scalar offset + immediate)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_blob, lane2k, byte_off, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
#pragma unroll
    for (int i = 0; i < 8; i++) cur[i] = blob_load(16 * i);
    Fr29 leaf0 = eval_table_load(rs, 0, lane16, lane4), leaf1 = eval_table_load(rs, 1, lane16, lane4);
    // Register diet for three wavefronts per SIMD (168 VGPRs): a pair's two elements are CONVERTED (byte swap, canonical check,
    // radix 2^29, sum and difference: 18 registers) before anything else of the iteration, which frees the 16 registers of
    // its half line - and the next line's half is requested into exactly those.  The line registers are then live once
    // (32) instead of twice (64), and the table entries are requested one product ahead of their use instead of a whole
    // iteration ahead.  Memory operations keep their order (sched_barrier 0x7: only ALU instructions may cross).
    struct PairOps {
        Fr29 s, u;
    };
    auto convert = [&](const uint4& a_hi, const uint4& a_lo, const uint4& b_hi, const uint4& b_lo) {
        Fr wa = fr_from_be_words(a_hi, a_lo), wb = fr_from_be_words(b_hi, b_lo);
        if (__any((wa.l[7] >= consts::FR_MOD[7]) | (wb.l[7] >= consts::FR_MOD[7]))) bad |= FrF::geq_mod(wa) | FrF::geq_mod(wb);
        const Fr29 pa = fr29_from_words(wa.l), pb = fr29_from_words(wb.l);
        PairOps o;
        o.s = fr29_add(pa, pb);
        o.u = fr29_sub_biased4(pa, pb);
        psum = fr29_add(psum, o.s);
        return o;
    };
    for (int j = 0; j < 16; j++) {  // pairs q = 2j, 2j + 1
        const int jn = j < 15 ? j + 1 : 15;
        const PairOps p0 = convert(cur[0], cur[1], cur[2], cur[3]);
        __builtin_amdgcn_sched_barrier(0x7);
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = blob_load(128 * jn + 16 * i);
        __builtin_amdgcn_sched_barrier(0x7);
        const Fr29 n0 = fr29_mul2(p0.s, zd, p0.u, leaf0);  // z s + roots[2k] u, k = 32 lane + q: ONE reduction (fr29.hpp)
        const PairOps p1 = convert(cur[4], cur[5], cur[6], cur[7]);
        __builtin_amdgcn_sched_barrier(0x7);
#pragma unroll
        for (int i = 4; i < 8; i++) cur[i] = blob_load(128 * jn + 16 * i);
        Fr29 root = eval_table_load(rs, 32 + j, lane16, lane4);  // the level-1 merge of the two pairs
        __builtin_amdgcn_sched_barrier(0x7);
        const Fr29 n1 = fr29_mul2(p1.s, zd, p1.u, leaf1);
        psum = fr29_normalize(psum);  // limbs: 2^29 + 2 * 2^30 < 2^32 between normalisations
        __builtin_amdgcn_sched_barrier(0x7);
        leaf0 = eval_table_load(rs, 2 * jn, lane16, lane4);
        leaf1 = eval_table_load(rs, 2 * jn + 1, lane16, lane4);
        Fr29 root_next = eval_table_load(rs, 48 + (j >> 1), lane16, lane4);  // the level-2 merge that follows when j is odd
        __builtin_amdgcn_sched_barrier(0x7);
        n = fr29_mul2(fr29_add(n0, n1), Z[1], fr29_sub_biased4(n0, n1), root);
        root = root_next;
        int level = 2;  // level of the merge that may follow: node index at level L is (32 lane + 2j + 1) >> L
        for (int jj = j; jj & 1; jj >>= 1) {
            Fr29 na;
            const uint4 h0 = stack_a[level - 2][lane], h1 = stack_b[level - 2][lane];
            na.l[0] = h0.x; na.l[1] = h0.y; na.l[2] = h0.z; na.l[3] = h0.w;
            na.l[4] = h1.x; na.l[5] = h1.y; na.l[6] = h1.z; na.l[7] = h1.w;
            na.l[8] = stack_c[level - 2][lane];
            root_next = eval_table_load(rs, 64 - (32 >> level) + (j >> level), lane16, lane4);  // for the merge one level up, if it follows (always a valid slot)
            const Fr29 sum = fr29_add(na, n), dif = fr29_sub_biased4(na, n);
            n = fr29_mul2(sum, Z[level], dif, root);
            root = root_next;
            level++;
        }
        if (j != 15) {  // level <= 5 here
            stack_a[level - 2][lane] = make_uint4(n.l[0], n.l[1], n.l[2], n.l[3]);
            stack_b[level - 2][lane] = make_uint4(n.l[4], n.l[5], n.l[6], n.l[7]);
            stack_c[level - 2][lane] = n.l[8];
        }
    }
    } else {
#pragma unroll
    for (int i = 0; i < 8; i++) cur[i] = src[i];
    Fr29 leaf0 = eval_table_load(tab, 0, lane), leaf1 = eval_table_load(tab, 1, lane);
    auto leaf_pair = [&](const uint4& a_hi, const uint4& a_lo, const uint4& b_hi, const uint4& b_lo, const Fr29& w) {
        Fr wa = fr_from_be_words(a_hi, a_lo), wb = fr_from_be_words(b_hi, b_lo);
        // canonical check (>= r -> BadArgs): the full 8-limb compare only when some lane's top word reaches r's
        if (__any((wa.l[7] >= consts::FR_MOD[7]) | (wb.l[7] >= consts::FR_MOD[7]))) bad |= FrF::geq_mod(wa) | FrF::geq_mod(wb);
        const Fr29 pa = fr29_from_words(wa.l), pb = fr29_from_words(wb.l);
        const Fr29 s = fr29_add(pa, pb), u = fr29_sub_biased4(pa, pb);
        psum = fr29_add(psum, s);
        return fr29_mul2(s, zd, u, w);  // z s + roots[2k] u, k = 32 lane + q: ONE reduction (fr29.hpp)
    };
    for (int j = 0; j < 16; j++) {  // pairs q = 2j, 2j + 1
        const int jn = j < 15 ? j + 1 : 15;
        // Loads complete in issue order (one vmcnt counter): what this iteration needs is requested first, what the next
        // one needs last, and no load sits in a branch (a control-flow merge makes the wait counters conservative).
        Fr29 root = eval_table_load(tab, 32 + j, lane);             // the level-1 merge of the two pairs
        Fr29 root_next = eval_table_load(tab, 48 + (j >> 1), lane);  // the level-2 merge that follows when j is odd
        __builtin_amdgcn_sched_barrier(0x7);
        uint4 nxt[8];
#pragma unroll
        for (int i = 0; i < 8; i++) nxt[i] = src[8 * jn + i];
        const Fr29 leaf0_next = eval_table_load(tab, 2 * jn, lane), leaf1_next = eval_table_load(tab, 2 * jn + 1, lane);
        __builtin_amdgcn_sched_barrier(0x7);
        const Fr29 n0 = leaf_pair(cur[0], cur[1], cur[2], cur[3], leaf0);
        const Fr29 n1 = leaf_pair(cur[4], cur[5], cur[6], cur[7], leaf1);
        psum = fr29_normalize(psum);  // limbs: 2^29 + 2 * 2^30 < 2^32 between normalisations
        n = fr29_mul2(fr29_add(n0, n1), Z[1], fr29_sub_biased4(n0, n1), root);
        root = root_next;
        int level = 2;  // level of the merge that may follow: node index at level L is (32 lane + 2j + 1) >> L
        for (int jj = j; jj & 1; jj >>= 1) {
            Fr29 na;
            const uint4 h0 = stack_a[level - 2][lane], h1 = stack_b[level - 2][lane];
            na.l[0] = h0.x; na.l[1] = h0.y; na.l[2] = h0.z; na.l[3] = h0.w;
            na.l[4] = h1.x; na.l[5] = h1.y; na.l[6] = h1.z; na.l[7] = h1.w;
            na.l[8] = stack_c[level - 2][lane];
            root_next = eval_table_load(tab, 64 - (32 >> level) + (j >> level), lane);  // for the merge one level up, if it follows (always a valid slot)
            const Fr29 sum = fr29_add(na, n), dif = fr29_sub_biased4(na, n);
            n = fr29_mul2(sum, Z[level], dif, root);
            root = root_next;
            level++;
        }
        if (j != 15) {  // level <= 5 here
            stack_a[level - 2][lane] = make_uint4(n.l[0], n.l[1], n.l[2], n.l[3]);
            stack_b[level - 2][lane] = make_uint4(n.l[4], n.l[5], n.l[6], n.l[7]);
            stack_c[level - 2][lane] = n.l[8];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = nxt[i];
        leaf0 = leaf0_next;
        leaf1 = leaf1_next;
    }
    }
    // n = N0_{6,lane}; fold across lanes
    for (int L = 6; L < 12; L++) {
        int sh = L - 6;
        Fr29 other = fr29_shfl_xor(n, 1 << sh);
        bool left = ((lane >> sh) & 1) == 0;  // node lane >> sh at level L
        Fr29 na, nb;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            na.l[i] = left ? n.l[i] : other.l[i];
            nb.l[i] = left ? other.l[i] : n.l[i];
        }
        const Fr29 sum = fr29_add(na, nb), dif = fr29_sub_biased4(na, nb);
        n = fr29_mul2(sum, Z[L], dif, eval_table_load(tab, 63 + sh, lane));
    }
    // S R' per lane (below 2r), then the sum over the 64 lanes (below 128 r), normalised every second step
    Fr29 sm = fr29_mul(psum, fr29_const(c29::FR29_R2));
    for (int sh = 1; sh < 64; sh <<= 1) {
        sm = fr29_add(sm, fr29_shfl_xor(sm, sh));
        if (sh & 0x2A) sm = fr29_normalize(sm);  // after steps 2, 4, 6
    }
    unsigned long long any_bad = __ballot(bad);
    sm = fr29_mul(sm, fr29_const(c29::FR29_ONE));  // same residue, value back below 3r (every lane holds the same sum)
    if (lane < 9) {
        uint32_t vn = n.l[0], vs = sm.l[0];
#pragma unroll
        for (int i = 1; i < 9; i++) {
            vn = lane == i ? n.l[i] : vn;
            vs = lane == i ? sm.l[i] : vs;
        }
        my[14 * 9 + lane] = vn;
        my[15 * 9 + lane] = vs;
    }
    if (lane == 0 && any_bad) atomicOr(&status[blob_u], 1u);
    kstamp_out(ktime);
}

// ---------------------------------------------------------------- quotient polynomial (prover side)
// c-kzg-4844's compute_kzg_proof_impl: the proof of p(z) = y is pi = sum_i q_i * g1_points[i] with, in evaluation form,
//     q_i = (p_i - y) / (w_i - z)   (i != m),      and if z is the root w_m:   q_m = sum_{i != m} (p_i - y) w_i / (z (z - w_i)).
// One wavefront per (blob, z); lane t owns elements 64t..64t+63.  The 4096 inversions are one Montgomery batch
// inversion: per-lane prefix products (parked in the output buffer), a product scan across the lanes, ONE Fermat
// inversion per blob, and the backward sweep.  Not on the verification path: plain 8x32 arithmetic, no tuning.
// z_in / y_in: plain little-endian limbs, canonical.  out: 4096 plain scalars per blob.  status |= 1: element >= r.
__device__ __forceinline__ Fr fr_shfl(const Fr& a, int src) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl((int)a.l[i], src, 64);
    return r;
}
__device__ inline Fr fr_inverse_mont(const Fr& a) {  // a != 0, Montgomery in and out
    return pow_limbs<Fr, 8>(a, consts::FR_R_MINUS_2, FrF::one(), [](const Fr& x, const Fr& y) { return FrF::mul(x, y); });
}
__global__ __launch_bounds__(64) void k_blob_quotient(const uint8_t* __restrict__ blobs, const Fr* __restrict__ z_in,
                                                      const Fr* __restrict__ y_in, const Fr* __restrict__ M, Fr* __restrict__ out,
                                                      uint32_t* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const Fr zM = FrF::to_mont(z_in[b]), y = y_in[b];
    const uint4* src = reinterpret_cast<const uint4*>(blobs + (size_t)b * BLOB_BYTES) + (size_t)lane * 128;
    Fr* o = out + (size_t)b * FE_PER_BLOB + 64 * lane;
    const Fr* Ml = M + 64 * lane;
    // pass 1: prefix products of d_k = w_k - z within the lane (a zero factor, z = w_m, is replaced by one)
    Fr acc = FrF::one();
    int special = -1;
    for (int k = 0; k < 64; k++) {
        Fr d = FrF::sub(Ml[k], zM);
        if (FrF::is_zero(d)) {
            special = k;
            d = FrF::one();
        }
        o[k] = acc;
        acc = FrF::mul(acc, d);
    }
    // products across lanes: incl[l] = T_0 .. T_l (inclusive scan), then the exclusive prefix and the grand total
    Fr incl = acc;
    for (int sh = 1; sh < 64; sh <<= 1) {
        Fr up = fr_shfl(incl, lane >= sh ? lane - sh : lane);
        if (lane >= sh) incl = FrF::mul(incl, up);
    }
    const Fr total = fr_shfl(incl, 63);
    Fr excl = fr_shfl(incl, lane ? lane - 1 : 0);
    if (lane == 0) excl = FrF::one();
    const Fr inv_total = fr_inverse_mont(total);
    // suffix products S_l = T_(l+1) .. T_63 by the same scan mirrored
    Fr sfx = acc;
    for (int sh = 1; sh < 64; sh <<= 1) {
        Fr dn = fr_shfl(sfx, lane + sh < 64 ? lane + sh : lane);
        if (lane + sh < 64) sfx = FrF::mul(sfx, dn);
    }
    Fr sfx_excl = fr_shfl(sfx, lane < 63 ? lane + 1 : 63);
    if (lane == 63) sfx_excl = FrF::one();
    // running = 1 / (everything up to and including this lane's elements); walk the lane backwards
    Fr running = FrF::mul(inv_total, sfx_excl);
    bool bad = false;
    Fr wsum = FrF::zero();  // sum_i q_i w_i (plain), only used when z is a root
    for (int k = 63; k >= 0; k--) {
        Fr d = FrF::sub(Ml[k], zM);
        if (k == special) d = FrF::one();
        const Fr inv_d = FrF::mul(running, FrF::mul(excl, o[k]));  // 1 / d_k
        running = FrF::mul(running, d);
        const Fr p = fr_from_be_words(src[2 * k], src[2 * k + 1]);
        bad |= FrF::geq_mod(p);
        Fr q = FrF::mul(FrF::sub(p, y), inv_d);  // plain * Montgomery = plain
        if (k == special) q = FrF::zero();
        o[k] = q;
        wsum = FrF::add(wsum, FrF::mul(q, Ml[k]));
    }
    const unsigned long long any_special = __ballot(special >= 0);
    if (any_special) {  // z = w_m: q_m = -(1/z) sum_{i != m} q_i w_i
        for (int sh = 1; sh < 64; sh <<= 1) wsum = FrF::add(wsum, fr_shfl_xor(wsum, sh));
        if (special >= 0) o[special] = FrF::neg(FrF::mul(wsum, fr_inverse_mont(zM)));
    }
    if (__ballot(bad) && lane == 0) atomicOr(&status[b], 1u);
}

}  // namespace kzg
