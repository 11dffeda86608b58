// msm.hpp - the random-linear-combination step of verify_kzg_proof_batch on gfx950
// (reference src/kzg_proof.rs:399-444; compute_powers :279-289).
//
// The reference computes three size-n MSMs plus n full scalar multiplications G*y_i (:419-430).
// The same group elements are obtained here from ONE pass:
//     A = sum r^i pi_i
//     B = sum r^i C_i + sum (r^i z_i) pi_i + (-(sum r^i y_i)) G
// i.e. a multi-scalar multiplication over the term list
//     t in [0,n)    : (pi_t,  a_t = r^t)        -> A
//     t in [n,2n)   : (pi_t', b_t = r^t' z_t')  -> B
//     t in [2n,3n)  : (C_t'', a_t'')            -> B
//     t = 3n        : (G,     g = -sum r^i y_i) -> B
//
// Pippenger, window c = 8 bits over 4 x 64-bit scalar chunks (see k_g1_multiples).
// One workgroup per (window, chunk, output):
//   1. counting sort of the output's terms by their 8-bit digit (LDS histogram + cursor),
//      so that afterwards every lane walks its own bucket list and all lanes add at once;
//   2. thread b accumulates bucket b with mixed Jacobian+affine additions;
//   3. sum_b b*B_b through row / column sums of the 16 x 16 bucket matrix (Jacobian points staged in LDS);
// then k_msm_combine folds the chunk sums and the 8 windows (Horner, 8 doublings per window).
#pragma once
#include <algorithm>
#include "dyn_lds.hpp"
#include "g1.hpp"
#include "g1_29.hpp"

namespace kzg {

constexpr int MSM_C = 8;
constexpr int MSM_WINDOWS = 32;  // windows per output: chunks x windows per chunk (4 x 8 or 16 x 2)
constexpr int MSM_BUCKETS = 256;

struct MsmTerm {
    uint32_t point;  // index into the affine point array
};

// ---------------------------------------------------------------- scalars
// thread i: rp = r^(offset+i); a[i] = rp (plain), b[i] = rp*z_i (plain); per-block partial of
// sum rp*y_i (Montgomery) to partial[blockIdx].   zs/ys: plain limbs.   r: plain limbs.
// `offset` is the global index of this shard's first blob (multi-GPU: rank k starts its power
// table at r^(k n/N) with one square-and-multiply per thread; compute_powers, src/kzg_proof.rs:279-289).
// blockIdx.y = batch b of a launch group: r[b], zs/ys + b*n, scalars + b*(2n+1) (a at +0, b at +n), partial + b*gridDim.x.
__global__ __launch_bounds__(256) void k_batch_scalars(const Fr* __restrict__ r_plain, const Fr* __restrict__ zs,
                                                       const Fr* __restrict__ ys, Fr* __restrict__ scalars,
                                                       Fr* __restrict__ partial, int n, unsigned long long offset) {
    __shared__ Fr red[256];
    const int bt = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    zs += (size_t)bt * n;
    ys += (size_t)bt * n;
    Fr* a_out = scalars + (size_t)bt * (2 * n + 1);
    Fr* b_out = a_out + n;
    partial += (size_t)bt * gridDim.x;
    Fr acc = FrF::zero();
    if (i < n) {
        Fr r = FrF::to_mont(r_plain[bt]);
        Fr rp = FrF::one();
        unsigned long long e = offset + (unsigned long long)i;
        for (int b = 63 - __clzll(e | 1ull); b >= 0; b--) {
            rp = FrF::sqr(rp);
            if ((e >> b) & 1) rp = FrF::mul(rp, r);
        }
        a_out[i] = FrF::from_mont(rp);
        b_out[i] = FrF::mul(rp, zs[i]);        // (r^i R) z R^-1 = r^i z, plain
        acc = FrF::mul(rp, FrF::to_mont(ys[i]));  // Montgomery
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = FrF::add(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// g = -(sum of partials), plain limbs; block b = batch b: partial + b*nparts -> scalars[b*(2n+1) + 2n]
__global__ void k_finish_g(const Fr* __restrict__ partial, int nparts, Fr* __restrict__ scalars, int n) {
    if (threadIdx.x) return;
    const int bt = blockIdx.x;
    Fr s = FrF::zero();
    for (int i = 0; i < nparts; i++) s = FrF::add(s, partial[(size_t)bt * nparts + i]);
    scalars[(size_t)bt * (2 * n + 1) + 2 * n] = FrF::from_mont(FrF::neg(s));
}

// ---------------------------------------------------------------- scalar chunks and their points
// A 255-bit scalar is split into FOUR 64-bit chunks so that the MSM needs only 8 windows of 8 bits and the
// serial window combine 56 doublings instead of 248:
//     k = k1 + k2 * L,  L = x^2 (GLV: on G1, phi(P) = (beta x, y) = -[x^2]P, so [L]P = -phi(P)),  k1, k2 < 2^128
//     k P = k1_lo P + k1_hi (2^64 P) + k2_lo (-phi P) + k2_hi (-phi(2^64 P)).
// mult[0..3][p] = P, 2^64 P, -phi(P), -phi(2^64 P) (Jacobian): 64 doublings + 2 multiplications per point, which
// depend only on the INPUT points.  On the verification path they come out of the decode pass for free
// (k_g1_decode_multiples below shares the doubling chain with the subgroup test, beside the SHA-256 challenge chain);
// k_g1_multiples is the stand-alone form for points that are already decoded (the generator, kzg_g1_msm).
// Points must lie in G1 (all callers decode with the subgroup check).
//
// Two layouts.  CHUNKS = 4 (64-bit chunks, 8 windows each) is the default: the 2^64 multiple falls out of the decode
// pass for free and the window combine is 56 doublings.  CHUNKS = 32 (8-bit chunks, ONE window each: multiples
// 2^8 P .. 2^120 P and their -phi images) costs 56 more doublings per point in the decode pass - which runs beside the
// longer SHA-256 chain when a single batch is verified - and leaves the combine no doubling at all, a tree over the 32
// chunk sums: the LATENCY layout, used for small launches only (it would add a quarter to the decode work of a
// throughput launch and its tables are 32 x 192 bytes per point).  (Round 1 had 16 chunks of 16 bits: 8 doublings in
// the combine, ~0.1 ms of a single batch.)
constexpr int MSM_CHUNKS = 4;
constexpr int MSM_CHUNKS_LATENCY = 32;
constexpr int MSM_CHUNKS_PROOFS = 16;  // the proof-tuple entry points have no SHA-256 chain to hide the decode pass behind: 16-bit
                                       // chunks (two windows each, 8 doublings in the combine) keep that pass 0.6 ms shorter
constexpr int MSM_ENTRY_CHUNK_SHIFT = 27;  // a sorted-list entry: chunk << 27 | point index
constexpr uint32_t MSM_ENTRY_POINT_MASK = (1u << MSM_ENTRY_CHUNK_SHIFT) - 1;
__device__ __forceinline__ G1Jac g1_neg_phi(const G1Jac& p) {
    G1Jac r;
    r.x = fp_mul(p.x, fp_const(consts::FP_BETA_MONT));
    r.y = fp_neg(p.y);
    r.z = p.z;
    return r;
}
// table j of `chunks` tables: j < chunks/2 -> 2^(step j) P, j >= chunks/2 -> -phi of table j - chunks/2; step = 256 / chunks
__global__ __launch_bounds__(64, 4) void k_g1_multiples(const G1Aff* __restrict__ points, const uint32_t* __restrict__ pflag,
                                                     G1Jac* __restrict__ mult, int n, int stride, int chunks) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int half = chunks / 2, step = 256 / chunks;
    G1Jac acc = pflag[i] ? g1_identity() : g1_from_affine(points[i]);
    for (int j = 0; j < half; j++) {
        mult[(size_t)j * stride + i] = acc;
        mult[(size_t)(half + j) * stride + i] = g1_neg_phi(acc);
        if (j + 1 < half)
            for (int k = 0; k < step; k++) acc = g1_dbl(acc);
    }
}

// Decode + subgroup check + multiples in one pass: the subgroup test's first scalar multiplication walks the same
// doubling chain that produces 2^64 P (g1.hpp g1_in_subgroup_with_multiple), which saves the 64 doublings of a
// separate k_g1_multiples pass.  bytes0 holds points [0, n0), bytes1 points [n0, n).
template <int CHUNKS>
__global__ __launch_bounds__(64, 2) void k_g1_decode_multiples(const uint8_t* __restrict__ bytes0, const uint8_t* __restrict__ bytes1,
                                                               int n0, G1Aff* __restrict__ points, uint32_t* __restrict__ pflag,
                                                               G1Jac* __restrict__ mult, int n, int stride) {
    constexpr int HALF = CHUNKS / 2, STEP = 256 / CHUNKS;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = i < n0 ? bytes0 + (size_t)i * 48 : bytes1 + (size_t)(i - n0) * 48;
    G1Aff a;
    uint32_t st = g1_decompress(a, src, false);
    if (st == G1_OK) {
        const G1Jac p = g1_from_affine(a);
        mult[i] = p;
        mult[(size_t)HALF * stride + i] = g1_neg_phi(p);
        const bool in = g1_in_subgroup_with_multiples<STEP>(a, [&](int k, const G1Jac& m) {
            mult[(size_t)k * stride + i] = m;
            mult[(size_t)(HALF + k) * stride + i] = g1_neg_phi(m);
        });
        if (!in) st = G1_INVALID;
    }
    if (st != G1_OK) {
        a.x = FpF::zero();
        a.y = FpF::zero();
        const G1Jac id = g1_identity();
        for (int k = 0; k < CHUNKS; k++) mult[(size_t)k * stride + i] = id;
    }
    points[i] = a;
    pflag[i] = st;
}

// diagnostic (tools/prof/decode_placement.py): where the first wavefront of the last latency decode ran, and for how long
__device__ unsigned long long g_decode_dbg[4];  // HW_ID | XCC_ID | shader cycles | wall-clock ticks (100 MHz)

#ifndef KZG_DECODE_OCC
#define KZG_DECODE_OCC 2
#endif
// The same pass in the radix-2^29 field (fp29.hpp, g1_29.hpp): tables written as G1Jac29Mem (lazy values), the affine
// point as a canonical 12x32 element like the kernel above.
// AFF (CHUNKS = 4 only): the AFFINE table layout of the throughput path.  P and -phi(P) are affine as they are; the one
// Jacobian multiple 2^64 P goes to jtmp[i] and k_mult_to_affine29 below turns it into table rows 1 and 3, so that the
// window kernel's bucket additions are mixed additions (8M + 3S instead of 12M + 4S).
template <int CHUNKS, bool AFF>
__global__ __launch_bounds__(256, KZG_DECODE_OCC) void k_g1_decode_multiples29(const uint8_t* __restrict__ bytes0, const uint8_t* __restrict__ bytes1,
                                                                 int n0, G1Aff* __restrict__ points, uint32_t* __restrict__ pflag,
                                                                 void* __restrict__ mult_, G1Jac29Mem* __restrict__ jtmp, int n, int stride,
                                                                 unsigned long long* __restrict__ ktime = nullptr) {
    static_assert(!AFF || CHUNKS == 4, "the affine layout has one Jacobian multiple per point");
    constexpr int HALF = CHUNKS / 2, STEP = 256 / CHUNKS;
    extern __shared__ __attribute__((aligned(16))) uint4 park4[];  // PARK_UINT4_PER_THREAD per thread (g1_29.hpp LdsPark)
    const LdsPark pk = lds_park(park4 + threadIdx.x, blockDim.x);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kstamp_in(ktime);
    const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_t0 = wall_clock64();
    const uint8_t* src = i < n0 ? bytes0 + (size_t)i * 48 : bytes1 + (size_t)(i - n0) * 48;
    Fp29 x, y;
    uint32_t st = g1_decompress29(x, y, src, pk);
    G1Aff a;
    a.x = FpF::zero();
    a.y = FpF::zero();
    if (st == G1_OK) {
        bool in;
        if constexpr (AFF) {
            G1Aff29Mem* mult = static_cast<G1Aff29Mem*>(mult_);
            g1a29_store(mult[i], x, y);
            g1a29_store(mult[(size_t)HALF * stride + i], fp29_mul(x, fp29_const(cp29::FP29_BETA_MONT)), fp29_neg<3>(y));  // y < 4p
            in = g1j29_in_subgroup_with_multiples<STEP>(x, y, pk, [&](int, const G1Jac29& m) { g1j29_store(jtmp[i], m); });
        } else {
            G1Jac29Mem* mult = static_cast<G1Jac29Mem*>(mult_);
            G1Jac29 p;
            p.x = x;
            p.y = y;
            p.z = fp29_const(cp29::FP29_ONE);
            g1j29_store(mult[i], p);
            g1j29_store(mult[(size_t)HALF * stride + i], g1j29_neg_phi(p));
            in = g1j29_in_subgroup_with_multiples<STEP>(x, y, pk, [&](int k, const G1Jac29& m) {
                g1j29_store(mult[(size_t)k * stride + i], m);
                g1j29_store(mult[(size_t)(HALF + k) * stride + i], g1j29_neg_phi(m));
            });
        }
        if (in) {
            unpark_xy(pk, x, y);  // (x, y) were parked across the chains, not kept in registers
            a.x = fp29_to_std(x);
            a.y = fp29_to_std(y);
        } else {
            st = G1_INVALID;
        }
    }
    if (st != G1_OK && !AFF) {  // (affine entries of a flagged point are never read)
        G1Jac29Mem* mult = static_cast<G1Jac29Mem*>(mult_);
        const G1Jac29 id = g1j29_identity();
        for (int k = 0; k < CHUNKS; k++) g1j29_store(mult[(size_t)k * stride + i], id);
    }
    points[i] = a;
    pflag[i] = st;
    if (!AFF && i == 0) {
        g_decode_dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_decode_dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_decode_dbg[2] = __builtin_readcyclecounter() - dbg_c0;
        g_decode_dbg[3] = wall_clock64() - dbg_t0;
    }
    kstamp_out(ktime);
}

// jtmp[i] = 2^64 P_i (Jacobian) -> table rows 1 and 3 of the affine layout: (X / Z^2, Y / Z^3) and its -phi image.
// Thread t owns the K points t, t + nthreads, ...: ONE inversion per thread (Montgomery's trick; the prefix products
// are parked in the x slot of the output row until the backward sweep overwrites it).  Flagged points are skipped.
constexpr int AFFINE_BATCH = 16;
__global__ __launch_bounds__(64, 2) void k_mult_to_affine29(const G1Jac29Mem* __restrict__ jtmp, const uint32_t* __restrict__ pflag,
                                                           G1Aff29Mem* __restrict__ mult, int n, int stride) {
    const int nthreads = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    G1Aff29Mem* row1 = mult + stride;
    G1Aff29Mem* row3 = mult + (size_t)3 * stride;
    Fp29 acc = fp29_const(cp29::FP29_ONE);
#pragma unroll 1
    for (int k = 0; k < AFFINE_BATCH; k++) {
        const int i = t + k * nthreads;
        if (i >= n || pflag[i]) continue;
        fp29_store(row1[i].x, acc);
        acc = fp29_mul(acc, fp29_load(jtmp[i].z));  // Z of a point of G1 that is not the identity: never 0 mod p
    }
    Fp29 inv = fp29_inverse(acc);
    const Fp29 beta = fp29_const(cp29::FP29_BETA_MONT);
#pragma unroll 1
    for (int k = AFFINE_BATCH - 1; k >= 0; k--) {
        const int i = t + k * nthreads;
        if (i >= n || pflag[i]) continue;
        const Fp29 zi = fp29_mul(inv, fp29_load(row1[i].x));  // 1 / Z_i
        inv = fp29_mul(inv, fp29_load(jtmp[i].z));
        const Fp29 zi2 = fp29_sqr(zi), zi3 = fp29_mul(zi2, zi);
        const Fp29 x = fp29_mul(fp29_load(jtmp[i].x), zi2), y = fp29_mul(fp29_load(jtmp[i].y), zi3);  // < 2p
        g1a29_store(row1[i], x, y);
        g1a29_store(row3[i], fp29_mul(x, beta), fp29_neg<2>(y));  // < 4p
    }
}

// Jacobian table entries -> affine ones, one inversion each (set-up time only: the generator's multiples)
__global__ void k_jac29_to_aff29(const G1Jac29Mem* __restrict__ in, G1Aff29Mem* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Jac29 p = g1j29_load(in[i]);
    const Fp29 zi = fp29_inverse(p.z), zi2 = fp29_sqr(zi);
    g1a29_store(out[i], fp29_mul(p.x, zi2), fp29_mul(p.y, fp29_mul(zi2, zi)));
}

// tables made in the 12x32 form (k_g1_multiples: the generator, kzg_g1_msm) -> the radix-2^29 table format
__global__ void k_jac_to_jac29(const G1Jac* __restrict__ in, G1Jac29Mem* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1j29_store(out[i], g1j29_from_std(in[i]));
}

// in place: canonical k (< r) -> k1 (limbs 0..3) | k2 (limbs 4..7) with k = k1 + k2 * x^2.
// Barrett with M = floor(2^383 / x^2): the quotient estimate is low by at most 1 for k < 2^255.
// (Round 5 tried a DIGIT-major copy of the split scalars - digits_t[b * count + i] = byte b of scalar i - so that the window
// kernel's sort reads consecutive bytes for consecutive terms instead of one byte per 32-byte scalar: the window kernel's HBM
// bytes went from 1.182 to 1.169 GB per launch group and its time and SQ_WAIT_ANY share did not move - those byte reads were L2
// hits, the kernel's traffic is its table rows (4 x 128 B per term, 0.40 GB per group, read ~1.6x) and the bucket sums between
// the passes (0.20 GB).  Not kept.  profiles/r5_pmc.json vs r4_pmc.json.)
__global__ __launch_bounds__(256) void k_glv_split(Fr* __restrict__ scalars, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr k = scalars[i];
    uint32_t prod[16];
#pragma unroll
    for (int j = 0; j < 16; j++) prod[j] = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        uint64_t c = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t t = (uint64_t)k.l[a] * consts::GLV_M[b] + prod[a + b] + c;
            prod[a + b] = (uint32_t)t;
            c = t >> 32;
        }
        prod[a + 8] = (uint32_t)c;
    }
    uint32_t q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = (prod[11 + j] >> 31) | (prod[12 + j] << 1);
    // rem = k - q * L  (fits in 160 bits: rem < 2 L)
    uint32_t ql[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ql[j] = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        uint64_t c = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint64_t t = (uint64_t)q[a] * consts::GLV_L[b] + ql[a + b] + c;
            ql[a + b] = (uint32_t)t;
            c = t >> 32;
        }
        ql[a + 4] = (uint32_t)c;
    }
    uint32_t rem[5], borrow = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) rem[j] = subb(k.l[j], ql[j], borrow);
    // if rem >= L: rem -= L, q += 1
    uint32_t d[5], bo = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) d[j] = subb(rem[j], j < 4 ? consts::GLV_L[j] : 0u, bo);
    const bool ge = bo == 0;
    uint32_t carry = ge ? 1u : 0u;
    Fr out;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        out.l[j] = ge ? d[j] : rem[j];
        out.l[4 + j] = addc(q[j], 0u, carry);
    }
    scalars[i] = out;
}

// ---------------------------------------------------------------- bucket accumulation + reduction
// Term t of output o is (point index, scalar index) given by two small tables:
//   term_point[o][t], term_scalar[o][t]  (indices into points[] / scalars[]); nterms[o].
// pflag[p] != 0 marks the identity / an invalid point (skipped).
struct MsmDesc {
    const void* mult;             // [chunks][stride] precomputed multiples: G1Jac (Curve32) or G1Jac29Mem (Curve29)
    const uint32_t* pflag;
    const Fr* scalars;            // plain little-endian limbs
    const uint32_t* term_point;   // [2][max_terms]
    const uint32_t* term_scalar;  // [2][max_terms]
    uint32_t* sorted;             // [2][32][max_terms] scratch
    G1Jac* window_sums;           // [2][4][8]
    int nterms[2];
    int max_terms;
    int stride;
    int slices;                   // term slices per output (power of two, >= 1): blockIdx.z = (2 batch + output) * slices +
                                  // slice; slice k sums terms [k nt / slices, (k + 1) nt / slices) into its own window sums
                                  // [2B][slots][slices][W], folded by k_msm_fold_slices - what keeps ONE large batch
                                  // (tens of thousands of terms per output) from running on a few dozen workgroups
    int chunks;                   // MSM_CHUNKS or MSM_CHUNKS_LATENCY; windows per chunk = 32 / chunks = gridDim.x
    uint32_t* save;               // [blocks of this launch][WORDS][256] words: the block's bucket sums between the row and the
                                  // column trees of the reduction (kept in registers they were spilled: 2.7 KB of scratch
                                  // writes per thread)
    int z0;                       // logical blockIdx.z of this launch's first z-layer (a grid too large for the save area
                                  // is launched in pieces)
    int flags;                    // MSM_FLAG_*
    unsigned long long* ktime;    // (optional) the window kernel's execution interval (field.hpp kstamp_in / kstamp_out)
    int chunks_per_block;         // 1: a block per (window, chunk) - most parallel, lowest latency;  4: a block per window
                                  // sums the four chunks' terms into ONE bucket set - a quarter of the reductions and
                                  // fuller, better balanced buckets (throughput mode).  gridDim.y = chunks / chunks_per_block
};

// MSM_FLAG_ROTATE: which wavefront of a block takes the fullest buckets rotates with the block.  The buckets are handed out
//   by rank (wavefront 0 the 64 fullest ... wavefront 3 the 64 emptiest) and a workgroup's wavefronts are dealt to the CU's
//   four SIMDs in order, so without the rotation one SIMD of every CU collects the long bucket lists of every block resident
//   there and another one the short ones.
// MSM_FLAG_XCD: the 8 window blocks of one (batch, output) run on the SAME XCD, one after the other in dispatch order - they
//   read the same table rows (each row once per window), which then come out of that XCD's L2 instead of crossing the fabric 8
//   times.  Workgroups go to the 8 XCDs round-robin by linear id, so (with gridDim = (8, 1, gz), gz a multiple of 8) the block
//   with linear id L works on window (L / 8) % 8 of z = (L / 64) * 8 + L % 8.
constexpr int MSM_SAVE2_WORDS = 48;  // a point of the two-pass form's save area: X | Y | Z, 16 words each (14 limbs + 2 of padding)
constexpr int MSM_FLAG_ROTATE = 1, MSM_FLAG_XCD = 2, MSM_FLAG_BFIRST = 4;
// MSM_FLAG_BFIRST (set by msm_window_launch for a one-piece launch without slices): the blocks of the outputs B (2 n + 1 terms,
//   twice the work) are dispatched before those of the outputs A, so that the launch ends on short blocks.

__device__ __forceinline__ void lds_store_jac(uint32_t* base, int slot, const G1Jac& p) {
    uint32_t* d = base + slot * 36;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        d[i] = p.x.l[i];
        d[12 + i] = p.y.l[i];
        d[24 + i] = p.z.l[i];
    }
}
__device__ __forceinline__ G1Jac lds_load_jac(const uint32_t* base, int slot) {
    const uint32_t* d = base + slot * 36;
    G1Jac p;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p.x.l[i] = d[i];
        p.y.l[i] = d[12 + i];
        p.z.l[i] = d[24 + i];
    }
    return p;
}

// ---- the sum of a small batch's window sums with FOUR LANES PER ADDITION (the latency layout: W = 1, up to 128 sums per
// output).  A Jacobian addition is 16 products of dependency depth 5; on one lane of a wavefront that is alone on its SIMD
// it takes ~15 us, and the fold over the slices and the tree over the chunks were 7 of them in a row (k_msm_fold_slices +
// k_msm_combine: 0.2 ms of a 5 ms batch).  Here the quad (lanes 4q .. 4q + 3) runs one addition as five ROUNDS of one
// product per lane; a round's results go through three 16-word LDS slots per lane (same wavefront: the LDS executes its
// instructions in order, no barrier inside an addition) and the next round's operands are read back by per-lane address:
//     round 1   Z1Z1 = Z1^2        Z2Z2 = Z2^2        Y1Z2 = Y1 Z2         Y2Z1 = Y2 Z1
//     round 2   U1 = X1 Z2Z2       U2 = X2 Z1Z1       S1 = Y1Z2 Z2Z2       S2 = Y2Z1 Z1Z1
//               H = U2 - U1 (lanes 0, 1)              R = S2 - S1 (lanes 2, 3)
//     round 3   HH = H^2           Z1H = Z1 H         RR = R^2             (RR again)
//     round 4   HHH = H HH         Z3 = Z1H Z2        V = U1 HH            (V again)
//               X3 = RR - HHH - 2V (lane 2)
//     round 5   T = S1 HHH         -                  R (V - X3)           -
//               Y3 = R (V - X3) - T (lane 2)
// (the formulas and bounds of g1j29_add, g1_29_formulas.hpp).  An identity operand or equal x (P + P, P - P) shows in
// Z1Z1, Z2Z2, HH; such a quad lets its lane 0 run the complete g1j29_add afterwards.
constexpr int SUMQ_POINT_WORDS = 48;   // X | Y | Z, 16 words each (14 limbs + 2 of padding: ds_*_b128)
// per quad: 4 lanes x 3 slots x 16 words, the lanes 68 words apart and the quads 272: a b128 access of sixteen lanes (four
// quads) then touches sixteen different groups of four banks (at 48 / 192 words the four quads of a pass hit the same banks)
constexpr int SUMQ_LANE_WORDS = 68;
constexpr int SUMQ_SCRATCH_WORDS = 4 * SUMQ_LANE_WORDS;
constexpr int SUMQ_MAX_POINTS = 128;
__device__ __forceinline__ Fp29 sumq_load(const uint32_t* w) {
    const uint4* q = reinterpret_cast<const uint4*>(w);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    Fp29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void sumq_store(uint32_t* w, const Fp29& v) {
    uint4* q = reinterpret_cast<uint4*>(w);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
// point layouts in LDS: coordinate c of the point at p
struct SumqLayout16 {  // X | Y | Z at 16-word strides (this file's sum kernel)
    __device__ static __forceinline__ Fp29 ld(const uint32_t* p, int c) { return sumq_load(p + 16 * c); }
    __device__ static __forceinline__ void st(uint32_t* p, int c, const Fp29& v) { sumq_store(p + 16 * c, v); }
};
struct SumqLayout14 {  // lds_store_jac29's 42-word slots (the window kernel's bucket arrays)
    __device__ static __forceinline__ Fp29 ld(const uint32_t* p, int c) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = p[14 * c + i];
        return r;
    }
    __device__ static __forceinline__ void st(uint32_t* p, int c, const Fp29& v) {
#pragma unroll
        for (int i = 0; i < 14; i++) p[14 * c + i] = v.l[i];
    }
};
// OUT <- P + Q (points at P and Q; OUT may be P; scr: the quad's scratch, r = lane & 3).  All four lanes of the quad call
// this together.
template <class LY>
__device__ __forceinline__ void g1j29_add_quad(const uint32_t* P, const uint32_t* Q, uint32_t* OUT, uint32_t* scr, int r, int lane) {
    auto slot = [&](int ln, int which) { return scr + ln * SUMQ_LANE_WORDS + which * 16; };
    const bool r0 = r == 0, r1 = r == 1, r2 = r == 2;
    // round 1
    const Fp29 a1 = LY::ld((r0 || r2) ? P : Q, (r0 || r1) ? 2 : 1);  // Z1 | Z2 | Y1 | Y2
    const Fp29 b1 = LY::ld((r1 || r2) ? Q : P, 2);                    // Z1 | Z2 | Z2 | Z1
    const Fp29 t1 = fp29_mul(a1, b1);
    const unsigned long long zb1 = __ballot(fp29_is_zero_mod_p(t1));
    sumq_store(slot(r, 0), t1);
    // round 2
    const Fp29 x2 = LY::ld(r0 ? P : Q, 0);  // (only lanes 0, 1 use it)
    Fp29 a2;
#pragma unroll
    for (int i = 0; i < 14; i++) a2.l[i] = r < 2 ? x2.l[i] : t1.l[i];
    const Fp29 b2 = sumq_load(slot((r & 1) ? 0 : 1, 0));  // Z2Z2 for lanes 0, 2; Z1Z1 for lanes 1, 3
    const Fp29 t2 = fp29_mul(a2, b2);
    sumq_store(slot(r, 1), t2);
    const Fp29 o2 = sumq_load(slot(r ^ 1, 1));
    Fp29 hi, lo;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        hi.l[i] = (r & 1) ? t2.l[i] : o2.l[i];
        lo.l[i] = (r & 1) ? o2.l[i] : t2.l[i];
    }
    const Fp29 D = fp29_sub<2>(hi, lo);  // H on lanes 0, 1; R on lanes 2, 3; below 6p
    // round 3
    const Fp29 z1 = LY::ld(P, 2);
    Fp29 b3;
#pragma unroll
    for (int i = 0; i < 14; i++) b3.l[i] = r1 ? z1.l[i] : D.l[i];
    const Fp29 t3 = fp29_mul(D, b3);  // HH | Z1 H | RR | RR
    const unsigned long long zb3 = __ballot(fp29_is_zero_mod_p(t3));
    sumq_store(slot(r, 0), t3);  // (every lane has read its round-2 operand from slot 0 by now)
    // round 4
    const Fp29 u1 = sumq_load(slot(0, 1)), hh = sumq_load(slot(0, 0)), z2 = LY::ld(Q, 2);
    Fp29 a4, b4;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a4.l[i] = r0 ? D.l[i] : r1 ? t3.l[i] : u1.l[i];
        b4.l[i] = r0 ? t3.l[i] : r1 ? z2.l[i] : hh.l[i];
    }
    const Fp29 t4 = fp29_mul(a4, b4);  // HHH | Z3 | V | V
    sumq_store(slot(r, 2), t4);
    const Fp29 hhh = sumq_load(slot(0, 2));
    const Fp29 X3 = fp29_sub<3>(fp29_sub<2>(t3, hhh), fp29_dbl(t4));  // lanes 2, 3: RR + 4p - HHH + 8p - 2V < 14p
    // round 5
    const Fp29 s1 = sumq_load(slot(2, 1));
    const Fp29 vx = fp29_sub<5>(t4, X3);
    Fp29 a5, b5;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a5.l[i] = r0 ? s1.l[i] : D.l[i];
        b5.l[i] = r0 ? t4.l[i] : vx.l[i];
    }
    const Fp29 t5 = fp29_mul(a5, b5);  // T | - | R (V - X3) | -
    sumq_store(slot(r, 0), t5);
    const Fp29 T = sumq_load(slot(0, 0));
    const Fp29 Y3 = fp29_sub<2>(t5, T);  // lane 2: below 6p
    // the quad's flags: bit 0 Z1Z1 = 0, bit 1 Z2Z2 = 0 (round 1); bit 0 HH = 0 (round 3)
    const int q0 = lane & ~3;
    const bool special = (((zb1 >> q0) & 3ull) != 0) || (((zb3 >> q0) & 1ull) != 0);
    if (!special) {
        if (r2) {
            LY::st(OUT, 0, X3);
            LY::st(OUT, 1, Y3);
        }
        if (r1) LY::st(OUT, 2, t4);
    } else if (r0) {
        G1Jac29 p, q;
        p.x = LY::ld(P, 0); p.y = LY::ld(P, 1); p.z = LY::ld(P, 2);
        q.x = LY::ld(Q, 0); q.y = LY::ld(Q, 1); q.z = LY::ld(Q, 2);
        const G1Jac29 s = g1j29_add(p, q);
        LY::st(OUT, 0, s.x);
        LY::st(OUT, 1, s.y);
        LY::st(OUT, 2, s.z);
    }
}
// OUT <- 2 P with three lanes of a quad (dbl-2009-l as in g1j29_dbl, same bounds): three rounds instead of seven products in
// a row.   round 1: B = Y^2 | A = X^2 | YZ        round 2: C = B^2 | F = (3A)^2 | t = (X + B)^2        round 3 (lane 1):
// E (D + 512p - X3), with D = 2 (t - A - C) + 16p, X3 = F + 128p - 2D; lane 1 writes X3, Y3 and lane 2 Z3 = 2 YZ.
// scr: the quad's scratch (slots 0 and 1 of every lane are used).  OUT may be P.  No special cases: Z = 0 stays so.
template <class LY>
__device__ __forceinline__ void g1j29_dbl_quad(const uint32_t* P, uint32_t* OUT, uint32_t* scr, int r) {
    auto slot = [&](int ln, int which) { return scr + ln * SUMQ_LANE_WORDS + which * 16; };
    const bool r0 = r == 0, r1 = r == 1;
    // round 1
    const Fp29 a1 = LY::ld(P, r1 ? 0 : 1);             // Y | X | Y | Y
    const Fp29 b1 = LY::ld(P, r0 ? 1 : r1 ? 0 : 2);    // Y | X | Z | Z
    const Fp29 t1 = fp29_mul(a1, b1);                  // B | A | YZ | (YZ)
    sumq_store(slot(r, 0), t1);
    // round 2
    const Fp29 x = LY::ld(P, 0), Bv = sumq_load(slot(0, 0));
    Fp29 E, a2;  // E = 3A on lane 1
#pragma unroll
    for (int i = 0; i < 14; i++) E.l[i] = (t1.l[i] << 1) + t1.l[i];  // words < 2^31
    E = fp29_normalize(E);                                             // < 6p
    const Fp29 xb = fp29_add(x, Bv);
#pragma unroll
    for (int i = 0; i < 14; i++) a2.l[i] = r0 ? t1.l[i] : r1 ? E.l[i] : xb.l[i];
    const Fp29 t2 = fp29_sqr(a2);                      // C | F | t | (t)
    sumq_store(slot(r, 1), t2);
    // lane 1: D, X3, then round 3
    const Fp29 C = sumq_load(slot(0, 1)), t = sumq_load(slot(2, 1));
    Fp29 D, X, C8;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        D.l[i] = ((t.l[i] - t1.l[i] - C.l[i]) << 1) + cp29::FP29_BIASX4[i];  // lane 1: t1 = A
        C8.l[i] = C.l[i] << 3;
    }
    D = fp29_normalize(D);    // < 20p
    C8 = fp29_normalize(C8);  // < 16p
#pragma unroll
    for (int i = 0; i < 14; i++) X.l[i] = t2.l[i] + cp29::FP29_BIASW7[i] - (D.l[i] << 1);  // lane 1: t2 = F
    X = fp29_normalize(X);    // < 130p
    const Fp29 Y3 = fp29_sub<5>(fp29_mul(E, fp29_sub<9>(D, X)), C8);  // < 34p
    if (r1) {
        LY::st(OUT, 0, X);
        LY::st(OUT, 1, Y3);
    }
    if (r == 2) LY::st(OUT, 2, fp29_dbl(t1));  // < 4p
}

// The latency form of the pass (Jacobian tables; launches that leave most of the chip idle): EIGHT LANES PER POINT, two quads.
// Lane 0 decodes (the square root is one chain of products), then quad A walks the doubling chain R = 2^k P with the point in
// LDS - doublings in three product rounds (g1j29_dbl_quad), the additions into ACC = [|x|]P in five (g1j29_add_quad) - and
// from k = 64 on, where the doublings that extend the multiples and the chain [|x|]ACC of the subgroup test no longer depend
// on each other, quad B runs the second chain IN THE SAME INSTRUCTIONS as quad A's doublings (the two quads are lanes of one
// wavefront: same code, different LDS slots).  ~500 product times on the critical path instead of ~1 450 (+ the square root).
// 64 threads = 8 points per workgroup; dynamic LDS per point: R | ACC | R2 (48 words each) | two quad scratches | x, y (32),
// then 18 uint4 of parking per point for the decode.
constexpr int DECQ_POINT_WORDS = 3 * 48 + 2 * SUMQ_SCRATCH_WORDS + 32 + 16;  // (+16: consecutive points 32 banks apart)
constexpr int DECQ_POINTS_PER_BLOCK = 8;  // ONE wavefront per workgroup, and one workgroup per CU (DECQ_LDS_BYTES is padded for
                                          // that): the quads talk through LDS, ~30 b128 accesses of every lane per product
                                          // round - four such wavefronts on a CU spend as long waiting for the LDS as multiplying
constexpr size_t DECQ_LDS_USED = DECQ_POINTS_PER_BLOCK * (DECQ_POINT_WORDS * 4 + PARK_UINT4_PER_THREAD * 16);
constexpr size_t DECQ_LDS_BYTES = 96 * 1024;
template <int CHUNKS>
__global__ __launch_bounds__(64, 1) void k_g1_decode_multiples29_quads(const uint8_t* __restrict__ bytes0, const uint8_t* __restrict__ bytes1, int n0,
                                                                     G1Aff* __restrict__ points, uint32_t* __restrict__ pflag,
                                                                     G1Jac29Mem* __restrict__ mult, int n, int stride) {
    constexpr int HALF = CHUNKS / 2, STEP = 256 / CHUNKS;
    static_assert(64 % STEP == 0, "a multiple is due at k = 64");
    extern __shared__ __attribute__((aligned(16))) uint32_t decq_lds[];
    const int lane = threadIdx.x & 63, r = threadIdx.x & 3, qb = (threadIdx.x >> 2) & 1, pt = threadIdx.x >> 3;
    static_assert(DECQ_LDS_USED <= DECQ_LDS_BYTES, "LDS");
    const int i = blockIdx.x * DECQ_POINTS_PER_BLOCK + pt;
    if (i >= n) return;  // (whole groups of eight lanes; nothing below synchronises across groups)
    uint32_t* const mine = decq_lds + pt * DECQ_POINT_WORDS;
    uint32_t *const R = mine, *const ACC = mine + 48, *const R2 = mine + 96, *const scr = mine + 144 + qb * SUMQ_SCRATCH_WORDS;
    uint32_t* const XY = mine + 144 + 2 * SUMQ_SCRATCH_WORDS;
    const bool first = qb == 0 && r == 0;
    uint32_t st = G1_INVALID;
    if (first) {
        const LdsPark pk = lds_park(reinterpret_cast<const uint4*>(decq_lds + DECQ_POINTS_PER_BLOCK * DECQ_POINT_WORDS) + pt, DECQ_POINTS_PER_BLOCK);
        Fp29 x, y;
        st = g1_decompress29(x, y, i < n0 ? bytes0 + (size_t)i * 48 : bytes1 + (size_t)(i - n0) * 48, pk);
        if (st == G1_OK) {
            sumq_store(XY, x);
            sumq_store(XY + 16, y);
            sumq_store(R, x);
            sumq_store(R + 16, y);
            sumq_store(R + 32, fp29_const(cp29::FP29_ONE));
        }
    }
    st = (uint32_t)__shfl((int)st, lane & ~7, 64);
    bool in = false;
    if (st == G1_OK) {
        auto emit = [&](int k) {  // (quad A) rows k and HALF + k: lane 0 the point, lane 1 its -phi image
            if (r < 2) {
                G1Jac29 m;
                m.x = sumq_load(R);
                m.y = sumq_load(R + 16);
                m.z = sumq_load(R + 32);
                if (r == 0) g1j29_store(mult[(size_t)k * stride + i], m);
                else g1j29_store(mult[(size_t)(HALF + k) * stride + i], g1j29_neg_phi(m));
            }
        };
        if (qb == 0) {
            // R = 2^k P with the multiples on the way; ACC = [|x|]P accumulated right to left (the highest set bit of |x| is 63)
            emit(0);
#pragma unroll 1
            for (int k = 0; k < 64; k++) {
                if (k && k % STEP == 0) emit(k / STEP);
                if ((BLS_X_ABS >> k) & 1) {
                    if (k == 16) {  // the lowest set bit
                        if (r < 3) sumq_store(ACC + 16 * r, sumq_load(R + 16 * r));
                    } else {
                        g1j29_add_quad<SumqLayout16>(ACC, R, ACC, scr, r, lane);
                    }
                }
                g1j29_dbl_quad<SumqLayout16>(R, R, scr, r);
            }
        } else if (r < 3) {
            // (quad B waits - its lanes are masked off while quad A runs - and then starts [|x|]ACC from ACC)
        }
        if (qb == 1 && r < 3) sumq_store(R2 + 16 * r, sumq_load(ACC + 16 * r));
        // k = 64 + j on quad A (emit, double until the last multiple is out), bit 62 - j of |x| on quad B (double, add ACC)
        bool a_done = false;
#pragma unroll 1
        for (int j = 0; j < 63; j++) {
            const int k = 64 + j;
            if (qb == 0 && !a_done) {
                if (k % STEP == 0) {
                    emit(k / STEP);
                    if (k + STEP >= 128) a_done = true;
                }
            }
            if (qb == 1 || !a_done) g1j29_dbl_quad<SumqLayout16>(qb ? R2 : R, qb ? R2 : R, scr, r);
            if (qb == 1 && ((BLS_X_ABS >> (62 - j)) & 1)) g1j29_add_quad<SumqLayout16>(R2, ACC, R2, scr, r, lane);
        }
        // phi(P) == -R2  <=>  X = beta x Z^2  and  Y + y Z^3 = 0 (as in g1j29_in_subgroup_with_multiples)
        if (qb == 1 && r == 0) {
            const Fp29 qx = sumq_load(R2), qy = sumq_load(R2 + 16), qz = sumq_load(R2 + 32);
            const Fp29 zz = fp29_sqr(qz);
            if (!fp29_is_zero_mod_p(zz)) {
                const Fp29 xx = sumq_load(XY), yy = sumq_load(XY + 16);
                const Fp29 zzz = fp29_mul(zz, qz), one = fp29_const(cp29::FP29_ONE);
                const Fp29 bx = fp29_mul(xx, fp29_const(cp29::FP29_BETA_MONT));
                const Fp29 dx = fp29_mul(fp29_sub<2>(qx, fp29_mul(bx, zz)), one);
                const Fp29 dy = fp29_mul(fp29_add(qy, fp29_mul(yy, zzz)), one);
                in = fp29_is_zero_mod_p(dx) && fp29_is_zero_mod_p(dy);
            }
        }
    }
    if (!(qb == 1 && r == 0)) return;  // the lane that ran the test writes the verdict
    G1Aff a;
    a.x = FpF::zero();
    a.y = FpF::zero();
    if (st == G1_OK) {
        if (in) {
            a.x = fp29_to_std(sumq_load(XY));
            a.y = fp29_to_std(sumq_load(XY + 16));
        } else {
            st = G1_INVALID;
        }
    }
    if (st != G1_OK) {  // (quad A's row stores of this point have been issued by the same wavefront before these)
        const G1Jac29 id = g1j29_identity();
        for (int k = 0; k < CHUNKS; k++) g1j29_store(mult[(size_t)k * stride + i], id);
    }
    points[i] = a;
    pflag[i] = st;
    if (i == 0) {
        g_decode_dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_decode_dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// grid (W windows per chunk, chunks / chunks_per_block, 2 outputs x batches), 256 threads: block (w, g, o) handles digit
// byte W j + w of every scalar of output o for its chunks j, against the table of chunk j (W = 32 / chunks).
__device__ __forceinline__ void lds_store_jac29(uint32_t* base, int slot, const G1Jac29& p) {
    uint32_t* d = base + slot * 42;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d[i] = p.x.l[i];
        d[14 + i] = p.y.l[i];
        d[28 + i] = p.z.l[i];
    }
}
__device__ __forceinline__ G1Jac29 lds_load_jac29(const uint32_t* base, int slot) {
    const uint32_t* d = base + slot * 42;
    G1Jac29 p;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        p.x.l[i] = d[i];
        p.y.l[i] = d[14 + i];
        p.z.l[i] = d[28 + i];
    }
    return p;
}

// The window kernel is written once over a curve policy: Curve29 (radix-2^29 field, lazy reduction: the default) or
// Curve32 (the 12x32 field of field.hpp; KZG_FP29=0, kept for A/B measurement and as a cross-check).
struct Curve32 {
    using Pt = G1Jac;
    using Mem = G1Jac;
    using Entry = G1Jac;
    static constexpr int WORDS = 36;
    static constexpr bool SPLIT = false;  // the 12x32 A/B variant keeps the complete formulas in its loops
    static constexpr bool QUADS = false;
    __device__ static __forceinline__ Pt add_entry(const Pt& a, const Entry& b) { return g1_add(a, b); }
    __device__ static __forceinline__ Pt identity() { return g1_identity(); }
    __device__ static __forceinline__ Pt add(const Pt& a, const Pt& b) { return g1_add(a, b); }
    __device__ static __forceinline__ Pt dbl(const Pt& a) { return g1_dbl(a); }
    __device__ static __forceinline__ Pt load(const Mem& m) { return m; }
    __device__ static __forceinline__ void lds_store(uint32_t* b, int s, const Pt& p) { lds_store_jac(b, s, p); }
    __device__ static __forceinline__ Pt lds_load(const uint32_t* b, int s) { return lds_load_jac(b, s); }
    __device__ static __forceinline__ G1Jac to_std(const Pt& p) { return p; }
};
struct Curve29 {
    using Pt = G1Jac29;
    using Mem = G1Jac29Mem;
    using Entry = G1Jac29;
    static constexpr int WORDS = 42;
    // the additions of the loops in two halves (g1_29_formulas.hpp): special(h) -> the pair needs the complete formula
    static constexpr bool SPLIT = true;
    static constexpr bool QUADS = false;  // reduction trees with four lanes per addition (Curve29Quads)
    using EntryHead = G1AddHead;
    // (accumulator and table entries are finite in the bucket loop; an accumulator that a P - P turned into the identity
    // has Z = 0, hence H = 0 - U1 ... not necessarily 0: the flags are part of "special")
    __device__ static __forceinline__ EntryHead entry_head(const Pt& a, const Entry& b) {
        Fp29 Z1Z1, Z2Z2;
        bool p_inf, q_inf;
        g1j29_inf_flags(a, b, Z1Z1, Z2Z2, p_inf, q_inf);
        EntryHead h = g1j29_add_head(a, b, Z1Z1, Z2Z2);
        if (p_inf | q_inf) h.HH = fp29_zero();  // -> special
        return h;
    }
    __device__ static __forceinline__ bool entry_special(const EntryHead& h) { return g1j29_add_same_x(h); }
    __device__ static __forceinline__ Pt entry_tail(const Pt&, const Entry&, const EntryHead& h) { return g1j29_add_tail(h); }
    __device__ static __forceinline__ Pt from_entry(const Entry& b) { return b; }
    __device__ static __forceinline__ Pt add_entry(const Pt& a, const Entry& b) { return g1j29_add(a, b); }
    __device__ static __forceinline__ Pt identity() { return g1j29_identity(); }
    __device__ static __forceinline__ Pt add(const Pt& a, const Pt& b) { return g1j29_add(a, b); }
    __device__ static __forceinline__ Pt dbl(const Pt& a) { return g1j29_dbl(a); }
    __device__ static __forceinline__ Pt load(const Mem& m) { return g1j29_load(m); }
    __device__ static __forceinline__ void lds_store(uint32_t* b, int s, const Pt& p) { lds_store_jac29(b, s, p); }
    __device__ static __forceinline__ Pt lds_load(const uint32_t* b, int s) { return lds_load_jac29(b, s); }
    __device__ static __forceinline__ G1Jac to_std(const Pt& p) { return g1j29_to_std(p); }
};

// Curve29 for the latency layout: the levels of the bucket reduction that have at most 64 additions run them with four lanes
// each (g1j29_add_quad; 192 words of dynamic LDS per quad + a second R | C vector for the scan levels: MSM_QUADS_LDS_BYTES)
struct Curve29Quads : Curve29 {
    static constexpr bool QUADS = true;
};
constexpr size_t MSM_QUADS_LDS_BYTES = 4 * (64 * SUMQ_SCRATCH_WORDS + 32 * Curve29::WORDS + 64 * Curve29::WORDS);  // quad scratch | R, C | staged table rows
constexpr uint32_t MSM_QUADS_BUCKET_CAP = 3;  // entries of a bucket that its own thread adds (the first is a copy)

// Curve29 with AFFINE table entries (k_mult_to_affine29): bucket accumulation by mixed additions
struct Curve29Aff : Curve29 {
    using Mem = G1Aff29Mem;
    using Entry = G1Aff29;
    using EntryHead = G1MaddHead;
    __device__ static __forceinline__ EntryHead entry_head(const Pt& a, const Entry& b) { return g1j29_madd_head(a, b); }
    __device__ static __forceinline__ bool entry_special(const EntryHead& h) { return g1j29_madd_special(h); }
    __device__ static __forceinline__ Pt entry_tail(const Pt& a, const Entry&, const EntryHead& h) { return g1j29_madd_tail(a, h); }
    __device__ static __forceinline__ Pt from_entry(const Entry& b) {
        Pt r;
        r.x = b.x;
        r.y = b.y;
        r.z = fp29_const(cp29::FP29_ONE);
        return r;
    }
    __device__ static __forceinline__ Pt add_entry(const Pt& a, const Entry& b) { return g1j29_add_affine(a, b); }
    __device__ static __forceinline__ Entry load(const Mem& m) { return g1a29_load(m); }
};

#ifndef KZG_MSM_OCC
#define KZG_MSM_OCC 3
#endif
// LDSSORT: the sorted term list of the block lives in LDS, in the region that holds the bucket points afterwards
// (256 x WORDS words; the host picks it when chunks_per_block x the block's terms fit: msm_lds_sort_fits).  The global
// list is the general form (one large batch in few slices): its 4-byte scattered stores each cost a whole line of HBM
// write traffic once the launch's lists outgrow the L2 (measured: 25 M stores -> 3.2 GB written per launch group).
template <class CV>
constexpr int msm_lds_sort_capacity() { return MSM_BUCKETS * CV::WORDS; }
// A general addition for k_msm_reduce.  SAFE = false: straight-line code for finite operands with different x, an identity
// operand passed through, and a same-x pair (P + P, P - P) only REPORTED - the complete formula inlined beside the fast path
// cost the kernel 787 spilled registers at two wavefronts per SIMD, out of line its two 168-byte arguments were written to
// scratch before every addition (1.5 GB per launch).  SAFE = true: the complete formula, for the rare redo.
template <bool SAFE>
__device__ __forceinline__ G1Jac29 g1j29_add_fast(const G1Jac29& x, const G1Jac29& y, bool& same_x_seen) {
    if constexpr (SAFE) {
        return g1j29_add(x, y);
    } else {
        Fp29 Z1Z1, Z2Z2;
        bool p_inf, q_inf;
        g1j29_inf_flags(x, y, Z1Z1, Z2Z2, p_inf, q_inf);
        if (p_inf | q_inf) return p_inf ? y : x;  // (empty buckets: common only in small batches)
        const G1AddHead h = g1j29_add_head(x, y, Z1Z1, Z2Z2);
        same_x_seen |= g1j29_add_same_x(h);  // P + P or P - P: this result is wrong; the SAFE pass redoes the window
        return g1j29_add_tail(h);
    }
}
// TWO PASSES (round 3; the radix-2^29 throughput variants - every CV with SPLIT and without QUADS): the window kernel ends
// with the bucket sums, written to the save area, and k_msm_reduce turns them into the window sum.  In one kernel the 16
// levels of the reduction ran on one or two of a block's four wavefronts while the block held its 43 KB of LDS - a third
// of a block's life at a quarter of its lanes (SIMD utilisation of the kernel 58 %: profiles/r2_pmc.json, 7.4 cycles per
// instruction against 4.3); as a kernel of its own the reduction needs no LDS and no barrier, half the additions, and the
// bucket kernel's blocks leave as soon as their longest bucket is done.
template <class CV>
constexpr bool msm_two_pass() { return CV::SPLIT && !CV::QUADS; }
template <class CV, bool LDSSORT = false>
__global__ __launch_bounds__(256, CV::QUADS ? 1 : KZG_MSM_OCC) void k_msm_window(MsmDesc d) {  // (the latency variant has a CU to itself)
    using Pt = typename CV::Pt;
    constexpr bool TWOPASS = msm_two_pass<CV>();
    kstamp_in(d.ktime);
    // blockIdx.z = (2*batch + output) * slices + slice
    int zz = (int)blockIdx.z + d.z0;  // logical z: (2 batch + output) * slices + slice
    int w = blockIdx.x;
    if ((d.flags & MSM_FLAG_XCD) && gridDim.x == 8 && gridDim.y == 1 && (gridDim.z & 7) == 0) {
        const unsigned L = blockIdx.x + 8u * blockIdx.z, i = L >> 3;
        w = (int)(i & 7);
        zz = (int)((i >> 3) * 8 + (L & 7)) + d.z0;
    }
    if (d.flags & MSM_FLAG_BFIRST) {  // (z0 = 0, S = 1, an even number of z-layers: the host checked)
        const int half = (int)(gridDim.z >> 1);
        zz = zz < half ? 2 * zz + 1 : 2 * (zz - half);
    }
    const int S = d.slices, bo = zz / S, slice = zz % S;
    const int o = bo & 1, tid = threadIdx.x;
    const int cpb = d.chunks_per_block, j0 = blockIdx.y * cpb;
    const int W = gridDim.x;                                              // windows (digit bytes) per chunk
    const int wi = ((bo * gridDim.y + blockIdx.y) * S + slice) * W + w;  // window slot
    const int t0 = (int)((long long)d.nterms[o] * slice / S), nt = (int)((long long)d.nterms[o] * (slice + 1) / S) - t0;
    const uint32_t* tp = d.term_point + (size_t)bo * d.max_terms + t0;
    const uint32_t* tsc = d.term_scalar + (size_t)bo * d.max_terms + t0;
    // scratch of this block: the (window, chunk group) region of its output, then the slice's share of it
    __shared__ uint32_t cnt[MSM_BUCKETS], off[MSM_BUCKETS + 1], cur[MSM_BUCKETS];
    __shared__ uint32_t long_buckets[MSM_BUCKETS], n_long;  // (CV::QUADS) buckets with more than MSM_QUADS_BUCKET_CAP entries
    extern __shared__ __attribute__((aligned(16))) uint32_t msm_dyn_lds[];  // (CV::QUADS) quad scratch | second R | C vector
    // 36 / 42 KiB: one Jacobian point per thread for the reduction levels; before that the block's sorted term list (LDSSORT).
    // The two-pass form keeps only the list here.
    // (two-pass form: the list lies in DYNAMIC shared memory sized to the launch's longest list - 33 KB for the 8 196 entries of
    // a 1 024-blob batch's output B instead of the 43 KB of a bucket-point array - so that a CU takes a fourth block when
    // the wavefronts of the resident ones have begun to leave)
    __shared__ uint32_t pts[TWOPASS ? 1 : MSM_BUCKETS * CV::WORDS];
    uint32_t* const lst = TWOPASS ? msm_dyn_lds : pts;
    uint32_t* const sorted_global = d.sorted + ((size_t)(bo * gridDim.y + blockIdx.y) * W + w) * cpb * d.max_terms + (size_t)cpb * t0;
    // (compile-time choice: an LDS pointer or a global one, never a flat one)
    cnt[tid] = 0;
    cur[tid] = 0;
    if (tid == 0) n_long = 0;
    __syncthreads();
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(d.scalars);

    // 1. counting sort by digit of the cpb * nt (chunk, term) pairs; entry = chunk << 27 | point index (32 chunks at most, 2^27 points)
    for (int c = 0; c < cpb; c++) {
        const int byte = W * (j0 + c) + w;
        for (int t = tid; t < nt; t += 256) {
            uint32_t dig = d.pflag[tp[t]] ? 0u : sb[(size_t)tsc[t] * 32 + byte];
            atomicAdd(&cnt[dig], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t s = 0;
        for (int b = 0; b < MSM_BUCKETS; b++) {
            off[b] = s;
            s += cnt[b];
        }
        off[MSM_BUCKETS] = s;
    }
    __syncthreads();
    for (int c = 0; c < cpb; c++) {
        const int byte = W * (j0 + c) + w;
        for (int t = tid; t < nt; t += 256) {
            uint32_t dig = d.pflag[tp[t]] ? 0u : sb[(size_t)tsc[t] * 32 + byte];
            uint32_t pos = atomicAdd(&cur[dig], 1u);
            if constexpr (LDSSORT) lst[off[dig] + pos] = tp[t] | (uint32_t)(j0 + c) << MSM_ENTRY_CHUNK_SHIFT;
            else sorted_global[off[dig] + pos] = tp[t] | (uint32_t)(j0 + c) << MSM_ENTRY_CHUNK_SHIFT;
        }
    }
    __threadfence_block();
    __syncthreads();
    // 2. one bucket per thread, the buckets handed out in order of decreasing size: a wave runs as long as its largest
    //    bucket, so wave 0 takes the 64 fullest buckets, wave 3 the 64 emptiest (Poisson-distributed sizes: ~25% fewer
    //    wave-level additions than bucket tid -> thread tid).  Digit 0 contributes nothing and ranks last.
    {
        const uint32_t mine = tid ? cnt[tid] : 0u;
        uint32_t rank = 0;
        for (int b = 0; b < MSM_BUCKETS; b++) {
            const uint32_t c = b ? cnt[b] : 0u;
            rank += (c > mine) | ((c == mine) & (b < tid));
        }
        cur[rank] = tid;  // cur[] is free after the sort: bucket handled by thread `rank`
    }
    __syncthreads();
    const int rot = (d.flags & MSM_FLAG_ROTATE) ? (int)((blockIdx.x + blockIdx.z) & 3) : 0;
    const int bucket = cur[(tid + 64 * rot) & 255];
    const typename CV::Mem* mult = static_cast<const typename CV::Mem*>(d.mult);
    auto sorted_at = [&](uint32_t k) -> uint32_t {
        if constexpr (LDSSORT) return lst[k];
        else return sorted_global[k];
    };
    auto sorted_put = [&](uint32_t k, uint32_t v) {
        if constexpr (LDSSORT) lst[k] = v;
        else sorted_global[k] = v;
    };
    Pt acc = CV::identity();
    if constexpr (TWOPASS) {
        // The common addition (accumulator finite, different x) is straight-line code; a pair that needs the complete
        // formula - a repeated point (P + P) or its negative - is not added in the hot loop: its entry moves to the front of
        // the range and the loop behind it takes care of it.  (Inlined, the complete formula's five exits kept the
        // accumulator in scratch memory: 112 bytes loaded and stored per addition.)
        auto accumulate = [&](uint32_t k, const uint32_t kend) -> Pt {  // the sum of list entries [k, kend); only this thread touches them
            Pt a = CV::identity();
            const uint32_t first = k + 1;
            uint32_t wr = k;
            if (k < kend) {  // the first entry is a copy, not an addition to the identity
                const uint32_t e = sorted_at(k++);
                a = CV::from_entry(CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]));
                wr = k;
            }
            for (; k < kend; k++) {
                const uint32_t e = sorted_at(k);
                const typename CV::Entry q = CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]);
                const typename CV::EntryHead h = CV::entry_head(a, q);
                if (CV::entry_special(h)) sorted_put(wr++, e);
                else a = CV::entry_tail(a, q, h);
            }
            for (uint32_t j = first; j < wr; j++) {  // rare
                const uint32_t e = sorted_at(j);
                a = CV::add_entry(a, CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]));
            }
            return a;
        };
        // the block's slot of the save area, POINT-major (256 points of MSM_SAVE2_WORDS words, X | Y | Z at 16-word strides):
        // k_msm_reduce takes the bucket sums from there - its row lanes read bucket 16 hi + t, its column lanes bucket 16 t + lo,
        // and with every point in lines of its own both patterns fetch exactly the bytes they use (word-major, the row
        // pattern pulled a 64-byte sector per word: 5.5 GB of traffic for 0.35 GB of points, profiles/r3b_pmc.json)
        const size_t slot = ((size_t)(zz - d.z0) * gridDim.y + blockIdx.y) * W + w;
        uint32_t* const out = d.save + slot * MSM_SAVE2_WORDS * 256;
        auto put = [&](int col, const Pt& p) {
            uint4* const o4 = reinterpret_cast<uint4*>(out + (size_t)col * MSM_SAVE2_WORDS);
            const Fp29* const c[3] = {&p.x, &p.y, &p.z};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                o4[4 * j] = make_uint4(c[j]->l[0], c[j]->l[1], c[j]->l[2], c[j]->l[3]);
                o4[4 * j + 1] = make_uint4(c[j]->l[4], c[j]->l[5], c[j]->l[6], c[j]->l[7]);
                o4[4 * j + 2] = make_uint4(c[j]->l[8], c[j]->l[9], c[j]->l[10], c[j]->l[11]);
                o4[4 * j + 3] = make_uint4(c[j]->l[12], c[j]->l[13], 0u, 0u);
            }
        };
        // (Tried and measured, profiles/r3_ab_msm_halves.txt: two threads per PAIR of buckets - rank t and rank 255 - t, each
        // thread one half of either list, the halves exchanged through the save slot and completed with one general addition -
        // evens out the threads' work (a block lives as long as its fullest bucket: 49 entries at an average of 32), but the
        // second accumulator cost the hot loop 100 spilled registers and 4 GB of scratch writes: +0.2 %.  Not kept.)
        put(bucket, accumulate(bucket > 0 ? off[bucket] : 0u, bucket > 0 ? off[bucket + 1] : 0u));
        kstamp_out(d.ktime);
        return;
    } else if constexpr (CV::SPLIT) {
        // (the latency variant: Curve29Quads) the same loop with the long buckets' tails left to the quads
        uint32_t k = bucket > 0 ? off[bucket] : 0u, w = k;
        const uint32_t kend_all = bucket > 0 ? off[bucket + 1] : 0u;
        // (CV::QUADS) a thread takes the first MSM_QUADS_BUCKET_CAP entries of its bucket; what a long bucket has beyond them
        // is added afterwards by a quad per bucket, while the short buckets' threads would only wait
        const uint32_t kend = (CV::QUADS && kend_all - k > MSM_QUADS_BUCKET_CAP) ? k + MSM_QUADS_BUCKET_CAP : kend_all;
        if (k < kend) {  // the first entry of a bucket is a copy, not an addition to the identity
            const uint32_t e = sorted_at(k++);
            acc = CV::from_entry(CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]));
            w = k;
        }
        for (; k < kend; k++) {
            const uint32_t e = sorted_at(k);
            const typename CV::Entry q = CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]);
            const typename CV::EntryHead h = CV::entry_head(acc, q);
            if (CV::entry_special(h)) sorted_put(w++, e);  // w <= k: only this thread reads or writes its bucket's list
            else acc = CV::entry_tail(acc, q, h);
        }
        const uint32_t first = bucket > 0 ? off[bucket] + 1 : 0u;
        for (uint32_t j = first; j < w; j++) {  // rare
            const uint32_t e = sorted_at(j);
            acc = CV::add_entry(acc, CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]));
        }
        if constexpr (CV::QUADS) {
            if (kend < kend_all) {
                if constexpr (LDSSORT)  // the list's LDS region is about to become the bucket points: the rest of it to the global copy
                    for (uint32_t j = kend; j < kend_all; j++) sorted_global[j] = lst[j];
                long_buckets[atomicAdd(&n_long, 1u)] = (uint32_t)bucket;
            }
        }
    } else if (bucket > 0) {
        for (uint32_t k = off[bucket]; k < off[bucket + 1]; k++) {
            const uint32_t e = sorted_at(k);
            acc = CV::add_entry(acc, CV::load(mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)]));
        }
    }
    // 3. sum_b b*B_b with b = 16 hi + lo:   16 * sum_hi hi*R_hi + sum_lo lo*C_lo,
    //    R_hi = row sums, C_lo = column sums of the 16 x 16 bucket matrix.  Two 4-level trees (ops packed into
    //    the low threads so idle waves skip the additions), then two 16-element weighted sums done by a
    //    suffix scan + tree on 32 threads, then 4 doublings.  ~19 wave-level point additions per block instead
    //    of the 64 of a 256-wide scan + tree.
    //    All 21 levels run through ONE loop with a single inlined point addition (level = kind + stride):
    //      kind 0 rows  : dst = 16 hi + 2 s k, src = dst + s          (ops = 128 / s)
    //      kind 1 cols  : dst = 16 (2 s k) + lo, src = dst + 16 s     (ops = 128 / s)   on a fresh copy of the buckets
    //      kind 2 scan  : suffix scan over the two 16-element vectors R (slots 0..15) and C (slots 16..31)
    //      kind 3 tree  : tree sum of the two scanned vectors (slot 0 / 16 of each zeroed first)
    if constexpr (LDSSORT) __syncthreads();  // every list has been read: the region becomes the bucket points
    CV::lds_store(pts, bucket, acc);
    __syncthreads();
    if constexpr (CV::QUADS) {
        const uint32_t nl = n_long;
        if (nl) {
            for (uint32_t it = tid >> 2; it < nl; it += 64) {
                const uint32_t b = long_buckets[it];
                uint32_t* const P = pts + b * CV::WORDS;
                for (uint32_t j = off[b] + MSM_QUADS_BUCKET_CAP; j < off[b + 1]; j++) {
                    const uint32_t e = sorted_global[j];
                    // the table row into the quad's staging slot (lanes 0..2: one coordinate each), then an LDS-to-LDS addition
                    uint32_t* const Qs = msm_dyn_lds + 64 * SUMQ_SCRATCH_WORDS + 32 * CV::WORDS + (tid >> 2) * CV::WORDS;
                    const typename CV::Mem& row = mult[(size_t)(e >> MSM_ENTRY_CHUNK_SHIFT) * d.stride + (e & MSM_ENTRY_POINT_MASK)];
                    if ((tid & 3) < 3) SumqLayout14::st(Qs, tid & 3, fp29_load((tid & 3) == 0 ? row.x : (tid & 3) == 1 ? row.y : row.z));
                    g1j29_add_quad<SumqLayout14>(P, Qs, P, msm_dyn_lds + (tid >> 2) * SUMQ_SCRATCH_WORDS, tid & 3, tid & 63);
                }
            }
            __syncthreads();
        }
    }
    // From here on thread tid looks after bucket tid.  The row trees run in place and destroy the bucket sums the column
    // trees need: a copy waits in global memory (d.save; word-major, so a wavefront writes and reads whole lines).
    // R and C get their own small LDS vectors for the scans.
    __shared__ uint32_t rc[TWOPASS ? 1 : 32 * CV::WORDS];  // R_0..R_15 | C_0..C_15
    uint32_t* const save = d.save + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * CV::WORDS) * 256 + tid;
#pragma unroll
    for (int i = 0; i < CV::WORDS; i++) save[(size_t)i * 256] = pts[tid * CV::WORDS + i];
    uint32_t* rc_cur = rc;
    uint32_t* rc_nxt = msm_dyn_lds + 64 * SUMQ_SCRATCH_WORDS;
#pragma unroll 1
    for (int lvl = 0; lvl < 16; lvl++) {
        const int kind = lvl >> 2, s = kind == 3 ? (8 >> (lvl & 3)) : (1 << (lvl & 3));
        if (lvl == 4) {  // rows done: R to its vector, the buckets back for the column trees
            if (tid < 16) CV::lds_store(rc, tid, CV::lds_load(pts, 16 * tid));
            __syncthreads();
#pragma unroll
            for (int i = 0; i < CV::WORDS; i++) pts[tid * CV::WORDS + i] = save[(size_t)i * 256];
            __syncthreads();
        } else if (lvl == 8) {  // columns done: C beside R
            if (tid >= 16 && tid < 32) CV::lds_store(rc, tid, CV::lds_load(pts, tid - 16));
            __syncthreads();
        } else if (lvl == 12) {  // scans done: S_0 is not part of sum_{k>=1} S_k
            if (tid < 32 && (tid & 15) == 0) CV::lds_store(rc, tid, CV::identity());
            __syncthreads();
        }
        if constexpr (CV::QUADS) {
            // 16 R_hi instead of 16 (sum hi R_hi): the four doublings run on sixteen lanes of the fourth wavefront, which the
            // column levels leave idle (two during the 128 single-lane additions of level 4, one each beside levels 6, 7)
            if ((lvl == 4 || lvl == 6 || lvl == 7) && tid >= 192 && tid < 208) {
                Pt R = CV::lds_load(rc, tid - 192);
                R = CV::dbl(R);
                if (lvl == 4) R = CV::dbl(R);
                CV::lds_store(rc, tid - 192, R);
            }
        }
        const int ops = kind < 2 ? 128 / s : 32;  // additions of this level (some of the 32 of a scan / tree level idle)
        // k = the operation this thread works on: its own, or - with at most 64 of them - its quad's
        const bool quads = CV::QUADS && ops <= 64;
        const int k = quads ? tid >> 2 : tid;
        bool active;
        int dst, src;
        if (kind == 0) {
            const int per_row = 8 / s;
            active = k < ops;
            dst = 16 * (k / per_row) + 2 * s * (k % per_row);
            src = dst + s;
        } else if (kind == 1) {
            active = k < ops;
            dst = 16 * (2 * s * (k >> 4)) + (k & 15);
            src = dst + 16 * s;
        } else if (kind == 2) {
            active = k < 32 && (k & 15) + s < 16;
            dst = k;
            src = k + s;
        } else {
            active = k < 32 && (k & 15) < s;
            dst = k;
            src = k + s;
        }
        if constexpr (CV::QUADS) {
            if (quads) {
                uint32_t* const qs = msm_dyn_lds + (tid >> 2) * SUMQ_SCRATCH_WORDS;
                const int r = tid & 3;
                if (kind == 2) {
                    // a scan level reads slots that their owners rewrite in the same level: from one vector into the other
                    if (active) g1j29_add_quad<SumqLayout14>(rc_cur + dst * CV::WORDS, rc_cur + src * CV::WORDS, rc_nxt + dst * CV::WORDS, qs, r, tid & 63);
                    else if (k < 32 && r == 0) CV::lds_store(rc_nxt, k, CV::lds_load(rc_cur, k));
                    __syncthreads();
                    uint32_t* const t = rc_cur;
                    rc_cur = rc_nxt;
                    rc_nxt = t;  // (four scan levels: back in rc at level 12)
                } else {
                    uint32_t* const arr = kind < 2 ? pts : rc;
                    if (active) g1j29_add_quad<SumqLayout14>(arr + dst * CV::WORDS, arr + src * CV::WORDS, arr + dst * CV::WORDS, qs, r, tid & 63);
                    __syncthreads();
                }
                continue;
            }
        }
        uint32_t* const arr = kind < 2 ? pts : rc;
        Pt x = CV::identity(), y = CV::identity();
        if (active) {
            x = CV::lds_load(arr, dst);
            y = CV::lds_load(arr, src);
        }
        __syncthreads();  // scan levels read a slot that its owner rewrites in the same level
        if constexpr (CV::SPLIT) {
            // every path ends in its own store: no point value is merged across the paths (see the bucket loop)
            if (active) {
                Fp29 Z1Z1, Z2Z2;
                bool p_inf, q_inf;
                g1j29_inf_flags(x, y, Z1Z1, Z2Z2, p_inf, q_inf);
                if (p_inf | q_inf) {
                    CV::lds_store(arr, dst, p_inf ? y : x);  // empty buckets: common in small batches
                } else {
                    const G1AddHead h = g1j29_add_head(x, y, Z1Z1, Z2Z2);  // x and y are dead from here
                    if (g1j29_add_same_x(h)) CV::lds_store(arr, dst, g1j29_add_same_x_result(h));
                    else CV::lds_store(arr, dst, g1j29_add_tail(h));
                }
            }
        } else {
            if (active) CV::lds_store(arr, dst, CV::add(x, y));
        }
        __syncthreads();
    }
    if constexpr (CV::QUADS) {
        // rc[0] = sum hi (16 R_hi), rc[16] = sum lo C_lo: one more addition, then a lane per coordinate for the way out
        if (tid < 4) g1j29_add_quad<SumqLayout14>(rc, rc + 16 * CV::WORDS, rc, msm_dyn_lds, tid, tid);
        if (tid < 3) {  // (same wavefront as the quad: its LDS instructions execute in order)
            const Fp c = fp29_to_std(SumqLayout14::ld(rc, tid));
            G1Jac& o = d.window_sums[wi];
            *(tid == 0 ? &o.x : tid == 1 ? &o.y : &o.z) = c;
        }
    } else {
        if (tid == 0) {
            Pt r = CV::lds_load(rc, 0);  // sum hi * R_hi
#pragma unroll 1
            for (int k = 0; k < 4; k++) r = CV::dbl(r);
            d.window_sums[wi] = CV::to_std(CV::add(r, CV::lds_load(rc, 16)));
        }
    }
    kstamp_out(d.ktime);
}

// ---- pass 2 of the two-pass form: sum_b b B_b of one window block from its 256 bucket sums in the save area.
// b = 16 hi + lo:  16 * sum_hi hi R_hi + sum_lo lo C_lo  with the row sums R_hi and the column sums C_lo of the 16 x 16 bucket
// matrix.  HALF A WAVEFRONT PER WINDOW BLOCK: lane hi < 16 sums row hi, lane 16 + lo sums column lo - 15 additions each, all
// 32 lanes busy - then the two weighted sums by a suffix scan and a tree over the 16 lanes of each vector (8 levels, the
// partner's point fetched with 42 shuffles - no LDS, no barrier), four doublings and one addition on the half's first lane.
// 28 addition times per two window blocks instead of 21 wave-level additions + 16 barriers per block.
// The bucket sums are any Jacobian points (empty buckets: the identity, common only in small batches); the rare endings
// (identity operand, same x) leave the straight-line path through the complete formula, out of line.
__device__ __forceinline__ G1Jac29 g1j29_shfl(const G1Jac29& p, int src_lane) {
    G1Jac29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = (uint32_t)__shfl((int)p.x.l[i], src_lane, 64);
        r.y.l[i] = (uint32_t)__shfl((int)p.y.l[i], src_lane, 64);
        r.z.l[i] = (uint32_t)__shfl((int)p.z.l[i], src_lane, 64);
    }
    return r;
}
// slot q of this launch piece = the window block ((q / W / gy) + z0, (q / W) % gy, q % W) of k_msm_window.
// SAFE = false is the pass that always runs; it leaves flags[q] = 1 where it met a same-x pair.  SAFE = true runs behind it
// over the same grid, returns at once unless one of its wavefront's two windows is flagged, and redoes those with the
// complete formula.
template <bool SAFE>
__global__ __launch_bounds__(256, SAFE ? 1 : 2) void k_msm_reduce(const uint32_t* __restrict__ save, uint32_t* __restrict__ flags,
                                                                    G1Jac* __restrict__ window_sums, int W, int gy, int S, int z0, int nslots) {
    const int tid = threadIdx.x, lane = tid & 63, l = tid & 31;
    int q = (int)blockIdx.x * 8 + (tid >> 5);
    const bool live = q < nslots;
    if (!live) q = nslots - 1;  // (redundant work: the shuffles below want whole wavefronts)
    if constexpr (SAFE) {
        if (!__any((int)flags[q])) return;
    }
    bool bad = false;
    const bool is_col = l >= 16;
    const int k = l & 15;
    const uint32_t* base = save + (size_t)q * MSM_SAVE2_WORDS * 256;
    G1Jac29 acc = g1j29_identity();
#pragma unroll 1
    for (int t = 0; t < 16; t++) {
        const int b = is_col ? 16 * t + k : 16 * k + t;
        const uint4* const i4 = reinterpret_cast<const uint4*>(base + (size_t)b * MSM_SAVE2_WORDS);
        G1Jac29 pnt;
        Fp29* const c[3] = {&pnt.x, &pnt.y, &pnt.z};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint4 a = i4[4 * j], bb = i4[4 * j + 1], cc = i4[4 * j + 2], dd = i4[4 * j + 3];
            c[j]->l[0] = a.x; c[j]->l[1] = a.y; c[j]->l[2] = a.z; c[j]->l[3] = a.w;
            c[j]->l[4] = bb.x; c[j]->l[5] = bb.y; c[j]->l[6] = bb.z; c[j]->l[7] = bb.w;
            c[j]->l[8] = cc.x; c[j]->l[9] = cc.y; c[j]->l[10] = cc.z; c[j]->l[11] = cc.w;
            c[j]->l[12] = dd.x; c[j]->l[13] = dd.y;
        }
        acc = t == 0 ? pnt : g1j29_add_fast<SAFE>(acc, pnt, bad);
    }
    // suffix scan over each 16-lane vector: V_k <- sum_{j >= k} V_j
#pragma unroll 1
    for (int s = 1; s < 16; s <<= 1) {
        const G1Jac29 other = g1j29_shfl(acc, (lane + s) & 63);
        if (k + s < 16) acc = g1j29_add_fast<SAFE>(acc, other, bad);
    }
    // sum_{k >= 1} S_k = sum_j j V_j: slot 0 leaves the sum, then a tree
    if (k == 0) acc = g1j29_identity();
#pragma unroll 1
    for (int s = 8; s >= 1; s >>= 1) {
        const G1Jac29 other = g1j29_shfl(acc, (lane + s) & 63);
        if (k < s) acc = g1j29_add_fast<SAFE>(acc, other, bad);
    }
    const G1Jac29 colsum = g1j29_shfl(acc, (lane + 16) & 63);  // lane 0 of the half: sum lo C_lo from lane 16
    if (l == 0) {
#pragma unroll 1
        for (int i = 0; i < 4; i++) acc = g1j29_dbl(acc);
        acc = g1j29_add_fast<SAFE>(acc, colsum, bad);
    }
    // a same-x pair anywhere in the window's 32 lanes flags the window
    const unsigned long long any_bad = __ballot((int)bad);
    const bool win_bad = ((any_bad >> (lane & 32)) & 0xffffffffull) != 0;
    if (l == 0 && live) {
        if constexpr (!SAFE) flags[q] = win_bad ? 1u : 0u;
        if (SAFE ? flags[q] != 0 : !win_bad) {
            const int w = q % W, by = (q / W) % gy, zz = q / (W * gy) + z0;
            const int bo = zz / S, slice = zz % S;
            window_sums[((size_t)(bo * gy + by) * S + slice) * W + w] = g1j29_to_std(acc);
        }
    }
}

// Host side: launch the window kernel over grid (gx, gy, gz) in z-pieces that fit the save area (save_bytes >= one z-layer).
// (Measured and not kept, round 5: the z-layers of ONE large sum in four pieces with the bucket reduction of piece k on a second
// stream beside the window kernel of piece k + 1 - the reduction is a 1 ms chain of ~28 point additions per slot whatever the
// number of slots - made a 2^20-term sum SLOWER, 10.5 ms against 8.5: four launches of long workgroups have four tails, and the
// reduction's 198-VGPR wavefronts take SIMD slots from the window kernel.  profiles/r5_config4_experiments.txt.)
template <class CV, bool LDSSORT>
inline void msm_window_launch(MsmDesc d, unsigned gx, unsigned gy, unsigned gz, uint32_t* save, size_t save_bytes, hipStream_t st, bool with_reduce = true) {
    const size_t layer = (size_t)gx * gy * (256 * (msm_two_pass<CV>() ? MSM_SAVE2_WORDS : CV::WORDS) + 1) * 4;  // (+ 1: the window's flag)
    unsigned per = (unsigned)std::min<size_t>(gz, std::max<size_t>(1, save_bytes / layer));
    d.save = save;
    if (CV::QUADS)  // static + dynamic LDS pass 64 KB (raised once per device: dyn_lds.hpp)
        (void)DYN_LDS((k_msm_window<CV, LDSSORT>), MSM_QUADS_LDS_BYTES);
    if (msm_two_pass<CV>() && per == gz && d.slices == 1 && (gz & 1) == 0 && d.nterms[1] > d.nterms[0]) d.flags |= MSM_FLAG_BFIRST;
    for (unsigned z = 0; z < gz; z += per) {
        d.z0 = (int)z;
        const unsigned nz = std::min(per, gz - z);
        // dynamic LDS: the quads' scratch (latency variant), or the two-pass form's sorted list (chunks_per_block x the longest slice, + 1: slice boundaries round either way)
        const size_t list_bytes = (msm_two_pass<CV>() && LDSSORT) ? 4 * ((size_t)d.chunks_per_block * ((size_t)(d.max_terms + d.slices - 1) / d.slices + 1) + 4) : 0;
        hipLaunchKernelGGL((k_msm_window<CV, LDSSORT>), dim3(gx, gy, nz), dim3(256), CV::QUADS ? MSM_QUADS_LDS_BYTES : list_bytes, st, d);
        if constexpr (msm_two_pass<CV>()) {
            if (!with_reduce) continue;  // (the caller takes the bucket sums of the whole grid from the save area: msm_large_tail)
            const int nslots = (int)(gx * gy * nz);
            uint32_t* const flags = save + (size_t)per * gx * gy * 256 * MSM_SAVE2_WORDS;  // behind the points of a full piece
            hipLaunchKernelGGL(k_msm_reduce<false>, dim3((unsigned)((nslots + 7) / 8)), dim3(256), 0, st, (const uint32_t*)save, flags, d.window_sums, (int)gx,
                               (int)gy, d.slices, (int)z, nslots);
            hipLaunchKernelGGL(k_msm_reduce<true>, dim3((unsigned)((nslots + 7) / 8)), dim3(256), 0, st, (const uint32_t*)save, flags, d.window_sums, (int)gx,
                               (int)gy, d.slices, (int)z, nslots);
        }
    }
}
constexpr size_t msm_save_layer_bytes(unsigned gx, unsigned gy, int words) { return (size_t)gx * gy * (256 * words + 1) * 4; }

// sums[(g * slices + k) * W + w] over the slices k -> out[g * W + w]; one 64-thread workgroup per (g, w) of every output
__global__ __launch_bounds__(64) void k_msm_fold_slices(const G1Jac* __restrict__ sums, G1Jac* __restrict__ out, int slices, int W) {
    const int gw = blockIdx.x, g = gw / W, w = gw % W, tid = threadIdx.x;  // g runs over (2B x slots)
    __shared__ uint32_t pts[64 * 36];
    if (tid < slices) lds_store_jac(pts, tid, sums[((size_t)g * slices + tid) * W + w]);
    int width = 1;  // (any count up to 64: the tree runs over the next power of two, the missing leaves stay out of it)
    while (width < slices) width <<= 1;
    __syncthreads();
    for (int half = width >> 1; half >= 1; half >>= 1) {
        const bool active = tid < half && tid + half < slices;
        G1Jac x = g1_identity(), y = g1_identity();
        if (active) {
            x = lds_load_jac(pts, tid);
            y = lds_load_jac(pts, tid + half);
        }
        __syncthreads();
        if (active) lds_store_jac(pts, tid, g1_add(x, y));
        __syncthreads();
    }
    if (tid == 0) out[(size_t)g * W + w] = lds_load_jac(pts, 0);
}

// out[o] = sum_w 2^(8w) (sum_g S[o][g][w]), w < W: the nslots chunk groups of every window are folded by a tree over
// nslots * W (<= 32) threads, then one Horner chain of 8 (W - 1) doublings (56 for the default layout, 8 for the
// latency layout).  nslots is a power of two.
__global__ __launch_bounds__(64) void k_msm_combine(const G1Jac* __restrict__ window_sums, G1Jac* __restrict__ out, int nslots, int W) {
    const int o = blockIdx.x, tid = threadIdx.x;
    __shared__ uint32_t pts[32 * 36];
    const int total = nslots * W;  // slot (g, w) at index g * W + w
    if (tid < total) lds_store_jac(pts, tid, window_sums[o * total + tid]);
    __syncthreads();
    for (int half = nslots >> 1; half >= 1; half >>= 1) {  // fold group g + half into g
        const bool active = tid < half * W;
        G1Jac x = g1_identity(), y = g1_identity();
        if (active) {
            x = lds_load_jac(pts, tid);
            y = lds_load_jac(pts, tid + half * W);
        }
        __syncthreads();
        if (active) lds_store_jac(pts, tid, g1_add(x, y));
        __syncthreads();
    }
    if (tid) return;
    G1Jac acc = lds_load_jac(pts, W - 1);
    for (int w = W - 2; w >= 0; w--) {
        for (int k = 0; k < MSM_C; k++) acc = g1_dbl(acc);
        acc = g1_add(acc, lds_load_jac(pts, w));
    }
    out[o] = acc;
}

// out[o] = the sum of the npts points sums[o * npts ..]; npts a power of two, 2 <= npts <= 128; one 256-thread workgroup
// per output; dynamic LDS: SUMQ_MAX_POINTS points + 64 quad scratches = 72 KB
constexpr size_t SUMQ_LDS_BYTES = 4 * (SUMQ_MAX_POINTS * SUMQ_POINT_WORDS + 64 * SUMQ_SCRATCH_WORDS);
__global__ __launch_bounds__(256) void k_msm_sum_quads(const G1Jac* __restrict__ sums, G1Jac* __restrict__ out, int npts) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sumq_lds[];
    uint32_t* pts = sumq_lds;
    uint32_t* scr = sumq_lds + SUMQ_MAX_POINTS * SUMQ_POINT_WORDS;
    const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (tid < npts) {
        const G1Jac p = sums[(size_t)o * npts + tid];
        sumq_store(pts + tid * SUMQ_POINT_WORDS, fp29_from_std(p.x));
        sumq_store(pts + tid * SUMQ_POINT_WORDS + 16, fp29_from_std(p.y));
        sumq_store(pts + tid * SUMQ_POINT_WORDS + 32, fp29_from_std(p.z));
    }
    __syncthreads();
    const int q = tid >> 2, r = tid & 3;
    for (int half = npts >> 1; half >= 1; half >>= 1) {
        if (q < half)
            g1j29_add_quad<SumqLayout16>(pts + q * SUMQ_POINT_WORDS, pts + (q + half) * SUMQ_POINT_WORDS, pts + q * SUMQ_POINT_WORDS, scr + q * SUMQ_SCRATCH_WORDS, r, lane);
        __syncthreads();
    }
    if (tid < 3) {  // one coordinate each
        const Fp c = fp29_to_std(sumq_load(pts + 16 * tid));
        Fp* dst = tid == 0 ? &out[o].x : tid == 1 ? &out[o].y : &out[o].z;
        *dst = c;
    }
}

// out[o] = sum_w 2^(8 w) S[o][w] for ONE chunk group (nslots = 1) with FOUR LANES per point operation: the Horner chain of a
// single output is 56 doublings and 7 additions in a row - 0.8 ms of a 2^20-term MSM on one lane of k_msm_combine - and a quad
// does a doubling in three product rounds and an addition in five (g1j29_dbl_quad / g1j29_add_quad).  One wavefront per output.
__global__ __launch_bounds__(64) void k_msm_combine_quad(const G1Jac* __restrict__ window_sums, G1Jac* __restrict__ out, int W) {
    __shared__ __attribute__((aligned(16))) uint32_t pts[(32 + 1) * 48 + SUMQ_SCRATCH_WORDS];
    const int o = blockIdx.x, tid = threadIdx.x, r = tid & 3;
    uint32_t* const acc = pts + 32 * 48;
    uint32_t* const scr = pts + 33 * 48;
    if (tid < 3 * W) {  // thread 3 w + c: coordinate c of S_w
        const int w = tid / 3, c = tid % 3;
        const G1Jac& p = window_sums[(size_t)o * W + w];
        sumq_store(pts + w * 48 + 16 * c, fp29_from_std(c == 0 ? p.x : c == 1 ? p.y : p.z));
    }
    __syncthreads();
    if (tid >= 4) return;  // (one quad: its LDS instructions execute in order)
    if (r < 3) sumq_store(acc + 16 * r, sumq_load(pts + (W - 1) * 48 + 16 * r));
#pragma unroll 1
    for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
        for (int k = 0; k < MSM_C; k++) g1j29_dbl_quad<SumqLayout16>(acc, acc, scr, r);
        g1j29_add_quad<SumqLayout16>(acc, pts + w * 48, acc, scr, r, tid);
    }
    if (r < 3) {
        const Fp c = fp29_to_std(sumq_load(acc + 16 * r));
        Fp* dst = r == 0 ? &out[o].x : r == 1 ? &out[o].y : &out[o].z;
        *dst = c;
    }
}

// ---- the tail of ONE LARGE sum (kzg_g1_msm from ~50 000 terms on: Z = 2 S layers of (window, slice) workgroups, Z in the
// hundreds at 2^20 terms).  k_msm_reduce is a chain of ~28 point additions per slot whatever the number of slots (1.0 ms,
// one lane per addition at ~35 us each), the fold of the Z window sums six more, the Horner chain 63: 1.55 ms behind a 7 ms
// window kernel.  Here the SLICES are folded first, bucket by bucket - B[w][b] = sum_z B[z][w][b]: Z - 1 additions per
// bucket, all 256 W buckets at once: work, not latency - and only W slots are reduced, with four lanes per addition:
//   k_msm_bucket_fold      thread (w, b, g) adds the layers [g per, (g + 1) per) of its bucket one after the other
//                          (SAFE = false: a same-x pair only flags the workgroup, SAFE = true redoes flagged workgroups with
//                          the complete formula - the two passes of k_msm_reduce)
//   k_msm_bucket_sum_quads one wavefront per (w, b): the tree over its gp partial sums, sixteen quads
//   k_msm_reduce_quads     one workgroup per window: sum_b b B_b as 16 sum_hi hi R_hi + sum_lo lo C_lo (the scheme of
//                          k_msm_window's latency form), 64 quads
// then k_msm_combine_quad.  Points travel as the save area's 48-word records (X | Y | Z, radix 2^29, 16 words each).
__device__ __forceinline__ G1Jac29 save2_load(const uint32_t* p) {
    const uint4* const i4 = reinterpret_cast<const uint4*>(p);
    G1Jac29 pnt;
    Fp29* const c[3] = {&pnt.x, &pnt.y, &pnt.z};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint4 a = i4[4 * j], bb = i4[4 * j + 1], cc = i4[4 * j + 2], dd = i4[4 * j + 3];
        c[j]->l[0] = a.x; c[j]->l[1] = a.y; c[j]->l[2] = a.z; c[j]->l[3] = a.w;
        c[j]->l[4] = bb.x; c[j]->l[5] = bb.y; c[j]->l[6] = bb.z; c[j]->l[7] = bb.w;
        c[j]->l[8] = cc.x; c[j]->l[9] = cc.y; c[j]->l[10] = cc.z; c[j]->l[11] = cc.w;
        c[j]->l[12] = dd.x; c[j]->l[13] = dd.y;
    }
    return pnt;
}
__device__ __forceinline__ void save2_store(uint32_t* p, const G1Jac29& v) {
    sumq_store(p, v.x);
    sumq_store(p + 16, v.y);
    sumq_store(p + 32, v.z);
}
constexpr int MSM_FOLD_MAX_GROUPS = 128;
// save: [Z][W][256] bucket sums (the window kernel's slots, gy = 1, one piece); part: [W][256][gp] partial sums, the groups
// beyond `groups` identities; flags: one word per workgroup, zeroed by the caller.  Grid W gp x 256 threads.
template <bool SAFE>
__global__ __launch_bounds__(256) void k_msm_bucket_fold(const uint32_t* __restrict__ save, uint32_t* __restrict__ part, uint32_t* __restrict__ flags, int W,
                                                         int Z, int per, int groups, int gp) {
    const int w = blockIdx.x % W, g = blockIdx.x / W, b = threadIdx.x;
    if constexpr (SAFE) {
        if (!flags[blockIdx.x]) return;
    }
    uint32_t* const dst = part + (((size_t)w * 256 + b) * gp + g) * MSM_SAVE2_WORDS;
    if (g >= groups) {
        save2_store(dst, g1j29_identity());
        return;
    }
    const int z0 = g * per, z1 = min(Z, z0 + per);
    bool bad = false;
    G1Jac29 acc = save2_load(save + (((size_t)z0 * W + w) * 256 + b) * MSM_SAVE2_WORDS);
#pragma unroll 1
    for (int z = z0 + 1; z < z1; z++) acc = g1j29_add_fast<SAFE>(acc, save2_load(save + (((size_t)z * W + w) * 256 + b) * MSM_SAVE2_WORDS), bad);
    save2_store(dst, acc);
    if constexpr (!SAFE) {
        if (__any((int)bad) && (threadIdx.x & 63) == 0) flags[blockIdx.x] = 1u;  // (every writer stores the same word)
    }
}
// out[p] = the sum of part[p][0 .. gp), gp a power of two <= MSM_FOLD_MAX_GROUPS; dynamic LDS: gp points + 16 quad scratches
constexpr size_t msm_bucket_sum_lds_bytes(int gp) { return 4 * ((size_t)gp * SUMQ_POINT_WORDS + 16 * SUMQ_SCRATCH_WORDS); }
__global__ __launch_bounds__(64) void k_msm_bucket_sum_quads(const uint32_t* __restrict__ part, uint32_t* __restrict__ out, int gp) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bsum_lds[];
    uint32_t* const pts = bsum_lds;
    const int tid = threadIdx.x, q = tid >> 2, r = tid & 3;
    uint32_t* const scr = bsum_lds + gp * SUMQ_POINT_WORDS + q * SUMQ_SCRATCH_WORDS;
    const uint4* const src = reinterpret_cast<const uint4*>(part + (size_t)blockIdx.x * gp * SUMQ_POINT_WORDS);
    for (int i = tid; i < gp * (SUMQ_POINT_WORDS / 4); i += 64) reinterpret_cast<uint4*>(pts)[i] = src[i];
    __syncthreads();
#pragma unroll 1
    for (int half = gp >> 1; half >= 1; half >>= 1) {
#pragma unroll 1
        for (int k = q; k < half; k += 16)
            g1j29_add_quad<SumqLayout16>(pts + k * SUMQ_POINT_WORDS, pts + (k + half) * SUMQ_POINT_WORDS, pts + k * SUMQ_POINT_WORDS, scr, r, tid);
        __syncthreads();
    }
    if (tid < SUMQ_POINT_WORDS / 4) reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * SUMQ_POINT_WORDS)[tid] = reinterpret_cast<const uint4*>(pts)[tid];
}
// window_sums[w] = sum_b b B[w][b] from buckets[w][256] (48-word records); one 256-thread workgroup (64 quads) per window.
// Row trees in place, the bucket sums fetched again for the column trees, then the two weighted 16-element sums by a suffix
// scan (from one vector into the other: a level reads slots that their owners rewrite) and a tree, four doublings, one
// addition: ~22 quad additions in a row.
constexpr size_t MSM_REDQ_LDS_BYTES = 4 * ((size_t)(256 + 64) * SUMQ_POINT_WORDS + 64 * SUMQ_SCRATCH_WORDS);
__global__ __launch_bounds__(256) void k_msm_reduce_quads(const uint32_t* __restrict__ buckets, G1Jac* __restrict__ window_sums) {
    extern __shared__ __attribute__((aligned(16))) uint32_t redq_lds[];
    constexpr int PW = SUMQ_POINT_WORDS, P4 = SUMQ_POINT_WORDS / 4;
    const int tid = threadIdx.x, q = tid >> 2, r = tid & 3, lane = tid & 63;
    uint32_t* const pts = redq_lds;
    uint32_t* rc_cur = redq_lds + 256 * PW;
    uint32_t* rc_nxt = rc_cur + 32 * PW;
    uint32_t* const scr = redq_lds + (256 + 64) * PW + q * SUMQ_SCRATCH_WORDS;
    const uint4* const src = reinterpret_cast<const uint4*>(buckets + (size_t)blockIdx.x * 256 * PW);
    for (int i = tid; i < 256 * P4; i += 256) reinterpret_cast<uint4*>(pts)[i] = src[i];
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s < 16; s <<= 1) {  // rows: R_hi ends in slot 16 hi
        const int ops = 128 / s, per_row = 8 / s;
#pragma unroll 1
        for (int k = q; k < ops; k += 64) {
            const int dst = 16 * (k / per_row) + 2 * s * (k % per_row);
            g1j29_add_quad<SumqLayout16>(pts + dst * PW, pts + (dst + s) * PW, pts + dst * PW, scr, r, lane);
        }
        __syncthreads();
    }
    for (int i = tid; i < 16 * P4; i += 256) reinterpret_cast<uint4*>(rc_cur)[i] = reinterpret_cast<const uint4*>(pts)[(16 * (i / P4)) * P4 + i % P4];
    __syncthreads();
    for (int i = tid; i < 256 * P4; i += 256) reinterpret_cast<uint4*>(pts)[i] = src[i];
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s < 16; s <<= 1) {  // columns: C_lo ends in slot lo
        const int ops = 128 / s;
#pragma unroll 1
        for (int k = q; k < ops; k += 64) {
            const int dst = 16 * (2 * s * (k >> 4)) + (k & 15);
            g1j29_add_quad<SumqLayout16>(pts + dst * PW, pts + (dst + 16 * s) * PW, pts + dst * PW, scr, r, lane);
        }
        __syncthreads();
    }
    for (int i = tid; i < 16 * P4; i += 256) reinterpret_cast<uint4*>(rc_cur)[16 * P4 + i] = reinterpret_cast<const uint4*>(pts)[i];
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s < 16; s <<= 1) {  // suffix scans of R (slots 0..15) and C (16..31): V_k <- sum_{j >= k} V_j
        if (q < 32) {
            if ((q & 15) + s < 16) g1j29_add_quad<SumqLayout16>(rc_cur + q * PW, rc_cur + (q + s) * PW, rc_nxt + q * PW, scr, r, lane);
            else if (r < 3) sumq_store(rc_nxt + q * PW + 16 * r, sumq_load(rc_cur + q * PW + 16 * r));
        }
        __syncthreads();
        uint32_t* const t = rc_cur;
        rc_cur = rc_nxt;
        rc_nxt = t;
    }
    if (tid < 2) save2_store(rc_cur + 16 * tid * PW, g1j29_identity());  // sum_{k >= 1} S_k = sum_j j V_j: S_0 stays out
    __syncthreads();
#pragma unroll 1
    for (int s = 8; s >= 1; s >>= 1) {
        if (q < 32 && (q & 15) < s) g1j29_add_quad<SumqLayout16>(rc_cur + q * PW, rc_cur + (q + s) * PW, rc_cur + q * PW, scr, r, lane);
        __syncthreads();
    }
    if (tid >= 4) return;  // (one quad: its LDS instructions execute in order)
#pragma unroll 1
    for (int i = 0; i < 4; i++) g1j29_dbl_quad<SumqLayout16>(rc_cur, rc_cur, scr, r);
    g1j29_add_quad<SumqLayout16>(rc_cur, rc_cur + 16 * PW, rc_cur, scr, r, lane);
    if (r < 3) {
        const Fp c = fp29_to_std(sumq_load(rc_cur + 16 * r));
        G1Jac& o = window_sums[blockIdx.x];
        *(r == 0 ? &o.x : r == 1 ? &o.y : &o.z) = c;
    }
}
// Host side of the tail: `save` holds the bucket sums of all Z layers (msm_window_launch(..., with_reduce = false) in ONE
// piece); tmp: msm_large_tail_bytes(W, gp) of device memory; leaves window_sums[0 .. W).
constexpr size_t msm_large_tail_bytes(unsigned W, int gp) { return 4 * ((size_t)W * 256 * ((size_t)gp + 1) * MSM_SAVE2_WORDS + (size_t)W * MSM_FOLD_MAX_GROUPS); }
inline int msm_large_tail_groups(unsigned Z, int per, int* gp_out) {
    const int groups = (int)((Z + per - 1) / per);
    int gp = 2;
    while (gp < groups) gp <<= 1;
    *gp_out = gp;
    return groups;
}
inline hipError_t msm_large_tail(const uint32_t* save, unsigned W, unsigned Z, int per, uint32_t* tmp, G1Jac* window_sums, hipStream_t st) {
    int gp;
    const int groups = msm_large_tail_groups(Z, per, &gp);
    uint32_t* const flags = tmp;                                   // [W gp]
    uint32_t* const part = tmp + (size_t)W * MSM_FOLD_MAX_GROUPS;  // [W][256][gp] records
    uint32_t* const buckets = part + (size_t)W * 256 * gp * MSM_SAVE2_WORDS;  // [W][256] records
    hipError_t e = hipMemsetAsync(flags, 0, 4 * (size_t)W * gp, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_bucket_fold<false>, dim3(W * gp), dim3(256), 0, st, save, part, flags, (int)W, (int)Z, per, groups, gp);
    hipLaunchKernelGGL(k_msm_bucket_fold<true>, dim3(W * gp), dim3(256), 0, st, save, part, flags, (int)W, (int)Z, per, groups, gp);
    if ((e = DYN_LDS(k_msm_bucket_sum_quads, msm_bucket_sum_lds_bytes(MSM_FOLD_MAX_GROUPS))) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_bucket_sum_quads, dim3(W * 256), dim3(64), msm_bucket_sum_lds_bytes(gp), st, (const uint32_t*)part, buckets, gp);
    if ((e = DYN_LDS(k_msm_reduce_quads, MSM_REDQ_LDS_BYTES)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_reduce_quads, dim3(W), dim3(256), MSM_REDQ_LDS_BYTES, st, (const uint32_t*)buckets, window_sums);
    return hipGetLastError();
}

// The same sum with ONE LANE per output, for launches with many outputs (a launch group of batches): the Horner chain
// is serial either way, and a workgroup per output spends a whole wavefront's issue slots on its single busy lane.
__global__ __launch_bounds__(64) void k_msm_combine_lanes(const G1Jac* __restrict__ window_sums, G1Jac* __restrict__ out, int nslots, int W,
                                                          int nout) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    const G1Jac* S = window_sums + (size_t)o * nslots * W;  // slot (g, w) at index g * W + w
    G1Jac acc = g1_identity();
#pragma unroll 1
    for (int w = W - 1; w >= 0; w--) {
        if (w != W - 1)
            for (int k = 0; k < MSM_C; k++) acc = g1_dbl(acc);
#pragma unroll 1
        for (int g = 0; g < nslots; g++) acc = g1_add(acc, S[g * W + w]);
    }
    out[o] = acc;
}

}  // namespace kzg
