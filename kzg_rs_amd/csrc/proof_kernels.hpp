// proof_kernels.hpp - the two small kernels of the ONE-PROOF path (KzgProof::verify_kzg_proof, src/kzg_proof.rs:353-397).
//
// The reference checks  e(C - [y]G, G2) == e(pi, [tau]G2 - [z]G2).  Its scalars z and y are known before any point is
// decoded, so the two scalar multiplications and the Miller-loop lines of the per-call G2 point run BESIDE the point decode
// (latency program SCALARS, kzg_rs_amd/slp/gen_pairing.py) and the pairing (program VERIFY3) starts as soon as the two square
// roots are done - the subgroup test of C and pi runs beside the pairing and only its verdict is awaited.  Critical path:
// max(SCALARS, two square roots in one wavefront) + VERIFY3, against decode -> MSM -> pairing in the general path.
//   k_proof_select      the digits of z and y pick the programs' table entries (fixed-base tables of the two generators,
//                       tools/gen_fixed_base.py: [d 2^(8w)]G2 / [d 2^(8w)]G1 homogeneous, (0 : 1 : 0) for d = 0) - a copy
//   k_proof_decompress  C and pi on two lanes of ONE wavefront (the same instructions take both square roots), written as
//                       VERIFY3's inputs: (x, y, 1), or (0, 1, 0) for the point at infinity; no subgroup test here
#pragma once
#include "g1_29.hpp"

namespace kzg {

constexpr int FB_WINDOWS = 32, FB_DIGITS = 256;
constexpr size_t FB_G2_FP = 6, FB_G1_FP = 3;  // coordinates (Fp elements) per table entry
constexpr size_t FB_TABLE_BYTES = (size_t)FB_WINDOWS * FB_DIGITS * (FB_G2_FP + FB_G1_FP) * sizeof(Fp);
constexpr int SCALARS_INPUTS = FB_WINDOWS * (int)(FB_G2_FP + FB_G1_FP) + 4;  // + [tau]G2 affine
constexpr int VERIFY3_INPUTS = 9 + 68 * 6 + 2;                                 // pi, C, [y]G, the lines of Q, Z of Q
constexpr int VERIFY3_OUTPUTS = 8;                                             // the six pairing coefficients, Z.c0 and Z.c1 of Q

// One workgroup per proof.  z, y: 8 little-endian 32-bit limbs each (canonical; pinned host memory or device memory), instance i
// at z + i * stride_words / y + i * stride_words; out: SCALARS' input record of instance i at out + i * SCALARS_INPUTS.
__global__ __launch_bounds__(128) void k_proof_select(const uint32_t* __restrict__ z, const uint32_t* __restrict__ y, uint32_t stride_words,
                                                      const Fp* __restrict__ table, const Fp* __restrict__ tau4, Fp* __restrict__ out) {
    const int t = threadIdx.x;
    z += (size_t)blockIdx.x * stride_words;
    y += (size_t)blockIdx.x * stride_words;
    out += (size_t)blockIdx.x * SCALARS_INPUTS;
    if (t < 2 * FB_WINDOWS) {
        const bool g1 = t >= FB_WINDOWS;
        const int w = g1 ? t - FB_WINDOWS : t;
        const uint32_t digit = ((g1 ? y : z)[w >> 2] >> (8 * (w & 3))) & 255u;
        const size_t nfp = g1 ? FB_G1_FP : FB_G2_FP;
        const Fp* src = table + (g1 ? (size_t)FB_WINDOWS * FB_DIGITS * FB_G2_FP : 0) + ((size_t)w * FB_DIGITS + digit) * nfp;
        Fp* dst = out + (g1 ? (size_t)FB_WINDOWS * FB_G2_FP + (size_t)w * FB_G1_FP : (size_t)w * FB_G2_FP);
        for (size_t k = 0; k < nfp; k++) dst[k] = src[k];
    } else if (t < 2 * FB_WINDOWS + 4) {
        out[FB_WINDOWS * (FB_G2_FP + FB_G1_FP) + (t - 2 * FB_WINDOWS)] = tau4[t - 2 * FB_WINDOWS];
    }
}

// Two lanes per proof: lane 2 i decompresses commitments[i], lane 2 i + 1 proofs[i] (48 compressed bytes each) - all the
// square roots of a wavefront in the same instructions.  v3in: VERIFY3's input record of proof i at v3in + i * VERIFY3_INPUTS,
// [0..2] = pi, [3..5] = C.  flags[2 i] = status of C, flags[2 i + 1] = status of pi (G1_OK / G1_INFINITY / G1_INVALID; the
// subgroup is NOT tested here).
__global__ __launch_bounds__(64) void k_proof_decompress(const uint8_t* __restrict__ commitments, const uint8_t* __restrict__ proofs, int n,
                                                         Fp* __restrict__ v3in, uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint4 proof_park4[];  // PARK_UINT4_PER_THREAD per thread
    const int t = blockIdx.x * 64 + threadIdx.x, inst = t >> 1, which = t & 1;
    if (inst >= n) return;
    const LdsPark pk = lds_park(proof_park4 + threadIdx.x, blockDim.x);
    Fp29 x, y;
    const uint32_t st = g1_decompress29(x, y, (which ? proofs : commitments) + (size_t)48 * inst, pk);
    Fp* const o = v3in + (size_t)inst * VERIFY3_INPUTS + (which ? 0 : 3);
    if (st == G1_OK) {
        o[0] = fp29_to_std(x);
        o[1] = fp29_to_std(y);
        o[2] = FpF::one();
    } else {  // the identity (and a rejected encoding: the proof fails on its flag, the pairing's result is not looked at)
        o[0] = FpF::zero();
        o[1] = FpF::one();
        o[2] = FpF::zero();
    }
    flags[2 * inst + which] = st;
}

}  // namespace kzg
