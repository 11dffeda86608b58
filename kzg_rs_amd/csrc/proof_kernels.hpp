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

// z, y: 8 little-endian 32-bit limbs each (canonical; pinned host memory or device memory)
__global__ __launch_bounds__(128) void k_proof_select(const uint32_t* __restrict__ z, const uint32_t* __restrict__ y, const Fp* __restrict__ table,
                                                      const Fp* __restrict__ tau4, Fp* __restrict__ out) {
    const int t = threadIdx.x;
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

// cp: the 48 compressed bytes of C, then those of pi.  v3in[0..2] = pi, v3in[3..5] = C (VERIFY3's first six inputs);
// flags[0] = status of C, flags[1] = status of pi (G1_OK / G1_INFINITY / G1_INVALID; the subgroup is NOT tested here).
__global__ __launch_bounds__(64) void k_proof_decompress(const uint8_t* __restrict__ cp, Fp* __restrict__ v3in, uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint4 proof_park4[];  // PARK_UINT4_PER_THREAD per thread
    const int lane = threadIdx.x;
    if (lane >= 2) return;
    const LdsPark pk = lds_park(proof_park4 + lane, blockDim.x);
    Fp29 x, y;
    const uint32_t st = g1_decompress29(x, y, cp + 48 * lane, pk);
    Fp* const o = v3in + (lane == 0 ? 3 : 0);
    if (st == G1_OK) {
        o[0] = fp29_to_std(x);
        o[1] = fp29_to_std(y);
        o[2] = FpF::one();
    } else {  // the identity (and a rejected encoding: the call fails on its flag, the pairing's result is not looked at)
        o[0] = FpF::zero();
        o[1] = FpF::one();
        o[2] = FpF::zero();
    }
    flags[lane] = st;
}

}  // namespace kzg
