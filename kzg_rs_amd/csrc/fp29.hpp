// fp29.hpp - BLS12-381 Fp in radix 2^29 (14 limbs of 29 bits in 32-bit words), Montgomery radix R'' = 2^406.
// The representation of the point-decode and MSM kernels; the rest of the library keeps the 12x32 form of field.hpp
// (conversions at the kernels' edges).
//
// Why (profiles/r1_issuebench_valu_issue_cost.txt, tools/microbench/fp29bench.hip): a 12x32 Montgomery product is 288
// multiply-adds each followed by a carry instruction, both 4.2-cycle class: 2 612 SIMD cycles per wave-product.  With
// 29-bit limbs a whole column (<= 28 products of < 2^58) accumulates in ONE 64-bit register with no carry instruction:
// 2 029 cycles per product and 1 726 per SQUARE (cross products taken once against a doubled operand) - and three
// quarters of the field operations of point decoding are squarings.
//
// Lazy reduction.  406 - 381 = 25 spare bits: values may grow to ~6000 p before a product of two of them leaves the
// range in which the Montgomery output stays below 2p (a b < 2^406 p), so nothing is ever reduced conditionally:
//   fp29_mul / fp29_sqr : inputs NORMALISED (limbs 0..12 < 2^29) and below 6000 p;  output normalised, below 2p.
//   fp29_add            : limb-wise + carry propagation (normalised output), value = a + b.
//   fp29_sub<E>(a, b)   : a + 2^E p - b, for b normalised with value(b) <= 2^(E-1) p; the constant is stored with
//                         every limb below the top one pre-biased by 2^29, so no limb ever borrows.
// Equality with zero mod p is only ever asked of product OUTPUTS (below 2p: the value is 0 or p) - the point formulas
// test H^2, R^2 and Z^2, which they compute anyway, instead of H, R and Z.
// The file compiles for the host too (plain C++): tests/test_fp29_host.py checks results and bounds without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FP29_FN __host__ __device__ __forceinline__
#else
#define FP29_FN inline
#endif

namespace kzg {
namespace cp29 {
#define KZG_CONST static constexpr
#include "constants.inc"
#undef KZG_CONST
}  // namespace cp29

constexpr uint32_t FP29_MASK = 0x1FFFFFFFu;
struct Fp29 {
    uint32_t l[14];
};

FP29_FN Fp29 fp29_const(const uint32_t (&c)[14]) {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = c[i];
    return r;
}
FP29_FN Fp29 fp29_zero() {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = 0;
    return r;
}

// modulus limbs / -p^-1 as opaque scalar registers (hipcc strength-reduces multiplications by literals it can see)
FP29_FN uint32_t fp29_opaque(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(v));
#endif
    return v;
}

// carry propagation: limbs 0..12 back below 2^29, the excess collects in the top limb (same value)
FP29_FN Fp29 fp29_normalize(const Fp29& a) {
    Fp29 r = a;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        r.l[i + 1] += r.l[i] >> 29;
        r.l[i] &= FP29_MASK;
    }
    return r;
}
FP29_FN Fp29 fp29_add(const Fp29& a, const Fp29& b) {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    return fp29_normalize(r);
}
FP29_FN Fp29 fp29_dbl(const Fp29& a) { return fp29_add(a, a); }
// a + 2^E p - b   (b normalised, value(b) <= 2^(E-1) p)
template <int E>
FP29_FN Fp29 fp29_sub(const Fp29& a, const Fp29& b) {
    static_assert(E >= 1 && E <= 10, "bias table holds 2^1 p .. 2^10 p");
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + cp29::FP29_BIAS[E - 1][i] - b.l[i];
    return fp29_normalize(r);
}
template <int E>
FP29_FN Fp29 fp29_neg(const Fp29& b) {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = cp29::FP29_BIAS[E - 1][i] - b.l[i];
    return fp29_normalize(r);
}

// Montgomery product a * b * 2^-406 mod p (SQR: a * a, b ignored).  Inputs normalised; output normalised, < 2p.
template <bool SQR>
FP29_FN Fp29 fp29_mul_impl(const Fp29& a, const Fp29& b) {
    uint64_t acc = 0;
    uint32_t m[14], mod[14], a2[14];
    Fp29 out;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        mod[i] = fp29_opaque(cp29::FP29_MOD[i]);
        a2[i] = SQR ? a.l[i] << 1 : 0u;
    }
    const uint32_t pinv = fp29_opaque(FP29_INV);
#pragma unroll
    for (int k = 0; k < 28; k++) {
        const int lo = k < 14 ? 0 : k - 13, hi = k < 14 ? k : 13;
        if (SQR) {
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                const int j = k - i;
                if (i < j) acc += (uint64_t)a.l[i] * a2[j];
                else if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
            }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k >= 14 || i < k) acc += (uint64_t)m[i] * mod[k - i];
        if (k < 14) {
            m[k] = ((uint32_t)acc * pinv) & FP29_MASK;
            acc += (uint64_t)m[k] * mod[0];  // the low 29 bits are now zero
        } else {
            out.l[k - 14] = (uint32_t)acc & FP29_MASK;
        }
        acc >>= 29;
    }
    out.l[13] |= (uint32_t)acc << 29;  // zero for in-range inputs (output < 2p < 2^382)
    return out;
}
FP29_FN Fp29 fp29_mul(const Fp29& a, const Fp29& b) { return fp29_mul_impl<false>(a, b); }
FP29_FN Fp29 fp29_sqr(const Fp29& a) { return fp29_mul_impl<true>(a, a); }

// t = a product output (normalised, below 2p): is t = 0 mod p, i.e. t in {0, p} ?
FP29_FN bool fp29_is_zero_mod_p(const Fp29& t) {
    uint32_t z = 0, q = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        z |= t.l[i];
        q |= t.l[i] ^ cp29::FP29_MOD[i];
    }
    return z == 0 || q == 0;
}

// 12 little-endian 32-bit words (an integer < 2^384) <-> 14 limbs of 29 bits (the top limb takes bits 377..383)
FP29_FN Fp29 fp29_from_words(const uint32_t (&w)[12]) {
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 29 * i, q = bit >> 5, sh = bit & 31;
        uint32_t v = w[q] >> sh;
        if (sh > 3 && q + 1 < 12) v |= w[q + 1] << (32 - sh);
        r.l[i] = i < 13 ? (v & FP29_MASK) : v;
    }
    return r;
}
// limbs 0..12 < 2^29, value < 2^384
FP29_FN void fp29_to_words(uint32_t (&w)[12], const Fp29& a) {
#pragma unroll
    for (int q = 0; q < 12; q++) {
        // word q holds bits [32 q, 32 q + 32): pieces of limb i = (32 q) / 29 and the next one or two
        const int bit = 32 * q, i = bit / 29, off = bit - 29 * i;
        uint32_t v = a.l[i] >> off;
        if (i + 1 < 14) v |= a.l[i + 1] << (29 - off);
        if (29 - off + 29 < 32 && i + 2 < 14) v |= a.l[i + 2] << (58 - off);
        w[q] = v;
    }
}

}  // namespace kzg
