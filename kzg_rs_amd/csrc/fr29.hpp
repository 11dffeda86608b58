// fr29.hpp - BLS12-381 Fr in radix 2^29 (9 limbs of 29 bits in 32-bit words), Montgomery radix R' = 2^261.
// The representation of the evaluation kernel (fr_kernels.hpp); everything else keeps the 8x32 form of field.hpp.
//
// Why: on gfx950 v_mad_u64_u32, v_add_co_u32 and v_addc_co_u32 all cost ~4.2 issue cycles per wave-instruction and
// plain v_add_u32 / v_and_b32 / v_lshrrev_b32 ~2.4 (profiles/r1_issuebench_valu_issue_cost.txt).  A saturated 8x32
// Montgomery product is 136 multiply-adds PLUS one carry instruction per multiply-add (272 x 4.2 cycles); with 29-bit
// limbs a whole column of products (each < 2^60.7) accumulates in ONE 64-bit register with no carry instruction at
// all, r = 1 mod 2^29 makes the Montgomery factor a negation, and additions are limb-wise with no carry chain and no
// conditional subtraction: 162 multiply-adds + ~45 cheap instructions, measured ~0.6x the cycles of the 8x32 product.
//
// Value discipline (no reduction inside the tree; checked by tests/test_fr29_host.py with worst-case inputs):
//   fr29_mul(a, b): a "wide" (every limb <= 2^31.33), b "narrow" (every limb < 2^29).  Column sums stay below
//     9 * 2^31.33 * 2^29 + 9 * 2^58 + 2^35.5 < 2^63.8.  Output limbs < 2^29 (top limb: value >> 232), and
//     value < (value(a) * value(b) / (70 r^2) + 1) r   because 2^261 > 70 r;  all call sites keep that below 3 r.
//   fr29_add: limb-wise.   fr29_sub_biased(a, b) = a + 8r - b with 8r written so that no limb borrows while
//     b's limbs 0..7 are <= 2^30 - 2 and b's top limb <= 2 * (2r >> 232)  (b = sum of two product outputs).
//   fr29_mul2(a, b, c, d) = a b + c d under one reduction - the tree's merge step z^(2^L) (na + nb) + w (na - nb):
//     sum limbs < 2^30, difference (fr29_sub_biased4: + 4r) limbs < 3 * 2^29, columns below 2^63.8; nodes stay below 1.2 r.
// The file compiles for the host too (plain C++), which is how the bounds are unit-tested without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FR29_FN __host__ __device__ __forceinline__
#else
#define FR29_FN inline
#endif

namespace kzg {
namespace c29 {
#define KZG_CONST static constexpr
#include "constants.inc"
#undef KZG_CONST
}  // namespace c29

constexpr uint32_t FR29_MASK = 0x1FFFFFFFu;
struct Fr29 {
    uint32_t l[9];
};

FR29_FN Fr29 fr29_const(const uint32_t (&c)[9]) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[i];
    return r;
}

// 8 little-endian 32-bit words (an integer < 2^256) <-> 9 limbs of 29 bits
FR29_FN Fr29 fr29_from_words(const uint32_t (&w)[8]) {
    Fr29 r;
    r.l[0] = w[0] & FR29_MASK;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const int sh = 32 - 3 * i;  // bit 29 i = 32 (i - 1) + sh; a funnel shift (v_alignbit_b32)
        r.l[i] = ((w[i - 1] >> sh) | (w[i] << (32 - sh))) & FR29_MASK;
    }
    r.l[8] = w[7] >> 8;
    return r;
}
// limbs < 2^29, value < 2^256
FR29_FN void fr29_to_words(uint32_t (&w)[8], const Fr29& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (a.l[i] >> (3 * i)) | (a.l[i + 1] << (29 - 3 * i));
}

FR29_FN Fr29 fr29_add(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
FR29_FN Fr29 fr29_sub_biased(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + c29::FR29_BIAS8[i] - b.l[i];
    return r;
}
// a + 4r - b, for b a product output (limbs 0..7 < 2^29, value < 2r): result limbs below 2^29 + 2^30 when a's are below 2^29
FR29_FN Fr29 fr29_sub_biased4(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + c29::FR29_BIAS4[i] - b.l[i];
    return r;
}
// carry propagation: limbs 0..7 back below 2^29, the excess collects in the top limb (same value)
FR29_FN Fr29 fr29_normalize(const Fr29& a) {
    Fr29 r = a;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.l[i + 1] += r.l[i] >> 29;
        r.l[i] &= FR29_MASK;
    }
    return r;
}

// acc += m * 1: on the device a multiply-add (one 4-cycle instruction; hipcc would turn `acc += m` into a two
// instruction 64-bit add with carry)
FR29_FN void fr29_acc_add(uint64_t& acc, uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(m) : "vcc");
#else
    acc += m;
#endif
}

// a modulus limb as an opaque scalar register: hipcc otherwise strength-reduces m * 0x1ffffff8 into two 64-bit
// shift-adds, twice the cost of the multiply-add it replaces
FR29_FN uint32_t fr29_mod_limb(int i) {
    uint32_t v = c29::FR29_MOD[i];
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(v));
#endif
    return v;
}

// Montgomery product a * b * 2^-261 mod r; a wide, b narrow (see the header comment)
FR29_FN Fr29 fr29_mul(const Fr29& a, const Fr29& b) {
    uint64_t acc = 0;
    uint32_t m[9], mod[9];
    Fr29 out;
#pragma unroll
    for (int i = 1; i < 9; i++) mod[i] = fr29_mod_limb(i);
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * mod[k - i];
        m[k] = (0u - (uint32_t)acc) & FR29_MASK;  // -r^-1 = -1 mod 2^29
        fr29_acc_add(acc, m[k]);                  // m_k * r_0, r_0 = 1: the low 29 bits are now zero
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * mod[k - i];
        }
        out.l[k - 9] = (uint32_t)acc & FR29_MASK;
        acc >>= 29;
    }
    out.l[8] = (uint32_t)acc;
    return out;
}

// (a b + c d) 2^-261 mod r with ONE Montgomery reduction for the two products: 162 + 72 + 9 multiply-adds instead of
// 2 x (81 + 72 + 9) and one set of shifts and masks instead of two.  b, d narrow (limbs < 2^29); a, c moderately wide:
// (largest limb of a) + (largest limb of c) <= 5 * 2^29, which keeps every column below 9 * 5 * 2^58 + 9 * 2^58 + carry
// < 2^63.8.  Output limbs < 2^29, value < ((value(a) value(b) + value(c) value(d)) / (70 r^2) + 1) r.
FR29_FN Fr29 fr29_mul2(const Fr29& a, const Fr29& b, const Fr29& c, const Fr29& d) {
    uint64_t acc = 0;
    uint32_t m[9], mod[9];
    Fr29 out;
#pragma unroll
    for (int i = 1; i < 9; i++) mod[i] = fr29_mod_limb(i);
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * mod[k - i];
        m[k] = (0u - (uint32_t)acc) & FR29_MASK;
        fr29_acc_add(acc, m[k]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
            acc += (uint64_t)m[i] * mod[k - i];
        }
        out.l[k - 9] = (uint32_t)acc & FR29_MASK;
        acc >>= 29;
    }
    out.l[8] = (uint32_t)acc;
    return out;
}

}  // namespace kzg
