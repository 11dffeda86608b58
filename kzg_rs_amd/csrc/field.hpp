// field.hpp - BLS12-381 Fr (8x32) / Fp (12x32) Montgomery arithmetic for gfx950, per-thread.
//
// Why 32-bit limbs: the CDNA4 VALU has no 64x64 multiply; the widest integer multiply-add is
// v_mad_u64_u32 (32x32 + 64 -> 64).  With 32-bit limbs every step of the CIOS inner loop
// `(u64)a*b + t + c` is exactly one v_mad_u64_u32 plus one 64-bit add and can never overflow
// ((2^32-1)^2 + 2(2^32-1) = 2^64-1).  Memory layout is little-endian limbs, so 8x32 / 12x32 is
// byte-identical to the reference crates' 4x64 / 6x64 in-memory Montgomery structs
// (reference build.rs:195-206).
//
// Reference semantics implemented here (sp1_bls12_381 is un-vendored; call sites in
// /root/reference/src/kzg_proof.rs): Scalar::from_bytes (:36), Scalar::from_raw (:90),
// Scalar::to_bytes (:321), + - * (:105-131, :172-198).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kzg {
// A kernel's own execution interval, for the launches that share the chip with other launch groups (a HIP-event pair around
// such a launch also times the wait for free CUs): kt[0] = max over the waves of ~start, kt[1] = max of end, in ticks of the
// 100 MHz s_memrealtime counter; the first lane of every wavefront stamps, the host zeroes kt before the launch and reads
// end - start = kt[1] - ~kt[0].  kt == nullptr: no stamp.
__device__ __forceinline__ void kstamp_in(unsigned long long* kt) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (kt && (threadIdx.x & 63) == 0) atomicMax(&kt[0], ~__builtin_amdgcn_s_memrealtime());
#else
    (void)kt;
#endif
}
__device__ __forceinline__ void kstamp_out(unsigned long long* kt) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (kt && (threadIdx.x & 63) == 0) atomicMax(&kt[1], __builtin_amdgcn_s_memrealtime());
#else
    (void)kt;
#endif
}
namespace consts {
#define KZG_CONST static constexpr
#include "constants.inc"
#undef KZG_CONST
}  // namespace consts

#define KZG_DEV __device__ __forceinline__

template <int N>
struct alignas(16) Limbs {
    uint32_t l[N];
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = FR_INV32;
    KZG_DEV static constexpr uint32_t mod(int i) { return consts::FR_MOD[i]; }
    KZG_DEV static constexpr uint32_t one(int i) { return consts::FR_ONE[i]; }
    KZG_DEV static constexpr uint32_t r2(int i) { return consts::FR_R2[i]; }
};
struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = FP_INV32;
    KZG_DEV static constexpr uint32_t mod(int i) { return consts::FP_MOD[i]; }
    KZG_DEV static constexpr uint32_t one(int i) { return consts::FP_ONE[i]; }
    KZG_DEV static constexpr uint32_t r2(int i) { return consts::FP_R2[i]; }
};

// ---------------------------------------------------------------- carry helpers
// clang's carry builtins lower to real v_add_co_u32 / v_addc_co_u32 (v_sub_co / v_subb_co) chains on
// gfx950: one VALU instruction per limb.  (The portable `(uint64_t)a + b + carry` idiom is compiled by
// hipcc into 64-bit adds plus register-pair shuffling, ~7 instructions per limb.)
KZG_DEV uint32_t addc(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned c = carry;
    uint32_t r = __builtin_addc(a, b, c, &c);
    carry = c;
    return r;
}
KZG_DEV uint32_t subb(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned c = borrow;
    uint32_t r = __builtin_subc(a, b, c, &c);
    borrow = c;
    return r;
}

#include "mac_chains.inc"

template <class P>
struct Field {
    static constexpr int N = P::N;
    using E = Limbs<N>;

    KZG_DEV static E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    KZG_DEV static E one() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::one(i);
        return r;
    }
    KZG_DEV static E modulus() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::mod(i);
        return r;
    }
    KZG_DEV static bool is_zero(const E& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= a.l[i];
        return o == 0;
    }
    KZG_DEV static bool eq(const E& a, const E& b) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= a.l[i] ^ b.l[i];
        return o == 0;
    }
    // a >= modulus ?  (plain integer compare)
    KZG_DEV static bool geq_mod(const E& a) {
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) (void)subb(a.l[i], P::mod(i), borrow);
        return borrow == 0;
    }
    // r = a - modulus if a >= modulus (a < 2*modulus, possibly with an extra carry bit `hi`)
    KZG_DEV static E reduce_once(const E& a, uint32_t hi = 0) {
        E d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) d.l[i] = subb(a.l[i], P::mod(i), borrow);
        bool take = (hi != 0) | (borrow == 0);
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = take ? d.l[i] : a.l[i];
        return r;
    }
    KZG_DEV static E add(const E& a, const E& b) {
        E s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) s.l[i] = addc(a.l[i], b.l[i], c);
        return reduce_once(s, c);
    }
    KZG_DEV static E sub(const E& a, const E& b) {
        E d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) d.l[i] = subb(a.l[i], b.l[i], borrow);
        E t, r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = addc(d.l[i], P::mod(i), c);
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
        return r;
    }
    KZG_DEV static E neg(const E& a) { return sub(zero(), a); }
    KZG_DEV static E dbl(const E& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod m (CIOS, fully unrolled; requires a*b < m*R).
    // Kept for A/B measurement (tools/microbench): hipcc spends half of it on v_mov.
    KZG_DEV static E mul_cios(const E& a, const E& b) {
        uint32_t t[N + 2];
#pragma unroll
        for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
                t[j] = (uint32_t)x;
                c = x >> 32;
            }
            uint64_t x = (uint64_t)t[N] + c;
            t[N] = (uint32_t)x;
            t[N + 1] = (uint32_t)(x >> 32);
            uint32_t q = t[0] * P::INV;
            x = (uint64_t)q * P::mod(0) + t[0];
            c = x >> 32;
#pragma unroll
            for (int j = 1; j < N; j++) {
                x = (uint64_t)q * P::mod(j) + t[j] + c;
                t[j - 1] = (uint32_t)x;
                c = x >> 32;
            }
            x = (uint64_t)t[N] + c;
            t[N - 1] = (uint32_t)x;
            t[N] = t[N + 1] + (uint32_t)(x >> 32);
        }
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        return reduce_once(r, t[N]);
    }
    // Same product, finely-integrated product scanning (column-wise): the 96-bit column
    // accumulator is {acc (aligned 64-bit VGPR pair), hi}; each partial product is ONE
    // v_mad_u64_u32 that accumulates in place (no register-pair shuffling) plus one 32-bit
    // v_addc for the carry-out.  4N^2 + O(N) VALU instructions instead of ~10N^2.
    KZG_DEV static void mac(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b) {
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(acc), "+v"(hi)
            : "v"(a), "v"(b)
            : "vcc");
    }
    KZG_DEV static E mul_ps(const E& a, const E& b) {
        uint64_t acc = 0;
        uint32_t hi = 0;
        uint32_t m[N];
        E r;
#pragma unroll
        for (int k = 0; k < N; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) mac(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
            for (int i = 0; i < k; i++) mac(acc, hi, m[i], P::mod(k - i));
            m[k] = (uint32_t)acc * P::INV;
            mac(acc, hi, m[k], P::mod(0));
            acc = (acc >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#pragma unroll
        for (int k = N; k < 2 * N; k++) {
#pragma unroll
            for (int i = k - N + 1; i < N; i++) mac(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
            for (int i = k - N + 1; i < N; i++) mac(acc, hi, m[i], P::mod(k - i));
            r.l[k - N] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        return reduce_once(r, (uint32_t)acc);
    }
    // Product scanning with each column's chain fused into (at most) two asm statements.
    KZG_DEV static E mul(const E& a, const E& b) {
        uint64_t acc = 0;
        uint32_t hi = 0;
        uint32_t m[N], x[N], y[N];
        E r;
#pragma unroll
        for (int k = 0; k < N; k++) {
            // a_i * b_(k-i), i = 0..k
#pragma unroll
            for (int i = 0; i <= k; i++) { x[i] = a.l[i]; y[i] = b.l[k - i]; }
            chain(acc, hi, x, y, k + 1);
            // m_i * p_(k-i), i = 0..k-1
#pragma unroll
            for (int i = 0; i < k; i++) { x[i] = m[i]; y[i] = P::mod(k - i); }
            chain(acc, hi, x, y, k);
            m[k] = (uint32_t)acc * P::INV;
            x[0] = m[k]; y[0] = P::mod(0);
            chain(acc, hi, x, y, 1);
            acc = (acc >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#pragma unroll
        for (int k = N; k < 2 * N; k++) {
            const int lo = k - N + 1, cnt = N - lo;
#pragma unroll
            for (int i = 0; i < cnt; i++) { x[i] = a.l[lo + i]; y[i] = b.l[k - lo - i]; }
            chain(acc, hi, x, y, cnt);
#pragma unroll
            for (int i = 0; i < cnt; i++) { x[i] = m[lo + i]; y[i] = P::mod(k - lo - i); }
            chain(acc, hi, x, y, cnt);
            r.l[k - N] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        return reduce_once(r, (uint32_t)acc);
    }
    KZG_DEV static void chain(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y, int len) {
        switch (len) {  // len is a compile-time constant after unrolling
            case 1: MacChain<1>::run(acc, hi, x, y); break;
            case 2: MacChain<2>::run(acc, hi, x, y); break;
            case 3: MacChain<3>::run(acc, hi, x, y); break;
            case 4: MacChain<4>::run(acc, hi, x, y); break;
            case 5: MacChain<5>::run(acc, hi, x, y); break;
            case 6: MacChain<6>::run(acc, hi, x, y); break;
            case 7: MacChain<7>::run(acc, hi, x, y); break;
            case 8: MacChain<8>::run(acc, hi, x, y); break;
            case 9: MacChain<9>::run(acc, hi, x, y); break;
            case 10: MacChain<10>::run(acc, hi, x, y); break;
            case 11: MacChain<11>::run(acc, hi, x, y); break;
            case 12: MacChain<12>::run(acc, hi, x, y); break;
            default: break;
        }
    }
    KZG_DEV static E sqr(const E& a) { return mul(a, a); }

    // plain integer (< m) -> Montgomery form; also reduces any N-limb value mod m (a*R2 < R*m).
    KZG_DEV static E to_mont(const E& a) {
        E r2;
#pragma unroll
        for (int i = 0; i < N; i++) r2.l[i] = P::r2(i);
        return mul(a, r2);
    }
    KZG_DEV static E from_mont(const E& a) {
        E o = zero();
        o.l[0] = 1;
        return mul(a, o);
    }
    // big-endian bytes -> little-endian limbs (plain integer)
    KZG_DEV static E from_be_bytes(const uint8_t* b) {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint8_t* p = b + 4 * (N - 1 - i);
            r.l[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
        }
        return r;
    }
    KZG_DEV static void to_be_bytes(uint8_t* b, const E& a) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint8_t* p = b + 4 * (N - 1 - i);
            p[0] = (uint8_t)(a.l[i] >> 24);
            p[1] = (uint8_t)(a.l[i] >> 16);
            p[2] = (uint8_t)(a.l[i] >> 8);
            p[3] = (uint8_t)a.l[i];
        }
    }
};

using FrF = Field<FrParams>;
using FpF = Field<FpParams>;
using Fr = FrF::E;
using Fp = FpF::E;

// generic exponentiation, exponent as NE 32-bit limbs (plain integer), MSB-first square-multiply.
// `mulfn` is a callable so heavy fields can pass a non-inlined multiply.
template <class E, int NE, class Mul>
__device__ inline E pow_limbs(const E& a, const uint32_t (&e)[NE], const E& one, Mul mulfn) {
    E acc = one;
    bool started = false;
    for (int i = 32 * NE - 1; i >= 0; i--) {
        if (started) acc = mulfn(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? mulfn(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

}  // namespace kzg
