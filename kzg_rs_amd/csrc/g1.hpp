// g1.hpp - per-thread Fp / G1 (and the little Fp2 needed to decompress [tau]G2) for gfx950.
//
// Implements, on the device, what the reference gets from sp1_bls12_381 at
//   G1Affine::from_compressed            src/kzg_proof.rs:18   (flags, x < p, sqrt, sign, subgroup)
//   G2Affine::from_compressed_unchecked  build.rs:73
// Encoding: ZCash/IETF compressed BLS12-381 points.
//
// Fp multiplication is kept out of line (fp_mul_raw is __noinline__): one Fp product is ~760 VALU
// instructions, and point formulas use dozens of them; inlining would blow the instruction cache.
// The point formulas themselves are inlined at (few) call sites: big struct arguments would go through scratch.
#pragma once
#include "field.hpp"

namespace kzg {

// Out-of-line Fp product.  Operands travel as 12-lane ext-vectors: AMDGPU passes vector arguments in VGPRs,
// whereas a 48-byte struct argument goes through scratch memory (a store + load round trip per call).
typedef uint32_t v12u __attribute__((ext_vector_type(12)));
__device__ __noinline__ v12u fp_mul_raw(v12u a, v12u b) {
    Fp x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    Fp r = FpF::mul(x, y);
    v12u o;
#pragma unroll
    for (int i = 0; i < 12; i++) o[i] = r.l[i];
    return o;
}
__device__ __forceinline__ Fp fp_mul(const Fp& a, const Fp& b) {
    v12u x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x[i] = a.l[i];
        y[i] = b.l[i];
    }
    v12u o = fp_mul_raw(x, y);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = o[i];
    return r;
}
__device__ __forceinline__ Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }
__device__ __forceinline__ Fp fp_add(const Fp& a, const Fp& b) { return FpF::add(a, b); }
__device__ __forceinline__ Fp fp_sub(const Fp& a, const Fp& b) { return FpF::sub(a, b); }
__device__ __forceinline__ Fp fp_neg(const Fp& a) { return FpF::neg(a); }
__device__ __forceinline__ Fp fp_dbl(const Fp& a) { return FpF::add(a, a); }

template <int NE>
__device__ inline Fp fp_pow(const Fp& a, const uint32_t (&e)[NE]) {
    Fp acc = FpF::one();
    bool started = false;
    for (int i = 32 * NE - 1; i >= 0; i--) {
        if (started) acc = fp_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fp_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

__device__ inline Fp fp_const(const uint32_t (&c)[12]) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = c[i];
    return r;
}

// a^-1 by Fermat (a != 0).  Used only in set-up / per-call tails, never per blob element.
__device__ inline Fp fp_inv(const Fp& a) { return fp_pow(a, consts::FP_P_MINUS_2); }

// a^-1 for a LONE LANE (a != 0): the binary extended Euclidean algorithm on the plain integers - ~760 halvings and ~380 subtractions of
// 12-limb numbers (~55 k instructions) against the 381 squarings + 190 products of the Fermat chain (~260 k): 0.1 ms instead of 1.0 ms
// when one lane converts one point (k_jac_compress: the tail of every kzg_g1_msm / kzg_g1_msm_setup call).  Variable time and full of
// data-dependent branches: right for a single lane, wasteful for a wavefront of independent inversions (those keep fp_inv).
// Input and output in Montgomery form: the algorithm inverts x = a R as an integer, t = a^-1 R^-1; two products by R^2 give a^-1 R.
__device__ inline Fp fp_inv_lone_lane(const Fp& a) {
    auto is_one = [](const Fp& x) {
        uint32_t o = x.l[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 12; i++) o |= x.l[i];
        return o == 0;
    };
    auto shr1 = [](Fp& x, uint32_t top) {
#pragma unroll
        for (int i = 0; i < 11; i++) x.l[i] = (x.l[i] >> 1) | (x.l[i + 1] << 31);
        x.l[11] = (x.l[11] >> 1) | (top << 31);
    };
    auto half_mod = [&](Fp& x) {  // x / 2 mod p for 0 <= x < p  (x + p < 2^382 stays inside 12 limbs, the carry is always 0)
        uint32_t c = 0;
        if (x.l[0] & 1u) {
#pragma unroll
            for (int i = 0; i < 12; i++) x.l[i] = addc(x.l[i], consts::FP_MOD[i], c);
        }
        shr1(x, c);
    };
    auto sub_mod = [](Fp& x, const Fp& y) {  // x - y mod p for 0 <= x, y < p
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) x.l[i] = subb(x.l[i], y.l[i], b);
        if (b) {
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) x.l[i] = addc(x.l[i], consts::FP_MOD[i], c);
        }
    };
    Fp u = a, v = fp_const(consts::FP_MOD), x1 = FpF::zero(), x2 = FpF::zero();
    x1.l[0] = 1u;
#pragma unroll 1
    while (!is_one(u) && !is_one(v)) {
#pragma unroll 1
        while ((u.l[0] & 1u) == 0) {
            shr1(u, 0u);
            half_mod(x1);
        }
#pragma unroll 1
        while ((v.l[0] & 1u) == 0) {
            shr1(v, 0u);
            half_mod(x2);
        }
        Fp d;
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) d.l[i] = subb(u.l[i], v.l[i], b);
        if (!b) {  // u >= v
            u = d;
            sub_mod(x1, x2);
        } else {
            uint32_t b2 = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) v.l[i] = subb(v.l[i], u.l[i], b2);
            sub_mod(x2, x1);
        }
    }
    const Fp t = is_one(u) ? x1 : x2;  // (a R)^-1 as a plain integer
    const Fp r2 = fp_const(consts::FP_R2);
    return fp_mul(fp_mul(t, r2), r2);
}

// a^e with a 3-bit sliding window over the CONSTANT exponent e (the same in every lane, so the window walk is
// wave-uniform and the table of odd powers a, a^3, a^5, a^7 lives in registers, picked by selects).  For the
// square-root exponent: 379 squarings + 107 multiplications + 4 for the table, against 379 + 229 bit by bit.
template <int NE>
__device__ inline Fp fp_pow_window3(const Fp& a, const uint32_t (&e)[NE]) {
    const Fp a2 = fp_sqr(a);
    Fp t[4];
    t[0] = a;
    for (int k = 1; k < 4; k++) t[k] = fp_mul(t[k - 1], a2);
    Fp acc = FpF::one();
    bool started = false;
    int i = 32 * NE - 1;
    while (i >= 0) {
        if (!((e[i >> 5] >> (i & 31)) & 1)) {
            if (started) acc = fp_sqr(acc);
            i--;
            continue;
        }
        // window [i .. j], j the lowest set bit within 3 positions
        int j = i - 2 < 0 ? 0 : i - 2;
        while (!((e[j >> 5] >> (j & 31)) & 1)) j++;
        uint32_t v = 0;
        for (int k = i; k >= j; k--) v = (v << 1) | ((e[k >> 5] >> (k & 31)) & 1);
        if (started)
            for (int k = i; k >= j; k--) acc = fp_sqr(acc);
        const uint32_t idx = v >> 1;  // v is odd: 1, 3, 5, 7
        Fp m;
#pragma unroll
        for (int w = 0; w < 12; w++) m.l[w] = idx == 0 ? t[0].l[w] : idx == 1 ? t[1].l[w] : idx == 2 ? t[2].l[w] : t[3].l[w];
        acc = started ? fp_mul(acc, m) : m;
        started = true;
        i = j - 1;
    }
    return acc;
}

// sqrt for p = 3 mod 4: candidate a^((p+1)/4); returns false if a is not a square
__device__ inline bool fp_sqrt(Fp& r, const Fp& a) {
    Fp c = fp_pow_window3(a, consts::FP_SQRT_EXP);
    r = c;
    return FpF::eq(fp_sqr(c), a);
}

// plain-integer test  v > (p-1)/2  of a Montgomery-form element ("lexicographically largest")
__device__ inline bool fp_is_lex_largest(const Fp& a_mont) {
    Fp a = FpF::from_mont(a_mont);
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) (void)subb(consts::FP_HALF[i], a.l[i], borrow);  // half - a < 0  <=>  a > half
    return borrow != 0;
}

// ---------------------------------------------------------------- G1
struct G1Aff {
    Fp x, y;  // Montgomery form
};
struct G1Jac {
    Fp x, y, z;  // Jacobian, z == 0 <=> infinity
};

__device__ inline G1Jac g1_identity() {
    G1Jac r;
    r.x = FpF::zero();
    r.y = FpF::one();
    r.z = FpF::zero();
    return r;
}
__device__ inline bool g1_is_identity(const G1Jac& p) { return FpF::is_zero(p.z); }

__device__ inline G1Jac g1_from_affine(const G1Aff& a) {
    G1Jac r;
    r.x = a.x;
    r.y = a.y;
    r.z = FpF::one();
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S
__device__ __forceinline__ G1Jac g1_dbl(const G1Jac& p) {
    Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    Fp t = fp_sqr(fp_add(p.x, B));
    Fp D = fp_dbl(fp_sub(fp_sub(t, A), C));
    Fp E = fp_add(fp_dbl(A), A);
    Fp F = fp_sqr(E);
    G1Jac r;
    r.x = fp_sub(F, fp_dbl(D));
    Fp C8 = fp_dbl(fp_dbl(fp_dbl(C)));
    r.z = fp_dbl(fp_mul(p.y, p.z));
    r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), C8);
    return r;
}

// general Jacobian addition with every special case handled (identity operands, P+P, P-P)
__device__ __forceinline__ G1Jac g1_add(const G1Jac& p, const G1Jac& q) {
    if (g1_is_identity(p)) return q;
    if (g1_is_identity(q)) return p;
    Fp Z1Z1 = fp_sqr(p.z), Z2Z2 = fp_sqr(q.z);
    Fp U1 = fp_mul(p.x, Z2Z2), U2 = fp_mul(q.x, Z1Z1);
    Fp S1 = fp_mul(fp_mul(p.y, q.z), Z2Z2), S2 = fp_mul(fp_mul(q.y, p.z), Z1Z1);
    if (FpF::eq(U1, U2)) {
        if (FpF::eq(S1, S2)) return g1_dbl(p);
        return g1_identity();
    }
    Fp H = fp_sub(U2, U1), Rr = fp_sub(S2, S1);
    Fp HH = fp_sqr(H), HHH = fp_mul(H, HH), V = fp_mul(U1, HH);
    G1Jac r;
    r.x = fp_sub(fp_sub(fp_sqr(Rr), HHH), fp_dbl(V));
    r.y = fp_sub(fp_mul(Rr, fp_sub(V, r.x)), fp_mul(S1, HHH));
    r.z = fp_mul(fp_mul(p.z, q.z), H);
    return r;
}

// mixed addition p + (affine q, q != identity)
__device__ __forceinline__ G1Jac g1_add_affine(const G1Jac& p, const G1Aff& q) {
    if (g1_is_identity(p)) return g1_from_affine(q);
    Fp Z1Z1 = fp_sqr(p.z);
    Fp U2 = fp_mul(q.x, Z1Z1), S2 = fp_mul(fp_mul(q.y, p.z), Z1Z1);
    if (FpF::eq(p.x, U2)) {
        if (FpF::eq(p.y, S2)) return g1_dbl(p);
        return g1_identity();
    }
    Fp H = fp_sub(U2, p.x), Rr = fp_sub(S2, p.y);
    Fp HH = fp_sqr(H), HHH = fp_mul(H, HH), V = fp_mul(p.x, HH);
    G1Jac r;
    r.x = fp_sub(fp_sub(fp_sqr(Rr), HHH), fp_dbl(V));
    r.y = fp_sub(fp_mul(Rr, fp_sub(V, r.x)), fp_mul(p.y, HHH));
    r.z = fp_mul(p.z, H);
    return r;
}

// [|x|]P for the BLS parameter |x| = 0xd201000000010000 (63 doublings + 5 additions)
__device__ inline G1Jac g1_mul_xabs(const G1Jac& p) {
    G1Jac acc = p;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        acc = g1_dbl(acc);
        if ((BLS_X_ABS >> i) & 1) acc = g1_add(acc, p);
    }
    return acc;
}

// r-torsion test by the endomorphism: P in G1  <=>  phi(P) = -[x^2]P, phi(x, y) = (beta x, y)
// (Scott, "A note on group membership tests for G1, G2 and GT", eprint 2021/1130 - the same
// criterion zkcrypto/bls12_381 uses).  Accepts exactly the points with [r]P = O, which is what the
// CPU checker tests by definition; tests/test_gpu_parity.py::test_g1_decompress compares the two on and off the subgroup.
__device__ inline bool g1_in_subgroup(const G1Aff& p) {
    G1Jac q = g1_from_affine(p);
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++) q = g1_mul_xabs(q);  // [x^2]P (the two signs cancel); one copy of the loop body
    if (g1_is_identity(q)) return false;
    Fp zz = fp_sqr(q.z), zzz = fp_mul(zz, q.z);
    Fp bx = fp_mul(p.x, fp_const(consts::FP_BETA_MONT));
    // phi(P) == -q  <=>  X = beta*x*Z^2  and  Y = -y*Z^3
    return FpF::eq(q.x, fp_mul(bx, zz)) && FpF::eq(q.y, fp_neg(fp_mul(p.y, zzz)));
}

// The same test with its first [|x|] taken right-to-left, so that the doubling chain P, 2P, 4P, ... it walks passes
// through the multiples 2^(STEP k) P the MSM needs anyway (msm.hpp): STEP = 64 -> the chain ends in 2^64 P (64 doublings
// + 5 additions, nothing extra); STEP = 16 -> 2^16 P .. 2^112 P, the chain extended by 48 doublings past the test's
// 64.  `emit(k, point)` is called for k = 1 .. 128 / STEP - 1.  Then [|x|] of the result left-to-right as above.
// The multiples are valid whether or not the test passes.
template <int STEP, class Emit>
__device__ inline bool g1_in_subgroup_with_multiples(const G1Aff& p, Emit emit) {
    G1Jac r = g1_from_affine(p), q = r;  // q is overwritten at bit 16, the lowest set bit of |x|
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        if (i && i % STEP == 0) emit(i / STEP, r);
        if ((BLS_X_ABS >> i) & 1) q = (i == 16) ? r : g1_add(q, r);
        r = g1_dbl(r);
    }
#pragma unroll 1
    for (int i = 64; i < 128; i++) {
        if (i % STEP == 0) emit(i / STEP, r);
        if (i + STEP >= 128) break;  // the last multiple is out: no further doublings
        r = g1_dbl(r);
    }
    q = g1_mul_xabs(q);
    if (g1_is_identity(q)) return false;
    Fp zz = fp_sqr(q.z), zzz = fp_mul(zz, q.z);
    Fp bx = fp_mul(p.x, fp_const(consts::FP_BETA_MONT));
    return FpF::eq(q.x, fp_mul(bx, zz)) && FpF::eq(q.y, fp_neg(fp_mul(p.y, zzz)));
}

enum : uint32_t { G1_OK = 0, G1_INFINITY = 1, G1_INVALID = 2 };

// 48 compressed bytes -> affine (Montgomery).  Returns G1_OK / G1_INFINITY / G1_INVALID.
__device__ inline uint32_t g1_decompress(G1Aff& out, const uint8_t* b, bool check_subgroup) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint8_t* p = b + 4 * (11 - i);
        w[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
    }
    bool c_flag = (w[11] >> 31) & 1, i_flag = (w[11] >> 30) & 1, s_flag = (w[11] >> 29) & 1;
    w[11] &= 0x1fffffffu;
    out.x = FpF::zero();
    out.y = FpF::zero();
    if (!c_flag) return G1_INVALID;
    Fp x;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.l[i] = w[i];
        any |= w[i];
    }
    if (i_flag) return (s_flag || any) ? G1_INVALID : G1_INFINITY;
    if (FpF::geq_mod(x)) return G1_INVALID;
    x = FpF::to_mont(x);
    Fp y2 = fp_add(fp_mul(fp_sqr(x), x), fp_const(consts::FP_B_MONT));
    Fp y;
    if (!fp_sqrt(y, y2)) return G1_INVALID;
    if (fp_is_lex_largest(y) != s_flag) y = fp_neg(y);
    out.x = x;
    out.y = y;
    if (check_subgroup && !g1_in_subgroup(out)) return G1_INVALID;
    return G1_OK;
}

// affine (Montgomery) -> 48 compressed bytes; `inf` selects the identity encoding
__device__ inline void g1_compress(uint8_t* b, const G1Aff& a, bool inf) {
    if (inf) {
        for (int i = 0; i < 48; i++) b[i] = 0;
        b[0] = 0xc0;
        return;
    }
    Fp x = FpF::from_mont(a.x);
    FpF::to_be_bytes(b, x);
    b[0] |= 0x80;
    if (fp_is_lex_largest(a.y)) b[0] |= 0x20;
}

// Jacobian -> affine (one Fermat inversion); returns false for the identity
template <bool LONE_LANE = false>
__device__ inline bool g1_to_affine(G1Aff& out, const G1Jac& p) {
    if (g1_is_identity(p)) {
        out.x = FpF::zero();
        out.y = FpF::zero();
        return false;
    }
    Fp zi = LONE_LANE ? fp_inv_lone_lane(p.z) : fp_inv(p.z), zi2 = fp_sqr(zi);
    out.x = fp_mul(p.x, zi2);
    out.y = fp_mul(p.y, fp_mul(zi2, zi));
    return true;
}

// ---------------------------------------------------------------- Fp2 (only what G2 decompression needs)
struct Fp2 {
    Fp c0, c1;
};
__device__ inline Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    Fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
    Fp m = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    Fp2 r;
    r.c0 = fp_sub(t0, t1);
    r.c1 = fp_sub(fp_sub(m, t0), t1);
    return r;
}
__device__ inline Fp2 fp2_sqr(const Fp2& a) { return fp2_mul(a, a); }
__device__ inline bool fp2_eq(const Fp2& a, const Fp2& b) { return FpF::eq(a.c0, b.c0) && FpF::eq(a.c1, b.c1); }

// sqrt in Fp2 = Fp[u]/(u^2+1):  a = (x + y u)^2  =>  x^2 = (a0 +- sqrt(a0^2 + a1^2))/2,  y = a1/(2x)
__device__ inline bool fp2_sqrt(Fp2& r, const Fp2& a) {
    if (FpF::is_zero(a.c0) && FpF::is_zero(a.c1)) {
        r = a;
        return true;
    }
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1)), s;
    if (!fp_sqrt(s, n)) return false;
    Fp inv2 = fp_inv(fp_dbl(FpF::one()));
    for (int k = 0; k < 2; k++) {
        Fp x2 = fp_mul(k == 0 ? fp_add(a.c0, s) : fp_sub(a.c0, s), inv2), x;
        if (!fp_sqrt(x, x2) || FpF::is_zero(x)) continue;
        Fp y = fp_mul(a.c1, fp_inv(fp_dbl(x)));
        Fp2 c{x, y};
        if (fp2_eq(fp2_sqr(c), a)) {
            r = c;
            return true;
        }
    }
    return false;
}

struct G2Aff {
    Fp2 x, y;  // Montgomery
};

// 96 compressed bytes (x.c1 || x.c0, flags in byte 0) -> affine; subgroup NOT checked
// (trusted-setup data, like build.rs:73).  Returns G1_OK / G1_INFINITY / G1_INVALID.
__device__ inline uint32_t g2_decompress(G2Aff& out, const uint8_t* b) {
    bool c_flag = (b[0] >> 7) & 1, i_flag = (b[0] >> 6) & 1, s_flag = (b[0] >> 5) & 1;
    uint8_t xb[96];
    for (int i = 0; i < 96; i++) xb[i] = b[i];
    xb[0] &= 0x1f;
    if (!c_flag) return G1_INVALID;
    uint32_t any = 0;
    for (int i = 0; i < 96; i++) any |= xb[i];
    if (i_flag) return (s_flag || any) ? G1_INVALID : G1_INFINITY;
    Fp x1 = FpF::from_be_bytes(xb), x0 = FpF::from_be_bytes(xb + 48);
    if (FpF::geq_mod(x0) || FpF::geq_mod(x1)) return G1_INVALID;
    Fp2 x{FpF::to_mont(x0), FpF::to_mont(x1)};
    Fp b4 = fp_const(consts::FP_B_MONT);
    Fp2 y2 = fp2_mul(fp2_sqr(x), x);
    y2.c0 = fp_add(y2.c0, b4);  // + 4(1 + u)
    y2.c1 = fp_add(y2.c1, b4);
    Fp2 y;
    if (!fp2_sqrt(y, y2)) return G1_INVALID;
    bool largest = FpF::is_zero(y.c1) ? fp_is_lex_largest(y.c0) : fp_is_lex_largest(y.c1);
    if (largest != s_flag) {
        y.c0 = fp_neg(y.c0);
        y.c1 = fp_neg(y.c1);
    }
    out.x = x;
    out.y = y;
    return G1_OK;
}

}  // namespace kzg
