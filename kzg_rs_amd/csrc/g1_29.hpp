// g1_29.hpp - G1 point arithmetic over the radix-2^29 field of fp29.hpp (lazy reduction), for the point-decode and
// MSM kernels.  Same formulas as g1.hpp (dbl-2009-l, add-2007-bl); every value is normalised (limbs < 2^29) and carries
// a static bound in multiples of p, noted beside each line: products accept anything below ~6000 p and return < 2p,
// fp29_sub<E> adds 2^E p and needs its subtrahend <= 2^(E-1) p.  A coordinate is never compared or tested directly:
// "is it 0 mod p" is asked of the squares the formulas compute anyway (fp29_is_zero_mod_p on a product output).
#pragma once
#include "fp29.hpp"
#include "g1.hpp"

namespace kzg {

struct G1Jac29 {
    Fp29 x, y, z;  // Jacobian, lazy values; z = 0 mod p <=> infinity
};
// the table entry in global memory: coordinates padded to 64 bytes (b128 loads)
struct alignas(16) Fp29Mem {
    uint32_t l[16];
};
struct G1Jac29Mem {
    Fp29Mem x, y, z;
};
// an affine table entry (never the identity: the MSM skips flagged points before it looks at their entries)
struct G1Aff29 {
    Fp29 x, y;
};
struct G1Aff29Mem {
    Fp29Mem x, y;
};

__device__ __forceinline__ Fp29 fp29_load(const Fp29Mem& m) {
    const uint4* p = reinterpret_cast<const uint4*>(m.l);
    const uint4 a = p[0], b = p[1], c = p[2];
    const uint2 d = *reinterpret_cast<const uint2*>(m.l + 12);
    Fp29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void fp29_store(Fp29Mem& m, const Fp29& v) {
    uint4* p = reinterpret_cast<uint4*>(m.l);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    p[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
__device__ __forceinline__ G1Jac29 g1j29_load(const G1Jac29Mem& m) {
    G1Jac29 r;
    r.x = fp29_load(m.x);
    r.y = fp29_load(m.y);
    r.z = fp29_load(m.z);
    return r;
}
__device__ __forceinline__ void g1j29_store(G1Jac29Mem& m, const G1Jac29& p) {
    fp29_store(m.x, p.x);
    fp29_store(m.y, p.y);
    fp29_store(m.z, p.z);
}
__device__ __forceinline__ G1Aff29 g1a29_load(const G1Aff29Mem& m) {
    G1Aff29 r;
    r.x = fp29_load(m.x);
    r.y = fp29_load(m.y);
    return r;
}
__device__ __forceinline__ void g1a29_store(G1Aff29Mem& m, const Fp29& x, const Fp29& y) {
    fp29_store(m.x, x);
    fp29_store(m.y, y);
}

// ---- conversions to / from the 12x32 Montgomery form (radix 2^384) of field.hpp
// canonical 12x32 Montgomery element -> x R'' (below 2p)
__device__ __forceinline__ Fp29 fp29_from_std(const Fp& a) { return fp29_mul(fp29_from_words(a.l), fp29_const(cp29::FP29_FROM_STD)); }
// any lazy value -> canonical 12x32 Montgomery element
__device__ __forceinline__ Fp fp29_to_std(const Fp29& a) {
    const Fp29 t = fp29_mul(a, fp29_const(cp29::FP29_TO_STD));  // x 2^384 mod p as a plain integer, below 2p
    Fp r;
    fp29_to_words(r.l, t);
    return FpF::reduce_once(r);
}
__device__ __forceinline__ G1Jac29 g1j29_from_std(const G1Jac& p) {
    G1Jac29 r;
    r.x = fp29_from_std(p.x);
    r.y = fp29_from_std(p.y);
    r.z = fp29_from_std(p.z);
    return r;
}
__device__ __forceinline__ G1Jac g1j29_to_std(const G1Jac29& p) {
    G1Jac r;
    r.z = fp29_to_std(p.z);
    if (FpF::is_zero(r.z)) return g1_identity();
    r.x = fp29_to_std(p.x);
    r.y = fp29_to_std(p.y);
    return r;
}

__device__ __forceinline__ G1Jac29 g1j29_identity() {
    G1Jac29 r;
    r.x = fp29_zero();
    r.y = fp29_const(cp29::FP29_ONE);
    r.z = fp29_zero();
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S.  Inputs below 2^10 p; outputs X < 130p, Y < 34p, Z < 4p.  Z = 0 mod p stays so.
// The linear steps are taken limb-wise on un-normalised words and carried ONCE per result (8 carry passes instead of
// 14): every intermediate word stays below 2^32 by the bounds noted, the biases are 16p / 128p with limbs boosted by
// 2^31 / 2^30 so that no limb borrows (tools/gen_constants.py).
__device__ __forceinline__ G1Jac29 g1j29_dbl(const G1Jac29& p) {
    const Fp29 A = fp29_sqr(p.x), B = fp29_sqr(p.y), C = fp29_sqr(B);  // < 2p, limbs < 2^29
    const Fp29 t = fp29_sqr(fp29_add(p.x, B));                          // < 2p
    Fp29 D, E, X, C8;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        D.l[i] = ((t.l[i] - A.l[i] - C.l[i]) << 1) + cp29::FP29_BIASX4[i];  // 2t + 16p - 2A - 2C: words < 2^30 + 2^29 + 2^31
        E.l[i] = (A.l[i] << 1) + A.l[i];                                   // 3A: words < 2^31
        C8.l[i] = C.l[i] << 3;                                             // 8C: words < 2^32
    }
    D = fp29_normalize(D);    // < 20p
    E = fp29_normalize(E);    // < 6p
    C8 = fp29_normalize(C8);  // < 16p
    const Fp29 F = fp29_sqr(E);  // < 2p
#pragma unroll
    for (int i = 0; i < 14; i++) X.l[i] = F.l[i] + cp29::FP29_BIASW7[i] - (D.l[i] << 1);  // F + 128p - 2D: words < 2^31
    G1Jac29 r;
    r.x = fp29_normalize(X);                                            // < 130p
    r.z = fp29_dbl(fp29_mul(p.y, p.z));                                 // < 4p
    r.y = fp29_sub<5>(fp29_mul(E, fp29_sub<9>(D, r.x)), C8);            // E (D + 512p - X3) + 32p - 8C < 34p
    return r;
}

// general addition with every special case (identity operands, P + P, P - P).  Inputs below 2^10 p;
// outputs X < 14p, Y < 6p, Z < 2p (or a dbl / operand passed through).
__device__ __forceinline__ G1Jac29 g1j29_add(const G1Jac29& p, const G1Jac29& q) {
    const Fp29 Z1Z1 = fp29_sqr(p.z), Z2Z2 = fp29_sqr(q.z);
    if (fp29_is_zero_mod_p(Z1Z1)) return q;
    if (fp29_is_zero_mod_p(Z2Z2)) return p;
    const Fp29 U1 = fp29_mul(p.x, Z2Z2), U2 = fp29_mul(q.x, Z1Z1);
    const Fp29 S1 = fp29_mul(fp29_mul(p.y, q.z), Z2Z2), S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    const Fp29 H = fp29_sub<2>(U2, U1), Rr = fp29_sub<2>(S2, S1);  // < 6p
    const Fp29 HH = fp29_sqr(H), RR = fp29_sqr(Rr);
    if (fp29_is_zero_mod_p(HH)) {
        if (fp29_is_zero_mod_p(RR)) return g1j29_dbl(p);
        return g1j29_identity();
    }
    const Fp29 HHH = fp29_mul(H, HH), V = fp29_mul(U1, HH);
    G1Jac29 r;
    r.x = fp29_sub<3>(fp29_sub<2>(RR, HHH), fp29_dbl(V));                             // RR + 4p - HHH + 8p - 2V < 14p
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), fp29_mul(S1, HHH));          // < 6p
    r.z = fp29_mul(fp29_mul(p.z, q.z), H);                                            // < 2p
    return r;
}

// mixed addition p + q, q affine and not the identity: 8M + 3S instead of 12M + 4S (madd-2007-bl without the
// doubling tricks).  p below X < 256p, Y < 256p, Z < 2^10 p (every output of dbl / add / this function is); q below 8p.
// Outputs X < 14p, Y < 6p, Z < 2p (or a dbl / q passed through).
__device__ __forceinline__ G1Jac29 g1j29_add_affine(const G1Jac29& p, const G1Aff29& q) {
    const Fp29 Z1Z1 = fp29_sqr(p.z);
    if (fp29_is_zero_mod_p(Z1Z1)) {
        G1Jac29 r;
        r.x = q.x;
        r.y = q.y;
        r.z = fp29_const(cp29::FP29_ONE);
        return r;
    }
    const Fp29 U2 = fp29_mul(q.x, Z1Z1), S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    const Fp29 H = fp29_sub<9>(U2, p.x), Rr = fp29_sub<9>(S2, p.y);  // < 514p
    const Fp29 HH = fp29_sqr(H), RR = fp29_sqr(Rr);
    if (fp29_is_zero_mod_p(HH)) {
        if (fp29_is_zero_mod_p(RR)) return g1j29_dbl(p);
        return g1j29_identity();
    }
    const Fp29 HHH = fp29_mul(H, HH), V = fp29_mul(p.x, HH);
    G1Jac29 r;
    r.x = fp29_sub<3>(fp29_sub<2>(RR, HHH), fp29_dbl(V));                             // < 14p
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), fp29_mul(p.y, HHH));         // < 6p
    r.z = fp29_mul(p.z, H);                                                           // < 2p
    return r;
}

// -phi(P) = (beta x, -y, z); y below 64p in, below 128p out
__device__ __forceinline__ G1Jac29 g1j29_neg_phi(const G1Jac29& p) {
    G1Jac29 r;
    r.x = fp29_mul(p.x, fp29_const(cp29::FP29_BETA_MONT));
    r.y = fp29_neg<7>(p.y);
    r.z = p.z;
    return r;
}

// [|x|]P left to right (63 doublings + 5 additions)
__device__ inline G1Jac29 g1j29_mul_xabs(const G1Jac29& p) {
    G1Jac29 acc = p;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        acc = g1j29_dbl(acc);
        if ((BLS_X_ABS >> i) & 1) acc = g1j29_add(acc, p);
    }
    return acc;
}

// The subgroup test of g1.hpp (phi(P) = -[x^2]P) with the multiples 2^(STEP k) P emitted on the way, in this field.
// x, y: the affine point (x R'', y R'', below 2p).
template <int STEP, class Emit>
__device__ inline bool g1j29_in_subgroup_with_multiples(const Fp29& x, const Fp29& y, Emit emit) {
    G1Jac29 r;
    r.x = x;
    r.y = y;
    r.z = fp29_const(cp29::FP29_ONE);
    G1Jac29 q = r;  // overwritten at bit 16, the lowest set bit of |x|
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        if (i && i % STEP == 0) emit(i / STEP, r);
        if ((BLS_X_ABS >> i) & 1) q = (i == 16) ? r : g1j29_add(q, r);
        r = g1j29_dbl(r);
    }
#pragma unroll 1
    for (int i = 64; i < 128; i++) {
        if (i % STEP == 0) emit(i / STEP, r);
        if (i + STEP >= 128) break;
        r = g1j29_dbl(r);
    }
    q = g1j29_mul_xabs(q);
    const Fp29 zz = fp29_sqr(q.z);
    if (fp29_is_zero_mod_p(zz)) return false;
    const Fp29 zzz = fp29_mul(zz, q.z), one = fp29_const(cp29::FP29_ONE);
    const Fp29 bx = fp29_mul(x, fp29_const(cp29::FP29_BETA_MONT));
    // phi(P) == -q  <=>  X = beta x Z^2  and  Y + y Z^3 = 0; differences go through one more product to be testable
    const Fp29 dx = fp29_mul(fp29_sub<2>(q.x, fp29_mul(bx, zz)), one);
    const Fp29 dy = fp29_mul(fp29_add(q.y, fp29_mul(y, zzz)), one);
    return fp29_is_zero_mod_p(dx) && fp29_is_zero_mod_p(dy);
}

// a^e (e: little-endian 32-bit words, `top` = index of its highest set bit) with the 3-bit sliding window of g1.hpp's
// fp_pow_window3; a below 8p
__device__ inline Fp29 fp29_pow_window3(const Fp29& a, const uint32_t (&e)[12], int top) {
    const Fp29 a2 = fp29_sqr(a);
    Fp29 t[4];
    t[0] = a;
    for (int k = 1; k < 4; k++) t[k] = fp29_mul(t[k - 1], a2);
    Fp29 acc = fp29_const(cp29::FP29_ONE);
    bool started = false;
    int i = top;
    auto bit = [&](int k) { return (e[k >> 5] >> (k & 31)) & 1u; };
    while (i >= 0) {
        if (!bit(i)) {
            if (started) acc = fp29_sqr(acc);
            i--;
            continue;
        }
        int j = i - 2 < 0 ? 0 : i - 2;
        while (!bit(j)) j++;
        uint32_t v = 0;
        for (int k = i; k >= j; k--) v = (v << 1) | bit(k);
        if (started)
            for (int k = i; k >= j; k--) acc = fp29_sqr(acc);
        const uint32_t idx = v >> 1;
        Fp29 m;
#pragma unroll
        for (int w = 0; w < 14; w++) m.l[w] = idx == 0 ? t[0].l[w] : idx == 1 ? t[1].l[w] : idx == 2 ? t[2].l[w] : t[3].l[w];
        acc = started ? fp29_mul(acc, m) : m;
        started = true;
        i = j - 1;
    }
    return acc;
}
__device__ inline Fp29 fp29_sqrt_candidate(const Fp29& a) { return fp29_pow_window3(a, consts::FP_SQRT_EXP, 378); }  // (p+1)/4 has 379 bits
__device__ inline Fp29 fp29_inverse(const Fp29& a) { return fp29_pow_window3(a, consts::FP_P_MINUS_2, 380); }       // p - 2 has 381 bits; a != 0 mod p

// 48 compressed bytes -> affine point in this field (x R'', y R'' below 2p... y possibly 4p - y).  Subgroup NOT checked.
// Returns G1_OK / G1_INFINITY / G1_INVALID exactly like g1_decompress.
__device__ inline uint32_t g1_decompress29(Fp29& xo, Fp29& yo, const uint8_t* b) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint8_t* p = b + 4 * (11 - i);
        w[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
    }
    const bool c_flag = (w[11] >> 31) & 1, i_flag = (w[11] >> 30) & 1, s_flag = (w[11] >> 29) & 1;
    w[11] &= 0x1fffffffu;
    if (!c_flag) return G1_INVALID;
    Fp xs;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        xs.l[i] = w[i];
        any |= w[i];
    }
    if (i_flag) return (s_flag || any) ? G1_INVALID : G1_INFINITY;
    if (FpF::geq_mod(xs)) return G1_INVALID;
    const Fp29 x = fp29_mul(fp29_from_words(xs.l), fp29_const(cp29::FP29_R2));           // x R'', < 2p
    const Fp29 y2 = fp29_add(fp29_mul(fp29_sqr(x), x), fp29_const(cp29::FP29_B_MONT));    // < 3p
    Fp29 y = fp29_sqrt_candidate(y2);                                                     // < 2p
    const Fp29 one = fp29_const(cp29::FP29_ONE);
    if (!fp29_is_zero_mod_p(fp29_mul(fp29_sub<3>(fp29_sqr(y), y2), one))) return G1_INVALID;  // not a square
    // sign: "lexicographically largest" on the plain canonical value of y
    Fp29 plain_one = fp29_zero();
    plain_one.l[0] = 1;
    Fp yp;
    fp29_to_words(yp.l, fp29_mul(y, plain_one));  // y as a plain integer below 2p
    yp = FpF::reduce_once(yp);
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) (void)subb(consts::FP_HALF[i], yp.l[i], borrow);  // half - y < 0  <=>  y > half
    if ((borrow != 0) != s_flag) y = fp29_neg<2>(y);                               // 4p - y
    xo = x;
    yo = y;
    return G1_OK;
}

}  // namespace kzg
