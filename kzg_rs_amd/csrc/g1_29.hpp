// g1_29.hpp - G1 point arithmetic over the radix-2^29 field of fp29.hpp (lazy reduction), for the point-decode and
// MSM kernels.  Same formulas as g1.hpp (dbl-2009-l, add-2007-bl); every value is normalised (limbs < 2^29) and carries
// a static bound in multiples of p, noted beside each line: products accept anything below ~6000 p and return < 2p,
// fp29_sub<E> adds 2^E p and needs its subtrahend <= 2^(E-1) p.  A coordinate is never compared or tested directly:
// "is it 0 mod p" is asked of the squares the formulas compute anyway (fp29_is_zero_mod_p on a product output).
#pragma once
#include "fp29.hpp"
#include "g1.hpp"
#include "g1_29_formulas.hpp"

namespace kzg {

// the table entry in global memory: coordinates padded to 64 bytes (b128 loads)
struct alignas(16) Fp29Mem {
    uint32_t l[16];
};
struct G1Jac29Mem {
    Fp29Mem x, y, z;
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

// A per-thread parking place in LDS for values that are idle across a long chain of doublings: the decode pass holds
// the point (x, y), the running sum q and the chain value r at once, the doubling needs ~120 registers of its own, and
// at 256 VGPRs the allocator spilled q, x and y to scratch inside the chains (4.4 KB of HBM writes per point).  LDS is
// there for the taking (the kernel uses none otherwise): 18 uint4 per thread, element e of thread t at [e * stride + t]
// (one ds_write_b128 per element, conflict-free).
// The pointer carries the LDS address space in its TYPE.  As a plain (generic) pointer it survived only while everything
// was inlined: the out-of-line chain function got it as a 64-bit flat address, every park access became a flat_load /
// flat_store, and the counters showed them going through the L1 to the L2 like global accesses (TCP_TCC_WRITE_REQ 113 M
// per launch group, ~0.8 GB of it written on to HBM as the parked lines were evicted) - the "LDS parking" was parked
// in memory.
typedef uint32_t ParkVec __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ParkVec LdsParkVec;
struct LdsPark {
    LdsParkVec* base;  // already offset by the thread index
    unsigned stride;   // threads per workgroup
};
// from a pointer into a __shared__ array (the low 32 bits of its generic address are the LDS offset)
__device__ __forceinline__ LdsPark lds_park(const uint4* shared_ptr, unsigned stride) {
    return LdsPark{reinterpret_cast<LdsParkVec*>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(shared_ptr))), stride};
}
constexpr int PARK_UINT4_PER_THREAD = 18;  // (x, y): 7, a Jacobian point: 11
template <int NWORDS>
__device__ __forceinline__ void park_store(const LdsPark& pk, int at, const uint32_t (&w)[NWORDS]) {
#pragma unroll
    for (int e = 0; e < (NWORDS + 3) / 4; e++) {
        ParkVec v;
        v.x = w[4 * e];
        v.y = 4 * e + 1 < NWORDS ? w[4 * e + 1] : 0u;
        v.z = 4 * e + 2 < NWORDS ? w[4 * e + 2] : 0u;
        v.w = 4 * e + 3 < NWORDS ? w[4 * e + 3] : 0u;
        pk.base[(at + e) * pk.stride] = v;
    }
}
template <int NWORDS>
__device__ __forceinline__ void park_load(const LdsPark& pk, int at, uint32_t (&w)[NWORDS]) {
#pragma unroll
    for (int e = 0; e < (NWORDS + 3) / 4; e++) {
        const ParkVec v = pk.base[(at + e) * pk.stride];
        w[4 * e] = v.x;
        if (4 * e + 1 < NWORDS) w[4 * e + 1] = v.y;
        if (4 * e + 2 < NWORDS) w[4 * e + 2] = v.z;
        if (4 * e + 3 < NWORDS) w[4 * e + 3] = v.w;
    }
}
__device__ __forceinline__ void park_xy(const LdsPark& pk, const Fp29& x, const Fp29& y) {
    uint32_t w[28];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        w[i] = x.l[i];
        w[14 + i] = y.l[i];
    }
    park_store<28>(pk, 0, w);
}
__device__ __forceinline__ void unpark_xy(const LdsPark& pk, Fp29& x, Fp29& y) {
    uint32_t w[28];
    park_load<28>(pk, 0, w);
#pragma unroll
    for (int i = 0; i < 14; i++) {
        x.l[i] = w[i];
        y.l[i] = w[14 + i];
    }
}
__device__ __forceinline__ void park_point(const LdsPark& pk, const G1Jac29& p) {
    uint32_t w[42];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        w[i] = p.x.l[i];
        w[14 + i] = p.y.l[i];
        w[28 + i] = p.z.l[i];
    }
    park_store<42>(pk, 7, w);
}
__device__ __forceinline__ G1Jac29 unpark_point(const LdsPark& pk) {
    uint32_t w[42];
    park_load<42>(pk, 7, w);
    G1Jac29 p;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        p.x.l[i] = w[i];
        p.y.l[i] = w[14 + i];
        p.z.l[i] = w[28 + i];
    }
    return p;
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
// the same with P parked (unpark_point(pk)): it is needed five times in 63 iterations
// (forced inline: as a function of its own it returned the point through memory and kept the accumulator THERE - 168 bytes
// of scratch stores per iteration, the bulk of the decode pass's WRITE_SIZE)
__device__ __forceinline__ G1Jac29 g1j29_mul_xabs_parked(const LdsPark& pk) {
    G1Jac29 acc = unpark_point(pk);
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        acc = g1j29_dbl(acc);
        if ((BLS_X_ABS >> i) & 1) acc = g1j29_add(acc, unpark_point(pk));
    }
    return acc;
}

// The subgroup test of g1.hpp (phi(P) = -[x^2]P) with the multiples 2^(STEP k) P emitted on the way, in this field.
// x, y: the affine point (x R'', y R'', below 2p); they are parked in pk on return (unpark_xy), as is nothing else.
template <int STEP, class Emit>
__device__ inline bool g1j29_in_subgroup_with_multiples(const Fp29& x, const Fp29& y, const LdsPark& pk, Emit emit) {
    G1Jac29 r;
    r.x = x;
    r.y = y;
    r.z = fp29_const(cp29::FP29_ONE);
    park_xy(pk, x, y);
    // q = [|x|]P accumulated right to left lives in the parking place: it is touched at the six set bits of |x| only
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        if (i && i % STEP == 0) emit(i / STEP, r);
        if ((BLS_X_ABS >> i) & 1) {
            if (i == 16) park_point(pk, r);  // the lowest set bit of |x|
            else park_point(pk, g1j29_add(unpark_point(pk), r));
        }
        r = g1j29_dbl(r);
    }
#pragma unroll 1
    for (int i = 64; i < 128; i++) {
        if (i % STEP == 0) emit(i / STEP, r);
        if (i + STEP >= 128) break;
        r = g1j29_dbl(r);
    }
    const G1Jac29 q = g1j29_mul_xabs_parked(pk);
    const Fp29 zz = fp29_sqr(q.z);
    if (fp29_is_zero_mod_p(zz)) return false;
    Fp29 xx, yy;
    unpark_xy(pk, xx, yy);
    const Fp29 zzz = fp29_mul(zz, q.z), one = fp29_const(cp29::FP29_ONE);
    const Fp29 bx = fp29_mul(xx, fp29_const(cp29::FP29_BETA_MONT));
    // phi(P) == -q  <=>  X = beta x Z^2  and  Y + y Z^3 = 0; differences go through one more product to be testable
    const Fp29 dx = fp29_mul(fp29_sub<2>(q.x, fp29_mul(bx, zz)), one);
    const Fp29 dy = fp29_mul(fp29_add(q.y, fp29_mul(yy, zzz)), one);
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
// The same power with the window table's upper half (a^5, a^7) in the parking place, elements [at, at + 8): the table, the
// accumulator and the caller's x and y^2 on top of a squaring's own ~150 registers do not fit 248, and what the allocator
// then moved to scratch was the accumulator - a 56-byte store and load per squaring, 2 KB of it per point reaching HBM
// (that, not the table rows, was the decode pass's WRITE_SIZE).
__device__ inline Fp29 fp29_sqrt_candidate_parked(const Fp29& a, const LdsPark& pk, int at) {
    const uint32_t (&e)[12] = consts::FP_SQRT_EXP;
    const Fp29 a2 = fp29_sqr(a);
    const Fp29 t1 = fp29_mul(a, a2);
    {
        const Fp29 t2 = fp29_mul(t1, a2);
        park_store<14>(pk, at, t2.l);
        const Fp29 t3 = fp29_mul(t2, a2);
        park_store<14>(pk, at + 4, t3.l);
    }
    Fp29 acc = fp29_const(cp29::FP29_ONE);
    bool started = false;
    int i = 378;  // (p+1)/4 has 379 bits
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
        const uint32_t idx = v >> 1;  // uniform: the exponent is a constant
        Fp29 m;
        if (idx >= 2) {
            park_load<14>(pk, at + 4 * (int)(idx - 2), m.l);
        } else {
#pragma unroll
            for (int w = 0; w < 14; w++) m.l[w] = idx == 0 ? a.l[w] : t1.l[w];
        }
        acc = started ? fp29_mul(acc, m) : m;
        started = true;
        i = j - 1;
    }
    return acc;
}
__device__ inline Fp29 fp29_inverse(const Fp29& a) { return fp29_pow_window3(a, consts::FP_P_MINUS_2, 380); }       // p - 2 has 381 bits; a != 0 mod p

// 48 compressed bytes -> affine point in this field (x R'', y R'' below 2p... y possibly 4p - y).  Subgroup NOT checked.
// Returns G1_OK / G1_INFINITY / G1_INVALID exactly like g1_decompress.
// pk: the parking place (all of it is free here): x and y^2 wait there while the square root is taken.
__device__ inline uint32_t g1_decompress29(Fp29& xo, Fp29& yo, const uint8_t* b, const LdsPark& pk) {
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
    Fp29 x = fp29_mul(fp29_from_words(xs.l), fp29_const(cp29::FP29_R2));           // x R'', < 2p
    Fp29 y2 = fp29_add(fp29_mul(fp29_sqr(x), x), fp29_const(cp29::FP29_B_MONT));    // < 3p
    park_xy(pk, x, y2);                                                             // elements [0, 7)
    Fp29 y = fp29_sqrt_candidate_parked(y2, pk, 7);                                 // < 2p
    unpark_xy(pk, x, y2);
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
