// g1_29_formulas.hpp - the G1 point formulas over the radix-2^29 field of fp29.hpp: doubling, general and mixed addition
// with every special case, -phi.  Part of g1_29.hpp, kept apart because it is plain C++ (host + device): the exact code
// the kernels run is checked on the CPU against an integer model of the curve, with inputs at the documented bounds
// (tests/test_g1_29_host.py).  Same formulas as g1.hpp (dbl-2009-l, add-2007-bl, madd-2007-bl); every value is
// normalised (limbs < 2^29) and carries a static bound in multiples of p, noted beside each line.
#pragma once
#include "fp29.hpp"

namespace kzg {

struct G1Jac29 {
    Fp29 x, y, z;  // Jacobian, lazy values; z = 0 mod p <=> infinity
};
// an affine table entry (never the identity: the MSM skips flagged points before it looks at their entries)
struct G1Aff29 {
    Fp29 x, y;
};

FP29_FN G1Jac29 g1j29_identity() {
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
FP29_FN G1Jac29 g1j29_dbl(const G1Jac29& p) {
    G1Jac29 r;
    r.z = fp29_dbl(fp29_mul(p.y, p.z));                                 // < 4p; first, so that p.z and then p.y die early
    const Fp29 B = fp29_sqr(p.y), C = fp29_sqr(B), A = fp29_sqr(p.x);  // < 2p, limbs < 2^29
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
    r.x = fp29_normalize(X);                                            // < 130p
    r.y = fp29_sub<5>(fp29_mul(E, fp29_sub<9>(D, r.x)), C8);            // E (D + 512p - X3) + 32p - 8C < 34p
    return r;
}

// general addition with every special case (identity operands, P + P, P - P).  Inputs below 2^10 p;
// outputs X < 14p, Y < 6p, Z < 2p (or a dbl / operand passed through).
FP29_FN G1Jac29 g1j29_add(const G1Jac29& p, const G1Jac29& q) {
    const Fp29 Z1Z1 = fp29_sqr(p.z), Z2Z2 = fp29_sqr(q.z);
    if (fp29_is_zero_mod_p(Z1Z1)) return q;
    if (fp29_is_zero_mod_p(Z2Z2)) return p;
    // ordered so that every operand dies as early as it can
    const Fp29 U1 = fp29_mul(p.x, Z2Z2);
    const Fp29 H = fp29_sub<2>(fp29_mul(q.x, Z1Z1), U1);  // < 6p
    const Fp29 S1 = fp29_mul(fp29_mul(p.y, q.z), Z2Z2), S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    const Fp29 HH = fp29_sqr(H);
    if (fp29_is_zero_mod_p(HH)) {  // same x: P + P or P - P
        const Fp29 Rr = fp29_sub<2>(S2, S1);
        if (fp29_is_zero_mod_p(fp29_sqr(Rr))) return g1j29_dbl(p);
        return g1j29_identity();
    }
    G1Jac29 r;
    r.z = fp29_mul(fp29_mul(p.z, q.z), H);                                            // < 2p
    const Fp29 HHH = fp29_mul(H, HH), V = fp29_mul(U1, HH);
    const Fp29 Rr = fp29_sub<2>(S2, S1);                                              // < 6p
    r.x = fp29_sub<3>(fp29_sub<2>(fp29_sqr(Rr), HHH), fp29_dbl(V));                   // RR + 4p - HHH + 8p - 2V < 14p
    const Fp29 T = fp29_mul(S1, HHH);
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), T);                          // < 6p
    return r;
}

// mixed addition p + q, q affine and not the identity: 8M + 3S instead of 12M + 4S (madd-2007-bl without the
// doubling tricks).  p below X < 256p, Y < 256p, Z < 2^10 p (every output of dbl / add / this function is); q below 8p.
// Outputs X < 14p, Y < 6p, Z < 2p (or a dbl / q passed through).
FP29_FN G1Jac29 g1j29_add_affine(const G1Jac29& p, const G1Aff29& q) {
    const Fp29 Z1Z1 = fp29_sqr(p.z);
    if (fp29_is_zero_mod_p(Z1Z1)) {
        G1Jac29 r;
        r.x = q.x;
        r.y = q.y;
        r.z = fp29_const(cp29::FP29_ONE);
        return r;
    }
    // ordered so that every operand dies as early as it can (the bucket loop of the MSM window kernel is register-bound)
    const Fp29 U2 = fp29_mul(q.x, Z1Z1), S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    const Fp29 H = fp29_sub<9>(U2, p.x);  // < 514p
    const Fp29 HH = fp29_sqr(H);
    if (fp29_is_zero_mod_p(HH)) {  // same x: P + P or P - P
        const Fp29 Rr = fp29_sub<9>(S2, p.y);
        if (fp29_is_zero_mod_p(fp29_sqr(Rr))) return g1j29_dbl(p);
        return g1j29_identity();
    }
    G1Jac29 r;
    r.z = fp29_mul(p.z, H);                                                           // < 2p
    const Fp29 HHH = fp29_mul(H, HH), V = fp29_mul(p.x, HH);
    const Fp29 Rr = fp29_sub<9>(S2, p.y);                                             // < 514p
    r.x = fp29_sub<3>(fp29_sub<2>(fp29_sqr(Rr), HHH), fp29_dbl(V));                   // < 14p
    const Fp29 T = fp29_mul(p.y, HHH);
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), T);                          // < 6p
    return r;
}

// ---- the additions in two halves, for the MSM kernel's loops.  The complete formulas above return from five places; inlined
// into a loop the compiler merges those exits through memory (the bucket accumulator lived in scratch: 112 bytes loaded and
// stored per addition, 1.5 GB of HBM writes per launch group).  Here `head` computes everything the special cases are
// decided on, the caller branches (rare cases: out of the loop), and `tail` is straight-line code.  Same bounds as above.
struct G1MaddHead {
    Fp29 S2, H, HH;  // HH = 0 mod p: same x (P + P or P - P), or p at infinity - the complete formula must be used
};
FP29_FN G1MaddHead g1j29_madd_head(const G1Jac29& p, const G1Aff29& q) {
    G1MaddHead h;
    const Fp29 Z1Z1 = fp29_sqr(p.z);
    const Fp29 U2 = fp29_mul(q.x, Z1Z1);
    h.S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    h.H = fp29_sub<9>(U2, p.x);  // < 514p
    h.HH = fp29_sqr(h.H);
    return h;
}
FP29_FN bool g1j29_madd_special(const G1MaddHead& h) { return fp29_is_zero_mod_p(h.HH); }
// p + q for g1j29_madd_special(h) == false (p finite, different x): X < 14p, Y < 6p, Z < 2p
FP29_FN G1Jac29 g1j29_madd_tail(const G1Jac29& p, const G1MaddHead& h) {
    G1Jac29 r;
    r.z = fp29_mul(p.z, h.H);                                                         // < 2p
    const Fp29 HHH = fp29_mul(h.H, h.HH), V = fp29_mul(p.x, h.HH);
    const Fp29 Rr = fp29_sub<9>(h.S2, p.y);                                           // < 514p
    r.x = fp29_sub<3>(fp29_sub<2>(fp29_sqr(Rr), HHH), fp29_dbl(V));                   // < 14p
    const Fp29 T = fp29_mul(p.y, HHH);
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), T);                          // < 6p
    return r;
}
// General addition for the reduction trees, in a form whose later stages need NOTHING of the operands: the head leaves
// U1 = X1 Z2^2, S1 = Y1 Z2^3, S2, H, HH and ZZ = Z1 Z2; the tail works from those, and so does the rare same-x case -
// (U1, S1, ZZ) is the point p itself in another Jacobian representation (scaled by Z2), so P + P is its doubling.
// Caller's order: infinity flags first (pass the other operand through), then the head, then one of the two endings.
FP29_FN void g1j29_inf_flags(const G1Jac29& p, const G1Jac29& q, Fp29& Z1Z1, Fp29& Z2Z2, bool& p_inf, bool& q_inf) {
    Z1Z1 = fp29_sqr(p.z);
    Z2Z2 = fp29_sqr(q.z);
    p_inf = fp29_is_zero_mod_p(Z1Z1);
    q_inf = fp29_is_zero_mod_p(Z2Z2);
}
struct G1AddHead {
    Fp29 U1, S1, S2, H, HH, ZZ;
};
// p, q finite
FP29_FN G1AddHead g1j29_add_head(const G1Jac29& p, const G1Jac29& q, const Fp29& Z1Z1, const Fp29& Z2Z2) {
    G1AddHead h;
    h.ZZ = fp29_mul(p.z, q.z);
    h.U1 = fp29_mul(p.x, Z2Z2);
    h.H = fp29_sub<2>(fp29_mul(q.x, Z1Z1), h.U1);  // < 6p
    h.S1 = fp29_mul(fp29_mul(p.y, q.z), Z2Z2);
    h.S2 = fp29_mul(fp29_mul(q.y, p.z), Z1Z1);
    h.HH = fp29_sqr(h.H);
    return h;
}
FP29_FN bool g1j29_add_same_x(const G1AddHead& h) { return fp29_is_zero_mod_p(h.HH); }
// different x: X < 14p, Y < 6p, Z < 2p
FP29_FN G1Jac29 g1j29_add_tail(const G1AddHead& h) {
    G1Jac29 r;
    r.z = fp29_mul(h.ZZ, h.H);                                                        // < 2p
    const Fp29 HHH = fp29_mul(h.H, h.HH), V = fp29_mul(h.U1, h.HH);
    const Fp29 Rr = fp29_sub<2>(h.S2, h.S1);                                          // < 6p
    r.x = fp29_sub<3>(fp29_sub<2>(fp29_sqr(Rr), HHH), fp29_dbl(V));                   // < 14p
    const Fp29 T = fp29_mul(h.S1, HHH);
    r.y = fp29_sub<2>(fp29_mul(Rr, fp29_sub<5>(V, r.x)), T);                          // < 6p
    return r;
}
// same x: P + P (the doubling of p = (U1, S1, ZZ)) or P - P (the identity)
FP29_FN G1Jac29 g1j29_add_same_x_result(const G1AddHead& h) {
    const Fp29 Rr = fp29_sub<2>(h.S2, h.S1);
    if (!fp29_is_zero_mod_p(fp29_sqr(Rr))) return g1j29_identity();
    G1Jac29 t;
    t.x = h.U1;
    t.y = h.S1;
    t.z = h.ZZ;
    return g1j29_dbl(t);
}

// -phi(P) = (beta x, -y, z); y below 64p in, below 128p out
FP29_FN G1Jac29 g1j29_neg_phi(const G1Jac29& p) {
    G1Jac29 r;
    r.x = fp29_mul(p.x, fp29_const(cp29::FP29_BETA_MONT));
    r.y = fp29_neg<7>(p.y);
    r.z = p.z;
    return r;
}

}  // namespace kzg
