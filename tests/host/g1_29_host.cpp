// Host build of kzg_rs_amd/csrc/g1_29_formulas.hpp for tests/test_g1_29_host.py (no GPU needed: the header is plain C++).
// A point crosses the boundary as 42 words: x, y, z with 14 limbs each (28 words for an affine one).
#include "g1_29_formulas.hpp"
using namespace kzg;
static Fp29 ld(const uint32_t* p) { Fp29 r; for (int i = 0; i < 14; i++) r.l[i] = p[i]; return r; }
static void st(uint32_t* p, const Fp29& a) { for (int i = 0; i < 14; i++) p[i] = a.l[i]; }
static G1Jac29 ldj(const uint32_t* p) { G1Jac29 r; r.x = ld(p); r.y = ld(p + 14); r.z = ld(p + 28); return r; }
static void stj(uint32_t* p, const G1Jac29& a) { st(p, a.x); st(p + 14, a.y); st(p + 28, a.z); }
extern "C" {
void h_g1_dbl(uint32_t* o, const uint32_t* p) { stj(o, g1j29_dbl(ldj(p))); }
void h_g1_add(uint32_t* o, const uint32_t* p, const uint32_t* q) { stj(o, g1j29_add(ldj(p), ldj(q))); }
void h_g1_add_affine(uint32_t* o, const uint32_t* p, const uint32_t* q) {
    G1Aff29 a; a.x = ld(q); a.y = ld(q + 14);
    stj(o, g1j29_add_affine(ldj(p), a));
}
// the staged forms (what the MSM window kernel's loops run).  h_g1_add_split: always the complete result; returns 1 for the
// same-x ending, 2 when an identity operand was passed through.  h_g1_madd_split: returns 1 when the caller must use the
// complete formula (nothing is written then)
int h_g1_add_split(uint32_t* o, const uint32_t* pp, const uint32_t* qq) {
    const G1Jac29 p = ldj(pp), q = ldj(qq);
    Fp29 Z1Z1, Z2Z2;
    bool p_inf, q_inf;
    g1j29_inf_flags(p, q, Z1Z1, Z2Z2, p_inf, q_inf);
    if (p_inf || q_inf) { stj(o, p_inf ? q : p); return 2; }
    const G1AddHead h = g1j29_add_head(p, q, Z1Z1, Z2Z2);
    if (g1j29_add_same_x(h)) { stj(o, g1j29_add_same_x_result(h)); return 1; }
    stj(o, g1j29_add_tail(h));
    return 0;
}
int h_g1_madd_split(uint32_t* o, const uint32_t* pp, const uint32_t* qq) {
    const G1Jac29 p = ldj(pp);
    G1Aff29 a; a.x = ld(qq); a.y = ld(qq + 14);
    const G1MaddHead h = g1j29_madd_head(p, a);
    if (g1j29_madd_special(h)) return 1;
    stj(o, g1j29_madd_tail(p, h));
    return 0;
}
void h_g1_neg_phi(uint32_t* o, const uint32_t* p) { stj(o, g1j29_neg_phi(ldj(p))); }
void h_g1_identity(uint32_t* o) { stj(o, g1j29_identity()); }
}
