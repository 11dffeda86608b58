// Host build of kzg_rs_amd/csrc/fp29.hpp for tests/test_fp29_host.py (no GPU needed: the header is plain C++).
#include "fp29.hpp"
using namespace kzg;
static Fp29 ld(const uint32_t* p) { Fp29 r; for (int i = 0; i < 14; i++) r.l[i] = p[i]; return r; }
static void st(uint32_t* p, const Fp29& a) { for (int i = 0; i < 14; i++) p[i] = a.l[i]; }
extern "C" {
void h_fp29_mul(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fp29_mul(ld(a), ld(b))); }
void h_fp29_sqr(uint32_t* o, const uint32_t* a) { st(o, fp29_sqr(ld(a))); }
void h_fp29_add(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fp29_add(ld(a), ld(b))); }
void h_fp29_sub(uint32_t* o, const uint32_t* a, const uint32_t* b, int e) {
    Fp29 x = ld(a), y = ld(b), r;
    switch (e) {
        case 1: r = fp29_sub<1>(x, y); break; case 2: r = fp29_sub<2>(x, y); break; case 3: r = fp29_sub<3>(x, y); break;
        case 4: r = fp29_sub<4>(x, y); break; case 5: r = fp29_sub<5>(x, y); break; case 6: r = fp29_sub<6>(x, y); break;
        case 7: r = fp29_sub<7>(x, y); break; case 8: r = fp29_sub<8>(x, y); break; case 9: r = fp29_sub<9>(x, y); break;
        default: r = fp29_sub<10>(x, y); break;
    }
    st(o, r);
}
int h_fp29_is_zero_mod_p(const uint32_t* a) { return fp29_is_zero_mod_p(ld(a)) ? 1 : 0; }
void h_fp29_from_words(uint32_t* o, const uint32_t* w) { uint32_t t[12]; for (int i = 0; i < 12; i++) t[i] = w[i]; st(o, fp29_from_words(t)); }
void h_fp29_to_words(uint32_t* w, const uint32_t* a) { uint32_t t[12]; fp29_to_words(t, ld(a)); for (int i = 0; i < 12; i++) w[i] = t[i]; }
}
