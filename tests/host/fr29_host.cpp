// Host build of kzg_rs_amd/csrc/fr29.hpp for tests/test_fr29_host.py (no GPU needed: the header is plain C++).
#include "fr29.hpp"
using namespace kzg;
static Fr29 ld(const uint32_t* p) { Fr29 r; for (int i = 0; i < 9; i++) r.l[i] = p[i]; return r; }
static void st(uint32_t* p, const Fr29& a) { for (int i = 0; i < 9; i++) p[i] = a.l[i]; }
extern "C" {
void h_fr29_mul(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fr29_mul(ld(a), ld(b))); }
void h_fr29_mul2(uint32_t* o, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) { st(o, fr29_mul2(ld(a), ld(b), ld(c), ld(d))); }
void h_fr29_sub_biased4(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fr29_sub_biased4(ld(a), ld(b))); }
void h_fr29_add(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fr29_add(ld(a), ld(b))); }
void h_fr29_sub_biased(uint32_t* o, const uint32_t* a, const uint32_t* b) { st(o, fr29_sub_biased(ld(a), ld(b))); }
void h_fr29_normalize(uint32_t* o, const uint32_t* a) { st(o, fr29_normalize(ld(a))); }
void h_fr29_from_words(uint32_t* o, const uint32_t* w) { uint32_t t[8]; for (int i = 0; i < 8; i++) t[i] = w[i]; st(o, fr29_from_words(t)); }
void h_fr29_to_words(uint32_t* w, const uint32_t* a) { uint32_t t[8]; fr29_to_words(t, ld(a)); for (int i = 0; i < 8; i++) w[i] = t[i]; }
}
