// Host build of the latency interpreter's arithmetic (kzg_rs_amd/csrc/slp2.hpp): runs a whole schedule2.py program with the
// kernel's own step functions (slp2_lin / slp2_mul / slp2_is_zero over fp29.hpp), slots in an array, and checks the
// limb-level pre-conditions the kernel relies on at every step.  Test infrastructure (tests/test_slp2_host.py).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "slp2.hpp"

using namespace kzg;

static Fp29 get(const std::vector<uint32_t>& slots, uint32_t s) {
    Fp29 r;
    for (int i = 0; i < 14; i++) r.l[i] = slots[(size_t)SLP2_SLOT_WORDS * s + i];
    return r;
}
static bool normalised(const Fp29& v) {
    for (int i = 0; i < 13; i++)
        if (v.l[i] > FP29_MASK) return false;
    return true;
}

// inputs / settings_inputs: 16 words per element, ALREADY in the program's representation (radix 2^29, Montgomery 2^406,
// normalised) - the test converts with Python integers.  zero_out[i] = 1 where output i is 0 mod p.
// returns 0, or a negative code: -1 bad magic, -2 un-normalised operand, -3 write to slot 0 / slot out of range,
// -4 the limb-wise sum of a linear step left the 32-bit range
extern "C" int h_slp2_run(const uint32_t* blob, size_t n_words, const uint32_t* inputs, const uint32_t* settings_inputs, uint8_t* zero_out,
                          uint32_t* out_limbs /* n_out x 14, may be null */) {
    if (n_words < 16 || blob[0] != SLP2_MAGIC) return -1;
    const uint32_t lanes = blob[1], n_slots = blob[2], n_steps = blob[3], n_const = blob[4], n_out = blob[7];
    const uint32_t* consts = blob + 16;
    const uint32_t* outs = consts + 16 * (size_t)n_const;
    const uint32_t* desc = outs + ((n_out + 3) & ~3u);
    std::vector<uint32_t> slots((size_t)SLP2_SLOT_WORDS * n_slots, 0u);
    std::vector<uint8_t> written(n_slots, 0);
    written[0] = 1;
    std::vector<std::pair<uint32_t, Fp29>> pend;
    for (uint32_t s = 0; s < n_steps; s++) {
        pend.clear();
        for (uint32_t li = 0; li < lanes; li++) {
            const uint32_t* d = desc + 4 * ((size_t)s * lanes + li);
            const uint32_t kind = d[3] >> 30;
            if (!((d[3] >> 29) & 1u)) continue;
            const uint32_t dst = d[2] & 0xffffu, s0 = d[0] & 0xffffu, s1 = d[0] >> 16, s2 = d[1] & 0xffffu, s3 = d[1] >> 16;
            if (dst == 0 || dst >= n_slots) return -3;
            Fp29 r;
            if (kind == SLP2_LOAD) {
                const uint32_t src = (d[3] >> 26) & 7u, idx = d[3] & 0xffffu;
                const uint32_t* q = (src == SLP2_SRC_CONST ? consts : src == SLP2_SRC_INST ? inputs : settings_inputs) + 16 * (size_t)idx;
                for (int i = 0; i < 14; i++) r.l[i] = q[i];
            } else {
                for (uint32_t x : {s0, s1, s2, s3})
                    if (x >= n_slots || !written[x] || !normalised(get(slots, x))) return -2;
                if (kind == SLP2_MUL) {
                    r = slp2_mul(get(slots, s0), get(slots, s1), get(slots, s2), get(slots, s3));
                } else {
                    const uint32_t neg = (d[3] >> 16) & 15u, bs = d[2] >> 16;
                    if (bs >= n_slots || !written[bs]) return -2;
                    // exact check of the limb-wise range argument: every limb sum, as an integer, lies in [0, 2^32)
                    const Fp29 x[4] = {get(slots, s0), get(slots, s1), get(slots, s2), get(slots, s3)};
                    const Fp29 b = get(slots, bs);
                    for (int i = 0; i < 13; i++) {
                        int64_t t = b.l[i];
                        for (int k = 0; k < 4; k++) t += ((neg >> k) & 1u) ? -(int64_t)x[k].l[i] : (int64_t)x[k].l[i];
                        if (t < 0 || t >= ((int64_t)1 << 32)) return -4;
                    }
                    r = slp2_lin(x[0], x[1], x[2], x[3], b, neg);
                }
                if (!normalised(r)) return -2;
            }
            pend.emplace_back(dst, r);
        }
        for (auto& pr : pend) {
            for (int i = 0; i < 14; i++) slots[(size_t)SLP2_SLOT_WORDS * pr.first + i] = pr.second.l[i];
            written[pr.first] = 1;
        }
    }
    for (uint32_t o = 0; o < n_out; o++) {
        const Fp29 v = get(slots, outs[o]);
        zero_out[o] = slp2_is_zero(v) ? 1 : 0;
        if (out_limbs)
            for (int i = 0; i < 14; i++) out_limbs[14 * o + i] = v.l[i];
    }
    return 0;
}
