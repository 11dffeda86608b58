// slp2.hpp - multi-wave LATENCY interpreter for the straight-line Fp programs of kzg_rs_amd/slp/schedule2.py (format
// documented there).  One workgroup of prog.lanes threads (several wavefronts) runs one program instance.
//
// The one-wave interpreter of slp.hpp is the throughput form (many pairing checks side by side, fewest products).  For
// ONE check its 3.2 ms are dependency depth: ~11 dependent additions between two product levels, each a full step.
// Here the same check is traced with schoolbook towers (2.2x the products, every product level fits the 192 lanes) and
// every step does more:
//   LIN : dst = +-x1 +- x2 +- x3 +- x4 + bias      radix-2^29 limbs, lazy (no reduction; the scheduler proved the
//                                                   bounds and chose the bias 2^e p), one carry pass
//   MUL : dst = (a1 + a2)(b1 + b2) 2^-406 mod p     fp29.hpp product: no carry instructions, output below 2p
//   LOAD: constants / instance inputs (converted from the 12x32 Montgomery form) / settings inputs (pre-converted)
// 1 444 steps instead of 5 755; a workgroup barrier per step (LDS traffic crosses wavefronts).
//
// Slots are 14 limbs in a 20-word (80-byte) stride: 16-byte aligned for ds_read_b128, and 20 s mod 64 walks all 16
// four-bank groups of the LDS (a 64-byte stride would use 4 of them: 4-way conflicts on every operand read).
// The step functions are plain C++ over fp29.hpp, so tests/host/slp2_host.cpp runs whole programs through the kernel's
// own arithmetic on the CPU.
#pragma once
#include "fp29.hpp"
#if defined(__HIPCC__)
#include "field.hpp"
#endif

namespace kzg {

constexpr uint32_t SLP2_MAGIC = 0x32504c53u;
constexpr int SLP2_SLOT_WORDS = 20;
constexpr int SLP2_GROUP = 8;  // descriptors are fetched a group of steps ahead
enum : uint32_t { SLP2_LIN = 0, SLP2_MUL = 1, SLP2_LOAD = 2 };
enum : uint32_t { SLP2_SRC_CONST = 0, SLP2_SRC_INST = 1, SLP2_SRC_SET = 2 };

struct Slp2Desc {
    uint32_t w0, w1, w2, w3;
};
struct Slp2Program {
    uint32_t lanes, n_slots, n_steps, n_const, n_in, n_set, n_out, n_load_steps;
    uint32_t out_values;        // 0: the outputs are written as "is it 0 mod p" flags; 1: as canonical 12x32 Montgomery values
    const uint32_t* consts;     // [n_const][16]
    const uint32_t* out_slots;  // [n_out]
    const Slp2Desc* desc;       // [n_steps][lanes]
};

// dst = +-x0 +- x1 +- x2 +- x3 + bias, carries propagated.  neg: bit i set = x_i is subtracted (at most three; then bias
// is 2^e p with limbs 0..12 boosted by 2^31, so no limb goes negative; without subtrahends bias is the zero slot).
// All x_i normalised (limbs 0..12 < 2^29).  (x ^ m) - m with m = 0 / ~0 is the conditional negation; the -m terms
// are collected into one addend.
FP29_FN Fp29 slp2_lin(const Fp29& x0, const Fp29& x1, const Fp29& x2, const Fp29& x3, const Fp29& bias, uint32_t neg) {
    const uint32_t m0 = 0u - (neg & 1u), m1 = 0u - ((neg >> 1) & 1u), m2 = 0u - ((neg >> 2) & 1u), m3 = 0u - ((neg >> 3) & 1u);
    const uint32_t c = (neg & 1u) + ((neg >> 1) & 1u) + ((neg >> 2) & 1u) + ((neg >> 3) & 1u);
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t t = (bias.l[i] + c) + (x0.l[i] ^ m0) + (x1.l[i] ^ m1);
        r.l[i] = t + (x2.l[i] ^ m2) + (x3.l[i] ^ m3);
    }
    return fp29_normalize(r);
}

// dst = (a1 + a2)(b1 + b2) 2^-406 mod p, below 2p.  Operands normalised; the left sum stays un-normalised (limbs < 2^30:
// a column of the product is then < 14 2^59 + 14 2^58 < 2^64), the right one is carried.
FP29_FN Fp29 slp2_mul(const Fp29& a1, const Fp29& a2, const Fp29& b1, const Fp29& b2) {
    Fp29 a;
#pragma unroll
    for (int i = 0; i < 14; i++) a.l[i] = a1.l[i] + a2.l[i];
    return fp29_mul(a, fp29_add(b1, b2));
}

// a result of the program (any lazy value) -> "is it 0 mod p": through one product with R'' mod p (value unchanged,
// range back below 2p)
FP29_FN bool slp2_is_zero(const Fp29& v) { return fp29_is_zero_mod_p(fp29_mul(v, fp29_const(cp29::FP29_ONE))); }

#if defined(__HIPCC__)
__device__ __forceinline__ Fp29 slp2_load(const uint32_t* slots, uint32_t s) {
    const uint32_t* q = slots + SLP2_SLOT_WORDS * s;
    const uint4 a = *reinterpret_cast<const uint4*>(q), b = *reinterpret_cast<const uint4*>(q + 4), c = *reinterpret_cast<const uint4*>(q + 8);
    const uint2 d = *reinterpret_cast<const uint2*>(q + 12);
    Fp29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void slp2_store(uint32_t* slots, uint32_t s, const Fp29& v) {
    uint32_t* q = slots + SLP2_SLOT_WORDS * s;
    *reinterpret_cast<uint4*>(q) = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    *reinterpret_cast<uint4*>(q + 4) = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    *reinterpret_cast<uint4*>(q + 8) = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    *reinterpret_cast<uint2*>(q + 12) = make_uint2(v.l[12], v.l[13]);
}

// one LIN / MUL step for one lane.  The kind is replicated into every descriptor: a wave-uniform branch.  No global memory
// access in here: the step loop then carries no vmcnt wait, and the descriptor prefetch really runs ahead.
__device__ __forceinline__ void slp2_exec(uint32_t* slots, const uint4 d) {
    const uint32_t kind = __builtin_amdgcn_readfirstlane(d.w >> 30);
    const bool active = (d.w >> 29) & 1u;
    const uint32_t dst = d.z & 0xffffu;
    if (kind == SLP2_MUL) {
        const Fp29 a1 = slp2_load(slots, d.x & 0xffffu), a2 = slp2_load(slots, d.x >> 16);
        const Fp29 b1 = slp2_load(slots, d.y & 0xffffu), b2 = slp2_load(slots, d.y >> 16);
        const Fp29 r = slp2_mul(a1, a2, b1, b2);
        if (active) slp2_store(slots, dst, r);
    } else {
        const Fp29 x0 = slp2_load(slots, d.x & 0xffffu), x1 = slp2_load(slots, d.x >> 16);
        const Fp29 x2 = slp2_load(slots, d.y & 0xffffu), x3 = slp2_load(slots, d.y >> 16);
        const Fp29 bias = slp2_load(slots, d.z >> 16);
        const Fp29 r = slp2_lin(x0, x1, x2, x3, bias, (d.w >> 16) & 15u);
        if (active) slp2_store(slots, dst, r);
    }
    __syncthreads();  // step boundary: the next step's reads may come from another wavefront's writes
}

// one LOAD step for one lane (the first prog.n_load_steps steps of a program)
__device__ __forceinline__ void slp2_exec_load(uint32_t* slots, const uint4 d, const Slp2Program& prog, const Fp* my_in,
                                               const uint32_t* __restrict__ settings_inputs) {
    if ((d.w >> 29) & 1u) {
        const uint32_t src = (d.w >> 26) & 7u, idx = d.w & 0xffffu;
        Fp29 r;
        if (src == SLP2_SRC_INST) {
            r = fp29_mul(fp29_from_words(my_in[idx].l), fp29_const(cp29::FP29_FROM_STD));  // 12x32 Montgomery -> x R''
        } else {
            const uint32_t* q = (src == SLP2_SRC_CONST ? prog.consts : settings_inputs) + 16 * (size_t)idx;
            const uint4 a = *reinterpret_cast<const uint4*>(q), b = *reinterpret_cast<const uint4*>(q + 4), c = *reinterpret_cast<const uint4*>(q + 8);
            const uint2 e = *reinterpret_cast<const uint2*>(q + 12);
            r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
            r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
            r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
            r.l[12] = e.x; r.l[13] = e.y;
        }
        slp2_store(slots, d.z & 0xffffu, r);
    }
}

// inputs: [instances][n_in] Fp (12x32 Montgomery, as the MSM leaves them); settings_inputs: [n_set][16] words (radix 2^29,
// Montgomery 2^406: k_fp_to_fp29mem); outputs: [instances][n_out] Fp, all-zero words where the program's output is
// 0 mod p and a 1 otherwise (the host only asks "all zero?") - or, for a program with out_values set, the values themselves
// as canonical 12x32 Montgomery elements: the instance-input format, so that a program feeds the next one (SCALARS -> VERIFY3).
// Descriptor streaming as in slp.hpp: a group of steps ahead into registers, parked in a per-lane LDS ring at the group
// boundary, read back one step ahead.  Dynamic LDS = slots | ring.
// (one workgroup per CU whatever happens - the LDS footprint; a whole SIMD's registers are there for the taking, and the
// prefetched descriptors must stay in them: spilled, the loads would be waited for at once)
template <int LANES>
__global__ __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_slp2_run(Slp2Program prog, const Fp* __restrict__ inputs, const uint32_t* __restrict__ settings_inputs,
                                                    Fp* __restrict__ outputs, uint32_t in_stride, uint32_t out_stride) {
    // in_stride / out_stride: elements between two instances' inputs / outputs (>= n_in / n_out: one program's outputs can land
    // inside the next one's input records)
    extern __shared__ __attribute__((aligned(16))) uint32_t slots2[];
    const uint32_t tid = threadIdx.x, inst = blockIdx.x, n_steps = prog.n_steps - prog.n_load_steps;
    uint4* ring = reinterpret_cast<uint4*>(slots2 + (size_t)SLP2_SLOT_WORDS * prog.n_slots);  // [SLP2_GROUP][LANES]
    const Fp* my_in = inputs + (size_t)inst * in_stride;
    if (tid < SLP2_SLOT_WORDS) slots2[tid] = 0u;  // slot 0: the constant zero
    // the LOAD prefix (a handful of steps; independent of one another: one barrier after the last)
    for (uint32_t st = 0; st < prog.n_load_steps; st++)
        slp2_exec_load(slots2, reinterpret_cast<const uint4*>(prog.desc)[(size_t)st * LANES + tid], prog, my_in, settings_inputs);
    const uint4* desc = reinterpret_cast<const uint4*>(prog.desc) + (size_t)prog.n_load_steps * LANES;  // the LIN / MUL steps
    const uint32_t last = n_steps - 1;
    // the first group's descriptors go straight into the ring; from then on a group is requested while the one before it
    // runs.  Named registers, not an array: the compiler keeps an indexed array of loaded values in scratch, and a
    // scratch store waits for its load on the spot - which is the opposite of a prefetch.
    static_assert(SLP2_GROUP == 8, "eight named prefetch registers below");
#define SLP2_DESC_AT(k) desc[(size_t)((base_next + (k)) < last ? (base_next + (k)) : last) * LANES + tid]
    {
        const uint32_t base_next = 0;
#pragma unroll
        for (int k = 0; k < SLP2_GROUP; k++) ring[(size_t)k * LANES + tid] = SLP2_DESC_AT(k);
    }
    __syncthreads();
    const uint32_t n_groups = (n_steps + SLP2_GROUP - 1) / SLP2_GROUP;
    for (uint32_t g = 0; g < n_groups; g++) {
        const uint32_t base = g * SLP2_GROUP, base_next = base + SLP2_GROUP;
        const uint4 r0 = SLP2_DESC_AT(0), r1 = SLP2_DESC_AT(1), r2 = SLP2_DESC_AT(2), r3 = SLP2_DESC_AT(3);
        const uint4 r4 = SLP2_DESC_AT(4), r5 = SLP2_DESC_AT(5), r6 = SLP2_DESC_AT(6), r7 = SLP2_DESC_AT(7);
        const uint4* cur = ring + tid;
        const uint32_t cnt = n_steps - base < (uint32_t)SLP2_GROUP ? n_steps - base : (uint32_t)SLP2_GROUP;
        uint4 d = cur[0];
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; k++) {
            const uint4 dn = cur[(size_t)(k + 1 < (uint32_t)SLP2_GROUP ? k + 1 : k) * LANES];
            slp2_exec(slots2, d);
            d = dn;
        }
        // every read of this group's ring entries has been issued (LDS is in order per wave; entries are per lane)
        ring[(size_t)0 * LANES + tid] = r0; ring[(size_t)1 * LANES + tid] = r1; ring[(size_t)2 * LANES + tid] = r2; ring[(size_t)3 * LANES + tid] = r3;
        ring[(size_t)4 * LANES + tid] = r4; ring[(size_t)5 * LANES + tid] = r5; ring[(size_t)6 * LANES + tid] = r6; ring[(size_t)7 * LANES + tid] = r7;
    }
#undef SLP2_DESC_AT
    for (uint32_t oi = tid; oi < prog.n_out; oi += LANES) {
        const Fp29 v = slp2_load(slots2, prog.out_slots[oi]);
        Fp o;
        if (prog.out_values) {
            fp29_to_words(o.l, fp29_mul(v, fp29_const(cp29::FP29_TO_STD)));  // x 2^384 mod p as a plain integer, below 2p
            o = FpF::reduce_once(o);
        } else {
            const bool z = slp2_is_zero(v);
#pragma unroll
            for (int i = 0; i < 12; i++) o.l[i] = 0u;
            o.l[0] = z ? 0u : 1u;
        }
        outputs[(size_t)inst * out_stride + oi] = o;
    }
}

// canonical 12x32 Montgomery elements -> the settings-input table of a latency program (16 words each)
__global__ void k_fp_to_fp29mem(const Fp* __restrict__ in, uint32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp29 v = fp29_mul(fp29_from_words(in[i].l), fp29_const(cp29::FP29_FROM_STD));
    uint32_t* q = out + 16 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 14; k++) q[k] = v.l[k];
    q[14] = q[15] = 0u;
}
#endif  // __HIPCC__

}  // namespace kzg
