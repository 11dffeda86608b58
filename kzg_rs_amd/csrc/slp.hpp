// slp.hpp - LDS-resident interpreter for the straight-line Fp programs emitted by
// kzg_rs_amd/slp/schedule.py (format documented there).  One workgroup of `lanes` threads runs
// one program instance; blockIdx.x selects the instance (many independent pairing checks can run
// side by side).  Every step each lane performs one Fp operation on 48-byte LDS slots; a
// value's slot is never rewritten in the step in which it is read, so one barrier per step
// suffices.
//
// This is how the pairing (reference src/pairings.rs:5-9: multi_miller_loop +
// final_exponentiation) becomes a CDNA4 kernel: its ~21 000 Fp products have a dependency depth
// of ~480, so they run as ~480 wave-wide multiply steps instead of a 30 ms single-lane chain.
#pragma once
#include "field.hpp"

namespace kzg {

enum : uint32_t { SLP_NOP = 0, SLP_ADD = 1, SLP_SUB = 2, SLP_MUL = 3, SLP_LOADC = 4, SLP_LOADI = 5, SLP_LOADS = 6 };
constexpr uint32_t SLP_MAGIC = 0x31504c53u;

struct SlpProgram {
    uint32_t lanes, n_slots, n_steps, n_const, n_in, n_set, n_out;
    const Fp* consts;           // device
    const uint32_t* out_slots;  // device
    const uint32_t* kinds;      // device
    const uint2* desc;          // device
};

__device__ __forceinline__ Fp slp_load(const uint32_t* slots, uint32_t s) {
    const uint4* p = reinterpret_cast<const uint4*>(slots + 12 * s);
    uint4 a = p[0], b = p[1], c = p[2];
    Fp r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
__device__ __forceinline__ void slp_store(uint32_t* slots, uint32_t s, const Fp& v) {
    uint4* p = reinterpret_cast<uint4*>(slots + 12 * s);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}

// inputs: [instances][n_in] Fp (Montgomery); settings_inputs: [n_set] Fp; outputs: [instances][n_out] Fp
__global__ void k_slp_run(SlpProgram prog, const Fp* __restrict__ inputs, const Fp* __restrict__ settings_inputs,
                          Fp* __restrict__ outputs) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];
    const uint32_t tid = threadIdx.x, inst = blockIdx.x, lanes = prog.lanes;
    const Fp* my_in = inputs + (size_t)inst * prog.n_in;
    uint2 d = prog.desc[tid];
    for (uint32_t s = 0; s < prog.n_steps; s++) {
        uint2 dn = make_uint2(0, 0);
        if (s + 1 < prog.n_steps) dn = prog.desc[(size_t)(s + 1) * lanes + tid];  // prefetch
        const uint32_t op = d.y >> 16, dst = d.y & 0xffffu;
        const uint32_t kind = prog.kinds[s];  // wave-uniform
        Fp r;
        if (kind == 1) {
            Fp a = slp_load(slots, d.x & 0xffffu), b = slp_load(slots, d.x >> 16);
            r = FpF::mul(a, b);
        } else {
            if (op == SLP_ADD || op == SLP_SUB) {
                Fp a = slp_load(slots, d.x & 0xffffu), b = slp_load(slots, d.x >> 16);
                r = (op == SLP_ADD) ? FpF::add(a, b) : FpF::sub(a, b);
            } else if (op == SLP_LOADC) {
                r = prog.consts[d.x];
            } else if (op == SLP_LOADI) {
                r = my_in[d.x];
            } else if (op == SLP_LOADS) {
                r = settings_inputs[d.x];
            }
        }
        if (op != SLP_NOP) slp_store(slots, dst, r);
        __syncthreads();
        d = dn;
    }
    for (uint32_t o = tid; o < prog.n_out; o += lanes) outputs[(size_t)inst * prog.n_out + o] = slp_load(slots, prog.out_slots[o]);
}

}  // namespace kzg
