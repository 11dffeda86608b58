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

// one step for one lane; d = this lane's descriptor.  The three step kinds are separate wave-uniform
// branches, each with its own store, so that add/sub and multiply steps carry no vmcnt dependence.
template <bool MULTI_WAVE>
__device__ __forceinline__ void slp_exec(uint32_t* slots, const uint2 d, const SlpProgram& prog, const Fp* my_in,
                                         const Fp* settings_inputs) {
    const uint32_t op = (d.y >> 16) & 0x3fffu, dst = d.y & 0xffffu;
    const uint32_t kind = __builtin_amdgcn_readfirstlane(d.y >> 30);  // identical in every lane
    if (kind == 1) {
        Fp a = slp_load(slots, d.x & 0xffffu), b = slp_load(slots, d.x >> 16);
        Fp r = FpF::mul(a, b);
        if (op != SLP_NOP) slp_store(slots, dst, r);
    } else if (kind == 0) {
        Fp a = slp_load(slots, d.x & 0xffffu), b = slp_load(slots, d.x >> 16);
        Fp r = (op == SLP_ADD) ? FpF::add(a, b) : FpF::sub(a, b);
        if (op != SLP_NOP) slp_store(slots, dst, r);
    } else {
        if (op != SLP_NOP) {
            const Fp* src = op == SLP_LOADC ? prog.consts : op == SLP_LOADI ? my_in : settings_inputs;
            Fp r = src[d.x];
            slp_store(slots, dst, r);
        }
    }
    // Step boundary.  With lanes == 64 the workgroup is ONE wavefront: LDS instructions of a wave execute
    // in issue order, so the next step's ds_reads see this step's ds_writes without any barrier.
    if (MULTI_WAVE) __syncthreads();
    else asm volatile("" ::: "memory");
}

// inputs: [instances][n_in] Fp (Montgomery); settings_inputs: [n_set] Fp; outputs: [instances][n_out] Fp
//
// Descriptor streaming: a step lasts 0.15-1.5 us, shorter than a dependent global load, so descriptors are
// fetched a GROUP (16 steps) ahead into registers while the current group executes, parked in a small
// per-lane LDS ring at the group boundary (the only place that waits on vmcnt), and the (non-unrolled)
// step loop reads its descriptor from LDS one step ahead.  Dynamic LDS = slots | descriptor ring.
constexpr int SLP_GROUP = 16;
template <bool MULTI_WAVE>
__global__ __launch_bounds__(MULTI_WAVE ? 256 : 64) void k_slp_run(SlpProgram prog, const Fp* __restrict__ inputs, const Fp* __restrict__ settings_inputs,
                          Fp* __restrict__ outputs) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];
    const uint32_t tid = threadIdx.x, inst = blockIdx.x, lanes = prog.lanes, n_steps = prog.n_steps;
    uint2* ring = reinterpret_cast<uint2*>(slots + (size_t)12 * prog.n_slots);  // [2][SLP_GROUP][lanes]
    const Fp* my_in = inputs + (size_t)inst * prog.n_in;
    const uint32_t last = n_steps - 1;
    uint2 r[SLP_GROUP];
#pragma unroll
    for (int k = 0; k < SLP_GROUP; k++) {
        uint32_t st = (uint32_t)k < last ? (uint32_t)k : last;
        ring[(size_t)k * lanes + tid] = prog.desc[(size_t)st * lanes + tid];
    }
    if (MULTI_WAVE) __syncthreads();
    const uint32_t n_groups = (n_steps + SLP_GROUP - 1) / SLP_GROUP;
    for (uint32_t g = 0; g < n_groups; g++) {
        const uint32_t base = g * SLP_GROUP;
        // issue the next group's descriptor loads (clamped: always SLP_GROUP loads, so nothing is conditional)
#pragma unroll
        for (int k = 0; k < SLP_GROUP; k++) {
            uint32_t st = base + SLP_GROUP + k;
            st = st < last ? st : last;
            r[k] = prog.desc[(size_t)st * lanes + tid];
        }
        const uint2* cur = ring + (size_t)(g & 1) * SLP_GROUP * lanes + tid;
        const uint32_t cnt = n_steps - base < (uint32_t)SLP_GROUP ? n_steps - base : (uint32_t)SLP_GROUP;
        uint2 d = cur[0];
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; k++) {
            const uint2 dn = cur[(size_t)(k + 1 < (uint32_t)SLP_GROUP ? k + 1 : k) * lanes];
            slp_exec<MULTI_WAVE>(slots, d, prog, my_in, settings_inputs);
            d = dn;
        }
        uint2* nxt = ring + (size_t)((g + 1) & 1) * SLP_GROUP * lanes + tid;
#pragma unroll
        for (int k = 0; k < SLP_GROUP; k++) nxt[(size_t)k * lanes] = r[k];
    }
    if (MULTI_WAVE) __syncthreads();
    for (uint32_t o = tid; o < prog.n_out; o += lanes) outputs[(size_t)inst * prog.n_out + o] = slp_load(slots, prog.out_slots[o]);
}

}  // namespace kzg
