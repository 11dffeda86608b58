// capi_settings.hpp - KzgSettings / Workspace and the construction of a handle (trusted-setup text parser, tables, prepared pairing lines).
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
// ---------------------------------------------------------------- settings
struct DevProgram {
    SlpProgram p{};
    void* blob = nullptr;  // device copy of the whole program
};

struct DevProgram2 {  // a latency program (slp2.hpp)
    Slp2Program p{};
    void* blob = nullptr;
};

constexpr size_t MAX_WORLD = 64;
constexpr size_t LATENCY_PAIRING_MAX = 32;  // launches of up to this many pairing checks run the multi-wave latency program
constexpr size_t MAX_BATCHES_PER_LAUNCH = 16384;
constexpr unsigned MSM_MAX_SLICES = 32;
constexpr size_t SLICED_MAX_BLOBS = 16384;  // host batches up to this size arrive in slices across the blobs (the range of the two-lane challenge form, which runs in segments)
constexpr size_t LATENCY_MAX_BLOBS = 4096;  // launches up to this size: CU-split stream pair + the latency MSM layout
struct Workspace {
    size_t cap_n = 0;       // batch capacity
    size_t cap_b = 0;       // batches-per-group capacity
    size_t pending_n = 0, pending_b = 0, finish_b = 0;  // group currently in flight on this handle
    int chunks = MSM_CHUNKS;                              // MSM layout of the group in flight (msm.hpp)
    size_t off_r = 0, off_part = 0, off_out = 0, off_parts = 0;  // pinned-buffer layout
    size_t cap_stage = 0, cap_stage_cp = 0;   // staged host-input capacity: blobs | (commitment, proof) pairs
    Fr *d_z = nullptr, *d_y = nullptr, *d_scalars = nullptr, *d_partial = nullptr, *d_r = nullptr;
    uint32_t *d_status = nullptr, *d_pflag = nullptr, *d_term_point = nullptr, *d_term_scalar = nullptr, *d_sorted = nullptr;
    G1Aff* d_points = nullptr;
    G1Jac *d_window = nullptr, *d_window_sl = nullptr, *d_ab = nullptr, *d_parts = nullptr;
    G1Jac* d_send = nullptr;  // a shard's contribution to the RCCL all-gather of partial sums: up to MAX_WORLD pieces x (A, B) (capi_multi.hpp)
    uint32_t* d_msm_save = nullptr;  // the window kernel's bucket sums between its row and column trees (msm.hpp MsmDesc::save)
    size_t cap_msm_save = 0;         // bytes
    void* d_mult = nullptr;  // MSM tables: G1Jac29Mem / G1Aff29Mem / G1Jac entries (fp29_enabled(), msm_affine_enabled())
    // in-kernel stamps of the last launch group (field.hpp kstamp_in / kstamp_out): 16 words - [0, 4) the throughput-form
    // challenge kernel (its interval and its clock, fr_kernels.hpp), [4, 6) k_blob_evaluate, [6, 8) the decode pass, [8, 10) the
    // MSM window kernel
    unsigned long long* d_ktime = nullptr;
    bool ktime_valid = false;   // [0, 4) were written (the challenge took its throughput form)
    bool kstamps_valid = false; // [4, 10) belong to the group in flight
    G1Jac29Mem* d_jtmp = nullptr;  // 2^64 P of every decoded point on its way to the affine table (k_mult_to_affine29)
    bool mult_affine = false;      // format of d_mult as the last decode left it
    Fp *d_slp_in = nullptr, *d_slp_out = nullptr;
    uint8_t* d_g1msm = nullptr;  // kzg_g1_msm / kzg_g1_msm_setup: window sums, fold trees, the large-sum tail (grow-only; a per-call hipFree stalls every lane of the device)
    size_t cap_g1msm = 0;
    uint8_t *d_stage_blobs = nullptr, *d_stage_cp = nullptr, *d_bytes = nullptr, *d_records = nullptr;
    uint32_t* d_sha_mid = nullptr;  // SHA-256 midstates between the segments of a sliced challenge chain, 32 B per blob
    // host-fed stream of batches (kzg_verify_blob_kzg_proof_batches): two staging sets, each [blobs | commitments | proofs]
    // of one chunk, filled on the copy stream while the other one is verified
    uint8_t* d_hstage[2] = {nullptr, nullptr};
    size_t cap_hstage = 0;  // blobs per staging set
    // pinned host mirrors
    uint8_t* h_buf = nullptr;
    size_t h_cap = 0;
};

#include "small_queue.hpp"

struct KzgSettings {
    int device = 0;
    Fr *d_M = nullptr, *d_DM = nullptr;            // roots of unity, 8x32 Montgomery (R, R^2 scalings)
    Fr29Mem *d_M29 = nullptr, *d_DM29 = nullptr;   // the same in radix 2^29 (fr29.hpp)
    uint4 *d_eval_a = nullptr, *d_eval_b = nullptr;  // ... in the order k_blob_evaluate reads them (k_eval_tables)
    uint32_t* d_eval_c = nullptr;
    Fp* d_tau4 = nullptr;   // [tau]G2 affine (x.c0 x.c1 y.c0 y.c1), Montgomery
    Fp* d_prep = nullptr;   // prepared lines: [tau]G2 then generator (2 * 408 Fp)
    void* d_gen_mult = nullptr;   // the generator's MSM tables: [0, 4) the default layout, [4, 36) the latency layout, [36, 52) the proofs layout (msm.hpp)
    G1Aff29Mem* d_gen_mult_aff = nullptr;  // [0, 4) as affine entries
    // full trusted setup (kzg_settings_load_trusted_setup only; not needed by verification):
    G1Aff* d_g1 = nullptr;            // g1_points, bit-reversal permuted (build.rs:79,89-105), 4096 entries
    uint32_t* d_g1_flag = nullptr;    // 0 finite / 1 identity (unchecked decode, build.rs:68)
    void* d_g1_mult = nullptr;        // their MSM multiples (msm.hpp), valid iff g1_in_subgroup
    G1Aff29Mem* d_g1_mult_aff = nullptr;  // the same as AFFINE rows (the throughput layout of the verification path's MSM): kzg_g1_msm_setup's 8-bit form
    int n_g1 = 0;                     // number of G1 Lagrange points (4096)
    mutable G1Aff29Mem* d_g1_fb_rows = nullptr;  // fixed-base rows 2^(16 v) P_j (msm_fixed.hpp), made by the first large kzg_g1_msm_setup call
    mutable uint32_t* d_fb_plan = nullptr;       // ... and that form's device-side plan words
    bool g1_in_subgroup = false;      // every G1 point lies in the r-torsion (what the GLV multiples need)
    Fp* d_g2 = nullptr;               // g2_points (monomial), n_g2 x 4 Fp
    size_t n_g2 = 0;
    uint8_t g1_first[2][48] = {};     // g1_points[0], [1] of the FILE order, for the monomial-form check (build.rs:107-129)
    DevProgram prep, verify;
    DevProgram2 verify2;            // VERIFY scheduled for one check at a time (kzg_rs_amd/slp/schedule2.py)
    DevProgram2 scalars, verify3;   // the one-proof path (proof_kernels.hpp): [y]G, the lines of [tau]G2 - [z]G2 | the pairing behind them
    Fp* d_fixed_base = nullptr;     // fixed-base tables of the two generators (tools/gen_fixed_base.py), 3.4 MB
    uint32_t* d_prep29 = nullptr;   // d_prep in the latency program's format (radix 2^29, 16 words per element)
    // s1 / s2: the two streams the current launch uses (challenge chain | point decode).  They point at the plain pair,
    // or - for a small launch (a single batch) - at a pair confined to disjoint halves of the CUs: the 16 two-wave
    // workgroups of the challenge chain and the 32 decode waves otherwise land on the same first CUs of every XCD and,
    // run to run, share SIMDs (the chain then takes 4.9 ms instead of 3.5 ms).  Measured: one 1 024-blob batch 9.1 ms on
    // the split pair, 10.1-11.5 ms on the plain pair; option cu_mask=0 disables the split pair.
    mutable hipStream_t s1 = nullptr, s2 = nullptr, s_sha = nullptr;  // main | point decode | challenge chain (select_streams)
    hipStream_t s_plain[2] = {nullptr, nullptr};
    mutable hipStream_t s_half[2] = {nullptr, nullptr};
    mutable bool s_half_tried = false;
    int n_cus = 0;
    hipEvent_t ev[12] = {};
    mutable hipStream_t s_copy = nullptr;  // host -> device staging copies of the host-fed stream (made on first use)
    mutable hipStream_t s_aux = nullptr;   // the one-proof path's third stream: the subgroup test beside the pairing (made on first use)
    mutable Fp* d_proof = nullptr;         // ... and its device buffers: SCALARS' inputs | VERIFY3's inputs (made on first use)
    mutable Fp *d_proofs = nullptr, *d_proofs_out = nullptr;  // the same for MANY independent proofs (kzg_verify_kzg_proofs): [cap] records each
    mutable uint8_t* h_proofs = nullptr;   // ... and their pinned mirror
    mutable size_t cap_proofs = 0;
    mutable hipEvent_t ev_copy[2] = {nullptr, nullptr};
    mutable hipEvent_t ev_slice[17] = {};  // a host Vec<Blob> arriving in slices: [0] commitments + proofs landed, [1 + j] slice j landed
    mutable std::mutex mu;
    mutable Workspace ws;
    mutable uint32_t* d_eval_scratch = nullptr;  // between the three evaluation kernels (launch_evaluate)
    mutable size_t eval_scratch_cap = 0;
    mutable float timings[8] = {};
    mutable double tsum[8] = {};   // the same, summed over every group finished on this handle since the last reset
    mutable uint64_t tcount = 0;
    mutable struct ProverBufs* prover = nullptr;  // the prover-side entry points' buffers, made by the first of those calls (capi_prover.hpp)
    mutable double clk_sum[2] = {};  // shader cycles | 100 MHz reference ticks of the throughput-form challenge kernel's waves
    // the kernels' own execution intervals (in-kernel stamps), ms: challenge | evaluate | decode + multiples | MSM window - of the
    // last launch group, and summed over the groups finished since the last reset (kzg_kernel_stamp_totals)
    mutable float kstamp_ms[4] = {};
    mutable double kstamp_sum[4] = {};
    mutable uint64_t kstamp_count = 0;
    // A multi-device handle (capi_multi.hpp): this handle is shard 0 on the first device of the list and a complete
    // single-device handle in its own right; `peers` are the (private) single-device handles of the other entries, `multi`
    // the device list and the exchange (in-process RCCL communicators, or host staging).
    std::vector<KzgSettings*> peers;
    struct MultiState* multi = nullptr;
    // further private handles on THIS device (capi_pipeline.hpp): one per launch group kept in flight beyond the first
    mutable std::vector<KzgSettings*> lanes;
    bool borrowed = false;          // a lane: the tables and programs above belong to the handle it was made from (settings_lane)
    uint8_t tau_g2_bytes[96] = {};  // g2_points[1] as given
    mutable float multi_ms[8] = {};  // host wall-clock stages of the last sharded call (kzg_multi_last_timings)
    mutable uint8_t multi_last_r[32] = {};  // the batch challenge of the last sharded call, little-endian (test hook kzg_debug_multi_last_r)
    // concurrent small calls on this handle (capi_coalesce.hpp): their queue and the private lanes their launches run on; null
    // on the handle's own lanes and peers, and with option coalesce=0
    mutable SmallQueue* small = nullptr;
    int stream_priority = 0;         // of the handle's own streams (stream_make)
    bool proof_two_streams = false;  // a lane of that queue: the one-proof chains on two streams (proof_reserve)
    std::string note;  // what a caller may want to know about a handle that was made successfully (kzg_settings_note)
};
static void multi_free(KzgSettings* s);
static void small_free(KzgSettings* s);

static KzgRet upload_program(DevProgram& dp, const unsigned char* begin, const unsigned char* end) {
    size_t len = (size_t)(end - begin);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(begin);
    if (len < 64 || w[0] != SLP_MAGIC) return fail(KZG_ERROR, "embedded SLP program is corrupt");
    HIPCHK(hipMalloc(&dp.blob, len));
    HIPCHK(hipMemcpy(dp.blob, begin, len, hipMemcpyHostToDevice));
    SlpProgram& p = dp.p;
    p.lanes = w[1]; p.n_slots = w[2]; p.n_steps = w[3]; p.n_const = w[4]; p.n_in = w[5]; p.n_set = w[6]; p.n_out = w[7];
    const uint32_t* d = reinterpret_cast<const uint32_t*>(dp.blob);
    size_t off = 16;
    p.consts = reinterpret_cast<const Fp*>(d + off);
    off += (size_t)12 * p.n_const;
    p.out_slots = d + off;
    off += p.n_out;
    p.kinds = d + off;
    off += p.n_steps;
    p.desc = reinterpret_cast<const uint2*>(d + off);
    if ((off + (size_t)2 * p.lanes * p.n_steps) * 4 != len) return fail(KZG_ERROR, "embedded SLP program has the wrong size");
    return KZG_OK;
}

static KzgRet upload_program2(DevProgram2& dp, const unsigned char* begin, const unsigned char* end) {
    size_t len = (size_t)(end - begin);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(begin);
    if (len < 64 || w[0] != SLP2_MAGIC) return fail(KZG_ERROR, "embedded latency program is corrupt");
    HIPCHK(hipMalloc(&dp.blob, len));
    HIPCHK(hipMemcpy(dp.blob, begin, len, hipMemcpyHostToDevice));
    Slp2Program& p = dp.p;
    p.lanes = w[1]; p.n_slots = w[2]; p.n_steps = w[3]; p.n_const = w[4]; p.n_in = w[5]; p.n_set = w[6]; p.n_out = w[7]; p.n_load_steps = w[8]; p.out_values = w[9];
    if (p.n_load_steps >= p.n_steps) return fail(KZG_ERROR, "embedded latency program has no compute steps");
    const uint32_t* d = reinterpret_cast<const uint32_t*>(dp.blob);
    size_t off = 16;
    p.consts = d + off;
    off += (size_t)16 * p.n_const;
    p.out_slots = d + off;
    off += (p.n_out + 3) & ~3u;  // padded: the descriptors are read as 16-byte vectors
    p.desc = reinterpret_cast<const Slp2Desc*>(d + off);
    if ((off + (size_t)4 * p.lanes * p.n_steps) * 4 != len) return fail(KZG_ERROR, "embedded latency program has the wrong size");
    if (p.lanes != 128 && p.lanes != 192 && p.lanes != 256) return fail(KZG_ERROR, "embedded latency program: 128, 192 or 256 lanes expected");
    return KZG_OK;
}

// the latency form: one workgroup of 2 to 4 wavefronts per instance (what the program was scheduled for)
template <int LANES>
static KzgRet launch_program2(const DevProgram2& dp, const Fp* d_in, const uint32_t* d_set29, Fp* d_out, int instances, hipStream_t st, uint32_t in_stride,
                              uint32_t out_stride) {
    const size_t lds = (size_t)dp.p.n_slots * SLP2_SLOT_WORDS * 4 + (size_t)SLP2_GROUP * LANES * sizeof(uint4);  // slots | descriptor ring
    HIPCHK(DYN_LDS(k_slp2_run<LANES>, lds));
    hipLaunchKernelGGL(k_slp2_run<LANES>, dim3(instances), dim3(LANES), lds, st, dp.p, d_in, d_set29, d_out, in_stride ? in_stride : dp.p.n_in,
                       out_stride ? out_stride : dp.p.n_out);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}
// in_stride / out_stride (elements between two instances' records; 0 = the program's own counts)
static KzgRet run_program2(const DevProgram2& dp, const Fp* d_in, const uint32_t* d_set29, Fp* d_out, int instances, hipStream_t st, uint32_t in_stride = 0,
                           uint32_t out_stride = 0) {
    return dp.p.lanes == 128 ? launch_program2<128>(dp, d_in, d_set29, d_out, instances, st, in_stride, out_stride)
         : dp.p.lanes == 192 ? launch_program2<192>(dp, d_in, d_set29, d_out, instances, st, in_stride, out_stride)
                             : launch_program2<256>(dp, d_in, d_set29, d_out, instances, st, in_stride, out_stride);
}
// which form runs a launch of `instances` checks: option pairing=1 | 2 forces the one-wave / the latency program (A/B, cross-check)
static bool pairing_latency_form(size_t instances) {
    static const int forced = (int)ab_int("pairing", 0);
    return forced == 2 || (forced != 1 && instances <= LATENCY_PAIRING_MAX);
}
static KzgRet run_verify(const KzgSettings* s, const Fp* d_in, Fp* d_out, int instances, hipStream_t st);

static KzgRet run_program(const DevProgram& dp, const Fp* d_in, const Fp* d_set, Fp* d_out, int instances, hipStream_t st) {
    size_t lds = (size_t)dp.p.n_slots * 48 + (size_t)2 * SLP_GROUP * dp.p.lanes * sizeof(uint2);  // slots | descriptor ring
    if (dp.p.lanes == 64) {
        HIPCHK(DYN_LDS(k_slp_run<false>, lds));
        hipLaunchKernelGGL(k_slp_run<false>, dim3(instances), dim3(64), lds, st, dp.p, d_in, d_set, d_out);
    } else {
        HIPCHK(DYN_LDS(k_slp_run<true>, lds));
        hipLaunchKernelGGL(k_slp_run<true>, dim3(instances), dim3(dp.p.lanes), lds, st, dp.p, d_in, d_set, d_out);
    }
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// the pairing check of `instances` (A, B) pairs against the handle's prepared lines, in the form that suits the launch size
static KzgRet run_verify(const KzgSettings* s, const Fp* d_in, Fp* d_out, int instances, hipStream_t st) {
    if (pairing_latency_form((size_t)instances)) return run_program2(s->verify2, d_in, s->d_prep29, d_out, instances, st);
    return run_program(s->verify, d_in, s->d_prep, d_out, instances, st);
}

static KzgRet settings_build(KzgSettings* s, const uint8_t tau_g2[96]);
// a handle from g2_points[1]; on any failure everything allocated so far is released
static KzgRet settings_common(KzgSettings** out, const uint8_t tau_g2[96]) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(KZG_ERROR, "no HIP device: this library has no CPU fallback");
    }
    KzgSettings* s = new KzgSettings();
    KzgRet rc = settings_build(s, tau_g2);
    if (rc != KZG_OK) {
        const std::string msg = g_err;  // kzg_settings_free may run HIP calls; keep the first error
        kzg_settings_free(s);
        g_err = msg;
        return rc;
    }
    if (opt_flag("coalesce", true)) {  // (a peer of a multi-device handle loses it again in multi_build: only the handle a caller holds has one)
        s->small = new SmallQueue();
        s->small->max_lanes = (size_t)std::max(1L, std::min((long)SMALL_LANES_MAX, opt_int("small_lanes", 2)));
        s->small->lane_two_streams = ab_int("small_streams", 3) == 2;
        s->small->lane_priority = ab_int("small_priority", 1) ? 1 : 0;
        s->small->linger_us = std::max(0L, std::min(2000L, opt_int("small_linger_us", 250)));
        s->small->linger_gap_us = std::max(1L, std::min(1000L, ab_int("small_linger_gap_us", 40)));
        s->small->cap_proofs = (size_t)std::max(1L, std::min(1024L, ab_int("small_cap_proofs", 1024)));  // (A/B: fewer tuples per launch; a request larger than the cap would never leave)
    }
    *out = s;
    return KZG_OK;
}
// what every handle owns, a lane included: its streams and events (the workspace grows on first use)
static KzgRet stream_make(hipStream_t* st, int priority) {  // priority 0: the default; 1: the device's highest
    if (priority) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest));
    } else HIPCHK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    return KZG_OK;
}
static KzgRet settings_streams(KzgSettings* s, bool single_stream, int priority = 0) {
    HIPCHK(hipGetDevice(&s->device));  // the calling thread's current device (multi_build sets it per shard)
    s->stream_priority = priority;
    KzgRet rc_s = stream_make(&s->s_plain[0], priority);
    if (rc_s != KZG_OK) return rc_s;
    s->s1 = s->s_sha = s->s_plain[0];
    HIPCHK(hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, s->device));
    // option single_stream=1 (profiling aid): run the point-decode chain on the same stream as the challenge chain, so
    // per-dispatch PMC counters are not polluted by a concurrent kernel
    if (single_stream) s->s2 = s->s1;
    else {
        if ((rc_s = stream_make(&s->s_plain[1], priority)) != KZG_OK) return rc_s;
        s->s2 = s->s_plain[1];
    }
    for (auto& e : s->ev) HIPCHK(hipEventCreate(&e));
    return KZG_OK;
}
// A LANE of a handle (capi_pipeline.hpp, capi_multi.hpp): a private handle on the same device - its own streams, events,
// workspace and timings - that READS the parent's tables and pairing programs (they are immutable after construction): making
// one costs two streams and a dozen events instead of the table kernels, three program uploads and a PREP pairing run.  The
// caller has set the parent's device.  Freed with the parent (kzg_settings_free), never handed out.
static KzgRet settings_lane(KzgSettings** out, const KzgSettings* parent, int priority = 0) {
    KzgSettings* l = new KzgSettings();
    l->borrowed = true;
    l->d_M = parent->d_M; l->d_DM = parent->d_DM; l->d_M29 = parent->d_M29; l->d_DM29 = parent->d_DM29;
    l->d_eval_a = parent->d_eval_a; l->d_eval_b = parent->d_eval_b; l->d_eval_c = parent->d_eval_c;
    l->d_tau4 = parent->d_tau4; l->d_prep = parent->d_prep; l->d_prep29 = parent->d_prep29;
    l->d_gen_mult = parent->d_gen_mult; l->d_gen_mult_aff = parent->d_gen_mult_aff;
    l->prep = parent->prep; l->verify = parent->verify; l->verify2 = parent->verify2;
    l->scalars = parent->scalars; l->verify3 = parent->verify3; l->d_fixed_base = parent->d_fixed_base;
    memcpy(l->tau_g2_bytes, parent->tau_g2_bytes, 96);
    KzgRet rc = settings_streams(l, /*single_stream=*/!parent->s_plain[1], priority);
    if (rc != KZG_OK) {
        const std::string msg = g_err;
        kzg_settings_free(l);
        g_err = msg;
        return rc;
    }
    *out = l;
    return KZG_OK;
}
static KzgRet settings_build(KzgSettings* s, const uint8_t tau_g2[96]) {
    memcpy(s->tau_g2_bytes, tau_g2, 96);
    KzgRet rc_streams = settings_streams(s, opt_flag("single_stream", false));
    if (rc_streams != KZG_OK) return rc_streams;
    HIPCHK(hipMalloc(&s->d_M, sizeof(Fr) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_DM, sizeof(Fr) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_M29, sizeof(Fr29Mem) * FE_PER_BLOB));
    HIPCHK(hipMalloc(&s->d_DM29, sizeof(Fr29Mem) * FE_PER_BLOB));
    hipLaunchKernelGGL(k_roots_tables, dim3(FE_PER_BLOB / 64), dim3(64), 0, s->s1, s->d_M, s->d_DM);
    hipLaunchKernelGGL(k_roots_tables29, dim3(FE_PER_BLOB / 64), dim3(64), 0, s->s1, s->d_M, s->d_M29, s->d_DM29);
    HIPCHK(hipMalloc(&s->d_eval_a, sizeof(uint4) * EVAL_SLOTS * 64));
    HIPCHK(hipMalloc(&s->d_eval_b, sizeof(uint4) * EVAL_SLOTS * 64));
    HIPCHK(hipMalloc(&s->d_eval_c, 4 * EVAL_SLOTS * 64));
    hipLaunchKernelGGL(k_eval_tables, dim3(EVAL_SLOTS), dim3(64), 0, s->s1, s->d_M29, s->d_DM29, s->d_eval_a, s->d_eval_b, s->d_eval_c);
    HIPCHK(hipGetLastError());
    KzgRet rc;
    if ((rc = upload_program(s->prep, kzg_slp_prep_begin, kzg_slp_prep_end)) != KZG_OK) return rc;
    if ((rc = upload_program(s->verify, kzg_slp_verify_begin, kzg_slp_verify_end)) != KZG_OK) return rc;
    if ((rc = upload_program2(s->verify2, kzg_slp_verify2_begin, kzg_slp_verify2_end)) != KZG_OK) return rc;
    if ((rc = upload_program2(s->scalars, kzg_slp_scalars_begin, kzg_slp_scalars_end)) != KZG_OK) return rc;
    if ((rc = upload_program2(s->verify3, kzg_slp_verify3_begin, kzg_slp_verify3_end)) != KZG_OK) return rc;
    if ((size_t)(kzg_fixed_base_end - kzg_fixed_base_begin) != FB_TABLE_BYTES) return fail(KZG_ERROR, "embedded fixed-base table has the wrong size");
    HIPCHK(hipMalloc(&s->d_fixed_base, FB_TABLE_BYTES));
    HIPCHK(hipMemcpy(s->d_fixed_base, kzg_fixed_base_begin, FB_TABLE_BYTES, hipMemcpyHostToDevice));
    // decompress [tau]G2 on the device, then prepare the lines of [tau]G2 and of the generator
    DevTmp t_bytes, t_flag, t_q;  // released on every path out of this function
    HIPCHK(hipMalloc(&t_bytes.p, 96));
    HIPCHK(hipMalloc(&t_flag.p, 4));
    HIPCHK(hipMalloc(&t_q.p, sizeof(Fp) * 8));  // 2 instances x 4 Fp
    uint8_t* d_bytes = t_bytes.as<uint8_t>();
    uint32_t* d_flag = t_flag.as<uint32_t>();
    Fp* d_q = t_q.as<Fp>();
    HIPCHK(hipMalloc(&s->d_tau4, sizeof(Fp) * 4));
    HIPCHK(hipMalloc(&s->d_prep, sizeof(Fp) * 2 * s->prep.p.n_out));
    HIPCHK(hipMemcpyAsync(d_bytes, tau_g2, 96, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g2_decompress, dim3(1), dim3(64), 0, s->s1, d_bytes, d_q, d_flag);
    hipLaunchKernelGGL(k_g2_generator, dim3(1), dim3(64), 0, s->s1, d_q + 4);
    HIPCHK(hipGetLastError());
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(s->d_tau4, d_q, sizeof(Fp) * 4, hipMemcpyDeviceToDevice, s->s1));
    if ((rc = run_program(s->prep, d_q, nullptr, s->d_prep, 2, s->s1)) != KZG_OK) return rc;
    {  // the same lines for the latency program
        const int np = (int)(2 * s->prep.p.n_out);
        HIPCHK(hipMalloc(&s->d_prep29, (size_t)64 * np));
        hipLaunchKernelGGL(k_fp_to_fp29mem, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, s->s1, s->d_prep, s->d_prep29, np);
        HIPCHK(hipGetLastError());
    }
    {  // multiples of the generator (msm.hpp): the same for every batch
        DevTmp t_g, t_gf, t_gm;
        HIPCHK(hipMalloc(&t_g.p, sizeof(G1Aff)));
        HIPCHK(hipMalloc(&t_gf.p, 4));
        constexpr int NG = MSM_CHUNKS + MSM_CHUNKS_LATENCY + MSM_CHUNKS_PROOFS;
        HIPCHK(hipMalloc(&t_gm.p, sizeof(G1Jac) * NG));
        G1Aff* d_g = t_g.as<G1Aff>();
        uint32_t* d_gf = t_gf.as<uint32_t>();
        G1Jac* d_gm = t_gm.as<G1Jac>();
        hipLaunchKernelGGL(k_set_generator, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, 0);
        hipLaunchKernelGGL(k_g1_multiples, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, d_gm, 1, 1, MSM_CHUNKS);
        hipLaunchKernelGGL(k_g1_multiples, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, d_gm + MSM_CHUNKS, 1, 1, MSM_CHUNKS_LATENCY);
        hipLaunchKernelGGL(k_g1_multiples, dim3(1), dim3(64), 0, s->s1, d_g, d_gf, d_gm + MSM_CHUNKS + MSM_CHUNKS_LATENCY, 1, 1, MSM_CHUNKS_PROOFS);
        if (fp29_enabled()) {
            HIPCHK(hipMalloc(&s->d_gen_mult, sizeof(G1Jac29Mem) * NG));
            hipLaunchKernelGGL(k_jac_to_jac29, dim3(1), dim3(64), 0, s->s1, d_gm, (G1Jac29Mem*)s->d_gen_mult, NG);
            HIPCHK(hipMalloc(&s->d_gen_mult_aff, sizeof(G1Aff29Mem) * MSM_CHUNKS));
            hipLaunchKernelGGL(k_jac29_to_aff29, dim3(1), dim3(64), 0, s->s1, (const G1Jac29Mem*)s->d_gen_mult, s->d_gen_mult_aff, MSM_CHUNKS);
        } else {
            s->d_gen_mult = d_gm;  // the handle owns it from here
            t_gm.p = nullptr;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->s1));  // before the temporaries of this scope are released
    }
    HIPCHK(hipStreamSynchronize(s->s1));
    if (flag != G1_OK) return fail(KZG_BAD_SETUP, "g2_points[1] is not a valid (finite) compressed G2 point");
    if (s->verify.p.n_set != 2 * s->prep.p.n_out || s->verify.p.n_in != 6 || s->prep.p.n_in != 4 || s->verify2.p.n_set != s->verify.p.n_set ||
        s->verify2.p.n_in != 6 || s->verify2.p.n_out != s->verify.p.n_out)
        return fail(KZG_ERROR, "embedded SLP programs do not fit together");
    if (s->scalars.p.n_in != (uint32_t)SCALARS_INPUTS || s->scalars.p.n_set != 0 || !s->scalars.p.out_values || s->scalars.p.n_out + 6 != (uint32_t)VERIFY3_INPUTS ||
        s->verify3.p.n_in != (uint32_t)VERIFY3_INPUTS || s->verify3.p.n_set != s->verify.p.n_set || s->verify3.p.n_out != (uint32_t)VERIFY3_OUTPUTS || s->verify3.p.out_values)
        return fail(KZG_ERROR, "embedded one-proof programs do not fit together");
    return KZG_OK;
}

// Device half of kzg_settings_load_trusted_setup: the G1 Lagrange points - unchecked decode (build.rs:66-70) for the table, and
// the decode + subgroup test + multiples pass of the MSM (msm.hpp) so that commitments can be computed against them - and
// all G2 monomial points (build.rs:72-75; verification itself reads only [1]).  Temporaries live in DevTmp, so every
// return path releases them; the caller releases the handle on failure.
static KzgRet settings_load_points(KzgSettings* s, const std::vector<uint8_t>& g1b, const std::vector<uint8_t>& g2b, int N, size_t n2) {
    DevTmp t_bytes, t_flag2, t_gflag, t_tmp;
    HIPCHK(hipMalloc(&t_bytes.p, std::max(g1b.size(), g2b.size())));
    HIPCHK(hipMalloc(&t_flag2.p, 4 * (size_t)N));
    HIPCHK(hipMalloc(&t_tmp.p, sizeof(G1Aff) * (size_t)N));
    uint8_t* d_bytes = t_bytes.as<uint8_t>();
    uint32_t* d_flag2 = t_flag2.as<uint32_t>();
    G1Aff* d_tmp = t_tmp.as<G1Aff>();
    HIPCHK(hipMalloc(&s->d_g1, sizeof(G1Aff) * (size_t)N));
    HIPCHK(hipMalloc(&s->d_g1_flag, 4 * (size_t)N));
    HIPCHK(hipMalloc(&s->d_g1_mult, MULT_ENTRY_BYTES * MSM_CHUNKS * (size_t)N));
    HIPCHK(hipMemcpyAsync(d_bytes, g1b.data(), g1b.size(), hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s->s1, d_bytes, d_bytes, N, s->d_g1, s->d_g1_flag, N, 0);
#if KZG_AB_VARIANTS
    if (!fp29_enabled())
        hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS>, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s->s1, d_bytes, d_bytes, N, d_tmp,
                           d_flag2, (G1Jac*)s->d_g1_mult, N, N);
    else
#endif
        hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, false>), dim3((unsigned)((N + 63) / 64)), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s1, d_bytes, d_bytes, N, d_tmp,
                           d_flag2, s->d_g1_mult, (G1Jac29Mem*)nullptr, N, N);
    HIPCHK(hipGetLastError());
    s->n_g1 = N;
    DevTmp t_jtmp, t_aff_pts;
    if (msm_affine_enabled()) {
        // the same points as affine table rows (P, 2^64 P, -phi(P), -phi(2^64 P)): sums over the setup's own points
        // (kzg_g1_msm_setup) then run on the verification path's mixed-addition window kernel without a per-call decode
        HIPCHK(hipMalloc(&s->d_g1_mult_aff, sizeof(G1Aff29Mem) * MSM_CHUNKS * (size_t)N));
        HIPCHK(hipMalloc(&t_jtmp.p, sizeof(G1Jac29Mem) * (size_t)N));
        HIPCHK(hipMalloc(&t_aff_pts.p, sizeof(G1Aff) * (size_t)N));
        hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, true>), dim3((unsigned)((N + 255) / 256)), dim3(256), 256 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s1, d_bytes,
                           d_bytes, N, t_aff_pts.as<G1Aff>(), d_flag2, (void*)s->d_g1_mult_aff, t_jtmp.as<G1Jac29Mem>(), N, N);
        const unsigned conv_blocks = (unsigned)((N + 64 * AFFINE_BATCH - 1) / (64 * AFFINE_BATCH));
        hipLaunchKernelGGL(k_mult_to_affine29, dim3(conv_blocks), dim3(64), 0, s->s1, t_jtmp.as<G1Jac29Mem>(), d_flag2, s->d_g1_mult_aff, N, N);
        HIPCHK(hipGetLastError());
    }
    std::vector<uint32_t> f1((size_t)N), f2((size_t)N);
    HIPCHK(hipMemcpyAsync(f1.data(), s->d_g1_flag, 4 * (size_t)N, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(f2.data(), d_flag2, 4 * (size_t)N, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    s->g1_in_subgroup = true;
    for (int i = 0; i < N; i++) {
        if (f1[i] == G1_INVALID) return fail(KZG_BAD_SETUP, "load_trusted_setup Invalid g1 bytes");
        if (f2[i] == G1_INVALID) s->g1_in_subgroup = false;
    }
    s->n_g2 = n2;
    HIPCHK(hipMalloc(&s->d_g2, sizeof(Fp) * 4 * n2));
    HIPCHK(hipMalloc(&t_gflag.p, 4 * n2));
    uint32_t* d_gflag = t_gflag.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(d_bytes, g2b.data(), g2b.size(), hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g2_decompress_n, dim3((unsigned)n2), dim3(64), 0, s->s1, d_bytes, s->d_g2, d_gflag);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> fg(n2);
    HIPCHK(hipMemcpyAsync(fg.data(), d_gflag, 4 * n2, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    for (size_t i = 0; i < n2; i++)
        if (fg[i] == G1_INVALID) return fail(KZG_BAD_SETUP, "load_trusted_setup Invalid g2 bytes");
    return KZG_OK;
}

// Device list of a new handle: explicit (the _devices constructors), or from the environment for the reference-shaped
// constructors - KZG_DEVICES = "all" | "0,1,2,..." makes the handle of an UNCHANGED caller a multi-device one (capi_multi.hpp);
// unset: the calling thread's current device, as before.
static KzgRet multi_build(KzgSettings* s, const uint8_t tau_g2[96], const std::vector<int>& devices);
static KzgRet device_list(std::vector<int>& out, const int* devices, size_t n_devices, bool from_env) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(KZG_ERROR, "no HIP device: this library has no CPU fallback");
    }
    out.clear();
    if (from_env) {
        const char* e = getenv("KZG_DEVICES");
        if (!e || !*e) return KZG_OK;  // current device only
        if (strcmp(e, "all") == 0) {
            for (int i = 0; i < ndev; i++) out.push_back(i);
        } else {
            for (const char* p = e; *p;) {
                char* end = nullptr;
                long v = strtol(p, &end, 10);
                if (end == p) return fail(KZG_BADARGS, "KZG_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
                out.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') return fail(KZG_BADARGS, "KZG_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
            }
        }
    } else if (!devices || !n_devices) {
        for (int i = 0; i < ndev; i++) out.push_back(i);  // every visible device
    } else {
        out.assign(devices, devices + n_devices);
    }
    if (out.size() > MAX_WORLD) return fail(KZG_BADARGS, "more than 64 devices in one handle");
    for (int d : out)
        if (d < 0 || d >= ndev) return fail(KZG_BADARGS, "device ordinal out of range");
    return KZG_OK;
}

// shard 0 on devs[0] (or the current device), then the peers; the caller's current device is restored
static KzgRet settings_on_devices(KzgSettings** out, const uint8_t tau_g2[96], const std::vector<int>& devs) {
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    if (!have_prev) (void)hipGetLastError();
    if (!devs.empty()) HIPCHK(hipSetDevice(devs[0]));
    KzgSettings* s = nullptr;
    KzgRet rc = settings_common(&s, tau_g2);
    if (rc == KZG_OK && !devs.empty() && (rc = multi_build(s, tau_g2, devs)) != KZG_OK) {
        const std::string msg = g_err;
        kzg_settings_free(s);
        g_err = msg;
        s = nullptr;
    }
    if (have_prev) (void)hipSetDevice(prev);
    if (rc == KZG_OK) *out = s;
    return rc;
}

static KzgRet load_trusted_setup_on(KzgSettings** out, const char* txt, size_t len, const std::vector<int>& devs) {
    std::vector<uint8_t> g1b, g2b;
    uint8_t first[2][48];
    long n1 = 0, n2 = 0;
    std::string perr;
    if (!hostparse::trusted_setup_text(txt, len, g1b, g2b, first, n1, n2, perr)) return fail(KZG_BAD_SETUP, perr);
    KzgSettings* s = nullptr;
    KzgRet rc = settings_on_devices(&s, g2b.data() + 96, devs);
    if (rc != KZG_OK) return rc;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(s->device);  // the full point tables live on shard 0 only (verification does not read them)
    memcpy(s->g1_first, first, sizeof first);
    rc = settings_load_points(s, g1b, g2b, (int)n1, (size_t)n2);
    (void)hipSetDevice(prev);
    if (rc != KZG_OK) {  // any failure below settings_common releases the whole handle (and keeps the first message)
        const std::string msg = g_err;
        kzg_settings_free(s);
        g_err = msg;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

// a constructor that succeeded under fewer than 8 hardware queues says so in the handle's note (kzg_settings_note); the error
// channel stays empty after a success (rounds 3-4 left the note in kzg_last_error(), where a caller that logs a non-empty last
// error saw an "error" after every successful constructor)
static KzgRet constructed(KzgRet rc, KzgSettings** out) {
    if (rc == KZG_OK) {
        const char* note = hw_queues_note();
        if (note) (*out)->note = (*out)->note.empty() ? note : (*out)->note + "; " + note;
        g_err.clear();
    }
    return rc;
}
extern "C" const char* kzg_settings_note(const KzgSettings* s) { return s ? s->note.c_str() : ""; }
extern "C" KzgRet kzg_settings_load_trusted_setup(KzgSettings** out, const char* txt, size_t len) {
    if (!out || !txt) return fail(KZG_BADARGS, "null argument");
    std::vector<int> devs;
    KzgRet rc = device_list(devs, nullptr, 0, /*from_env=*/true);
    return rc != KZG_OK ? rc : constructed(load_trusted_setup_on(out, txt, len, devs), out);
}
extern "C" KzgRet kzg_settings_load_trusted_setup_devices(KzgSettings** out, const char* txt, size_t len, const int* devices, size_t n_devices) {
    if (!out || !txt) return fail(KZG_BADARGS, "null argument");
    std::vector<int> devs;
    KzgRet rc = device_list(devs, devices, n_devices, false);
    return rc != KZG_OK ? rc : constructed(load_trusted_setup_on(out, txt, len, devs), out);
}

extern "C" KzgRet kzg_settings_from_tau_g2(KzgSettings** out, const uint8_t tau_g2[96]) {
    if (!out || !tau_g2) return fail(KZG_BADARGS, "null argument");
    std::vector<int> devs;
    KzgRet rc = device_list(devs, nullptr, 0, /*from_env=*/true);
    return rc != KZG_OK ? rc : constructed(settings_on_devices(out, tau_g2, devs), out);
}
extern "C" KzgRet kzg_settings_from_tau_g2_devices(KzgSettings** out, const uint8_t tau_g2[96], const int* devices, size_t n_devices) {
    if (!out || !tau_g2) return fail(KZG_BADARGS, "null argument");
    std::vector<int> devs;
    KzgRet rc = device_list(devs, devices, n_devices, false);
    return rc != KZG_OK ? rc : constructed(settings_on_devices(out, tau_g2, devs), out);
}

static void ws_free(Workspace& w) {
    void* ptrs[] = {w.d_z, w.d_y, w.d_scalars, w.d_partial, w.d_r, w.d_status, w.d_pflag, w.d_term_point, w.d_term_scalar,
                    w.d_sorted, w.d_points, w.d_window, w.d_window_sl, w.d_ab, w.d_send, w.d_mult, w.d_jtmp, w.d_ktime, w.d_parts, w.d_slp_in, w.d_slp_out, w.d_stage_blobs, w.d_stage_cp, w.d_bytes,
                    w.d_records, w.d_hstage[0], w.d_hstage[1], w.d_msm_save, w.d_sha_mid, w.d_g1msm};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (w.h_buf) (void)hipHostFree(w.h_buf);
    w = Workspace();
}

static void prover_release(const KzgSettings* s);  // (capi_prover.hpp)
extern "C" void kzg_settings_free(KzgSettings* s) {
    if (!s) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    small_free(s);  // the small-call queue's lanes (capi_coalesce.hpp)
    multi_free(s);  // communicators and peer handles first (each on its own device)
    for (KzgSettings* l : s->lanes) kzg_settings_free(l);
    s->lanes.clear();
    (void)hipSetDevice(s->device);
    ws_free(s->ws);
    if (s->d_eval_scratch) (void)hipFree(s->d_eval_scratch);
    if (s->d_proof) (void)hipFree(s->d_proof);
    if (s->d_proofs) (void)hipFree(s->d_proofs);
    if (s->d_proofs_out) (void)hipFree(s->d_proofs_out);
    if (s->h_proofs) (void)hipHostFree(s->h_proofs);
    prover_release(s);
    if (!s->borrowed) {  // (a lane reads its parent's tables)
        void* ptrs[] = {s->d_g1, s->d_g1_flag, s->d_g1_mult, s->d_g1_mult_aff, s->d_g1_fb_rows, s->d_fb_plan, s->d_g2, s->d_M, s->d_DM, s->d_M29, s->d_DM29, s->d_eval_a, s->d_eval_b, s->d_eval_c, s->d_tau4, s->d_prep, s->d_gen_mult, s->d_gen_mult_aff, s->prep.blob, s->verify.blob, s->verify2.blob, s->d_prep29, s->scalars.blob, s->verify3.blob, s->d_fixed_base};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : s->ev_copy)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : s->ev_slice)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {s->s_plain[0], s->s_plain[1], s->s_half[0], s->s_half[1], s->s_copy, s->s_aux})
        if (st) (void)hipStreamDestroy(st);
    delete s;
    if (prev >= 0) (void)hipSetDevice(prev);
    (void)hipGetLastError();
}
