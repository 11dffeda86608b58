// capi_pieces.hpp - pieces of the path exposed for parity tests and per-kernel benchmarks, settings accessors.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// ---------------------------------------------------------------- pieces
extern "C" KzgRet kzg_compute_challenges(uint8_t* z_out, const uint8_t* blobs, const uint8_t* commitments, size_t n,
                                         const KzgSettings* s) {
    if (!s || !z_out || !blobs || !commitments) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, STAGE_BLOBS);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    select_streams(s, n);  // before the staging copies: they must be on the stream the kernels of this launch run on
    const unsigned S = host_slices(n);
    if (S > 1) {  // the same sliced hand-over as the batch entry point (capi_verify.hpp): segments of the chain behind the slices
        const HostBatch host{blobs, commitments, nullptr};
        HIPCHK(hipEventRecord(s->ev[0], s->s1));
        if ((rc = sliced_points_copy(s, host, w.d_stage_cp, nullptr, n, s->ev[0])) != KZG_OK ||
            (rc = sliced_segments(s, host, w.d_stage_blobs, w.d_stage_cp, w.d_z, n, S, s->s1)) != KZG_OK) {
            const std::string msg = g_err;
            (void)hipStreamSynchronize(s->s_copy);
            (void)hipStreamSynchronize(s->s1);
            (void)hipGetLastError();
            g_err = msg;
            return rc;
        }
    } else {
        HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blobs, (size_t)BLOB_BYTES * n, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemcpyAsync(w.d_stage_cp, commitments, 48 * n, hipMemcpyHostToDevice, s->s1));
        if ((rc = launch_challenge(s, w.d_stage_blobs, w.d_stage_cp, w.d_z, n)) != KZG_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(w.h_buf, w.d_z, 32 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    for (size_t i = 0; i < n; i++) reverse32(z_out + 32 * i, w.h_buf + 32 * i);
    return KZG_OK;
}

static KzgRet evaluate_device_locked(void* d_y, const void* d_blobs, const void* d_z, size_t n, const KzgSettings* s,
                                     bool* any_bad) {
    Workspace& w = s->ws;
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4 * n, s->s1));
    HIPCHK(hipEventRecord(s->ev[7], s->s1));
    KzgRet rc = launch_evaluate(s, d_blobs, (const Fr*)d_z, (Fr*)d_y, w.d_status, n, /*alone=*/true);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[8], s->s1));
    uint32_t* h_status = reinterpret_cast<uint32_t*>(w.h_buf + 64 * n);
    HIPCHK(hipMemcpyAsync(h_status, w.d_status, 4 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[4], s->ev[7], s->ev[8]);
    *any_bad = false;
    for (size_t i = 0; i < n; i++) *any_bad |= h_status[i] != 0;
    return KZG_OK;
}

extern "C" KzgRet kzg_evaluate_polynomials_device(void* d_y, const void* d_blobs, const void* d_z, size_t n, const KzgSettings* s) {
    if (!s || !d_y || !d_blobs || !d_z) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    bool bad = false;
    if ((rc = evaluate_device_locked(d_y, d_blobs, d_z, n, s, &bad)) != KZG_OK) return rc;
    return bad ? fail(KZG_BADARGS, "Failed to parse G1Affine from bytes") : KZG_OK;
}

extern "C" KzgRet kzg_evaluate_polynomials(uint8_t* ys_out, const uint8_t* blobs, const uint8_t* zs, size_t n, const KzgSettings* s) {
    if (!s || !ys_out || !blobs || !zs) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, n, 1, STAGE_BLOBS);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    for (size_t i = 0; i < n; i++) reverse32(w.h_buf + 32 * i, zs + 32 * i);
    HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blobs, (size_t)BLOB_BYTES * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_z, w.h_buf, 32 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    bool bad = false;
    if ((rc = evaluate_device_locked(w.d_y, w.d_stage_blobs, w.d_z, n, s, &bad)) != KZG_OK) return rc;
    if (bad) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    HIPCHK(hipMemcpy(w.h_buf, w.d_y, 32 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) reverse32(ys_out + 32 * i, w.h_buf + 32 * i);
    return KZG_OK;
}

extern "C" KzgRet kzg_g1_decompress(uint8_t* status_out, uint8_t* xy_out, const uint8_t* points48, size_t n, const KzgSettings* s) try {
    if (!s || !status_out || !points48) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, (n + 1) / 2 + 1, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    HIPCHK(hipMemcpyAsync(w.d_bytes, points48, 48 * n, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, (int)n, w.d_points, w.d_pflag, (int)n, 1);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> st(n);
    HIPCHK(hipMemcpyAsync(st.data(), w.d_pflag, 4 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    for (size_t i = 0; i < n; i++) status_out[i] = (uint8_t)st[i];
    if (xy_out) {
        DevTmp xy;
        HIPCHK(hipMalloc(&xy.p, 96 * n));
        hipLaunchKernelGGL(k_aff_to_bytes, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_points, xy.as<uint8_t>(), (int)n);
        HIPCHK(hipMemcpyAsync(xy_out, xy.p, 96 * n, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
    }
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}

// ---------------------------------------------------------------- G1Projective::msm_variable_base (call sites src/kzg_proof.rs:419,429,430)
// Two entry points over one core:
//   kzg_g1_msm        n (point, scalar) pairs in host memory: decode + subgroup test + table rows of the call's points, then the sum
//   kzg_g1_msm_setup  n scalars over the HANDLE'S OWN Lagrange points (term i -> g1_points[i mod 4096]; capi_prover.hpp): the tables
//                     were made when the setup was loaded - no per-call decode - and large sums take the fixed-base form of
//                     msm_fixed.hpp (16-bit windows, half the bucket additions)
// The core runs on the SAME kernels as the verification path's MSM (rounds 1-4 built this entry's tables with the round-1
// kernels and accumulated full Jacobian additions from a global sorted list: 20 ns per term at n = 2^20 against the hot path's 7):
//   GLV split, then the window kernel in the hot path's SHAPE: the terms are cut into slices of at most 3 072 (a 1 024-blob
//   batch's output B holds 2 049) - so that every (window, slice) workgroup sorts its 4 x 3 072 list entries in LDS and
//   adds mixed (affine) entries, 8 windows x 2 S workgroups (the terms are dealt to the kernel's two outputs, halves of one
//   sum); the S window sums per window are folded by trees of 64 and the 8 windows combined as usual.
// timings: [2] the MSM (split, windows, reduce, folds, combine), [6] decode + tables (0 for the setup form).
constexpr size_t G1_MSM_SLICE_TERMS = 3072;
// nothing of a call may stay in flight on the stream when an error path releases the host buffers its copies read or write
struct StreamDrain {
    hipStream_t st;
    ~StreamDrain() {
        if (st) (void)hipStreamSynchronize(st);
    }
};
// the family's device scratch lives on the handle, grow-only (three or four hipMalloc + hipFree per call stalled every small-call
// lane of the device: hipFree waits for the whole device)
static KzgRet g1msm_scratch(const KzgSettings* s, size_t bytes, uint8_t** out) {
    Workspace& w = s->ws;
    if (bytes > w.cap_g1msm) {
        if (w.d_g1msm) (void)hipFree(w.d_g1msm);
        w.d_g1msm = nullptr;
        w.cap_g1msm = 0;
        HIPCHK(hipMalloc(&w.d_g1msm, bytes));
        w.cap_g1msm = bytes;
    }
    *out = w.d_g1msm;
    return KZG_OK;
}
struct G1MsmTables {
    const void* mult;       // table rows [MSM_CHUNKS][stride]
    const uint32_t* pflag;  // [stride]: non-zero -> the point adds nothing
    int stride;
    bool affine;            // G1Aff29Mem rows (Curve29Aff) | Jacobian rows of the handle's field form
    bool periodic;          // term i uses point i mod stride (the setup form) | point i
};
// the sum of n terms: canonical scalars at ws.d_scalars (destroyed: GLV split in place), compressed result -> out.  The caller holds
// the handle's lock, has reserved the workspace for (n + 1) / 2 + 1 "blobs" and recorded nothing on ev[2] / ev[3]; returns after the
// stream has drained.
static KzgRet g1_msm_core(const KzgSettings* s, size_t n, const G1MsmTables& tb, uint8_t out[48]) {
    Workspace& w = s->ws;
    // the terms dealt to the kernel's two outputs: [0, h) and [h, n); slices of at most slice_terms terms (option
    // g1_msm_slice_terms, default 3 072: 4 x 3 072 list entries = 48 KB of LDS, three workgroups per CU - measured at 2^20 terms
    // against 2 048: fewer (window, slice) workgroups to reduce and fold), their number a multiple of 4 (the XCD placement of
    // msm.hpp wants 2 S workgroup layers in eights)
    static const size_t slice_terms = (size_t)std::max(256L, std::min(3072L, ab_int("g1_msm_slice_terms", (long)G1_MSM_SLICE_TERMS)));
    const size_t h = (n + 1) / 2;
    unsigned S = (unsigned)((h + slice_terms - 1) / slice_terms);
    S = S <= 1 ? 1 : (S + 3) & ~3u;
    const unsigned W = MSM_WINDOWS / MSM_CHUNKS, gz = 2 * S, gz_pad = gz <= 64 ? gz : (gz + 63) & ~63u;
    KzgRet rc;
    if ((rc = msm_save_reserve(s, W, 1, gz)) != KZG_OK) return rc;
    // From 16 layers on (and while the save area holds them all: ~4 M terms) the tail of a LARGE sum: the slices folded bucket
    // by bucket, then W slots reduced with four lanes per addition (msm.hpp msm_large_tail) - 0.8 ms instead of 1.55 at 2^20
    // terms.  fold_per = the layers one thread adds in a row (more: fewer partial sums for the quads, a longer chain; measured at
    // 344 layers: 6 -> 0.166 + 0.160 ms, 11 -> 0.188 + 0.111, 16 -> 0.264 + 0.115, 22 -> 0.358 + 0.077 for fold + sum).
    static const int fold_per_opt = (int)std::max(2L, std::min(64L, ab_int("g1_msm_fold_per", 11)));
    const int fold_per = std::max(fold_per_opt, (int)((gz + MSM_FOLD_MAX_GROUPS - 1) / MSM_FOLD_MAX_GROUPS));
    int fold_gp = 0;
    (void)msm_large_tail_groups(gz, fold_per, &fold_gp);
    const bool large_tail = fp29_enabled() && gz >= 16 && ab_flag("g1_msm_large_tail", true) &&
                            w.cap_msm_save >= msm_save_layer_bytes(W, 1, MSM_SAVE2_WORDS) * gz;
    // scratch: window sums [gz_pad][W] | fold level A [gz_pad / 2][W] | fold level B [gz_pad / 4][W] | the large tail's
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_ws = up(sizeof(G1Jac) * (size_t)gz_pad * W), b_f0 = up(sizeof(G1Jac) * (size_t)std::max(1u, gz_pad / 2) * W),
                 b_f1 = up(sizeof(G1Jac) * (size_t)std::max(1u, gz_pad / 4) * W), b_tail = large_tail ? up(msm_large_tail_bytes(W, fold_gp)) : 0;
    uint8_t* scratch = nullptr;
    if ((rc = g1msm_scratch(s, b_ws + b_f0 + b_f1 + b_tail, &scratch)) != KZG_OK) return rc;
    G1Jac* const t_ws = reinterpret_cast<G1Jac*>(scratch);
    G1Jac* const t_f0 = reinterpret_cast<G1Jac*>(scratch + b_ws);
    G1Jac* const t_f1 = reinterpret_cast<G1Jac*>(scratch + b_ws + b_f0);
    uint32_t* const t_tail = reinterpret_cast<uint32_t*>(scratch + b_ws + b_f0 + b_f1);
    if (gz_pad != gz)  // (Z = 0: the identity) the padding of the window sums up to whole groups of 64
        HIPCHK(hipMemsetAsync(t_ws + (size_t)gz * W, 0, sizeof(G1Jac) * (size_t)(gz_pad - gz) * W, s->s1));
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    if (n) {
        hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->s1, w.d_scalars, (int)n);
        if (tb.periodic && (size_t)tb.stride == (size_t)FE_PER_BLOB)
            hipLaunchKernelGGL(k_commit_terms, dim3((unsigned)((2 * h + 255) / 256)), dim3(256), 0, s->s1, w.d_term_point, w.d_term_scalar, (int)(2 * h));
        else
            hipLaunchKernelGGL(k_plain_terms, dim3((unsigned)((2 * h + 255) / 256)), dim3(256), 0, s->s1, w.d_term_point, w.d_term_scalar, (int)(2 * h));
    }
    MsmDesc d{};
    d.mult = const_cast<void*>(tb.mult);
    d.pflag = const_cast<uint32_t*>(tb.pflag);
    d.scalars = w.d_scalars;
    d.term_point = w.d_term_point;  // output 1's list starts at entry max_terms = h: term_point[i] = i (or i mod 4096) serves both
    d.term_scalar = w.d_term_scalar;
    d.sorted = w.d_sorted;
    d.window_sums = t_ws;
    d.nterms[0] = (int)h;
    d.nterms[1] = (int)(n - h);
    d.max_terms = (int)(h ? h : 1);
    d.stride = tb.stride;
    d.slices = (int)S;
    d.chunks = MSM_CHUNKS;
    d.chunks_per_block = MSM_CHUNKS;  // one workgroup per (window, slice): the four chunks' entries in one sorted list
    d.flags = (gz & 7) == 0 ? MSM_FLAG_XCD : 0;
    if (tb.affine) msm_window_launch<Curve29Aff, true>(d, W, 1, gz, w.d_msm_save, w.cap_msm_save, s->s1, !large_tail);
    else if (fp29_enabled()) msm_window_launch<Curve29, true>(d, W, 1, gz, w.d_msm_save, w.cap_msm_save, s->s1, !large_tail);
#if KZG_AB_VARIANTS
    else msm_window_launch<Curve32, true>(d, W, 1, gz, w.d_msm_save, w.cap_msm_save, s->s1);
#endif
    if (large_tail) HIPCHK(msm_large_tail(w.d_msm_save, W, gz, fold_per, t_tail, t_ws, s->s1));
    // window sums [2 S (padded to whole groups of 64 with identities)][W] -> [1][W]: trees over up to 64 slices at a time.  A level
    // with more than 64 inputs reads them in whole groups of 64: the tail of the last group is zeroed here when the level below
    // wrote a count that is not a multiple of 64 (gz = 4 160 -> 65 partial sums: round 5 read 63 uninitialised points there)
    const G1Jac* cur = t_ws;
    G1Jac* bufs[2] = {t_f0, t_f1};
    int which = 0;
    for (unsigned left = large_tail ? 1 : gz_pad; left > 1;) {
        const unsigned f = std::min(left, 64u), groups = (left + f - 1) / f;
        if (groups > 64 && (groups & 63)) {
            const unsigned pad = ((groups + 63) & ~63u) - groups;
            HIPCHK(hipMemsetAsync(bufs[which] + (size_t)groups * W, 0, sizeof(G1Jac) * (size_t)pad * W, s->s1));
        }
        hipLaunchKernelGGL(k_msm_fold_slices, dim3(groups * W), dim3(64), 0, s->s1, cur, bufs[which], (int)f, (int)W);
        cur = bufs[which];
        which ^= 1;
        left = groups;
    }
    // 8 windows of one output: the Horner chain with four lanes per doubling / addition (0.8 -> ~0.2 ms of a 2^20-term call)
    if (fp29_enabled()) hipLaunchKernelGGL(k_msm_combine_quad, dim3(1), dim3(64), 0, s->s1, cur, w.d_ab, (int)W);
    else hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, s->s1, cur, w.d_ab, 1, (int)W);
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    hipLaunchKernelGGL(k_jac_compress, dim3(1), dim3(64), 0, s->s1, w.d_ab, w.d_bytes, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, w.d_bytes, 48, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    return KZG_OK;
}
// scalars: big-endian, any value below 2^256 -> canonical limbs at ws.d_scalars, reduced on the device (round 5 reduced and
// reversed them on the host: 10 ms of a 2^20-term call); staged behind `skip` bytes of ws.d_bytes
static KzgRet g1_msm_scalars_in(const KzgSettings* s, const uint8_t* scalars, size_t n, size_t skip) {
    Workspace& w = s->ws;
    uint8_t* const stage = w.d_bytes + ((skip + 15) & ~(size_t)15);
    HIPCHK(hipMemcpyAsync(stage, scalars, 32 * n, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_scalars_reduce_be, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->s1, stage, w.d_scalars, (int)n);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}
extern "C" KzgRet kzg_g1_msm(uint8_t out[48], const uint8_t* points48, const uint8_t* scalars, size_t n, const KzgSettings* s) try {
    if (!s || !out || (n && (!points48 || !scalars))) return fail(KZG_BADARGS, "null argument");
    if (n > ((size_t)1 << 26)) return fail(KZG_BADARGS, "kzg_g1_msm: more than 2^26 terms");  // (a list entry holds a 27-bit point index)
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, (n + 1) / 2 + 1, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    const int np = (int)(n ? n : 1);  // table stride
    const bool aff = msm_affine_enabled();
    std::vector<uint32_t> st(n);
    StreamDrain drain{s->s1};  // (declared after `st`: destroyed - and the stream drained - before it)
    HIPCHK(hipEventRecord(s->ev[5], s->s1));
    if (n) {
        HIPCHK(hipMemcpyAsync(w.d_bytes, points48, 48 * n, hipMemcpyHostToDevice, s->s1));
        if ((rc = g1_msm_scalars_in(s, scalars, n, 48 * n)) != KZG_OK) return rc;
        HIPCHK(hipEventRecord(s->ev[5], s->s1));
        const unsigned blocks256 = (unsigned)((n + 255) / 256);
        if (aff) {
            hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, true>), dim3(blocks256), dim3(256), 256 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s1, w.d_bytes,
                               w.d_bytes, (int)n, w.d_points, w.d_pflag, w.d_mult, w.d_jtmp, (int)n, np);
            const unsigned conv_blocks = (unsigned)((n + 64 * AFFINE_BATCH - 1) / (64 * AFFINE_BATCH));
            hipLaunchKernelGGL(k_mult_to_affine29, dim3(conv_blocks), dim3(64), 0, s->s1, w.d_jtmp, w.d_pflag, (G1Aff29Mem*)w.d_mult, (int)n, np);
        } else if (fp29_enabled()) {
            hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s1,
                               w.d_bytes, w.d_bytes, (int)n, w.d_points, w.d_pflag, w.d_mult, (G1Jac29Mem*)nullptr, (int)n, np);
        }
#if KZG_AB_VARIANTS
        else
            hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, (int)n, w.d_points,
                               w.d_pflag, (G1Jac*)w.d_mult, (int)n, np);
#endif
        HIPCHK(hipGetLastError());
        // (a point outside G1 has the digit 0 in every window - the kernel reads its flag - so the verdict on the inputs is
        // looked at after the sum: one wait at the end)
        HIPCHK(hipMemcpyAsync(st.data(), w.d_pflag, 4 * n, hipMemcpyDeviceToHost, s->s1));
    }
    HIPCHK(hipEventRecord(s->ev[6], s->s1));
    const G1MsmTables tb{w.d_mult, w.d_pflag, np, aff, false};
    if ((rc = g1_msm_core(s, n, tb, out)) != KZG_OK) return rc;
    elapsed(&s->timings[6], s->ev[5], s->ev[6]);
    for (size_t i = 0; i < n; i++)
        if (st[i] == G1_INVALID) return fail(KZG_BADARGS, "invalid G1 point");
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}

extern "C" KzgRet kzg_g1_mul_generator(uint8_t* out48, const uint8_t* scalars, size_t n, const KzgSettings* s) try {
    if (!s || (n && (!out48 || !scalars))) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    std::vector<uint8_t> le(32 * n);
    for (size_t i = 0; i < n; i++) {
        uint8_t t[32];
        memcpy(t, scalars + 32 * i, 32);
        while (be_geq_r(t)) be_sub_r(t);
        reverse32(le.data() + 32 * i, t);
    }
    DevTmp ts, to;
    HIPCHK(hipMalloc(&ts.p, 32 * n));
    HIPCHK(hipMalloc(&to.p, 48 * n));
    Fr* d_s = ts.as<Fr>();
    uint8_t* d_o = to.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(d_s, le.data(), 32 * n, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_mul_generator, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s->s1, d_s, d_o, (int)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out48, d_o, 48 * n, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}

extern "C" KzgRet kzg_pairing_check(bool* ok, const uint8_t a[48], const uint8_t b[48], const KzgSettings* s) {
    if (!ok || !a || !b || !s) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, 2, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    HIPCHK(hipMemcpyAsync(w.d_bytes, a, 48, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_bytes + 48, b, 48, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3(1), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, 2, w.d_points, w.d_pflag, 2, 0);
    hipLaunchKernelGGL(k_aff_to_slp, dim3(1), dim3(64), 0, s->s1, w.d_points, w.d_pflag, w.d_slp_in);
    HIPCHK(hipGetLastError());
    uint32_t* h = reinterpret_cast<uint32_t*>(w.h_buf);
    HIPCHK(hipMemcpyAsync(h, w.d_pflag, 8, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    if ((rc = run_verify(s, w.d_slp_in, w.d_slp_out, 1, s->s1)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[4], s->s1));
    HIPCHK(hipMemcpyAsync(h + 2, w.d_slp_out, sizeof(Fp) * 6, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[3], s->ev[3], s->ev[4]);
    if (h[0] == G1_INVALID || h[1] == G1_INVALID) return fail(KZG_BADARGS, "invalid G1 point");
    uint32_t any = 0;
    for (int i = 0; i < 72; i++) any |= h[2 + i];
    *ok = any == 0;
    return KZG_OK;
}

// pairings_verify(a1, a2, b1, b2) (src/pairings.rs:5-9, public at src/lib.rs:15) with ARBITRARY G2 arguments:
//     multi_miller_loop([(-a1, prep(a2)), (b1, prep(b2))]).final_exponentiation() == 1   <=>   e(a1, a2) == e(b1, b2).
// The reference takes decoded G1Affine / G2Affine; across the C ABI they are compressed bytes, decoded on the device like
// from_compressed_unchecked (on the curve, no subgroup test - a typed Rust value carries that invariant already); an
// undecodable point is KZG_BADARGS.  The PREP program that makes a G2 point's 68 line triples (the reference's
// G2Prepared::from, which it runs on every call) runs here on both G2 arguments - two instances, one launch - and VERIFY
// takes those lines instead of the handle's.  An identity G2 argument makes its pair contribute 1, as the reference's
// skipped pair does: its G1 partner is replaced by the identity and the lines by the generator's.
extern "C" KzgRet kzg_pairings_verify(bool* ok, const uint8_t a1[48], const uint8_t a2[96], const uint8_t b1[48], const uint8_t b2[96],
                                      const KzgSettings* s) {
    if (!ok || !a1 || !a2 || !b1 || !b2 || !s) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    KzgRet rc = ws_reserve(s, 2, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    const size_t n_lines = (size_t)2 * s->prep.p.n_out;
    DevTmp t_g2b, t_q, t_flag, t_lines, t_lines29;
    HIPCHK(hipMalloc(&t_g2b.p, 192));
    HIPCHK(hipMalloc(&t_q.p, sizeof(Fp) * 8));
    HIPCHK(hipMalloc(&t_flag.p, 8));
    HIPCHK(hipMalloc(&t_lines.p, sizeof(Fp) * n_lines));
    HIPCHK(hipMalloc(&t_lines29.p, (size_t)64 * n_lines));
    auto g2_is_identity_encoding = [](const uint8_t* b) {
        if (b[0] != 0xC0) return false;
        for (int i = 1; i < 96; i++)
            if (b[i]) return false;
        return true;
    };
    const bool inf[2] = {g2_is_identity_encoding(a2), g2_is_identity_encoding(b2)};
    uint8_t* h = w.h_buf;  // pinned: [a1 | b1 | a2 | b2] in, then flags and the program's output
    memcpy(h, a1, 48);
    memcpy(h + 48, b1, 48);
    memcpy(h + 96, a2, 96);
    memcpy(h + 192, b2, 96);
    HIPCHK(hipMemcpyAsync(w.d_bytes, h, 96, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(t_g2b.p, h + 96, 192, hipMemcpyHostToDevice, s->s1));
    hipLaunchKernelGGL(k_g1_decode, dim3(1), dim3(64), 0, s->s1, w.d_bytes, w.d_bytes, 2, w.d_points, w.d_pflag, 2, 0);
    hipLaunchKernelGGL(k_g2_decompress_n, dim3(2), dim3(64), 0, s->s1, t_g2b.as<uint8_t>(), t_q.as<Fp>(), t_flag.as<uint32_t>());
    HIPCHK(hipGetLastError());
    uint32_t* hf = reinterpret_cast<uint32_t*>(h + 320);  // [g1 flags 2 | g2 flags 2 | out 72]
    HIPCHK(hipMemcpyAsync(hf, w.d_pflag, 8, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipMemcpyAsync(hf + 2, t_flag.p, 8, hipMemcpyDeviceToHost, s->s1));
    static const uint32_t one_flag = G1_INFINITY;
    for (int k = 0; k < 2; k++)
        if (inf[k]) {  // e(P, O) = 1: the pair leaves the product
            hipLaunchKernelGGL(k_g2_generator, dim3(1), dim3(64), 0, s->s1, t_q.as<Fp>() + 4 * k);
            HIPCHK(hipMemcpyAsync(w.d_pflag + k, &one_flag, 4, hipMemcpyHostToDevice, s->s1));
        }
    hipLaunchKernelGGL(k_aff_to_slp, dim3(1), dim3(64), 0, s->s1, w.d_points, w.d_pflag, w.d_slp_in);
    HIPCHK(hipGetLastError());
    if ((rc = run_program(s->prep, t_q.as<Fp>(), nullptr, t_lines.as<Fp>(), 2, s->s1)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    if (pairing_latency_form(1)) {
        hipLaunchKernelGGL(k_fp_to_fp29mem, dim3((unsigned)((n_lines + 63) / 64)), dim3(64), 0, s->s1, t_lines.as<Fp>(), t_lines29.as<uint32_t>(), (int)n_lines);
        HIPCHK(hipGetLastError());
        rc = run_program2(s->verify2, w.d_slp_in, t_lines29.as<uint32_t>(), w.d_slp_out, 1, s->s1);
    } else {
        rc = run_program(s->verify, w.d_slp_in, t_lines.as<Fp>(), w.d_slp_out, 1, s->s1);
    }
    if (rc != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[4], s->s1));
    HIPCHK(hipMemcpyAsync(hf + 4, w.d_slp_out, sizeof(Fp) * 6, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[3], s->ev[3], s->ev[4]);
    if (hf[0] == G1_INVALID || hf[1] == G1_INVALID) return fail(KZG_BADARGS, "invalid G1 point");
    if (hf[2] == G1_INVALID || hf[3] == G1_INVALID) return fail(KZG_BADARGS, "invalid G2 point");
    uint32_t any = 0;
    for (int i = 0; i < 72; i++) any |= hf[4 + i];
    *ok = any == 0;
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_root_of_unity(const KzgSettings* s, size_t i, uint8_t out[32]) {
    if (!s || !out || i >= FE_PER_BLOB) return fail(KZG_BADARGS, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    // the table holds w*R mod r; strip the Montgomery factor with one host-side REDC (test/diagnostic path only)
    Fr m;
    HIPCHK(hipMemcpy(&m, s->d_M + i, sizeof(Fr), hipMemcpyDeviceToHost));
    // host Montgomery reduction of one element: t = m * R^-1 mod r with 64-bit arithmetic
    const uint32_t* MOD = consts::FR_MOD;
    uint32_t t[9] = {0};
    for (int k = 0; k < 8; k++) t[k] = m.l[k];
    for (int k = 0; k < 8; k++) {
        uint32_t q = t[0] * FR_INV32;
        uint64_t c = ((uint64_t)q * MOD[0] + t[0]) >> 32;
        for (int j = 1; j < 8; j++) {
            uint64_t x = (uint64_t)q * MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c;
        t[7] = (uint32_t)x;
        t[8] = (uint32_t)(x >> 32);
    }
    uint8_t be[32];
    for (int k = 0; k < 8; k++) {
        be[4 * (7 - k)] = (uint8_t)(t[k] >> 24); be[4 * (7 - k) + 1] = (uint8_t)(t[k] >> 16);
        be[4 * (7 - k) + 2] = (uint8_t)(t[k] >> 8); be[4 * (7 - k) + 3] = (uint8_t)t[k];
    }
    if (t[8] || be_geq_r(be)) be_sub_r(be);
    memcpy(out, be, 32);
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_tau_g2(const KzgSettings* s, uint8_t out[96]) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    DevTmp t;
    HIPCHK(hipMalloc(&t.p, 96));
    uint8_t* d = t.as<uint8_t>();
    hipLaunchKernelGGL(k_g2_compress, dim3(1), dim3(64), 0, s->s1, s->d_tau4, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 96, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_g1_point(const KzgSettings* s, size_t i, uint8_t out[48]) {
    if (!s || !out || i >= FE_PER_BLOB) return fail(KZG_BADARGS, "bad argument");
    if (!s->d_g1) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    DevTmp t;
    HIPCHK(hipMalloc(&t.p, 48));
    uint8_t* d = t.as<uint8_t>();
    hipLaunchKernelGGL(k_aff_compress, dim3(1), dim3(64), 0, s->s1, s->d_g1 + i, s->d_g1_flag + i, d, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 48, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    return KZG_OK;
}

extern "C" KzgRet kzg_settings_g2_point(const KzgSettings* s, size_t i, uint8_t out[96]) {
    if (!s || !out) return fail(KZG_BADARGS, "bad argument");
    if (!s->d_g2 || i >= s->n_g2) return fail(KZG_BADARGS, "no such G2 point in these settings");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    DevTmp t;
    HIPCHK(hipMalloc(&t.p, 96));
    uint8_t* d = t.as<uint8_t>();
    hipLaunchKernelGGL(k_g2_compress, dim3(1), dim3(64), 0, s->s1, s->d_g2 + 4 * i, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d, 96, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    return KZG_OK;
}

extern "C" KzgRet kzg_pairing_check(bool* ok, const uint8_t a[48], const uint8_t b[48], const KzgSettings* s);
// is_trusted_setup_in_lagrange_form (build.rs:107-129; its result is discarded by the reference's loader):
// e(g1[1], g2[0]) == e(g1[0], g2[1]) on the points in FILE order - true for a monomial-form G1 section, false for the
// Lagrange-form file the crate ships.
extern "C" KzgRet kzg_settings_is_monomial_form(bool* ok, const KzgSettings* s) {
    if (!s || !ok) return fail(KZG_BADARGS, "null argument");
    if (!s->d_g1) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    return kzg_pairing_check(ok, s->g1_first[0], s->g1_first[1], s);  // e(g1[0], [tau]G2) == e(g1[1], G2)
}
