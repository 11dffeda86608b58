// capi_verify.hpp - the verification path: kernel selection, workspace, MSM tail, the three phases as launch + wait, and the entry points.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// The evaluation kernel over T blobs on stream s1 (radix-2^29 form; option evaluate_kernel=32 of the A/B build selects the 8x32
// form, kept for measurement and as a cross-check).
// alone: the evaluation is the whole call (kzg_evaluate_polynomials*, BASELINE configs[2]) - nothing else wants the CUs.
// stamp: the kernel records its own execution interval in the workspace's stamp words (phase 1 has zeroed them)
static KzgRet launch_evaluate(const KzgSettings* s, const void* d_blobs, const Fr* d_z, Fr* d_y, uint32_t* d_status, size_t T, bool alone = false,
                              bool stamp = false) {
    unsigned long long* const kt = stamp && s->ws.d_ktime ? s->ws.d_ktime + 4 : nullptr;
#if KZG_AB_VARIANTS
    static const bool use32 = opt_is("evaluate_kernel", "32");
    if (use32) {
        hipLaunchKernelGGL(k_blob_evaluate32, dim3((unsigned)T), dim3(64), 0, s->s1, (const uint8_t*)d_blobs, d_z, s->d_M, s->d_DM, d_y, d_status);
        return KZG_OK;
    }
#endif
    if (T > s->eval_scratch_cap) {  // 576 bytes per blob between the three kernels (fr_kernels.hpp); callers hold the handle's lock
        if (s->d_eval_scratch) HIPCHK(hipFree(s->d_eval_scratch));  // waits for the kernels that may still read it
        s->d_eval_scratch = nullptr;
        s->eval_scratch_cap = 0;
        const size_t cap = T < 1024 ? 1024 : T;
        HIPCHK(hipMalloc(&s->d_eval_scratch, 4 * (size_t)EVAL_SCRATCH_WORDS * cap));
        s->eval_scratch_cap = cap;
    }
    const unsigned per_lane = (unsigned)((T + 63) / 64);
    hipLaunchKernelGGL(k_eval_powers, dim3(per_lane), dim3(64), 0, s->s1, d_z, s->d_eval_scratch, (int)T);
    // A launch that leaves CUs free asks for enough (unused) dynamic LDS that no CU takes a second workgroup: the dispatcher
    // otherwise pairs workgroups on half the CUs, two wavefronts per SIMD, and the single batch waits twice as long.
    const unsigned eval_blocks = (unsigned)((T + EVAL_BLOBS_PER_BLOCK - 1) / EVAL_BLOBS_PER_BLOCK);
    // A LARGE launch inside a verification asks for 16 KB of (unused) dynamic LDS per workgroup: 38.9 + 16 KB caps a CU at two
    // workgroups = two wavefronts per SIMD, although the kernel's 166 VGPRs would allow three.  Alone on the chip three are
    // faster (14.5 against 15.3 ms per 262 144 blobs), but beside the other launch groups' kernels three workgroups hold
    // 117 KB of a CU's 160 KB of LDS and the MSM window blocks (36-46 KB each) queue behind them: 4.74-4.82 M blobs/s with
    // three, 4.91-4.97 M with two (profiles/r4_ab_evaluate.txt).  Option eval_lds_pad=<bytes> overrides (measurement).
    static const long pad_opt = ab_int("eval_lds_pad", -1);
    const size_t lds_pad = pad_opt >= 0 ? (size_t)std::min(120L * 1024, pad_opt) : alone ? 0 : 16384;
    // (the attribute is a per-kernel maximum that dyn_lds_ensure only raises: concurrent launches with different paddings cannot
    // fail each other; a launch whose request could not be raised goes without its padding - slower placement, same result)
    size_t spread_lds = eval_blocks <= (unsigned)s->n_cus ? (size_t)EVAL_SPREAD_LDS : lds_pad;
    if (spread_lds) {
        bool attr_ok = DYN_LDS(k_blob_evaluate_t<true>, spread_lds) == hipSuccess;
#if KZG_AB_VARIANTS
        attr_ok = DYN_LDS(k_blob_evaluate_t<false>, spread_lds) == hipSuccess && attr_ok;
#endif
        if (!attr_ok) {
            (void)hipGetLastError();  // nothing sticky is left for the callers' checks
            spread_lds = 0;
        }
    }
#if KZG_AB_VARIANTS
    static const bool eval_r3 = opt_is("evaluate_kernel", "r3");  // round 3's kernel: 192 VGPRs, two wavefronts per SIMD (A/B measurement)
    if (eval_r3)
        hipLaunchKernelGGL(k_blob_evaluate_t<false>, dim3(eval_blocks), dim3(64 * EVAL_BLOBS_PER_BLOCK), spread_lds, s->s1,
                           (const uint8_t*)d_blobs, EvalTables{s->d_eval_a, s->d_eval_b, s->d_eval_c}, s->d_eval_scratch, d_status, (int)T, kt);
    else
#endif
        hipLaunchKernelGGL(k_blob_evaluate_t<true>, dim3(eval_blocks), dim3(64 * EVAL_BLOBS_PER_BLOCK), spread_lds, s->s1,
                           (const uint8_t*)d_blobs, EvalTables{s->d_eval_a, s->d_eval_b, s->d_eval_c}, s->d_eval_scratch, d_status, (int)T, kt);
    hipLaunchKernelGGL(k_eval_finish, dim3(per_lane), dim3(64), 0, s->s1, s->d_eval_scratch, d_y, (int)T);
    return KZG_OK;
}

// The challenge kernel over T blobs on stream s1: a producer / consumer form (the serial chain split over wavefronts, lowest
// latency) while every workgroup can have a CU to itself, the one-lane-per-blob form (highest throughput) beyond that.
// Small launches take the form with two lanes per blob on the consumer side (k_blob_challenge_split2), mid-size ones the
// one-lane consumer (k_blob_challenge_split); option challenge_kernel = lane | split | split2 forces a form (A/B
// measurement, cross-check in the tests).
static KzgRet launch_challenge(const KzgSettings* s, const void* d_blobs, const void* d_commitments, Fr* d_z, size_t T, hipStream_t st = nullptr) {
    if (!st) st = s->s1;
    static const int forced = opt_is("challenge_kernel", "lane") ? 1 : opt_is("challenge_kernel", "split") ? 2 : opt_is("challenge_kernel", "split2") ? 3 : 0;
    const uint8_t *bl = (const uint8_t*)d_blobs, *cm = (const uint8_t*)d_commitments;
    // measured on MI355X (tools/prof/challenge_forms_rate.py, ms for 1 024 / 16 384 / 32 768 / 49 152 blobs): lane 5.2 / 6.5 / 6.9 / 12.0,
    // split 3.5 / 5.0 / 5.4 / 8.8, split2 2.9 / 4.4 / 7.2 / 13.3 - three waves per 64 blobs stop paying once the CUs hold more
    // than one workgroup each; the one-lane-per-blob form wins when the launch fills every SIMD several times over
    const int form = forced ? forced : T <= 16384 ? 3 : T <= 49152 ? 2 : 1;
    s->ws.ktime_valid = false;
    if (form == 1) {
        unsigned long long* kt = s->ws.d_ktime;  // null for callers that never reserved the workspace
        if (kt) {
            HIPCHK(hipMemsetAsync(kt, 0, 32, st));
            s->ws.ktime_valid = true;
        }
        // 4 wavefronts per SIMD (114 VGPRs; the compiler's own choice was 146 = 3): the whole launch is resident at once.
        // Measured and not taken: 5 wavefronts (spills: -2.3 %), the launch as 2 / 4 back-to-back halves / quarters
        // so that the chain's waves leave registers for the other kernels of the pipeline (-2.6 % / -3.5 %).
#if KZG_AB_VARIANTS
        static const long occ = opt_int("challenge_occ", 4);
        if (occ == 3) hipLaunchKernelGGL(k_blob_challenge_t<3>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, bl, cm, d_z, (int)T, kt);
        else if (occ == 5) hipLaunchKernelGGL(k_blob_challenge_t<5>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, bl, cm, d_z, (int)T, kt);
        else
#endif
            hipLaunchKernelGGL(k_blob_challenge_t<4>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, bl, cm, d_z, (int)T, kt);
    } else if (form == 2) {
        hipLaunchKernelGGL(k_blob_challenge_split, dim3((unsigned)((T + 63) / 64)), dim3(128), 0, st, bl, cm, d_z, (int)T);
    } else {
        hipLaunchKernelGGL(k_blob_challenge_split2_t<false>, dim3((unsigned)((T + 63) / 64)), dim3(192), 0, st, bl, cm, d_z, (int)T, 0, 1024, (uint32_t*)nullptr);
    }
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// stage: host-input staging wanted - STAGE_BLOBS (T blobs + their commitments / proofs) or STAGE_CP (commitments / proofs only)
enum { STAGE_NONE = 0, STAGE_BLOBS = 1, STAGE_CP = 2 };
static KzgRet ws_reserve(const KzgSettings* s, size_t T, size_t B, int stage) {
    Workspace& w = s->ws;
    if (T > w.cap_n || B > w.cap_b) {
        size_t keep_stage = w.cap_stage, keep_cp = w.cap_stage_cp, keep_h = w.cap_hstage;
        uint8_t *sb = w.d_stage_blobs, *sc = w.d_stage_cp, *h0 = w.d_hstage[0], *h1 = w.d_hstage[1];
        w.d_stage_blobs = nullptr;
        w.d_stage_cp = nullptr;
        w.d_hstage[0] = w.d_hstage[1] = nullptr;
        size_t capT = T > w.cap_n ? T : w.cap_n, capB = B > w.cap_b ? B : w.cap_b;
        ws_free(w);
        w.d_stage_blobs = sb;
        w.d_stage_cp = sc;
        w.cap_stage = keep_stage;
        w.cap_stage_cp = keep_cp;
        w.d_hstage[0] = h0;
        w.d_hstage[1] = h1;
        w.cap_hstage = keep_h;
        if (capT < 16) capT = 16;
        const size_t np = 2 * capT + 1;            // points: C's, pi's, generator
        const size_t nsc = 2 * capT + capB;        // scalars: (2n+1) per batch
        const size_t nterm = 4 * capT + 2 * capB;  // term table rows: [2B][2n+1]
        HIPCHK(hipMalloc(&w.d_z, sizeof(Fr) * capT));
        HIPCHK(hipMalloc(&w.d_y, sizeof(Fr) * capT));
        HIPCHK(hipMalloc(&w.d_scalars, sizeof(Fr) * nsc));
        HIPCHK(hipMalloc(&w.d_partial, sizeof(Fr) * ((capT + 255) / 256 + capB)));
        HIPCHK(hipMalloc(&w.d_r, sizeof(Fr) * capB));
        HIPCHK(hipMalloc(&w.d_status, 4 * capT));
        HIPCHK(hipMalloc(&w.d_pflag, 4 * np));
        HIPCHK(hipMalloc(&w.d_term_point, 4 * nterm));
        HIPCHK(hipMalloc(&w.d_term_scalar, 4 * nterm));
        HIPCHK(hipMalloc(&w.d_sorted, 4 * MSM_WINDOWS * nterm));
        HIPCHK(hipMalloc(&w.d_points, sizeof(G1Aff) * np));
        HIPCHK(hipMalloc(&w.d_window, sizeof(G1Jac) * 2 * MSM_WINDOWS * capB));
        HIPCHK(hipMalloc(&w.d_window_sl, sizeof(G1Jac) * 2 * MSM_WINDOWS * MSM_MAX_SLICES * 4));  // sliced launches have <= 4 batches
        HIPCHK(hipMalloc(&w.d_mult, MULT_ENTRY_BYTES * std::max((size_t)MSM_CHUNKS * np, (size_t)MSM_CHUNKS_LATENCY * std::min(np, (size_t)(2 * LATENCY_MAX_BLOBS + 1)))));
        HIPCHK(hipMalloc(&w.d_ab, sizeof(G1Jac) * 2 * capB));
        HIPCHK(hipMalloc(&w.d_ktime, 128));
        if (msm_affine_enabled()) HIPCHK(hipMalloc(&w.d_jtmp, sizeof(G1Jac29Mem) * np));
        HIPCHK(hipMalloc(&w.d_parts, sizeof(G1Jac) * 2 * capB * MAX_WORLD));
        HIPCHK(hipMalloc(&w.d_send, sizeof(G1Jac) * 2 * MAX_WORLD));
        HIPCHK(hipMalloc(&w.d_slp_in, sizeof(Fp) * 6 * capB));
        HIPCHK(hipMalloc(&w.d_slp_out, sizeof(Fp) * 6 * capB));
        HIPCHK(hipMalloc(&w.d_bytes, 96 * np));
        HIPCHK(hipMalloc(&w.d_records, 160 * capT));
        HIPCHK(hipMalloc(&w.d_sha_mid, 32 * std::min(capT, (size_t)SLICED_MAX_BLOBS)));
        w.off_r = 256 * capT + 4096;                 // pinned layout: [per-blob area | r | own partials | out | gathered partials]
        w.off_part = w.off_r + 32 * capB;
        w.off_out = w.off_part + 288 * capB;
        w.off_parts = w.off_out + 288 * capB;
        w.h_cap = w.off_parts + 288 * capB * MAX_WORLD;
        HIPCHK(hipHostMalloc(&w.h_buf, w.h_cap));
        w.cap_n = capT;
        w.cap_b = capB;
    }
    if (stage != STAGE_NONE && T > w.cap_stage_cp) {  // 96 bytes per tuple: what the proof-tuple entry points stage
        if (w.d_stage_cp) (void)hipFree(w.d_stage_cp);
        w.d_stage_cp = nullptr;
        w.cap_stage_cp = 0;
        size_t cap = T < 4 ? 4 : T;
        HIPCHK(hipMalloc(&w.d_stage_cp, 96 * cap));
        w.cap_stage_cp = cap;
    }
    if (stage == STAGE_BLOBS && T > w.cap_stage) {  // 128 KiB per blob: only the host-blob entry points pay for it
        if (w.d_stage_blobs) (void)hipFree(w.d_stage_blobs);
        w.d_stage_blobs = nullptr;
        w.cap_stage = 0;
        size_t cap = T < 4 ? 4 : T;
        HIPCHK(hipMalloc(&w.d_stage_blobs, (size_t)BLOB_BYTES * cap));
        w.cap_stage = cap;
    }
    return KZG_OK;
}

// (window, chunk) blocks of the MSM: separate while the launch has few batches (latency), merged per window once the
// batch dimension alone fills the chip (msm.hpp MsmDesc::chunks_per_block); option msm_cpb = 1 | 2 | 4 overrides.
static int msm_chunks_per_block(size_t B) {
    static const int forced = [] {
        const int v = (int)ab_int("msm_cpb", 0);
        return (v == 1 || v == 2 || v == 4) ? v : 0;
    }();
    if (forced) return forced;
    return B >= 32 ? 4 : B >= 16 ? 2 : 1;
}

// The window kernel's save area (msm.hpp MsmDesc::save): grow-only, one z-layer at least, 512 MiB at most - a larger grid
// is launched in z-pieces by msm_window_launch.
static KzgRet msm_save_reserve(const KzgSettings* s, unsigned gx, unsigned gy, unsigned gz) {
    Workspace& w = s->ws;
    const size_t layer = msm_save_layer_bytes(gx, gy, fp29_enabled() ? MSM_SAVE2_WORDS : Curve32::WORDS);  // (the larger of the forms' point sizes)
    const size_t want = std::max(layer, std::min(layer * gz, (size_t)512 << 20));
    if (want > w.cap_msm_save) {
        if (w.d_msm_save) (void)hipFree(w.d_msm_save);
        w.d_msm_save = nullptr;
        w.cap_msm_save = 0;
        HIPCHK(hipMalloc(&w.d_msm_save, want));
        w.cap_msm_save = want;
    }
    return KZG_OK;
}

// ---------------------------------------------------------------- the tail: MSM + pairing
// Group of B batches of n blobs (T = B n).  scalars of batch b at b(2n+1): a [0,n), b [n,2n), g at 2n;
// points: C [0,T), pi [T,2T), G at 2T, multiples with stride 2T+1.  Leaves (A, B) of batch b in ws.d_ab[2b..].
static KzgRet run_msm(const KzgSettings* s, size_t n, size_t B) {
    Workspace& w = s->ws;
    const int T = (int)(n * B), mt = (int)(2 * n + 1);
    hipLaunchKernelGGL(k_batch_terms, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, s->s1, w.d_term_point,
                       w.d_term_scalar, (int)n, T, mt);
    MsmDesc d{};
    d.mult = w.d_mult;
    d.pflag = w.d_pflag;
    d.scalars = w.d_scalars;
    d.term_point = w.d_term_point;
    d.term_scalar = w.d_term_scalar;
    d.sorted = w.d_sorted;
    d.window_sums = w.d_window;
    d.nterms[0] = (int)n;
    d.nterms[1] = (int)(2 * n + 1);
    d.max_terms = mt;
    d.stride = 2 * T + 1;
    d.ktime = w.kstamps_valid ? w.d_ktime + 8 : nullptr;
    d.chunks = w.chunks;
    d.chunks_per_block = w.chunks == MSM_CHUNKS ? msm_chunks_per_block(B) : 1;
    // option msm_xcd=0: A/B measurement of the XCD placement (msm.hpp MSM_FLAG_XCD; profiles/r3_ab_msm.txt: +2 % throughput)
    static const int msm_flags = ab_flag("msm_xcd", true) ? MSM_FLAG_XCD : 0;
    d.flags = msm_flags;
    const unsigned slots = d.chunks / d.chunks_per_block, W = MSM_WINDOWS / d.chunks;
    // one large batch: slice the terms of an output over several workgroups until the launch has ~1000 of them
    // (each slice keeps >= 1024 terms of the smaller output)
    unsigned S = 1;
    while (S < MSM_MAX_SLICES && W * slots * 2 * B * S < 768 && n / (2 * S) >= 1024 && 2 * S * B <= 128) S *= 2;
    // a small launch (the latency layout: one batch, 64 blocks) is bound by its fullest bucket - ~18 of 2 049 terms, 15 us
    // a Jacobian addition at lone-wave speed: four slices quarter that chain for one more short fold
    static const unsigned latency_slices = [] {  // option msm_latency_slices = 1 | 2 | 4 | 8 (A/B measurement)
        const unsigned v = (unsigned)ab_int("msm_latency_slices", 0);
        return v == 1 || v == 2 || v == 4 || v == 8 ? v : 4u;
    }();
    if (d.chunks != MSM_CHUNKS && S == 1 && n >= 256 && B <= 4) S = latency_slices;
    // ... and until a block's sorted term list fits in LDS (msm.hpp LDSSORT): the global list costs a line of HBM write
    // traffic per 4-byte entry once the launch outgrows the L2
#if KZG_AB_VARIANTS
    const size_t lds_cap = fp29_enabled() ? msm_lds_sort_capacity<Curve29>() : msm_lds_sort_capacity<Curve32>();
#else
    const size_t lds_cap = msm_lds_sort_capacity<Curve29>();
#endif
    auto slice_terms = [&](unsigned S_) { return ((size_t)mt + S_ - 1) / S_ * (size_t)d.chunks_per_block; };
    while (S < MSM_MAX_SLICES && slice_terms(S) > lds_cap && n / (2 * S) >= 1024 && 2 * S * B <= 128) S *= 2;  // (d_window_sl holds S B <= 128 slice sets)
    const bool lds_sort = slice_terms(S) + 1 <= lds_cap;  // (+1: slice boundaries round either way)
    d.slices = (int)S;
    d.window_sums = S > 1 ? w.d_window_sl : w.d_window;
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    const int nsc = (int)(B * (2 * n + 1));
    hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, s->s1, w.d_scalars, nsc);
    const unsigned gz = (unsigned)(2 * B * S);
    KzgRet rc_save = msm_save_reserve(s, W, slots, gz);
    if (rc_save != KZG_OK) return rc_save;
    if (w.mult_affine) {
        if (lds_sort) msm_window_launch<Curve29Aff, true>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
        else msm_window_launch<Curve29Aff, false>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
    } else if (fp29_enabled()) {
        // the latency layout's few workgroups run their reduction trees with four lanes per addition (option msm_tree_quads=0: A/B)
        static const bool tree_quads = ab_flag("msm_tree_quads", true);
        if (tree_quads && d.chunks != MSM_CHUNKS && B <= 4) {
            if (lds_sort) msm_window_launch<Curve29Quads, true>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
            else msm_window_launch<Curve29Quads, false>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
        } else if (lds_sort) msm_window_launch<Curve29, true>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
        else msm_window_launch<Curve29, false>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
    }
#if KZG_AB_VARIANTS
    else {
        if (lds_sort) msm_window_launch<Curve32, true>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
        else msm_window_launch<Curve32, false>(d, W, slots, gz, w.d_msm_save, w.cap_msm_save, s->s1);
    }
#endif
    // the latency layout (one window per chunk) of a few batches: every output is the plain sum of its slots x slices window
    // sums - one workgroup per output, four lanes per addition (option msm_sum_quads=0: the fold + combine kernels, A/B)
    static const bool sum_quads = ab_flag("msm_sum_quads", true);
    if (sum_quads && W == 1 && fp29_enabled() && slots * S >= 2 && slots * S <= (unsigned)SUMQ_MAX_POINTS && 2 * B < 64) {
        HIPCHK(DYN_LDS(k_msm_sum_quads, SUMQ_LDS_BYTES));
        hipLaunchKernelGGL(k_msm_sum_quads, dim3((unsigned)(2 * B)), dim3(256), SUMQ_LDS_BYTES, s->s1, d.window_sums, w.d_ab, (int)(slots * S));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->ev[3], s->s1));
        return KZG_OK;
    }
    if (S > 1)
        hipLaunchKernelGGL(k_msm_fold_slices, dim3((unsigned)(2 * B * slots * W)), dim3(64), 0, s->s1, w.d_window_sl, w.d_window, (int)S, (int)W);
    if (2 * B >= 64)  // enough outputs to fill wavefronts with one lane each
        hipLaunchKernelGGL(k_msm_combine_lanes, dim3((unsigned)((2 * B + 63) / 64)), dim3(64), 0, s->s1, w.d_window, w.d_ab, (int)slots, (int)W, (int)(2 * B));
    else
        hipLaunchKernelGGL(k_msm_combine, dim3((unsigned)(2 * B)), dim3(64), 0, s->s1, w.d_window, w.d_ab, (int)slots, (int)W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    return KZG_OK;
}

// decode 2T points (all C then all pi) into ws.d_points / d_pflag together with their 2^(64j) multiples, generator
// (precomputed multiples) as point 2T - all on stream s2, beside the SHA-256 chain
static KzgRet launch_decode(const KzgSettings* s, const void* d_commitments, const void* d_proofs, size_t T, bool behind_sha = true) {
    Workspace& w = s->ws;
    const int np = (int)(2 * T + 1);
    unsigned blocks = (unsigned)((2 * T + 63) / 64);
    static const bool no_latency_layout = !ab_flag("msm_latency_layout", true);
#if KZG_AB_VARIANTS
    static const bool proofs_16 = opt_int("proofs_chunks", 0) == 16;  // (A/B measurement: round 1's sixteen 16-bit chunks for the proof-tuple entries)
#else
    constexpr bool proofs_16 = false;
#endif
    w.chunks = (T <= LATENCY_MAX_BLOBS && !no_latency_layout) ? ((behind_sha || !proofs_16) ? MSM_CHUNKS_LATENCY : MSM_CHUNKS_PROOFS) : MSM_CHUNKS;
    const uint8_t *c = (const uint8_t*)d_commitments, *p = (const uint8_t*)d_proofs;
    const int n2 = (int)(2 * T);
    const size_t gen_off = w.chunks == MSM_CHUNKS ? 0 : w.chunks == MSM_CHUNKS_LATENCY ? MSM_CHUNKS : MSM_CHUNKS + MSM_CHUNKS_LATENCY;
    w.mult_affine = msm_affine_enabled() && w.chunks == MSM_CHUNKS;
    if (w.mult_affine) {
        // affine tables: rows 0 and 2 straight from the decode pass, rows 1 and 3 from 2^64 P through one inversion per 16 points
        G1Aff29Mem* mult = (G1Aff29Mem*)w.d_mult;
        // 256-thread workgroups: their four waves are dealt one to each SIMD of a CU (single-wave workgroups are placed unevenly)
        hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, true>), dim3((unsigned)((2 * T + 255) / 256)), dim3(256), 256 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s2, c, p, (int)T, w.d_points, w.d_pflag, w.d_mult, w.d_jtmp, n2, np,
                           w.kstamps_valid ? w.d_ktime + 6 : nullptr);
        const unsigned conv_blocks = (unsigned)((n2 + 64 * AFFINE_BATCH - 1) / (64 * AFFINE_BATCH));
        hipLaunchKernelGGL(k_mult_to_affine29, dim3(conv_blocks), dim3(64), 0, s->s2, w.d_jtmp, w.d_pflag, mult, n2, np);
        HIPCHK(hipEventRecord(s->ev[10], s->s2));
        hipLaunchKernelGGL(k_set_generator_multiples<G1Aff29Mem>, dim3(1), dim3(64), 0, s->s2, w.d_points, w.d_pflag, mult,
                           (const G1Aff29Mem*)s->d_gen_mult_aff, n2, np, w.chunks);
    } else if (fp29_enabled()) {
        G1Jac29Mem* mult = (G1Jac29Mem*)w.d_mult;
        // (the generator's rows first: behind the decode kernel they would sit on the critical path of a proof-tuple call)
        hipLaunchKernelGGL(k_set_generator_multiples<G1Jac29Mem>, dim3(1), dim3(64), 0, s->s2, w.d_points, w.d_pflag, mult,
                           (const G1Jac29Mem*)s->d_gen_mult + gen_off, n2, np, w.chunks);
        // the latency layouts: eight lanes per point, one wavefront of 8 points per CU (option decode_quads=0: one lane, A/B)
        static const bool dec_quads = ab_flag("decode_quads", true);
        // (while every workgroup can have a CU to itself, and not beside the challenge chain: there the decode hides behind the
        // chain anyway and has only half the CUs)
        if (dec_quads && w.chunks != MSM_CHUNKS && !behind_sha && (2 * T + DECQ_POINTS_PER_BLOCK - 1) / DECQ_POINTS_PER_BLOCK <= (size_t)s->n_cus) {
            blocks = (unsigned)((2 * T + DECQ_POINTS_PER_BLOCK - 1) / DECQ_POINTS_PER_BLOCK);
#if KZG_AB_VARIANTS
            if (w.chunks != MSM_CHUNKS_LATENCY) {
                HIPCHK(DYN_LDS(k_g1_decode_multiples29_quads<MSM_CHUNKS_PROOFS>, DECQ_LDS_BYTES));
                hipLaunchKernelGGL(k_g1_decode_multiples29_quads<MSM_CHUNKS_PROOFS>, dim3(blocks), dim3(64), DECQ_LDS_BYTES, s->s2, c, p, (int)T, w.d_points, w.d_pflag, mult, n2, np);
            } else
#endif
            {
                HIPCHK(DYN_LDS(k_g1_decode_multiples29_quads<MSM_CHUNKS_LATENCY>, DECQ_LDS_BYTES));
                hipLaunchKernelGGL(k_g1_decode_multiples29_quads<MSM_CHUNKS_LATENCY>, dim3(blocks), dim3(64), DECQ_LDS_BYTES, s->s2, c, p, (int)T, w.d_points, w.d_pflag, mult, n2, np);
            }
        } else if (w.chunks == MSM_CHUNKS_LATENCY)
            hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS_LATENCY, false>), dim3(blocks), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s2, c, p, (int)T, w.d_points, w.d_pflag, w.d_mult, (G1Jac29Mem*)nullptr, n2, np);
#if KZG_AB_VARIANTS
        else if (w.chunks == MSM_CHUNKS_PROOFS)
            hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS_PROOFS, false>), dim3(blocks), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s2, c, p, (int)T, w.d_points, w.d_pflag, w.d_mult, (G1Jac29Mem*)nullptr, n2, np);
#endif
        else
            hipLaunchKernelGGL((k_g1_decode_multiples29<MSM_CHUNKS, false>), dim3(blocks), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), s->s2, c, p, (int)T, w.d_points, w.d_pflag, w.d_mult, (G1Jac29Mem*)nullptr, n2, np);
        HIPCHK(hipEventRecord(s->ev[10], s->s2));
    }
#if KZG_AB_VARIANTS
    else {
        G1Jac* mult = (G1Jac*)w.d_mult;
        if (w.chunks == MSM_CHUNKS_LATENCY)
            hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS_LATENCY>, dim3(blocks), dim3(64), 0, s->s2, c, p, (int)T, w.d_points, w.d_pflag, mult, n2, np);
        else if (w.chunks == MSM_CHUNKS_PROOFS)
            hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS_PROOFS>, dim3(blocks), dim3(64), 0, s->s2, c, p, (int)T, w.d_points, w.d_pflag, mult, n2, np);
        else
            hipLaunchKernelGGL(k_g1_decode_multiples<MSM_CHUNKS>, dim3(blocks), dim3(64), 0, s->s2, c, p, (int)T, w.d_points, w.d_pflag, mult, n2, np);
        HIPCHK(hipEventRecord(s->ev[10], s->s2));
        hipLaunchKernelGGL(k_set_generator_multiples<G1Jac>, dim3(1), dim3(64), 0, s->s2, w.d_points, w.d_pflag, mult,
                           (const G1Jac*)s->d_gen_mult + gen_off, n2, np, w.chunks);
    }
#endif
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// ---------------------------------------------------------------- the three phases, each as launch + wait
// A handle is a small state machine: phase1_launch -> phase1_wait -> phase2_launch -> phase2_wait ->
// finish_launch -> finish_wait.  Launch halves only enqueue work on the handle's streams (plus the host
// transcript hashes in phase 2); wait halves block on this handle's stream only.  A call processes a launch
// GROUP of B independent batches of n blobs each (own transcript, r, MSM and pairing instance per batch): at
// n = 1024 every phase is a latency-bound serial chain that uses a sliver of the chip, so the batch dimension
// inside the kernels is what fills the machine.

// The stream pair of a launch of T blobs (KzgSettings::s1/s2): the split pair, made on first use, for a small launch.
// Every earlier launch of the handle has been waited for by then, so switching pairs is safe.
static void select_streams(const KzgSettings* s, size_t T) {
    if (!s->s_plain[1]) return;  // KZG_SINGLE_STREAM
    const bool small = T <= LATENCY_MAX_BLOBS;
    if (small && !s->s_half_tried) {
        s->s_half_tried = true;
        hipDeviceProp_t prop;
        if (ab_flag("cu_mask", true) && hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount >= 64) {
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> lo(words, 0), hi(words, 0);
            for (int i = 0; i < ncu; i++) ((i < ncu / 2) ? lo : hi)[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&s->s_half[0], words, lo.data()) != hipSuccess ||
                hipExtStreamCreateWithCUMask(&s->s_half[1], words, hi.data()) != hipSuccess) {
                (void)hipGetLastError();
                if (s->s_half[0]) (void)hipStreamDestroy(s->s_half[0]);
                s->s_half[0] = s->s_half[1] = nullptr;
            }
        }
    }
    // Only the two kernels that run side by side live on the CU halves: the challenge chain (s_sha) and the point decode (s2).
    // What follows them (evaluation, MSM, pairing) has the chip to itself and runs on the unmasked s1 - on a half mask the
    // 256 workgroups of the evaluation and the MSM's (window, slice) grid sat two to a CU.
    const bool use_half = small && s->s_half[0];
    s->s1 = s->s_plain[0];
    s->s_sha = use_half ? s->s_half[0] : s->s_plain[0];
    s->s2 = use_half ? s->s_half[1] : s->s_plain[1];
}

// A batch that still lies in HOST memory (the by-value Vec<Blob> of src/kzg_proof.rs:472-477; pageable): phase 1 brings it
// over itself, into the staging buffers it is given as d_blobs / d_commitments / d_proofs.
struct HostBatch {
    const uint8_t *blobs, *commitments, *proofs;
    const uint8_t* z_le = nullptr;  // the challenges, computed on the host (host_blob_challenges): n x 32 little-endian bytes ...
    hostpool::Job* z_job = nullptr;  // ... by the hashing pool (host_only.hpp); phase 1 joins in where it needs them (the point decode runs meanwhile)
};
// HOST batches of up to this many blobs take their challenges from the host's SHA-NI cores (option host_challenge_max_blobs;
// 0: never): a blob's SHA-256 chain is 2.8 ms on the GPU however few blobs there are and 65 us on a host core, and the hashing
// (up to 16 threads, option host_threads) runs beside the point decode.  Measured on MI355X + its 256-core host, one
// verify_blob_kzg_proof_batch call, host / GPU hashing (tools/prof/small_host_batch_latency.py): n = 1: 1.72 / 4.44 ms (with the
// one-proof tail), 2: 3.03 / 4.45, 8: 3.01 / 4.42, 64: 3.24 / 4.59, 128: 3.37 / 4.39, 256: 3.70 / 4.48, 512: 4.33 / 4.58.
// Larger batches, and every device-resident one, hash on the GPU.
static size_t host_challenge_max_blobs() {
    static const size_t v = (size_t)std::max(0L, std::min(4096L, opt_int("host_challenge_max_blobs", 256)));
    return v;
}
// In how many slices ACROSS the blobs (slice j = bytes [j W, (j + 1) W) of every blob, W = 128 KiB / S) a host batch of n
// blobs crosses PCIe.  One copy of the whole array first and the 2.8 ms SHA-256 chains after it cost copy + chain; SHA-256
// consumes a blob front to back, so the chain's segment j (k_blob_challenge_split2_t<true>) runs while slice j + 1 is on the
// link and the call costs ~ copy + chain / S.  Measured on MI355X (profiles/r3_hostslicebench.txt): hipMemcpy2DAsync from
// pageable memory returns at once and 8 slices of a 128 MiB batch land 0.31 ms apart, 2.44 ms in all against 2.38 ms for
// one copy.  A slice costs a dispatch and an event wait (~20 us): small batches take fewer.  option host_slices = 1 | 2 | 4 |
// 8 | 16 forces S (1: one copy, the round-2 behaviour).
static unsigned host_slices(size_t n) {
    static const unsigned forced = [] {
        const unsigned v = (unsigned)opt_int("host_slices", 0);
        return v == 1 || v == 2 || v == 4 || v == 8 || v == 16 ? v : 0u;
    }();
    static const bool lane_forced = opt_str("challenge_kernel") && !opt_is("challenge_kernel", "split2");
    if (n > SLICED_MAX_BLOBS || lane_forced) return 1;
    if (forced) return forced;
    return n >= 512 ? 8 : n >= 128 ? 4 : 1;
}

// A sliced hand-over, in two steps.  The copies run on the handle's copy stream, the chain's segments on stream st.
// ORDER OF THE HOST CALLS MATTERS: a wait on an event of the copy stream is issued right after the event is recorded and
// BEFORE the next copy is enqueued there.  With all eight slice copies enqueued first and the waits issued afterwards, every
// waiting kernel (the point decode behind the 96-KB copy included) started only when the LAST slice had landed
// (profiles/r3_host_timeline.txt, first form: copies 0.08 .. 2.65 ms, first kernel at 2.67 ms) - the runtime resolves such a
// wait against what the copy stream holds when the wait is made, not against the recorded point.
//   sliced_points_copy: commitments and proofs (96 bytes per blob; d_proofs may be null) -> ev_slice[0]; the copy stream first
//                       waits for `after`, an event of the stream whose earlier work used the staging buffers
//   sliced_segments:    for j < S: slice j of every blob -> ev_slice[1 + j], st waits for it, segment j of the chains
static KzgRet sliced_points_copy(const KzgSettings* s, const HostBatch& host, void* d_commitments, void* d_proofs, size_t T, hipEvent_t after) {
    if (!s->s_copy) {
        HIPCHK(hipStreamCreateWithFlags(&s->s_copy, hipStreamNonBlocking));
        for (auto& e : s->ev_copy) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!s->ev_slice[0])
        for (auto& e : s->ev_slice) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipStreamWaitEvent(s->s_copy, after, 0));
    HIPCHK(hipMemcpyAsync(d_commitments, host.commitments, 48 * T, hipMemcpyHostToDevice, s->s_copy));
    if (d_proofs) HIPCHK(hipMemcpyAsync(d_proofs, host.proofs, 48 * T, hipMemcpyHostToDevice, s->s_copy));
    HIPCHK(hipEventRecord(s->ev_slice[0], s->s_copy));
    return KZG_OK;
}
static KzgRet sliced_segments(const KzgSettings* s, const HostBatch& host, void* d_blobs, const void* d_commitments, Fr* d_z, size_t T, unsigned S,
                              hipStream_t st) {
    s->ws.ktime_valid = false;
    HIPCHK(hipStreamWaitEvent(st, s->ev_slice[0], 0));  // (the last segment reads the commitments)
    const size_t W = (size_t)BLOB_BYTES / S;
    for (unsigned j = 0; j < S; j++) {
        HIPCHK(hipMemcpy2DAsync((uint8_t*)d_blobs + j * W, BLOB_BYTES, host.blobs + j * W, BLOB_BYTES, W, T, hipMemcpyHostToDevice, s->s_copy));
        HIPCHK(hipEventRecord(s->ev_slice[1 + j], s->s_copy));
        HIPCHK(hipStreamWaitEvent(st, s->ev_slice[1 + j], 0));
        hipLaunchKernelGGL(k_blob_challenge_split2_t<true>, dim3((unsigned)((T + 63) / 64)), dim3(192), 0, st, (const uint8_t*)d_blobs,
                           (const uint8_t*)d_commitments, d_z, (int)T, (int)(1024 / S * j), (int)(1024 / S * (j + 1)), s->ws.d_sha_mid);
    }
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// Phase 1 (no communication): point decode + multiples || (challenge -> evaluate) for all T = B n blobs.
static KzgRet phase1_launch_locked(const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n, size_t B,
                                   const KzgSettings* s, const HostBatch* host = nullptr) {
    if (B > MAX_BATCHES_PER_LAUNCH) return fail(KZG_BADARGS, "more than 16384 batches in one launch group");  // gridDim.z = 2 B
    Workspace& w = s->ws;
    const size_t T = n * B;
    KzgRet rc;
    select_streams(s, T);
    const unsigned S = host && !host->z_le ? host_slices(T) : 1;
    w.kstamps_valid = w.d_ktime != nullptr;
    if (w.kstamps_valid) HIPCHK(hipMemsetAsync(w.d_ktime, 0, 128, s->s1));  // (before ev[0]: every stream of the launch is ordered behind it)
    HIPCHK(hipEventRecord(s->ev[0], s->s1));
    if (host && S == 1) {  // one copy of everything, on the stream the kernels follow on
        HIPCHK(hipMemcpyAsync(const_cast<void*>(d_commitments), host->commitments, 48 * T, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemcpyAsync(const_cast<void*>(d_proofs), host->proofs, 48 * T, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemcpyAsync(const_cast<void*>(d_blobs), host->blobs, (size_t)BLOB_BYTES * T, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipEventRecord(s->ev[0], s->s1));  // (the kernels' interval starts when the data is there, as before)
    }
    if (host && S > 1) {
        if ((rc = sliced_points_copy(s, *host, const_cast<void*>(d_commitments), const_cast<void*>(d_proofs), T, s->ev[0])) != KZG_OK) return rc;
        HIPCHK(hipStreamWaitEvent(s->s2, s->ev_slice[0], 0));  // the point decode starts while the blobs are still crossing
    } else {
        HIPCHK(hipStreamWaitEvent(s->s2, s->ev[0], 0));
    }
    HIPCHK(hipEventRecord(s->ev[5], s->s2));
    // (with the challenges from the host there is no SHA-256 chain to hide behind: the eight-lane decode, as for proof tuples)
    if ((rc = launch_decode(s, d_commitments, d_proofs, T, /*behind_sha=*/!(host && host->z_le))) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[6], s->s2));
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4 * T, s->s1));
    if (s->s_sha != s->s1) {
        HIPCHK(hipEventRecord(s->ev[11], s->s1));
        HIPCHK(hipStreamWaitEvent(s->s_sha, s->ev[11], 0));
    }
    if (host && host->z_le) {  // the host hashes the blobs: z crosses as 32 bytes per blob (pinned mirror, bytes [192 T, 224 T))
        w.ktime_valid = false;
        if (host->z_job) hostpool::finish(*host->z_job);
        memcpy(w.h_buf + 192 * T, host->z_le, 32 * T);
        HIPCHK(hipMemcpyAsync(w.d_z, w.h_buf + 192 * T, 32 * T, hipMemcpyHostToDevice, s->s_sha));
    } else if (host && S > 1) {
        if ((rc = sliced_segments(s, *host, const_cast<void*>(d_blobs), d_commitments, w.d_z, T, S, s->s_sha)) != KZG_OK) return rc;
    } else if ((rc = launch_challenge(s, d_blobs, d_commitments, w.d_z, T, s->s_sha)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[7], s->s_sha));
    if (s->s_sha != s->s1) {  // both joins in one place: each wait on another stream's event is a ~13 us bubble, even when it has long fired
        HIPCHK(hipStreamWaitEvent(s->s1, s->ev[7], 0));
        HIPCHK(hipStreamWaitEvent(s->s1, s->ev[6], 0));
    }
    if ((rc = launch_evaluate(s, d_blobs, w.d_z, w.d_y, w.d_status, T, false, w.kstamps_valid)) != KZG_OK) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[8], s->s1));
    if (s->s_sha == s->s1) HIPCHK(hipStreamWaitEvent(s->s1, s->ev[6], 0));
    HIPCHK(hipEventRecord(s->ev[1], s->s1));
    // the transcript records, packed on the device; pinned host mirror: [records 160 T | status 4 T | point flags 8 T]
    uint8_t* h = w.h_buf;
    if (T <= LATENCY_MAX_BLOBS) {  // the kernel writes the mirror itself: every dispatch of a single batch's chain is ~10 us of latency
        hipLaunchKernelGGL(k_pack_records_mirror, dim3((unsigned)((43 * T + 255) / 256)), dim3(256), 0, s->s1, (const uint32_t*)d_commitments,
                           (const uint32_t*)d_proofs, (const uint32_t*)w.d_z, (const uint32_t*)w.d_y, (uint32_t*)w.d_records, (uint32_t*)h,
                           (const uint32_t*)w.d_status, (const uint32_t*)w.d_pflag, (int)T);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((40 * T + 255) / 256)), dim3(256), 0, s->s1, (const uint32_t*)d_commitments,
                           (const uint32_t*)d_proofs, (const uint32_t*)w.d_z, (const uint32_t*)w.d_y, (uint32_t*)w.d_records, (int)T);
        HIPCHK(hipGetLastError());
        uint32_t* h_status = reinterpret_cast<uint32_t*>(h + 160 * T);
        uint32_t* h_pflag = h_status + T;
        HIPCHK(hipMemcpyAsync(h, w.d_records, 160 * T, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(h_status, w.d_status, 4 * T, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(h_pflag, w.d_pflag, 8 * T, hipMemcpyDeviceToHost, s->s1));
    }
    if (w.ktime_valid || w.kstamps_valid) HIPCHK(hipMemcpyAsync(h + 176 * T, w.d_ktime, 64, hipMemcpyDeviceToHost, s->s1));  // challenge | evaluate | decode
    w.pending_n = n;
    w.pending_b = B;
    return KZG_OK;
}

// records_out (optional): [B][n] x 160 bytes  C(48) || z(32, LE) || y(32, LE) || pi(48) - exactly the per-blob slices
// of the batch transcripts of src/kzg_proof.rs:314-334.  The handle keeps them (pinned host + device) for phase 2.  bad_out (optional, B bytes): 1 where a batch holds an invalid input
// (then the call still returns KZG_OK); without bad_out any invalid input makes the whole call return KZG_BADARGS.
static KzgRet phase1_wait_locked(uint8_t* records_out, uint8_t* bad_out, const KzgSettings* s) {
    Workspace& w = s->ws;
    const size_t n = w.pending_n, B = w.pending_b, T = n * B;
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[1], s->ev[0], s->ev[1]);
    elapsed(&s->timings[4], s->ev[7], s->ev[8]);
    elapsed(&s->timings[5], s->ev[0], s->ev[7]);
    if (w.ktime_valid) {  // the throughput-form kernel stamps its own execution interval (100 MHz ticks): no queueing time in it
        const unsigned long long* kt = reinterpret_cast<const unsigned long long*>(w.h_buf + 176 * T);
        if (kt[0] && kt[1] && kt[1] > ~kt[0]) s->timings[5] = (float)((double)(kt[1] - ~kt[0]) * 1e-5);
        s->clk_sum[0] += (double)kt[2];  // shader cycles and 100 MHz reference ticks of the kernel's waves (kzg_debug_shader_clock)
        s->clk_sum[1] += (double)kt[3];
    }
    elapsed(&s->timings[6], s->ev[5], s->ev[10]);
    elapsed(&s->timings[7], s->ev[10], s->ev[6]);
    {  // the kernels' own intervals (100 MHz ticks): challenge (its throughput form only) | evaluate | decode + multiples
        const unsigned long long* kt = reinterpret_cast<const unsigned long long*>(w.h_buf + 176 * T);
        auto ms_of = [&](int k) { return kt[k] && kt[k + 1] && kt[k + 1] > ~kt[k] ? (float)((double)(kt[k + 1] - ~kt[k]) * 1e-5) : 0.f; };
        s->kstamp_ms[0] = w.ktime_valid ? ms_of(0) : 0.f;
        s->kstamp_ms[1] = w.kstamps_valid ? ms_of(4) : 0.f;
        s->kstamp_ms[2] = w.kstamps_valid ? ms_of(6) : 0.f;
    }
    uint8_t* h = w.h_buf;
    uint32_t* h_status = reinterpret_cast<uint32_t*>(h + 160 * T);
    uint32_t* h_pflag = h_status + T;
    // error order of the reference: commitments (:503), proofs (:508), then blobs (:263); all map to BadArgs
    bool any_bad = false;
    for (size_t b = 0; b < B; b++) {
        bool bad = false;
        for (size_t i = b * n; i < (b + 1) * n; i++) bad |= h_pflag[i] == G1_INVALID || h_pflag[T + i] == G1_INVALID || h_status[i] != 0;
        if (bad_out) bad_out[b] = bad;
        any_bad |= bad;
    }
    if (any_bad && !bad_out) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) src/kzg_proof.rs:19-23,38-40
    if (records_out) memcpy(records_out, h, 160 * T);  // the device limb arrays ARE Scalar::to_bytes() (little-endian), :321,:326
    return KZG_OK;
}

// Phase 2: per batch b, this shard's scalars r_b^(offset+i) and its partial sums (A, B)_b.  Requires phase 1 of the same group on
// this handle.  r_b comes from the batch's FULL transcript, in one of four ways:
//   r_le != NULL                      : given by the caller, B x 32 little-endian bytes (kzg_batch_challenges on some rank)
//   all_records != NULL, world == 0   : hashed here from [B][n_total] records in global blob order
//   all_records != NULL, world  > 0   : hashed here from [world][B][n] records (n_total = world n)
//   both NULL                         : hashed here from the handle's own records of phase 1 (single rank: n_total = n)
static KzgRet phase2_launch_locked(const uint8_t* all_records, size_t n_total, size_t offset, const KzgSettings* s, size_t world = 0,
                                   const uint8_t* r_le = nullptr, bool want_partials = true) {
    Workspace& w = s->ws;
    const size_t n = w.pending_n, B = w.pending_b;
    if (!all_records && !r_le) {
        if (n_total != n || offset != 0) return fail(KZG_BADARGS, "local phase 2 needs n_total == n_local");
        all_records = w.h_buf;
        world = 0;
    }
    if (world && n_total != world * n) return fail(KZG_BADARGS, "gathered phase 2 needs equal shards");
    if (n_total == 1) {
        // verify_blob_kzg_proof path (:482-489): r^0 = 1, no batch challenge
        hipLaunchKernelGGL(k_single_scalars, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_z, w.d_y, w.d_scalars);
    } else {
        uint8_t* h_r = w.h_buf + w.off_r;  // pinned staging for the async H2D copy
        if (r_le) memcpy(h_r, r_le, 32 * B);
        else if (!host_batch_challenges(h_r, all_records, B, n, n_total, world)) return fail(KZG_MALLOC, "batch transcript buffer");
        const Fr* d_r = w.d_r;
        if (n * B <= LATENCY_MAX_BLOBS) d_r = reinterpret_cast<const Fr*>(h_r);  // a small launch reads r where the host wrote it (pinned memory)
        else HIPCHK(hipMemcpyAsync(w.d_r, h_r, 32 * B, hipMemcpyHostToDevice, s->s1));
        unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_batch_scalars, dim3(blocks, (unsigned)B), dim3(256), 0, s->s1, d_r, w.d_z, w.d_y, w.d_scalars,
                           w.d_partial, (int)n, (unsigned long long)offset);
        hipLaunchKernelGGL(k_finish_g, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_partial, (int)blocks, w.d_scalars, (int)n);
    }
    HIPCHK(hipGetLastError());
    KzgRet rc = run_msm(s, n, B);
    if (rc != KZG_OK) return rc;
    if (want_partials) HIPCHK(hipMemcpyAsync(w.h_buf + w.off_part, w.d_ab, 288 * B, hipMemcpyDeviceToHost, s->s1));  // (the local callers go straight on to the pairing)
    return KZG_OK;
}

static KzgRet phase2_wait_locked(uint8_t* partial_out /* B x 288 */, const KzgSettings* s) {
    Workspace& w = s->ws;
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    if (partial_out) memcpy(partial_out, w.h_buf + w.off_part, 288 * w.pending_b);
    return KZG_OK;
}

// Finish: fold `world` partial sets ([world][B] x 288 B), or take the handle's own (A, B)_b when partials == nullptr,
// and run one pairing instance per batch.
// parts_on_device: the [world][B] partial sets already lie in ws.d_parts (an in-process RCCL all-gather put them there, capi_multi.hpp)
static KzgRet finish_launch_locked(const uint8_t* partials, size_t world, size_t B, const KzgSettings* s, bool parts_on_device = false) {
    Workspace& w = s->ws;
    if (partials || parts_on_device) {
        if (world > MAX_WORLD) return fail(KZG_BADARGS, "world size above 64");
        if (!parts_on_device) {
            memcpy(w.h_buf + w.off_parts, partials, 288 * world * B);
            HIPCHK(hipMemcpyAsync(w.d_parts, w.h_buf + w.off_parts, 288 * world * B, hipMemcpyHostToDevice, s->s1));
        }
        hipLaunchKernelGGL(k_fold_partials, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_parts, (int)world, (int)B, w.d_ab);
    }
    hipLaunchKernelGGL(k_jac_to_slp, dim3((unsigned)B), dim3(64), 0, s->s1, w.d_ab, w.d_slp_in);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[4], s->s1));
    // (a handful of instances write their 288 bytes straight into the pinned mirror: one dispatch less at the end of the chain)
    const bool direct = B <= LATENCY_PAIRING_MAX;
    KzgRet rc = run_verify(s, w.d_slp_in, direct ? reinterpret_cast<Fp*>(w.h_buf + w.off_out) : w.d_slp_out, (int)B, s->s1);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[9], s->s1));
    if (!direct) HIPCHK(hipMemcpyAsync(w.h_buf + w.off_out, w.d_slp_out, sizeof(Fp) * 6 * B, hipMemcpyDeviceToHost, s->s1));
    if (w.kstamps_valid && w.pending_n) HIPCHK(hipMemcpyAsync(w.h_buf + 176 * w.pending_n * w.pending_b + 64, w.d_ktime + 8, 16, hipMemcpyDeviceToHost, s->s1));  // the MSM window kernel's interval
    w.finish_b = B;
    return KZG_OK;
}

static KzgRet finish_wait_locked(bool* ok /* B */, const KzgSettings* s) {
    Workspace& w = s->ws;
    HIPCHK(hipStreamSynchronize(s->s1));
    const uint32_t* h = reinterpret_cast<const uint32_t*>(w.h_buf + w.off_out);
    for (size_t b = 0; b < w.finish_b; b++) {
        uint32_t any = 0;
        for (int i = 0; i < 72; i++) any |= h[72 * b + i];
        ok[b] = any == 0;
    }
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    elapsed(&s->timings[3], s->ev[4], s->ev[9]);
    elapsed(&s->timings[0], s->ev[0], s->ev[9]);
    for (int i = 0; i < 8; i++) s->tsum[i] += s->timings[i];
    s->tcount++;
    if (w.kstamps_valid && w.pending_n) {
        const unsigned long long* kt = reinterpret_cast<const unsigned long long*>(w.h_buf + 176 * w.pending_n * w.pending_b + 64);
        s->kstamp_ms[3] = kt[0] && kt[1] && kt[1] > ~kt[0] ? (float)((double)(kt[1] - ~kt[0]) * 1e-5) : 0.f;
        for (int i = 0; i < 4; i++) s->kstamp_sum[i] += s->kstamp_ms[i];
        s->kstamp_count++;
        w.kstamps_valid = false;
    }
    w.pending_n = w.pending_b = w.finish_b = 0;  // the group is done: the handle holds no live state (kzg_shard_finish_launch tests this)
    return KZG_OK;
}

static KzgRet batch_device_locked(bool* ok, const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n,
                                  const KzgSettings* s, const HostBatch* host = nullptr) {
    KzgRet rc;
    if ((rc = phase1_launch_locked(d_blobs, d_commitments, d_proofs, n, 1, s, host)) != KZG_OK) return rc;
    if ((rc = phase1_wait_locked(nullptr, nullptr, s)) != KZG_OK) return rc;
    if ((rc = phase2_launch_locked(nullptr, n, 0, s, 0, nullptr, false)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, 1, s)) != KZG_OK) return rc;  // same stream: no host round trip needed
    return finish_wait_locked(ok, s);
}

// ---- multi-GPU / grouped / pipelined entry points (include/kzg_rs_amd.h) ----
#define KZG_ENTER(cond)                                          \
    if (!(cond)) return fail(KZG_BADARGS, "bad argument");       \
    std::lock_guard<std::mutex> lk(s->mu);                       \
    HIPCHK(hipSetDevice(s->device));

extern "C" KzgRet kzg_shard_phase1_launch(const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n_local,
                                          size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && d_blobs && d_commitments && d_proofs && n_local && n_batches);
    KzgRet rc = ws_reserve(s, n_local * n_batches, n_batches, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    return phase1_launch_locked(d_blobs, d_commitments, d_proofs, n_local, n_batches, s);
}
extern "C" KzgRet kzg_shard_phase1_wait(uint8_t* records_out, uint8_t* bad_out, const KzgSettings* s) {
    KZG_ENTER(s && s->ws.pending_n);
    return phase1_wait_locked(records_out, bad_out, s);
}
extern "C" KzgRet kzg_shard_phase2_launch(const uint8_t* all_records, size_t n_total, size_t offset, const KzgSettings* s) {
    KZG_ENTER(s && s->ws.pending_n && offset + s->ws.pending_n <= n_total);
    return phase2_launch_locked(all_records, n_total, offset, s);
}
extern "C" KzgRet kzg_shard_phase2_launch_gathered(const uint8_t* gathered, size_t world, size_t rank, const KzgSettings* s) {
    KZG_ENTER(s && gathered && s->ws.pending_n && world && rank < world);
    return phase2_launch_locked(gathered, world * s->ws.pending_n, rank * s->ws.pending_n, s, world);
}
extern "C" KzgRet kzg_shard_phase2_launch_r(const uint8_t* r_le, size_t n_total, size_t offset, const KzgSettings* s) {
    KZG_ENTER(s && r_le && s->ws.pending_n && offset + s->ws.pending_n <= n_total);
    for (size_t b = 0; b < s->ws.pending_b; b++) {  // canonical little-endian scalars only
        uint8_t be[32];
        reverse32(be, r_le + 32 * b);
        if (be_geq_r(be)) return fail(KZG_BADARGS, "kzg_shard_phase2_launch_r: challenge not below r");
    }
    return phase2_launch_locked(nullptr, n_total, offset, s, 0, r_le);
}
// the hash half of compute_r_powers on the host, without a handle or a GPU: see include/kzg_rs_amd.h
extern "C" KzgRet kzg_batch_challenges(uint8_t* r_le_out, const uint8_t* records, size_t world, size_t n_batches, size_t n_local) {
    if (!r_le_out || !records || !n_batches || !n_local) return fail(KZG_BADARGS, "bad argument");
    // sizes that cannot be a transcript: (world n_local) blobs of 128 KiB each would not fit any memory, and the products must not wrap
    const size_t w1 = world ? world : 1, lim = (size_t)1 << 40;
    if (world > MAX_WORLD || n_local > lim / w1 || n_batches > lim || w1 * n_local > (lim / 160) / n_batches)
        return fail(KZG_BADARGS, "kzg_batch_challenges: sizes out of range");
    try {  // (the buffers are allocated without throwing; what is left is thread creation)
        if (!host_batch_challenges(r_le_out, records, n_batches, n_local, w1 * n_local, world))
            return fail(KZG_MALLOC, "kzg_batch_challenges: transcript buffer");
    } catch (const std::exception& e) {
        return fail(KZG_ERROR, std::string("kzg_batch_challenges: ") + e.what());
    }
    return KZG_OK;
}
extern "C" KzgRet kzg_shard_records_device(void* d_records_out, const KzgSettings* s) {
    KZG_ENTER(s && d_records_out && s->ws.pending_n);
    HIPCHK(hipMemcpyAsync(d_records_out, s->ws.d_records, 160 * s->ws.pending_n * s->ws.pending_b, hipMemcpyDeviceToDevice, s->s1));
    return KZG_OK;
}
extern "C" KzgRet kzg_shard_phase2_wait(uint8_t* partial_out, const KzgSettings* s) {
    KZG_ENTER(s && partial_out);
    return phase2_wait_locked(partial_out, s);
}
extern "C" KzgRet kzg_shard_finish_launch(const uint8_t* partials, size_t world, size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && n_batches && (partials ? world > 0 : true));  // partials == NULL: pair this handle's own sums (single rank)
    Workspace& w = s->ws;
    // The finish needs only buffers sized by the batch count.  A handle that ran phases 1-2 of this group holds LIVE state
    // (the sums in d_ab, pending_n / pending_b): its workspace must never be regrown here - ws_reserve frees and
    // reallocates everything.  Only a handle without a group in flight (a rank that folds partials it did not produce)
    // reserves, keeping its blob capacity.
    if (!partials && (!w.pending_b || n_batches != w.pending_b))
        return fail(KZG_BADARGS, "kzg_shard_finish_launch without partials needs the group of this handle's phase 2");
    if (n_batches > w.cap_b) {
        if (w.pending_b) return fail(KZG_BADARGS, "kzg_shard_finish_launch: more batches than the group in flight on this handle");
        KzgRet rc = ws_reserve(s, w.cap_n ? w.cap_n : 16, n_batches, STAGE_NONE);
        if (rc != KZG_OK) return rc;
    }
    return finish_launch_locked(partials, world, n_batches, s);
}
extern "C" KzgRet kzg_shard_finish_wait(bool* ok, const KzgSettings* s) {
    KZG_ENTER(s && ok);
    return finish_wait_locked(ok, s);
}
// blocking single-batch forms
extern "C" KzgRet kzg_shard_phase1(uint8_t* records_out, const void* d_blobs, const void* d_commitments, const void* d_proofs,
                                   size_t n_local, const KzgSettings* s) {
    KzgRet rc = kzg_shard_phase1_launch(d_blobs, d_commitments, d_proofs, n_local, 1, s);
    return rc != KZG_OK ? rc : kzg_shard_phase1_wait(records_out, nullptr, s);
}
extern "C" KzgRet kzg_shard_phase2(uint8_t partial_out[288], const uint8_t* all_records, size_t n_total, size_t offset,
                                   size_t n_local, const KzgSettings* s) {
    if (s && n_local != s->ws.pending_n) return fail(KZG_BADARGS, "kzg_shard_phase2 without a matching kzg_shard_phase1");
    KzgRet rc = kzg_shard_phase2_launch(all_records, n_total, offset, s);
    return rc != KZG_OK ? rc : kzg_shard_phase2_wait(partial_out, s);
}
extern "C" KzgRet kzg_shard_finish(bool* ok, const uint8_t* partials, size_t world, const KzgSettings* s) {
    KzgRet rc = kzg_shard_finish_launch(partials, world, 1, s);
    return rc != KZG_OK ? rc : kzg_shard_finish_wait(ok, s);
}

// B independent batches of n blobs each in ONE launch group: blobs / commitments / proofs are contiguous device
// arrays of B*n entries, batch b = entries [b n, (b+1) n); ok_out[b] and (optional) err_out[b] per batch.
static KzgRet batches_device_locked(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n,
                                    size_t n_batches, const KzgSettings* s);
static KzgRet multi_batches_device_locked(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments, const void* d_proofs,
                                          size_t n, size_t n_batches, const KzgSettings* s);
extern "C" KzgRet kzg_verify_blob_kzg_proof_batches_device(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments,
                                                           const void* d_proofs, size_t n, size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && ok_out && d_blobs && d_commitments && d_proofs && n && n_batches);
    // a handle over several devices: the launch group runs on the device that owns its memory (capi_multi.hpp)
    if (s->multi) return multi_batches_device_locked(ok_out, err_out, d_blobs, d_commitments, d_proofs, n, n_batches, s);
    return batches_device_locked(ok_out, err_out, d_blobs, d_commitments, d_proofs, n, n_batches, s);
}
// the calling thread has set s's device and owns s
static KzgRet batches_device_locked(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments, const void* d_proofs, size_t n,
                                    size_t n_batches, const KzgSettings* s) {
    KzgRet rc = ws_reserve(s, n * n_batches, n_batches, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    if ((rc = phase1_launch_locked(d_blobs, d_commitments, d_proofs, n, n_batches, s)) != KZG_OK) return rc;
    if ((rc = phase1_wait_locked(nullptr, err_out, s)) != KZG_OK) return rc;
    if ((rc = phase2_launch_locked(nullptr, n, 0, s, 0, nullptr, false)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, n_batches, s)) != KZG_OK) return rc;
    if ((rc = finish_wait_locked(ok_out, s)) != KZG_OK) return rc;
    if (err_out)
        for (size_t b = 0; b < n_batches; b++)
            if (err_out[b]) ok_out[b] = false;
    return KZG_OK;
}

// a handle over several devices (capi_multi.hpp): does a call of n blobs go through its shards, and the sharded call itself
static bool multi_takes(const KzgSettings* s, size_t n);
static KzgRet multi_array_locked(bool* ok, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n, bool host,
                                 const KzgSettings* s);

extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_device(bool* ok, const void* d_blobs, const void* d_commitments,
                                                         const void* d_proofs, size_t n, const KzgSettings* s) {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {  // src/kzg_proof.rs:478-480
        *ok = true;
        return KZG_OK;
    }
    if (!d_blobs || !d_commitments || !d_proofs) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    // several devices: the arrays lie on the first one, the other shards' slices cross xGMI (per-device resident shards:
    // kzg_verify_blob_kzg_proof_batch_sharded)
    if (multi_takes(s, n)) return multi_array_locked(ok, (const uint8_t*)d_blobs, (const uint8_t*)d_commitments, (const uint8_t*)d_proofs, n, false, s);
    KzgRet rc = ws_reserve(s, n, 1, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    return batch_device_locked(ok, d_blobs, d_commitments, d_proofs, n, s);
}

static KzgRet blob_single_locked(bool* ok, bool* general, const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof, const KzgSettings* s,
                                 hostpool::Job* job = nullptr);
// the small-call queue of a shared handle (capi_coalesce.hpp)
static bool small_enabled(const KzgSettings* s);
static KzgRet small_proofs(bool* ok, uint8_t* err, uint8_t* general, const uint8_t* c, const uint8_t* z, const uint8_t* y, const uint8_t* p, size_t n,
                           const KzgSettings* s);
static KzgRet small_blobs(bool* ok, uint8_t* err, uint8_t* general, const uint8_t* blobs, const uint8_t* c, const uint8_t* p, size_t n, const KzgSettings* s);
static KzgRet blobs_small_locked(bool* ok, bool* general, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n, const KzgSettings* s);
static void proof_drain(const KzgSettings* s);
extern "C" KzgRet kzg_verify_blob_kzg_proof_batch(bool* ok, const uint8_t* blobs, const uint8_t* commitments,
                                                  const uint8_t* proofs, size_t n, const KzgSettings* s) try {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {
        *ok = true;
        return KZG_OK;
    }
    if (!blobs || !commitments || !proofs) return fail(KZG_BADARGS, "null argument");
    const size_t host_max = host_challenge_max_blobs();
    static const bool msm_path = opt_is("proof_path", "msm");
    static const size_t small_max = (size_t)std::max(0L, std::min(256L, opt_int("small_batch_pairings_max", 256)));
    // The sizes a beacon node calls this with (one blob; the 6-9 blobs of a block; up to small_max): a request in the handle's
    // small-call queue - concurrent callers share launches on pooled lanes, nobody holds the handle's lock (capi_coalesce.hpp)
    if (small_enabled(s) && !multi_takes(s, n) && ((n == 1 && host_max >= 1) || (n >= 2 && n <= small_max && n <= host_max))) {
        bool each = false;
        uint8_t err = 0, general = 0;
        const KzgRet qrc = small_blobs(&each, &err, &general, blobs, commitments, proofs, n, s);
        if (qrc != KZG_OK) return qrc;
        if (err) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
        if (!general) {
            *ok = each;
            return KZG_OK;
        }
        // (some z_i = tau: the combined form below decides)
    }
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    // several devices: contiguous slices of the Vec<Blob>, each over its own device's PCIe link (capi_multi.hpp)
    if (multi_takes(s, n)) return multi_array_locked(ok, blobs, commitments, proofs, n, true, s);
    KzgRet rc;
    const bool queued = small_enabled(s);  // (the small forms ran above; what is left of them here is the z = tau fallback)
    if (n == 1 && host_max >= 1 && !msm_path && !queued) {  // verify_blob_kzg_proof (:446-470, :482-489): host hash, one-proof tail
        bool general = false;
        if ((rc = blob_single_locked(ok, &general, blobs, commitments, proofs, s)) != KZG_OK) {
            proof_drain(s);
            return rc;
        }
        if (!general) return KZG_OK;
    }
    if (n >= 2 && n <= small_max && n <= host_max && !msm_path && !queued) {  // a few blobs: one pairing each, side by side (blobs_small_locked)
        bool general = false;
        if ((rc = blobs_small_locked(ok, &general, blobs, commitments, proofs, n, s)) != KZG_OK) {
            proof_drain(s);
            return rc;
        }
        if (!general) return KZG_OK;
    }
    if ((rc = ws_reserve(s, n, 1, STAGE_BLOBS)) != KZG_OK) return rc;
    Workspace& w = s->ws;
    HostBatch host{blobs, commitments, proofs};  // phase 1 brings the batch over, in slices across the blobs (host_slices)
    std::vector<uint8_t> z_host;
    hostpool::JobRef z_job;
    hostpool::JoinOnExit joined;  // (declared after z_host: on an exception out of the batch the workers are joined before the buffer they write dies)
    if (n <= host_max) {  // a few blobs: their challenges from the host's SHA-NI cores, beside the point decode on the GPU
        z_host.resize(32 * n);
        host.z_le = z_host.data();
        z_job = hostpool::make(z_host.data(), blobs, commitments, n);
        joined.add(z_job);
        hostpool::post(z_job, (size_t)std::max(1L, std::min(16L, opt_int("host_threads", 16))));
        host.z_job = z_job.get();
    }
    rc = batch_device_locked(ok, w.d_stage_blobs, w.d_stage_cp, w.d_stage_cp + 48 * n, n, s, &host);
    if (z_job) hostpool::finish(*z_job);  // (an error path may not have come by: no worker may still read the caller's memory)
    if (rc != KZG_OK && s->s_copy) {  // nothing may still read the caller's memory when the error goes back
        const std::string msg = g_err;
        (void)hipStreamSynchronize(s->s_copy);
        (void)hipGetLastError();
        g_err = msg;
    }
    return rc;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}

// A STREAM of host-resident batches: n_batches independent verify_blob_kzg_proof_batch calls (src/kzg_proof.rs:472-525) of n
// blobs each, all three arrays in HOST memory (n_batches Vec<Blob>s back to back).  The batches cross PCIe in chunks on a copy
// stream into one of two staging sets while the previous chunk is verified as one launch group, so the link never waits for
// the GPU: the rate is the link's (~56 GB/s measured on MI355X = ~0.43 M blobs/s), not copy + compute.  A pageable
// hipMemcpyAsync holds the calling thread until the data has left (measured: 2.4 ms per 128 MiB either way), so each chunk's
// copy is issued in two halves around the host-side steps of the chunk in flight (flags, transcript hashes, launches).
// option host_chunk = batches per chunk (default: half the stream, at most 32).
static KzgRet multi_host_stream_locked(bool* ok_out, uint8_t* err_out, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs,
                                       size_t n, size_t n_batches, const KzgSettings* s);
static KzgRet host_stream_locked(bool* ok_out, uint8_t* err_out, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n,
                                 size_t n_batches, const KzgSettings* s);
extern "C" KzgRet kzg_verify_blob_kzg_proof_batches(bool* ok_out, uint8_t* err_out, const uint8_t* blobs, const uint8_t* commitments,
                                                    const uint8_t* proofs, size_t n, size_t n_batches, const KzgSettings* s) {
    KZG_ENTER(s && ok_out && blobs && commitments && proofs && n && n_batches);
    // a handle over several devices: contiguous ranges of whole batches, one per device, each streamed over that device's own
    // PCIe link on a thread of its own (capi_multi.hpp) - the stream form is bound by the link, so the links add up
    if (s->multi && n_batches >= 2) return multi_host_stream_locked(ok_out, err_out, blobs, commitments, proofs, n, n_batches, s);
    return host_stream_locked(ok_out, err_out, blobs, commitments, proofs, n, n_batches, s);
}
// the stream on ONE device (the handle's own); the caller holds the handle's lock (or owns the handle) and has set the device
static KzgRet host_stream_locked(bool* ok_out, uint8_t* err_out, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n,
                                 size_t n_batches, const KzgSettings* s) {
    static const size_t chunk_forced = [] {
        const long v = opt_int("host_chunk", 0);
        return (size_t)(v < 0 ? 0 : v > 4096 ? 4096 : v);
    }();
    // chunk: whole batches.  Large chunks on purpose: the driver pins and unpins the pageable source around every copy
    // (~0.6 ms per 128 MiB on top of the 2.3 ms of DMA: measured 47-49 GB/s for 2-4 GiB copies, 38 GB/s for 0.5 GiB ones),
    // and a chunk must outlast the latency-bound phases of the chunk before it (~6 ms) to hide them.  Half the stream, 32
    // batches at most (2 x 4 GiB of staging at n = 1024).
    size_t G = chunk_forced ? chunk_forced : std::max<size_t>(1, std::min<size_t>(32, (n_batches + 1) / 2));
    G = std::min(std::min(G, n_batches), (size_t)MAX_BATCHES_PER_LAUNCH);
    while (G > 1 && G * n > (size_t)1 << 17) G /= 2;  // <= 128 Ki blobs (16 GiB) per staging set
    Workspace& w = s->ws;
    KzgRet rc = ws_reserve(s, G * n, G, STAGE_NONE);
    if (rc != KZG_OK) return rc;
    const size_t set_blobs = G * n, set_bytes = set_blobs * ((size_t)BLOB_BYTES + 96);
    if (set_blobs > w.cap_hstage) {
        for (auto& p : w.d_hstage) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        w.cap_hstage = 0;
        HIPCHK(hipMalloc(&w.d_hstage[0], set_bytes));
        HIPCHK(hipMalloc(&w.d_hstage[1], set_bytes));
        w.cap_hstage = set_blobs;
    }
    if (!s->s_copy) {
        HIPCHK(hipStreamCreateWithFlags(&s->s_copy, hipStreamNonBlocking));
        for (auto& e : s->ev_copy) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t cap_set = w.cap_hstage;  // the layout of a staging set follows its capacity
    const size_t n_chunks = (n_batches + G - 1) / G;
    auto batches_of = [&](size_t c) { return std::min(G, n_batches - c * G); };
    // staging set layout: [blobs of the chunk | commitments | proofs]; `part` 0 / 1 = first / second half of the blobs
    // (the 96 bytes per blob of commitments and proofs travel with the first half)
    auto copy_part = [&](size_t c, int part) -> KzgRet {
        uint8_t* dst = w.d_hstage[c & 1];
        const size_t nb = batches_of(c) * n, first = c * G * n;
        const size_t half = (nb / 2) * (size_t)BLOB_BYTES, total = nb * (size_t)BLOB_BYTES;
        if (part == 0) {
            HIPCHK(hipMemcpyAsync(dst + cap_set * (size_t)BLOB_BYTES, commitments + 48 * first, 48 * nb, hipMemcpyHostToDevice, s->s_copy));
            HIPCHK(hipMemcpyAsync(dst + cap_set * ((size_t)BLOB_BYTES + 48), proofs + 48 * first, 48 * nb, hipMemcpyHostToDevice, s->s_copy));
            if (half) HIPCHK(hipMemcpyAsync(dst, blobs + first * (size_t)BLOB_BYTES, half, hipMemcpyHostToDevice, s->s_copy));
        } else {
            HIPCHK(hipMemcpyAsync(dst + half, blobs + first * (size_t)BLOB_BYTES + half, total - half, hipMemcpyHostToDevice, s->s_copy));
            HIPCHK(hipEventRecord(s->ev_copy[c & 1], s->s_copy));
        }
        return KZG_OK;
    };
    // any failure leaves copies and kernels in flight that read the caller's memory and the staging sets: drain both
    // streams before the error goes back (the first message is kept)
    auto drained = [&](KzgRet code) {
        const std::string msg = g_err;
        (void)hipStreamSynchronize(s->s_copy);
        (void)hipStreamSynchronize(s->s1);
        (void)hipStreamSynchronize(s->s2);
        if (s->s_sha) (void)hipStreamSynchronize(s->s_sha);  // the CU-masked challenge stream of a small chunk
        (void)hipGetLastError();
        g_err = msg;
        return code;
    };
    if ((rc = copy_part(0, 0)) != KZG_OK || (rc = copy_part(0, 1)) != KZG_OK) return drained(rc);
    for (size_t c = 0; c < n_chunks; c++) {
        const size_t B = batches_of(c);
        uint8_t* st = w.d_hstage[c & 1];
        select_streams(s, B * n);
        if (hipStreamWaitEvent(s->s1, s->ev_copy[c & 1], 0) != hipSuccess) return drained(fail(KZG_ERROR, "HIP: hipStreamWaitEvent"));  // this chunk has landed
        if ((rc = phase1_launch_locked(st, st + cap_set * (size_t)BLOB_BYTES, st + cap_set * ((size_t)BLOB_BYTES + 48), n, B, s)) != KZG_OK) return drained(rc);
        // the other staging set is free: its chunk (c - 1) was waited for to the end in the previous iteration
        if (c + 1 < n_chunks && (rc = copy_part(c + 1, 0)) != KZG_OK) return drained(rc);
        uint8_t* err = err_out ? err_out + c * G : nullptr;
        if ((rc = phase1_wait_locked(nullptr, err, s)) != KZG_OK) return drained(rc);
        if ((rc = phase2_launch_locked(nullptr, n, 0, s, 0, nullptr, false)) != KZG_OK) return drained(rc);
        if ((rc = finish_launch_locked(nullptr, 1, B, s)) != KZG_OK) return drained(rc);
        if (c + 1 < n_chunks && (rc = copy_part(c + 1, 1)) != KZG_OK) return drained(rc);
        if ((rc = finish_wait_locked(ok_out + c * G, s)) != KZG_OK) return drained(rc);
        if (err)
            for (size_t b = 0; b < B; b++)
                if (err[b]) ok_out[c * G + b] = false;
    }
    HIPCHK(hipStreamSynchronize(s->s_copy));
    // The staging sets are grow-only inside a call but not kept beyond it when they are large (2 x up to 16 GiB would starve
    // later workspaces on this device): above option hstage_keep_mib per set (default 4608 MiB - the default chunk of 32
    // batches of 1 024 blobs is 4.1 GiB) they are released here and allocated again by the next stream call, which costs
    // tens of ms per GiB (measured: a stream of 64 host batches fell from 0.37 M to 0.12 M blobs/s when every call
    // reallocated its 2 x 4.1 GiB); below it they stay with the handle until kzg_settings_free: at most 9 GiB retained.
    static const size_t keep_bytes = [] {
        const long v = opt_int("hstage_keep_mib", 4608);
        return (size_t)(v < 0 ? 0 : v) << 20;
    }();
    if (w.cap_hstage * ((size_t)BLOB_BYTES + 96) > keep_bytes) {
        for (auto& p : w.d_hstage) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        w.cap_hstage = 0;
    }
    return KZG_OK;
}

extern "C" KzgRet kzg_verify_blob_kzg_proof(bool* ok, const uint8_t* blob, const uint8_t commitment[48], const uint8_t proof[48],
                                            const KzgSettings* s) {
    // src/kzg_proof.rs:446-470; the batch entry's n == 1 branch is this very function (:482-489)
    return kzg_verify_blob_kzg_proof_batch(ok, blob, commitment, proof, 1, s);
}

extern "C" KzgRet kzg_verify_kzg_proof_batch(bool* ok, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                             const uint8_t* proofs, size_t n, const KzgSettings* s);
// ONE proof, in the reference's own form (src/kzg_proof.rs:384-396): e(C - [y]G, G2) == e(pi, [tau]G2 - [z]G2).  z and y are
// known before any point is decoded, so three chains run side by side on three streams (proof_kernels.hpp):
//   A: k_proof_select -> SCALARS ([y]G, the 68 line triples of Q = [tau]G2 - [z]G2) ......... then VERIFY3 (the pairing)
//   B: k_proof_decompress: the square roots of C and pi on two lanes of one wavefront -> VERIFY3's point inputs
//   C: the full decode of both points (subgroup test: 2 x 64 doublings) - only its verdict is awaited, beside the pairing
// Critical path max(A's SCALARS, B) + VERIFY3 instead of decode -> MSM -> pairing.  *general = true: Q is the identity
// (z = tau: possible only for who knows the setup's secret) - its lines mean nothing, the caller takes the general path.
struct ProofStreams {
    hipStream_t sa, sb, sc;
};
// the buffers and streams of the one-proof path, made on first use; the pinned mirror (w.h_buf) holds
//   [0, 64) z | y little-endian (verify_kzg_proof)   [64, 160) C | pi   [160, 168) status of the decompression
//   [176, 184) status of the full decode   [1024, 1408) VERIFY3's eight outputs
static KzgRet proof_reserve(ProofStreams& ps, const KzgSettings* s) {
    KzgRet rc = ws_reserve(s, 1, 1, STAGE_BLOBS);
    if (rc != KZG_OK) return rc;
    // The plain stream pair: these paths bring their own third stream and never use the CU-masked pair - and must not create
    // it: a CU-masked stream holds a hardware queue of its own, the process has 8 (GPU_MAX_HW_QUEUES), and streams that share a
    // queue run one behind the other.  Measured with 6-blob batches from T threads with a handle each (tools/prof/
    // concurrent_small_batches.py, profiles/r4_concurrent_small_batches.txt): with the masked pair made on every handle 778
    // batches/s at T = 8 and 414 at T = 16 (10 and 39 ms per call); without it 1 270 and 1 310 - 1 920 and 2 270 with 16 queues.
    select_streams(s, (size_t)-1);
    s->ws.kstamps_valid = false;  // (no launch group on this handle: the decode pass of these paths does not stamp)
    if (!s->d_proof) HIPCHK(hipMalloc(&s->d_proof, sizeof(Fp) * (SCALARS_INPUTS + VERIFY3_INPUTS)));
    const bool one_stream = !s->s_plain[1];  // option single_stream: everything in sequence (profiling)
    // A lane of the small-call queue runs chain C (the subgroup test) BEHIND chain B on B's stream: the square roots end at
    // ~0.65 ms and the test at ~1.5 ms, still before A's pairing (~1.7 ms), and a lane then holds two streams instead of three -
    // four lanes in flight fit the process's 8 hardware queues without two of their streams sharing one (streams that share a
    // queue run one behind the other: measured with three streams per lane, launches of 3 lanes took 2.6 ms each instead of 1.7).
    const bool two_streams = s->proof_two_streams && !one_stream;
    if (!one_stream && !two_streams && !s->s_aux) {
        const KzgRet rc_aux = stream_make(&s->s_aux, s->stream_priority);
        if (rc_aux != KZG_OK) return rc_aux;
    }
    ps.sa = s->s_plain[0];
    ps.sb = one_stream ? ps.sa : s->s_plain[1];
    ps.sc = one_stream ? ps.sa : two_streams ? ps.sb : s->s_aux;
    return KZG_OK;
}
// B and C: the two square roots -> VERIFY3's point inputs (event ev[6]); the full decode for the subgroup verdict
static KzgRet proof_points_launch(const ProofStreams& ps, const uint8_t* commitment, const uint8_t* proof, const KzgSettings* s) {
    Workspace& w = s->ws;
    uint8_t* const h = w.h_buf;
    memcpy(h + 64, commitment, 48);
    memcpy(h + 112, proof, 48);
    uint32_t* const h_pre = reinterpret_cast<uint32_t*>(h + 160);
    uint32_t* const h_full = reinterpret_cast<uint32_t*>(h + 176);
    h_pre[0] = h_pre[1] = h_full[0] = h_full[1] = G1_INVALID;
    Fp* const d_v3in = s->d_proof + SCALARS_INPUTS;
    hipLaunchKernelGGL(k_proof_decompress, dim3(1), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), ps.sb, h + 64, h + 112, 1, d_v3in, h_pre);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[6], ps.sb));
    struct RestoreS2 {
        const KzgSettings* s;
        hipStream_t keep;
        ~RestoreS2() { s->s2 = keep; }
    } restore{s, s->s2};
    s->s2 = ps.sc;
    KzgRet rc = launch_decode(s, h + 64, h + 112, 1, /*behind_sha=*/false);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipMemcpyAsync(h_full, w.d_pflag, 8, hipMemcpyDeviceToHost, ps.sc));
    return KZG_OK;
}
// A: the scalars' chain from z and y (8 little-endian limbs each; pinned host or device memory; ordered behind what stream
// sa already holds), then the pairing once the points are there; waits, and reads the verdicts
static KzgRet proof_tail_locked(bool* ok, bool* general, const ProofStreams& ps, const uint32_t* z, const uint32_t* y, const KzgSettings* s) {
    Workspace& w = s->ws;
    uint8_t* const h = w.h_buf;
    Fp* const d_scal_in = s->d_proof;
    Fp* const d_v3in = s->d_proof + SCALARS_INPUTS;
    Fp* const h_out = reinterpret_cast<Fp*>(h + 1024);
    hipLaunchKernelGGL(k_proof_select, dim3(1), dim3(128), 0, ps.sa, z, y, 0u, s->d_fixed_base, s->d_tau4, d_scal_in);
    KzgRet rc = run_program2(s->scalars, d_scal_in, nullptr, d_v3in + 6, 1, ps.sa);
    if (rc != KZG_OK) return rc;
    if (ps.sb != ps.sa) HIPCHK(hipStreamWaitEvent(ps.sa, s->ev[6], 0));
    HIPCHK(hipEventRecord(s->ev[4], ps.sa));
    if ((rc = run_program2(s->verify3, d_v3in, s->d_prep29, h_out, 1, ps.sa)) != KZG_OK) return rc;  // the eight outputs go straight into the mirror
    HIPCHK(hipEventRecord(s->ev[9], ps.sa));
    HIPCHK(hipStreamSynchronize(ps.sa));
    if (ps.sc != ps.sa) HIPCHK(hipStreamSynchronize(ps.sc));
    elapsed(&s->timings[3], s->ev[4], s->ev[9]);
    elapsed(&s->timings[0], s->ev[0], s->ev[9]);
    // the reference's order of errors: commitment (:372), then proof (:378); an encoding the decompression accepts can
    // still fail the subgroup test of the full decode
    const uint32_t* h_pre = reinterpret_cast<const uint32_t*>(h + 160);
    const uint32_t* h_full = reinterpret_cast<const uint32_t*>(h + 176);
    for (int i = 0; i < 2; i++)
        if (h_pre[i] == G1_INVALID || h_full[i] == G1_INVALID) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    const uint32_t* o = reinterpret_cast<const uint32_t*>(h_out);
    uint32_t any = 0, zq = 0;
    for (int i = 0; i < 72; i++) any |= o[i];
    for (int i = 72; i < 96; i++) zq |= o[i];
    if (!zq) {
        *general = true;
        return KZG_OK;
    }
    *ok = any == 0;
    return KZG_OK;
}
static void proof_drain(const KzgSettings* s) {  // nothing of the call stays in flight behind an error (the first message is kept)
    const std::string msg = g_err;
    for (hipStream_t st : {s->s_plain[0], s->s_plain[1], s->s_aux})
        if (st) (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    g_err = msg;
}

static KzgRet proof_single_locked(bool* ok, bool* general, const uint8_t* commitment, const uint8_t* z_be, const uint8_t* y_be, const uint8_t* proof,
                                  const KzgSettings* s) {
    *general = false;
    ProofStreams ps{};
    KzgRet rc = proof_reserve(ps, s);
    if (rc != KZG_OK) return rc;
    uint8_t* const h = s->ws.h_buf;
    reverse32(h, z_be);
    reverse32(h + 32, y_be);
    HIPCHK(hipEventRecord(s->ev[0], ps.sa));
    if ((rc = proof_points_launch(ps, commitment, proof, s)) != KZG_OK) return rc;
    return proof_tail_locked(ok, general, ps, reinterpret_cast<const uint32_t*>(h), reinterpret_cast<const uint32_t*>(h + 32), s);
}

// ONE blob from host memory (KzgProof::verify_blob_kzg_proof, src/kzg_proof.rs:446-470; the n == 1 branch of the batch form,
// :482-489): the point chains start at once; the host hashes the blob (65 us) while the blob crosses PCIe; z follows as 32
// bytes; the evaluation (one wavefront) gives y on the device; then the one-proof tail.  *general as in proof_single_locked.
static KzgRet blob_single_locked(bool* ok, bool* general, const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof, const KzgSettings* s,
                                 hostpool::Job* job) {
    *general = false;
    ProofStreams ps{};
    KzgRet rc = proof_reserve(ps, s);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    uint8_t* const h = w.h_buf;
    HIPCHK(hipEventRecord(s->ev[0], ps.sa));
    if ((rc = proof_points_launch(ps, commitment, proof, s)) != KZG_OK) return rc;
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4, ps.sa));
    HIPCHK(hipMemcpyAsync(w.d_stage_blobs, blob, BLOB_BYTES, hipMemcpyHostToDevice, ps.sa));  // (pageable: the call returns when the bytes have left)
    if (job) {  // (a caller that queued behind other calls hashed its blob while it waited: capi_coalesce.hpp)
        hostpool::finish(*job);
        memcpy(h + 192, job->z_le, 32);
    } else host_blob_challenge(h + 192, blob, commitment);
    HIPCHK(hipMemcpyAsync(w.d_z, h + 192, 32, hipMemcpyHostToDevice, ps.sa));
    if ((rc = launch_evaluate(s, w.d_stage_blobs, w.d_z, w.d_y, w.d_status, 1)) != KZG_OK) return rc;  // (on s->s1 = sa)
    HIPCHK(hipMemcpyAsync(h + 224, w.d_status, 4, hipMemcpyDeviceToHost, ps.sa));
    rc = proof_tail_locked(ok, general, ps, reinterpret_cast<const uint32_t*>(w.d_z), reinterpret_cast<const uint32_t*>(w.d_y), s);
    // A runtime failure inside the tail is reported as what it is: the mirrors below are only meaningful once the streams were
    // synchronised (before that they hold the G1_INVALID they were initialised with, or are still being written), and an
    // infrastructure failure must never look like the reference's Err(BadArgs) for an invalid input.
    if (rc != KZG_OK && rc != KZG_BADARGS) return rc;
    // the reference parses the commitment (:453), the blob (:454: a non-canonical element is BadArgs), then the proof (:455)
    const uint32_t* h_pre = reinterpret_cast<const uint32_t*>(h + 160);
    const uint32_t* h_full = reinterpret_cast<const uint32_t*>(h + 176);
    if (h_pre[0] == G1_INVALID || h_full[0] == G1_INVALID) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    if (*reinterpret_cast<const uint32_t*>(h + 224) != 0) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) :38-40
    return rc;
}

// MANY INDEPENDENT proofs, each with its own pairing and its own result ("verify_kzg_proof x N, each with its
// own pairing" - the revm precompile's workload when every transaction's proof needs its own verdict): the one-proof path with
// one program instance (= one workgroup, one CU) per proof - m proofs of a chunk run side by side (mirror layout: ProofsLaunch).
constexpr size_t PROOFS_CHUNK = 1024;
constexpr size_t SMALL_REQ_MAX_TUPLES = 256;  // a larger kzg_verify_kzg_proofs call fills launches by itself: it takes the handle's own path, not the small-call queue (capi_coalesce.hpp)  // proofs per launch (the eight-lane subgroup-test kernel takes up to 8 x 256 points)
constexpr size_t PROOFS_MIRROR_BYTES = 64 + 96 + 16 + 8 * sizeof(Fp);
// one launch of m proofs: the pinned mirror [m][z | y LE] | [m][48] C | [m][48] pi | [m][2] decompression status | [2 m] full-decode
// status (all C, then all pi) | [m][8] Fp VERIFY3's outputs, and the device buffers [m] SCALARS inputs | [m] VERIFY3 inputs
struct ProofsLaunch {
    size_t m;
    uint32_t* h_zy;
    uint8_t *h_c, *h_p;
    uint32_t *h_pre, *h_full;
    Fp *h_out, *d_scal_in, *d_v3in;
};
static KzgRet proofs_reserve(ProofStreams& ps, ProofsLaunch& pl, size_t m, int stage, const KzgSettings* s) {
    KzgRet rc = proof_reserve(ps, s);
    if (rc != KZG_OK) return rc;
    if (m > s->cap_proofs) {
        if (s->d_proofs) (void)hipFree(s->d_proofs);
        if (s->d_proofs_out) (void)hipFree(s->d_proofs_out);
        if (s->h_proofs) (void)hipHostFree(s->h_proofs);
        s->d_proofs = s->d_proofs_out = nullptr;
        s->h_proofs = nullptr;
        s->cap_proofs = 0;
        const size_t cap = std::max<size_t>(m, 64);
        HIPCHK(hipMalloc(&s->d_proofs, sizeof(Fp) * (SCALARS_INPUTS + VERIFY3_INPUTS) * cap));
        HIPCHK(hipMalloc(&s->d_proofs_out, sizeof(Fp) * VERIFY3_OUTPUTS * cap));
        HIPCHK(hipHostMalloc(&s->h_proofs, PROOFS_MIRROR_BYTES * cap));
        s->cap_proofs = cap;
    }
    if ((rc = ws_reserve(s, m, 1, stage)) != KZG_OK) return rc;
    uint8_t* const h = s->h_proofs;
    pl.m = m;
    pl.h_zy = reinterpret_cast<uint32_t*>(h);
    pl.h_c = h + 64 * m;
    pl.h_p = pl.h_c + 48 * m;
    pl.h_pre = reinterpret_cast<uint32_t*>(pl.h_p + 48 * m);
    pl.h_full = pl.h_pre + 2 * m;
    pl.h_out = reinterpret_cast<Fp*>(pl.h_full + 2 * m);
    pl.d_scal_in = s->d_proofs;
    pl.d_v3in = s->d_proofs + (size_t)SCALARS_INPUTS * m;
    return KZG_OK;
}
// streams B and C for m proofs: the square roots (two lanes per proof) -> VERIFY3's point inputs (event ev[6]); the full
// decode of all 2 m points for the subgroup verdicts
static KzgRet proofs_points_launch(const ProofStreams& ps, const ProofsLaunch& pl, const uint8_t* commitments, const uint8_t* proofs, const KzgSettings* s) {
    const size_t m = pl.m;
    if (commitments) {  // (null: the caller gathered the points of several requests into the mirror itself)
        memcpy(pl.h_c, commitments, 48 * m);
        memcpy(pl.h_p, proofs, 48 * m);
    }
    for (size_t i = 0; i < 2 * m; i++) pl.h_pre[i] = pl.h_full[i] = G1_INVALID;
    hipLaunchKernelGGL(k_proof_decompress, dim3((unsigned)((2 * m + 63) / 64)), dim3(64), 64 * PARK_UINT4_PER_THREAD * sizeof(uint4), ps.sb, pl.h_c, pl.h_p, (int)m,
                       pl.d_v3in, pl.h_pre);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[6], ps.sb));
    struct RestoreS2 {
        const KzgSettings* s;
        hipStream_t keep;
        ~RestoreS2() { s->s2 = keep; }
    } restore{s, s->s2};
    s->s2 = ps.sc;
    const KzgRet rc = launch_decode(s, pl.h_c, pl.h_p, m, /*behind_sha=*/false);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipMemcpyAsync(pl.h_full, s->ws.d_pflag, 8 * m, hipMemcpyDeviceToHost, ps.sc));
    return KZG_OK;
}
// stream A for m proofs: z and y (8 little-endian limbs each, stride_words apart; pinned host or device memory; behind what
// stream sa already holds) -> select -> SCALARS -> (points there) -> VERIFY3 -> the mirror; waits for A and C
static KzgRet proofs_tail_locked(const ProofStreams& ps, const ProofsLaunch& pl, const uint32_t* z, const uint32_t* y, uint32_t stride_words, const KzgSettings* s) {
    const size_t m = pl.m;
    hipLaunchKernelGGL(k_proof_select, dim3((unsigned)m), dim3(128), 0, ps.sa, z, y, stride_words, s->d_fixed_base, s->d_tau4, pl.d_scal_in);
    KzgRet rc = run_program2(s->scalars, pl.d_scal_in, nullptr, pl.d_v3in + 6, (int)m, ps.sa, 0, VERIFY3_INPUTS);
    if (rc != KZG_OK) return rc;
    if (ps.sb != ps.sa) HIPCHK(hipStreamWaitEvent(ps.sa, s->ev[6], 0));
    HIPCHK(hipEventRecord(s->ev[4], ps.sa));
    if ((rc = run_program2(s->verify3, pl.d_v3in, s->d_prep29, s->d_proofs_out, (int)m, ps.sa)) != KZG_OK) return rc;
    HIPCHK(hipMemcpyAsync(pl.h_out, s->d_proofs_out, sizeof(Fp) * VERIFY3_OUTPUTS * m, hipMemcpyDeviceToHost, ps.sa));
    HIPCHK(hipEventRecord(s->ev[9], ps.sa));
    HIPCHK(hipStreamSynchronize(ps.sa));
    if (ps.sc != ps.sa) HIPCHK(hipStreamSynchronize(ps.sc));
    elapsed(&s->timings[3], s->ev[4], s->ev[9]);
    elapsed(&s->timings[0], s->ev[0], s->ev[9]);
    return KZG_OK;
}
// proof i of the launch: a point the reference would not parse / the pairing's verdict / Q_i is the identity (z_i = tau)
struct ProofVerdict {
    bool bad_commitment, bad_proof, equal, q_identity;
};
static ProofVerdict proofs_verdict(const ProofsLaunch& pl, size_t i) {
    ProofVerdict r;
    r.bad_commitment = pl.h_pre[2 * i] == G1_INVALID || pl.h_full[i] == G1_INVALID;
    r.bad_proof = pl.h_pre[2 * i + 1] == G1_INVALID || pl.h_full[pl.m + i] == G1_INVALID;
    const uint32_t* o = reinterpret_cast<const uint32_t*>(pl.h_out + VERIFY3_OUTPUTS * i);
    uint32_t any = 0, zq = 0;
    for (int k = 0; k < 72; k++) any |= o[k];
    for (int k = 72; k < 96; k++) zq |= o[k];
    r.equal = any == 0;
    r.q_identity = zq == 0;
    return r;
}

static KzgRet proofs_independent_locked(bool* ok_out, uint8_t* err, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys, const uint8_t* proofs,
                                        size_t n, const KzgSettings* s, std::vector<size_t>& general) {
    for (size_t first = 0; first < n; first += PROOFS_CHUNK) {
        const size_t m = std::min(PROOFS_CHUNK, n - first);
        ProofStreams ps{};
        ProofsLaunch pl{};
        KzgRet rc = proofs_reserve(ps, pl, m, STAGE_CP, s);
        if (rc != KZG_OK) return rc;
        uint8_t* const h_zy = reinterpret_cast<uint8_t*>(pl.h_zy);
        for (size_t i = 0; i < m; i++) {
            const size_t g = first + i;
            err[g] = be_geq_r(zs + 32 * g) || be_geq_r(ys + 32 * g);  // (:360-371) - the instance still runs, its result is not looked at
            reverse32(h_zy + 64 * i, zs + 32 * g);
            reverse32(h_zy + 64 * i + 32, ys + 32 * g);
        }
        HIPCHK(hipEventRecord(s->ev[0], ps.sa));
        if ((rc = proofs_points_launch(ps, pl, commitments + 48 * first, proofs + 48 * first, s)) != KZG_OK) return rc;
        if ((rc = proofs_tail_locked(ps, pl, pl.h_zy, pl.h_zy + 8, 16u, s)) != KZG_OK) return rc;
        for (size_t i = 0; i < m; i++) {
            const size_t g = first + i;
            const ProofVerdict r = proofs_verdict(pl, i);
            if (r.bad_commitment || r.bad_proof) err[g] = 1;
            ok_out[g] = !err[g] && r.equal;
            if (!err[g] && r.q_identity) general.push_back(g);  // z = tau: the general path decides this one
        }
    }
    return KZG_OK;
}

// A FEW blobs from host memory (verify_blob_kzg_proof_batch, src/kzg_proof.rs:472-525, at the sizes a beacon node calls it
// with: the 6-9 blobs of one block): every blob gets its own pairing on its own CU instead of the random linear combination
// (:399-444) - the combination exists to save pairings on a CPU; here 64 pairings side by side cost what one does, and the
// decode -> MSM -> pairing chain of the combined form is the longer critical path (3.0 ms against 1.8 ms).  The result is the
// conjunction of the n verify_blob_kzg_proof verdicts - the statement the combined check tests probabilistically: it holds
// whenever this does, and could hold without it only if the hash-derived r hit one of at most n - 1 roots in Fr (< 2^-246).
// Host: per-blob challenges on SHA-NI threads while the blobs cross PCIe; device: evaluation -> y; then the m-proof tail with
// z and y read from device memory.  *general = true: some z_i = tau - the caller takes the combined path.
// ... generalised to the blobs of SEVERAL calls in one launch (capi_coalesce.hpp): part k = n blobs of one caller with its own
// verdict (the conjunction over ITS blobs), its own Err and its own z = tau flag; `job` (optional) = that caller's challenges
// on their way from the hashing pool (host_only.hpp), else they are hashed here.  m = the parts' blobs in all (<= 256).
struct BlobsPart {
    const uint8_t *blobs, *commitments, *proofs;
    size_t n;
    hostpool::Job* job;
    bool ok, bad, general;  // out
};
static KzgRet blobs_parts_locked(BlobsPart* parts, size_t n_parts, size_t m, const KzgSettings* s) {
    ProofStreams ps{};
    ProofsLaunch pl{};
    KzgRet rc = proofs_reserve(ps, pl, m, STAGE_BLOBS, s);
    if (rc != KZG_OK) return rc;
    Workspace& w = s->ws;
    uint8_t* const h_z = reinterpret_cast<uint8_t*>(pl.h_zy);  // (the mirror's z | y area: 32 m bytes of challenges)
    uint32_t* const h_status = pl.h_zy + 8 * m;
    std::vector<hostpool::JobRef> own(n_parts);
    hostpool::JoinOnExit joined;  // (every HIPCHK / rc return below leaves with the posted jobs finished: they read the callers' blobs)
    size_t off = 0;
    for (size_t k = 0; k < n_parts; k++) {
        BlobsPart& pt = parts[k];
        memcpy(pl.h_c + 48 * off, pt.commitments, 48 * pt.n);
        memcpy(pl.h_p + 48 * off, pt.proofs, 48 * pt.n);
        if (!pt.job) {  // hashed from here: the pool's workers start at once, this thread joins them after the copies below
            own[k] = hostpool::make(h_z + 32 * off, pt.blobs, pt.commitments, pt.n);
            static const size_t threads = (size_t)std::max(1L, std::min(16L, opt_int("host_threads", 16)));
            joined.add(own[k]);
            hostpool::post(own[k], threads);
        }
        off += pt.n;
    }
    HIPCHK(hipEventRecord(s->ev[0], ps.sa));
    if ((rc = proofs_points_launch(ps, pl, nullptr, nullptr, s)) != KZG_OK) return rc;
    HIPCHK(hipMemsetAsync(w.d_status, 0, 4 * m, ps.sa));
    off = 0;
    for (size_t k = 0; k < n_parts; k++) {  // (pageable: each call returns when its bytes have left)
        HIPCHK(hipMemcpyAsync(w.d_stage_blobs + off * (size_t)BLOB_BYTES, parts[k].blobs, parts[k].n * (size_t)BLOB_BYTES, hipMemcpyHostToDevice, ps.sa));
        off += parts[k].n;
    }
    off = 0;
    for (size_t k = 0; k < n_parts; k++) {
        if (parts[k].job) {
            hostpool::finish(*parts[k].job);
            memcpy(h_z + 32 * off, parts[k].job->z_le, 32 * parts[k].n);
        } else hostpool::finish(*own[k]);
        off += parts[k].n;
    }
    HIPCHK(hipMemcpyAsync(w.d_z, h_z, 32 * m, hipMemcpyHostToDevice, ps.sa));
    if ((rc = launch_evaluate(s, w.d_stage_blobs, w.d_z, w.d_y, w.d_status, m)) != KZG_OK) return rc;  // (on s->s1 = sa)
    HIPCHK(hipMemcpyAsync(h_status, w.d_status, 4 * m, hipMemcpyDeviceToHost, ps.sa));
    if ((rc = proofs_tail_locked(ps, pl, reinterpret_cast<const uint32_t*>(w.d_z), reinterpret_cast<const uint32_t*>(w.d_y), 8u, s)) != KZG_OK) return rc;
    off = 0;
    for (size_t k = 0; k < n_parts; k++) {
        BlobsPart& pt = parts[k];
        pt.ok = true;
        pt.bad = pt.general = false;
        for (size_t i = off; i < off + pt.n; i++) {
            const ProofVerdict r = proofs_verdict(pl, i);
            // (every parse failure of the batch form is the same BadArgs: commitments, blobs - a non-canonical element - and proofs)
            if (r.bad_commitment || r.bad_proof || h_status[i] != 0) pt.bad = true;
            if (r.q_identity) pt.general = true;
            pt.ok = pt.ok && r.equal;
        }
        off += pt.n;
    }
    return KZG_OK;
}
static KzgRet blobs_small_locked(bool* ok, bool* general, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n, const KzgSettings* s) {
    *general = false;
    BlobsPart part{blobs, commitments, proofs, n, nullptr, false, false, false};
    const KzgRet rc = blobs_parts_locked(&part, 1, n, s);
    if (rc != KZG_OK) return rc;
    if (part.bad) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    *general = part.general;
    *ok = part.ok;
    return KZG_OK;
}

extern "C" KzgRet kzg_verify_kzg_proof_batch(bool* ok, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                             const uint8_t* proofs, size_t n, const KzgSettings* s);
extern "C" KzgRet kzg_verify_kzg_proofs(bool* ok_out, uint8_t* err_out, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                        const uint8_t* proofs, size_t n, const KzgSettings* s) try {
    if (!ok_out || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) return KZG_OK;
    if (!commitments || !zs || !ys || !proofs) return fail(KZG_BADARGS, "null argument");
    std::vector<uint8_t> err_local;
    if (!err_out) err_local.resize(n);
    uint8_t* const err = err_out ? err_out : err_local.data();
    std::vector<size_t> general;
    if (small_enabled(s) && n <= SMALL_REQ_MAX_TUPLES) {  // a few proofs: a request in the shared handle's small-call queue (capi_coalesce.hpp)
        std::vector<uint8_t> gen(n);
        const KzgRet rc = small_proofs(ok_out, err, gen.data(), commitments, zs, ys, proofs, n, s);
        if (rc != KZG_OK) return rc;
        for (size_t i = 0; i < n; i++)
            if (gen[i] && !err[i]) general.push_back(i);
    } else {
        std::lock_guard<std::mutex> lk(s->mu);
        HIPCHK(hipSetDevice(s->device));
        const KzgRet rc = proofs_independent_locked(ok_out, err, commitments, zs, ys, proofs, n, s, general);
        if (rc != KZG_OK) {
            proof_drain(s);
            return rc;
        }
    }
    for (size_t g : general) {  // (z = tau under a setup whose secret the caller knows: test rigs)
        bool one = false;
        const KzgRet rc = kzg_verify_kzg_proof_batch(&one, commitments + 48 * g, zs + 32 * g, ys + 32 * g, proofs + 48 * g, 1, s);
        if (rc == KZG_BADARGS) err[g] = 1;
        else if (rc != KZG_OK) return rc;
        ok_out[g] = rc == KZG_OK && one;
    }
    if (!err_out)
        for (size_t i = 0; i < n; i++)
            if (err[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");
}

extern "C" KzgRet kzg_verify_kzg_proof(bool* ok, const uint8_t commitment[48], const uint8_t z[32], const uint8_t y[32],
                                       const uint8_t proof[48], const KzgSettings* s) try {
    // src/kzg_proof.rs:353-397.  One proof at a time takes the reference's own equation (proof_single_locked); option
    // proof_path=msm sends it through the batch form with the single scalar r^0 = 1 instead (round 3's path; A/B, cross-check):
    // e(pi, [tau]G2) == e(C - [y]G + [z]pi, G2)  <=>  e(pi, [tau - z]G2) == e(C - [y]G, G2)
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (!commitment || !z || !y || !proof) return fail(KZG_BADARGS, "null argument");
    static const bool msm_path = opt_is("proof_path", "msm");
    if (!msm_path) {
        if (be_geq_r(z) || be_geq_r(y)) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) :360-371
        bool general = false;
        if (small_enabled(s)) {  // a request in the shared handle's small-call queue: concurrent callers share launches (capi_coalesce.hpp)
            bool each = false;
            uint8_t err = 0, gen = 0;
            const KzgRet rc = small_proofs(&each, &err, &gen, commitment, z, y, proof, 1, s);
            if (rc != KZG_OK) return rc;
            if (err) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
            general = gen != 0;
            if (!general) *ok = each;
        } else {
            std::lock_guard<std::mutex> lk(s->mu);
            HIPCHK(hipSetDevice(s->device));
            const KzgRet rc = proof_single_locked(ok, &general, commitment, z, y, proof, s);
            if (rc != KZG_OK) {
                proof_drain(s);
                return rc;
            }
        }
        if (!general) return KZG_OK;
    }
    return kzg_verify_kzg_proof_batch(ok, commitment, z, y, proof, 1, s);
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (the queue's request list; nothing is thrown across the C ABI)
}

// KzgProof::verify_kzg_proof_batch (src/kzg_proof.rs:399-444) over byte inputs: n (commitment, z, y, proof) tuples checked
// with ONE random linear combination and ONE pairing.  Same pipeline as the blob batch minus challenge + evaluation.
extern "C" KzgRet kzg_verify_kzg_proof_batch(bool* ok, const uint8_t* commitments, const uint8_t* zs, const uint8_t* ys,
                                             const uint8_t* proofs, size_t n, const KzgSettings* s) try {
    if (!ok || !s) return fail(KZG_BADARGS, "null argument");
    if (n == 0) {  // compute_r_powers on an empty batch: both MSMs are the identity, e(O, .) == e(O, .)
        *ok = true;
        return KZG_OK;
    }
    if (!commitments || !zs || !ys || !proofs) return fail(KZG_BADARGS, "null argument");
    for (size_t i = 0; i < n; i++)
        if (be_geq_r(zs + 32 * i) || be_geq_r(ys + 32 * i)) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    // a few tuples: one pairing each, side by side on CUs of their own, and the conjunction of the verdicts (the reasoning at
    // blobs_small_locked) - 1.7 ms against the 2.9 ms of decode -> MSM -> pairing.  option small_batch_pairings_max=0: always combined
    static const size_t small_max = (size_t)std::max(0L, std::min(256L, opt_int("small_batch_pairings_max", 256)));
    static const bool msm_path = opt_is("proof_path", "msm");
    bool queued = false;
    if (n >= 2 && n <= small_max && small_enabled(s)) {  // ... as a request in the shared handle's small-call queue (capi_coalesce.hpp)
        std::vector<uint8_t> res(3 * n);
        bool* const each = reinterpret_cast<bool*>(res.data());
        const KzgRet qrc = small_proofs(each, res.data() + n, res.data() + 2 * n, commitments, zs, ys, proofs, n, s);
        if (qrc != KZG_OK) return qrc;
        bool all = true, any_general = false;
        for (size_t i = 0; i < n; i++) {
            if (res[n + i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
            any_general = any_general || res[2 * n + i];
            all = all && each[i];
        }
        if (!any_general) {
            *ok = all;
            return KZG_OK;
        }
        queued = true;  // (some z_i = tau: the combined path below decides)
    }
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    KzgRet rc;
    if (n >= 2 && n <= small_max && !msm_path && !queued) {
        std::vector<uint8_t> verdicts(2 * n);
        std::vector<size_t> general;
        bool* const each = reinterpret_cast<bool*>(verdicts.data());
        if ((rc = proofs_independent_locked(each, verdicts.data() + n, commitments, zs, ys, proofs, n, s, general)) != KZG_OK) {
            proof_drain(s);
            return rc;
        }
        if (general.empty()) {  // (some z_i = tau: the combined path below decides)
            bool all = true;
            for (size_t i = 0; i < n; i++) {
                if (verdicts[n + i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
                all = all && each[i];
            }
            *ok = all;
            return KZG_OK;
        }
    }
    select_streams(s, n);
    if ((rc = ws_reserve(s, n, 1, STAGE_CP)) != KZG_OK) return rc;
    Workspace& w = s->ws;
    w.kstamps_valid = false;
    HIPCHK(hipEventRecord(s->ev[0], s->s1));
    // z, y: big-endian -> the device's little-endian limb arrays (= the transcript's encoding)
    std::vector<uint8_t> records(160 * n);
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = records.data() + 160 * i;
        memcpy(o, commitments + 48 * i, 48);
        reverse32(o + 48, zs + 32 * i);
        reverse32(o + 80, ys + 32 * i);
        memcpy(o + 112, proofs + 48 * i, 48);
        memcpy(w.h_buf + 32 * i, o + 48, 32);
        memcpy(w.h_buf + 32 * n + 32 * i, o + 80, 32);
    }
    HIPCHK(hipMemcpyAsync(w.d_z, w.h_buf, 32 * n, hipMemcpyHostToDevice, s->s1));
    HIPCHK(hipMemcpyAsync(w.d_y, w.h_buf + 32 * n, 32 * n, hipMemcpyHostToDevice, s->s1));
    uint32_t* h_pflag = reinterpret_cast<uint32_t*>(w.h_buf + 64 * n);
    // A small call keeps the host out of the chain: the decode kernel reads the points where they lie (a pinned copy - the
    // caller's memory is pageable), the MSM waits for it through an event, and the point flags are looked at after the
    // pairing (flagged points have identity table rows: the work on them is wasted, not wrong).
    const bool chained = n <= LATENCY_MAX_BLOBS;
    struct RestoreS2 {  // the chained form aliases s2 to s1 for this call only
        const KzgSettings* s;
        hipStream_t keep;
        ~RestoreS2() { s->s2 = keep; }
    } restore_s2{s, s->s2};
    if (chained) {
        uint8_t* h_cp = w.h_buf + 72 * n;
        memcpy(h_cp, commitments, 48 * n);
        memcpy(h_cp + 48 * n, proofs, 48 * n);
        s->s2 = s->s1;  // nothing runs beside the decode here: one stream, no event between the kernels (select_streams resets the pair)
        if ((rc = launch_decode(s, h_cp, h_cp + 48 * n, n, /*behind_sha=*/false)) != KZG_OK) return rc;
        HIPCHK(hipMemcpyAsync(h_pflag, w.d_pflag, 8 * n, hipMemcpyDeviceToHost, s->s1));
    } else {
        HIPCHK(hipMemcpyAsync(w.d_stage_cp, commitments, 48 * n, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemcpyAsync(w.d_stage_cp + 48 * n, proofs, 48 * n, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        if ((rc = launch_decode(s, w.d_stage_cp, w.d_stage_cp + 48 * n, n, /*behind_sha=*/false)) != KZG_OK) return rc;
        HIPCHK(hipMemcpyAsync(h_pflag, w.d_pflag, 8 * n, hipMemcpyDeviceToHost, s->s2));
        HIPCHK(hipStreamSynchronize(s->s2));
        for (size_t i = 0; i < 2 * n; i++)
            if (h_pflag[i] == G1_INVALID) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    }
    w.pending_n = n;
    w.pending_b = 1;
    // n == 1: r^0 = 1 whatever the transcript hashes to, which is phase 2's n_total == 1 branch (scalars 1, z, -y)
    if ((rc = phase2_launch_locked(records.data(), n, 0, s, 0, nullptr, false)) != KZG_OK) return rc;
    if ((rc = finish_launch_locked(nullptr, 1, 1, s)) != KZG_OK) return rc;
    bool paired = false;  // *ok is written only once the inputs are known to be valid
    if ((rc = finish_wait_locked(&paired, s)) != KZG_OK) return rc;
    if (chained) {  // (finish_wait_locked has waited for the stream the flags were copied on)
        for (size_t i = 0; i < 2 * n; i++)
            if (h_pflag[i] == G1_INVALID) {
                *ok = false;
                return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
            }
    }
    *ok = paired;
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}
