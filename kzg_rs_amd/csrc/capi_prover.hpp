// capi_prover.hpp - prover side: commitments and proofs over the settings' Lagrange points.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// ---------------------------------------------------------------- prover side (not in the reference)
// c-kzg-4844's blob_to_kzg_commitment / compute_kzg_proof / compute_blob_kzg_proof: 4096-term MSMs over the settings'
// Lagrange points, PROVER_CHUNK blobs per launch of the MSM kernels.
constexpr size_t PROVER_CHUNK = 64;
struct ProverBufs {
    uint8_t *d_blobs = nullptr, *d_out = nullptr, *d_cm = nullptr;
    Fr *d_sc = nullptr, *d_z = nullptr, *d_y = nullptr;
    uint32_t *d_tp = nullptr, *d_ts = nullptr, *d_sorted = nullptr, *d_status = nullptr, *d_cflag = nullptr;
    G1Jac *d_win = nullptr, *d_res = nullptr;
    G1Aff* d_cpts = nullptr;
    G1Jac29Mem* d_cmult = nullptr;  // the commitments' multiples, a by-product of their eight-lane decode (never read)
    ~ProverBufs() {
        void* ptrs[] = {d_blobs, d_out, d_cm, d_sc, d_z, d_y, d_tp, d_ts, d_sorted, d_status, d_cflag, d_win, d_res, d_cpts, d_cmult};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
    }
    KzgRet alloc() {
        const size_t NT = (size_t)FE_PER_BLOB, CH = PROVER_CHUNK;
        HIPCHK(hipMalloc(&d_blobs, (size_t)BLOB_BYTES * CH));
        HIPCHK(hipMalloc(&d_sc, sizeof(Fr) * NT * CH));
        HIPCHK(hipMalloc(&d_tp, 4 * NT * CH));
        HIPCHK(hipMalloc(&d_ts, 4 * NT * CH));
        HIPCHK(hipMalloc(&d_sorted, 4 * NT * CH * MSM_WINDOWS));
        HIPCHK(hipMalloc(&d_status, 4 * CH));
        HIPCHK(hipMalloc(&d_win, sizeof(G1Jac) * MSM_WINDOWS * CH));
        HIPCHK(hipMalloc(&d_res, sizeof(G1Jac) * CH));
        HIPCHK(hipMalloc(&d_out, 48 * CH));
        HIPCHK(hipMalloc(&d_z, sizeof(Fr) * CH));
        HIPCHK(hipMalloc(&d_y, sizeof(Fr) * CH));
        HIPCHK(hipMalloc(&d_cm, 48 * CH));
        HIPCHK(hipMalloc(&d_cflag, 4 * CH));
        HIPCHK(hipMalloc(&d_cpts, sizeof(G1Aff) * CH));
        HIPCHK(hipMalloc(&d_cmult, sizeof(G1Jac29Mem) * MSM_CHUNKS_LATENCY * CH));
        return KZG_OK;
    }
};
static KzgRet prover_ready(const KzgSettings* s) {
    if (!s->d_g1_mult) return fail(KZG_BADARGS, "these settings were not loaded from a trusted-setup file");
    if (!s->g1_in_subgroup) return fail(KZG_BAD_SETUP, "a G1 setup point is outside the r-torsion subgroup");
    return KZG_OK;
}
// the prover's buffers live on the handle from the first prover call on (fourteen hipMallocs and frees were 0.8 ms of every
// call); the caller holds the handle's lock
static KzgRet prover_bufs(const KzgSettings* s, ProverBufs** out) {
    if (!s->prover) {
        ProverBufs* b = new ProverBufs();
        const KzgRet rc = b->alloc();
        if (rc != KZG_OK) {
            delete b;
            return rc;
        }
        s->prover = b;
    }
    *out = s->prover;
    return KZG_OK;
}
static void prover_release(const KzgSettings* s) {
    delete s->prover;
    s->prover = nullptr;
}
static KzgRet fb_rows_ready(const KzgSettings* s);
// m MSMs: out[b] = compress(sum_i sc[b][i] * g1_points[i]); sc = plain canonical scalars (destroyed: GLV split in place)
static KzgRet setup_msm(const KzgSettings* s, ProverBufs& b, size_t m) {
    const size_t NT = (size_t)FE_PER_BLOB;
    const int total = (int)(m * NT);
    // One or two blobs: every sum through the FIXED-BASE form of kzg_g1_msm_setup (msm_fixed.hpp: a 4 096-term sum in ~0.8-1.1 ms;
    // the window kernels below need ~1.6 ms for one blob and pay off from a handful of blobs per launch on).  The row sums stay on the
    // main stream here: the proof path's second stream is busy with the commitments' decode.
    static const size_t fb_max_blobs = (size_t)std::max(0L, std::min(8L, opt_int("prover_fixed_base_max_blobs", 2)));
    if (m <= fb_max_blobs && msm_affine_enabled() && s->d_g1_mult_aff && (size_t)s->n_g1 == NT && NT * 2 * FBM_WINDOWS <= 131072) {
        KzgRet rc = fb_rows_ready(s);
        if (rc != KZG_OK) return rc;
        const int L = FBM_SLICE_ENTRIES;
        const unsigned Z = fb_max_blocks((size_t)FBM_WINDOWS * NT, L);
        Workspace& w = s->ws;
        const size_t save_bytes = (size_t)Z * 256 * MSM_SAVE2_WORDS * 4;
        if (save_bytes > w.cap_msm_save) {
            if (w.d_msm_save) (void)hipFree(w.d_msm_save);
            w.d_msm_save = nullptr;
            w.cap_msm_save = 0;
            HIPCHK(hipMalloc(&w.d_msm_save, save_bytes));
            w.cap_msm_save = save_bytes;
        }
        int gp = 0;
        (void)msm_large_tail_groups(Z, std::max(11, (int)((Z + MSM_FOLD_MAX_GROUPS - 1) / MSM_FOLD_MAX_GROUPS)), &gp);
        uint8_t* tmp = nullptr;
        if ((rc = g1msm_scratch(s, fb_tail_bytes(gp), &tmp)) != KZG_OK) return rc;
        for (size_t k = 0; k < m; k++)
            HIPCHK(fb_msm_launch(b.d_sc + k * NT, s->d_g1_flag, (int)NT, s->n_g1, s->d_g1_fb_rows, s->d_fb_plan, b.d_sorted, w.d_msm_save, tmp, b.d_res + k, L, 11, nullptr, s->s1));
        hipLaunchKernelGGL(k_jac_compress_n, dim3(1), dim3(64), 0, s->s1, b.d_res, b.d_out, (int)m);
        HIPCHK(hipGetLastError());
        return KZG_OK;
    }
    hipLaunchKernelGGL(k_commit_terms, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_tp, b.d_ts, total);
    hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_sc, total);
    // the setup's AFFINE table rows when the handle has them (mixed additions: 8M + 3S instead of 12M + 4S per bucket entry), and the
    // (window, chunk group) workgroups sized so that their sorted lists stay in LDS: one chunk (4 096 entries) per workgroup for a few
    // blobs, two (8 192) from 16 blobs on - rounds 2-5 ran Jacobian rows with four chunks per workgroup and the list in global memory
    const bool aff = msm_affine_enabled() && s->d_g1_mult_aff != nullptr;
    MsmDesc d{};
    d.mult = aff ? (void*)s->d_g1_mult_aff : s->d_g1_mult;
    d.pflag = s->d_g1_flag;
    d.scalars = b.d_sc;
    d.term_point = b.d_tp;
    d.term_scalar = b.d_ts;
    d.sorted = b.d_sorted;
    d.window_sums = b.d_win;
    d.nterms[0] = d.nterms[1] = (int)NT;
    d.max_terms = (int)NT;
    d.stride = (int)NT;
    d.slices = 1;
    d.chunks = MSM_CHUNKS;
    d.chunks_per_block = aff ? (m >= 16 ? 2 : 1) : (m >= 16 ? 4 : 1);
    const unsigned slots = MSM_CHUNKS / d.chunks_per_block;
    KzgRet rc_save = msm_save_reserve(s, 8, slots, (unsigned)m);
    if (rc_save != KZG_OK) return rc_save;
#if KZG_AB_VARIANTS
    if (!fp29_enabled()) msm_window_launch<Curve32, false>(d, 8, slots, (unsigned)m, s->ws.d_msm_save, s->ws.cap_msm_save, s->s1);
    else
#endif
        if (aff) msm_window_launch<Curve29Aff, true>(d, 8, slots, (unsigned)m, s->ws.d_msm_save, s->ws.cap_msm_save, s->s1);
    else msm_window_launch<Curve29, false>(d, 8, slots, (unsigned)m, s->ws.d_msm_save, s->ws.cap_msm_save, s->s1);
    hipLaunchKernelGGL(k_msm_combine, dim3((unsigned)m), dim3(64), 0, s->s1, b.d_win, b.d_res, (int)slots, 8);
    hipLaunchKernelGGL(k_jac_compress_n, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s->s1, b.d_res, b.d_out, (int)m);
    HIPCHK(hipGetLastError());
    return KZG_OK;
}

// C_b = sum_i blob_b[i] * g1_points[i].  blobs: n * 131072 bytes, host memory; out: n * 48 bytes.
extern "C" KzgRet kzg_blob_to_kzg_commitment(uint8_t* out48, const uint8_t* blobs, size_t n, const KzgSettings* s) {
    if (!s || (n && (!out48 || !blobs))) return fail(KZG_BADARGS, "null argument");
    KzgRet rc = prover_ready(s);
    if (rc != KZG_OK || n == 0) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    ProverBufs* bp = nullptr;
    if ((rc = prover_bufs(s, &bp)) != KZG_OK) return rc;
    ProverBufs& b = *bp;
    for (size_t lo = 0; lo < n; lo += PROVER_CHUNK) {
        const size_t m = std::min(PROVER_CHUNK, n - lo);
        const int total = (int)(m * FE_PER_BLOB);
        HIPCHK(hipMemcpyAsync(b.d_blobs, blobs + (size_t)BLOB_BYTES * lo, (size_t)BLOB_BYTES * m, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemsetAsync(b.d_status, 0, 4 * m, s->s1));
        hipLaunchKernelGGL(k_blob_scalars, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_blobs, b.d_sc, b.d_status, total);
        if ((rc = setup_msm(s, b, m)) != KZG_OK) return rc;
        std::vector<uint32_t> st(m);
        HIPCHK(hipMemcpyAsync(out48 + 48 * lo, b.d_out, 48 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(st.data(), b.d_status, 4 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        for (size_t i = 0; i < m; i++)
            if (st[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) Blob::as_polynomial, src/dtypes.rs:48-57
    }
    return KZG_OK;
}

// Shared body of compute_kzg_proof (zs given) and compute_blob_kzg_proof (z = the Fiat-Shamir challenge of (blob,
// commitment), src/kzg_proof.rs:46-72): y = p(z), pi = sum_i q_i g1_points[i] with q the quotient (fr_kernels.hpp).
static KzgRet compute_proofs(uint8_t* proofs48, uint8_t* ys32, const uint8_t* blobs, const uint8_t* zs, const uint8_t* commitments,
                             size_t n, const KzgSettings* s) {
    KzgRet rc = prover_ready(s);
    if (rc != KZG_OK || n == 0) return rc;
    if (zs)
        for (size_t i = 0; i < n; i++)
            if (be_geq_r(zs + 32 * i)) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");  // (sic) :36-41
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);
    ProverBufs* bp = nullptr;
    if ((rc = prover_bufs(s, &bp)) != KZG_OK) return rc;
    ProverBufs& b = *bp;
    std::vector<uint8_t> le(32 * PROVER_CHUNK);
    hipStream_t const side = s->s2 && s->s2 != s->s1 ? s->s2 : nullptr;  // the commitments' validity check runs beside the chain
    struct DrainSide {  // (nothing of the call stays in flight on the side stream, whatever path leaves)
        hipStream_t st;
        ~DrainSide() {
            if (st) (void)hipStreamSynchronize(st);
        }
    } drain_side{side};
    // a few blobs: their Fiat-Shamir challenges from the host's SHA-NI cores while the blobs cross PCIe - a chain is 2.8 ms on
    // GPU lanes however few blobs there are and 65 us on a host core (the verifier's small host batches do the same)
    const size_t host_hash_max = std::min<size_t>(PROVER_CHUNK, host_challenge_max_blobs());
    for (size_t lo = 0; lo < n; lo += PROVER_CHUNK) {
        const size_t m = std::min(PROVER_CHUNK, n - lo);
        HIPCHK(hipMemcpyAsync(b.d_blobs, blobs + (size_t)BLOB_BYTES * lo, (size_t)BLOB_BYTES * m, hipMemcpyHostToDevice, s->s1));
        HIPCHK(hipMemsetAsync(b.d_status, 0, 4 * m, s->s1));
        if (zs) {
            for (size_t i = 0; i < m; i++) reverse32(le.data() + 32 * i, zs + 32 * (lo + i));
            HIPCHK(hipMemcpyAsync(b.d_z, le.data(), 32 * m, hipMemcpyHostToDevice, s->s1));
            HIPCHK(hipStreamSynchronize(s->s1));  // `le` is reused by the next chunk
        } else {
            HIPCHK(hipMemcpyAsync(b.d_cm, commitments + 48 * lo, 48 * m, hipMemcpyHostToDevice, s->s1));
            hipStream_t dec = s->s1;
            if (side) {  // decode + subgroup test of the commitments (2.5 ms on one lane each): only their verdict is needed
                HIPCHK(hipEventRecord(s->ev[5], s->s1));
                HIPCHK(hipStreamWaitEvent(side, s->ev[5], 0));
                dec = side;
            }
            // the commitments' validity (decompression + subgroup test): EIGHT lanes per point where every workgroup of eight points can
            // have a CU to itself (msm.hpp k_g1_decode_multiples29_quads: 1.2 ms instead of the one-lane kernel's 2.5 - the longest chain of
            // a one-blob proof call); its table multiples are a by-product nobody reads
            static const bool dec_quads = fp29_enabled() && ab_flag("decode_quads", true);
            const unsigned qblocks = (unsigned)((m + DECQ_POINTS_PER_BLOCK - 1) / DECQ_POINTS_PER_BLOCK);
            if (dec_quads && (int)qblocks <= s->n_cus && DYN_LDS(k_g1_decode_multiples29_quads<MSM_CHUNKS_LATENCY>, DECQ_LDS_BYTES) == hipSuccess)
                hipLaunchKernelGGL(k_g1_decode_multiples29_quads<MSM_CHUNKS_LATENCY>, dim3(qblocks), dim3(64), DECQ_LDS_BYTES, dec, (const uint8_t*)b.d_cm, (const uint8_t*)b.d_cm, (int)m,
                                   b.d_cpts, b.d_cflag, b.d_cmult, (int)m, (int)PROVER_CHUNK);
            else
                hipLaunchKernelGGL(k_g1_decode, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, dec, b.d_cm, b.d_cm, (int)m, b.d_cpts, b.d_cflag, (int)m, 1);
            if (m <= host_hash_max) {
                const size_t threads = (size_t)std::max(1L, std::min(16L, opt_int("host_threads", 16)));
                host_blob_challenges(le.data(), blobs + (size_t)BLOB_BYTES * lo, commitments + 48 * lo, m, threads);
                HIPCHK(hipMemcpyAsync(b.d_z, le.data(), 32 * m, hipMemcpyHostToDevice, s->s1));
                HIPCHK(hipStreamSynchronize(s->s1));  // `le` is reused by the next chunk
            } else if ((rc = launch_challenge(s, b.d_blobs, b.d_cm, b.d_z, m)) != KZG_OK) {
                return rc;
            }
        }
        if ((rc = launch_evaluate(s, b.d_blobs, b.d_z, b.d_y, b.d_status, m, /*alone=*/true)) != KZG_OK) return rc;
        hipLaunchKernelGGL(k_blob_quotient, dim3((unsigned)m), dim3(64), 0, s->s1, b.d_blobs, b.d_z, b.d_y, s->d_M, b.d_sc, b.d_status);
        HIPCHK(hipGetLastError());
        if ((rc = setup_msm(s, b, m)) != KZG_OK) return rc;
        std::vector<uint32_t> st(m), cf(m, 0);
        std::vector<uint8_t> yl(32 * m);
        HIPCHK(hipMemcpyAsync(proofs48 + 48 * lo, b.d_out, 48 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(st.data(), b.d_status, 4 * m, hipMemcpyDeviceToHost, s->s1));
        HIPCHK(hipMemcpyAsync(yl.data(), b.d_y, 32 * m, hipMemcpyDeviceToHost, s->s1));
        if (!zs) HIPCHK(hipMemcpyAsync(cf.data(), b.d_cflag, 4 * m, hipMemcpyDeviceToHost, side ? side : s->s1));
        HIPCHK(hipStreamSynchronize(s->s1));
        if (!zs && side) HIPCHK(hipStreamSynchronize(side));
        for (size_t i = 0; i < m; i++) {
            if (cf[i] == G1_INVALID || st[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
            if (ys32) reverse32(ys32 + 32 * (lo + i), yl.data() + 32 * i);
        }
    }
    return KZG_OK;
}
extern "C" KzgRet kzg_compute_kzg_proof(uint8_t* proofs48, uint8_t* ys32, const uint8_t* blobs, const uint8_t* zs, size_t n,
                                        const KzgSettings* s) {
    if (!s || (n && (!proofs48 || !ys32 || !blobs || !zs))) return fail(KZG_BADARGS, "null argument");
    return compute_proofs(proofs48, ys32, blobs, zs, nullptr, n, s);
}
extern "C" KzgRet kzg_compute_blob_kzg_proof(uint8_t* proofs48, const uint8_t* blobs, const uint8_t* commitments, size_t n,
                                             const KzgSettings* s) {
    if (!s || (n && (!proofs48 || !blobs || !commitments))) return fail(KZG_BADARGS, "null argument");
    return compute_proofs(proofs48, nullptr, blobs, nullptr, commitments, n, s);
}

// ---------------------------------------------------------------- G1 msm_variable_base over the handle's own points
// out = sum_i scalars[i] * g1_points[i mod N] (N = 4 096 Lagrange points, bit-reversal permuted as the handle keeps them:
// src/trusted_setup.rs:20-26; the shape of BASELINE.json configs[3], "2^20 trusted-setup points x random Fr scalars", and of the
// reference's own call sites src/kzg_proof.rs:419,429,430 when their points are setup points).  scalars: n x 32 big-endian bytes,
// any value below 2^256 (reduced mod r like Scalar::from_raw).  Nothing is decoded per call: the tables were made when the setup
// was loaded.  Two forms, same result bit for bit:
//   fixed   the fixed-base form of msm_fixed.hpp, the default at EVERY size: 16-bit signed windows over rows 2^(16 v) P_j (16 MB with the
//           doubled rows, made by the first call), 16 bucket additions per term instead of 32, one shared bucket set.  Measured against
//           the window form at 1 / 1 024 / 4 096 / 32 768 / 65 536 / 2^20 terms: 0.39 / 0.69 / 0.77 / 1.04 / 1.30 / 5.2 ms against
//           1.14 / 1.19 / 1.63 / 1.50 / 1.83 / 8.1 ms
//   window  the verification path's window kernel (GLV, 8-bit windows) over the setup's affine table rows: the fallback when the
//           fixed-base form's bucket sums would pass 2 GiB (above ~2^24.6 terms) or the handle has no affine rows (A/B build, fp29=0)
// option g1_msm_setup_form = window | fixed forces one (tests, A/B).  timings: [2] the MSM, [6] = 0 (no decode, no tables).
constexpr size_t FBM_MIN_TERMS = 1;
static KzgRet fb_rows_ready(const KzgSettings* s) {
    if (s->d_g1_fb_rows) return KZG_OK;
    const int N = s->n_g1;
    DevTmp t_jac;
    G1Aff29Mem* rows = nullptr;
    HIPCHK(hipMalloc(&t_jac.p, sizeof(G1Jac29Mem) * (size_t)(2 * FBM_WINDOWS - 1) * N));
    HIPCHK(hipMalloc(&rows, sizeof(G1Aff29Mem) * (size_t)2 * FBM_WINDOWS * N));
    DevTmp own;
    own.p = rows;  // (released on an error path)
    if (!s->d_fb_plan) HIPCHK(hipMalloc(&s->d_fb_plan, 4 * FBM_PLAN_WORDS));
    HIPCHK(hipMemcpyAsync(rows, s->d_g1_mult_aff, sizeof(G1Aff29Mem) * (size_t)N, hipMemcpyDeviceToDevice, s->s1));  // row 0 = P_j
    hipLaunchKernelGGL(k_fb_build_rows, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s->s1, (const G1Aff29Mem*)s->d_g1_mult_aff, (const uint32_t*)s->d_g1_flag,
                       t_jac.as<G1Jac29Mem>(), N);
    const int m = (2 * FBM_WINDOWS - 1) * N;
    hipLaunchKernelGGL(k_jac29_to_aff29, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s->s1, (const G1Jac29Mem*)t_jac.as<G1Jac29Mem>(), rows + N, m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->s1));
    own.p = nullptr;
    s->d_g1_fb_rows = rows;
    return KZG_OK;
}
extern "C" KzgRet kzg_g1_msm_setup(uint8_t out[48], const uint8_t* scalars, size_t n, const KzgSettings* s) try {
    if (!s || !out || (n && !scalars)) return fail(KZG_BADARGS, "null argument");
    if (n > ((size_t)1 << 26)) return fail(KZG_BADARGS, "kzg_g1_msm_setup: more than 2^26 terms");
    KzgRet rc = prover_ready(s);
    if (rc != KZG_OK) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    select_streams(s, (size_t)-1);  // stand-alone pieces run on the plain stream pair
    if ((rc = ws_reserve(s, (n + 1) / 2 + 1, 1, STAGE_NONE)) != KZG_OK) return rc;
    Workspace& w = s->ws;
    StreamDrain drain{s->s1};
    const bool aff = msm_affine_enabled() && s->d_g1_mult_aff;
    static const int forced = opt_is("g1_msm_setup_form", "window") ? 1 : opt_is("g1_msm_setup_form", "fixed") ? 2 : 0;
    static const int L = (int)std::max(1024L, std::min((long)FBM_SLICE_ENTRIES, ab_int("g1_msm_fb_slice", (long)FBM_SLICE_ENTRIES)));
    const unsigned Z = fb_max_blocks((size_t)FBM_WINDOWS * n, L);
    const size_t save_bytes = (size_t)Z * 256 * MSM_SAVE2_WORDS * 4;
    const bool fixed = aff && n > 0 && (size_t)s->n_g1 * 2 * FBM_WINDOWS <= 131072 && save_bytes <= ((size_t)2 << 30) && (forced == 2 || (forced == 0 && n >= FBM_MIN_TERMS));
    if (n && (rc = g1_msm_scalars_in(s, scalars, n, 0)) != KZG_OK) return rc;
    s->timings[6] = 0.0f;
    if (!fixed) {
        const G1MsmTables tb{aff ? (const void*)s->d_g1_mult_aff : (const void*)s->d_g1_mult, s->d_g1_flag, s->n_g1, aff, true};
        return g1_msm_core(s, n, tb, out);
    }
    if ((rc = fb_rows_ready(s)) != KZG_OK) return rc;
    if (save_bytes > w.cap_msm_save) {  // (grow-only, like msm_save_reserve; this form's layers are 48 KB each)
        if (w.d_msm_save) (void)hipFree(w.d_msm_save);
        w.d_msm_save = nullptr;
        w.cap_msm_save = 0;
        HIPCHK(hipMalloc(&w.d_msm_save, save_bytes));
        w.cap_msm_save = save_bytes;
    }
    static const int fold_per_opt = (int)std::max(2L, std::min(64L, ab_int("g1_msm_fold_per", 11)));
    int gp = 0;
    (void)msm_large_tail_groups(Z, std::max(fold_per_opt, (int)((Z + MSM_FOLD_MAX_GROUPS - 1) / MSM_FOLD_MAX_GROUPS)), &gp);
    uint8_t* tmp = nullptr;
    if ((rc = g1msm_scratch(s, fb_tail_bytes(gp), &tmp)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    HIPCHK(fb_msm_launch(w.d_scalars, s->d_g1_flag, (int)n, s->n_g1, s->d_g1_fb_rows, s->d_fb_plan, w.d_sorted, w.d_msm_save, tmp, w.d_ab, L, fold_per_opt,
                         w.d_ktime ? w.d_ktime + 8 : nullptr, s->s1, s->s2, s->ev[7], s->ev[8]));
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    hipLaunchKernelGGL(k_jac_compress, dim3(1), dim3(64), 0, s->s1, w.d_ab, w.d_bytes, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, w.d_bytes, 48, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    elapsed(&s->timings[2], s->ev[2], s->ev[3]);
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the call");  // (nothing is thrown across the C ABI)
}
