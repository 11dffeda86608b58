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
    ~ProverBufs() {
        void* ptrs[] = {d_blobs, d_out, d_cm, d_sc, d_z, d_y, d_tp, d_ts, d_sorted, d_status, d_cflag, d_win, d_res, d_cpts};
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
// m MSMs: out[b] = compress(sum_i sc[b][i] * g1_points[i]); sc = plain canonical scalars (destroyed: GLV split in place)
static KzgRet setup_msm(const KzgSettings* s, ProverBufs& b, size_t m) {
    const size_t NT = (size_t)FE_PER_BLOB;
    const int total = (int)(m * NT);
    hipLaunchKernelGGL(k_commit_terms, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_tp, b.d_ts, total);
    hipLaunchKernelGGL(k_glv_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->s1, b.d_sc, total);
    MsmDesc d{};
    d.mult = s->d_g1_mult;
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
    d.chunks_per_block = m >= 16 ? 4 : 1;
    const unsigned slots = MSM_CHUNKS / d.chunks_per_block;
    KzgRet rc_save = msm_save_reserve(s, 8, slots, (unsigned)m);
    if (rc_save != KZG_OK) return rc_save;
#if KZG_AB_VARIANTS
    if (!fp29_enabled()) msm_window_launch<Curve32, false>(d, 8, slots, (unsigned)m, s->ws.d_msm_save, s->ws.cap_msm_save, s->s1);
    else
#endif
        msm_window_launch<Curve29, false>(d, 8, slots, (unsigned)m, s->ws.d_msm_save, s->ws.cap_msm_save, s->s1);
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
