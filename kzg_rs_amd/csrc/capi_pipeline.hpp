// capi_pipeline.hpp - many independent batches through ONE call at the chip's rate: launch groups kept in flight inside the library,
// on every device of the handle.  Part of the single translation unit kzg_capi.hip; not a stand-alone header.  Host code only.
//
// At n = 1 024 every phase of a batch is a latency-bound serial chain, so batches share kernel launches (a launch GROUP of B
// batches) and several groups overlap.  Rounds 1-2 kept the groups in flight from Python
// (kzg_rs_amd/distributed.py PipelinedVerifier) - not an entry point a caller of the reference has.  Here the same fixed-order
// software pipeline runs behind one C call: the handle grows private "lanes" (handles on the same device with their own
// streams and workspace, reading the parent's tables), group t runs on lane t mod (F + 1), and iteration t of the host loop
// does, in this order,
//     phase1_launch(t);   phase1_wait(t - d1) + the group's transcript hashes + phase2_launch + finish_launch;   finish_wait(t - d1 - 1)
// with d1 = F - 1 groups between the first two steps (F = groups in flight; F = 1: everything in sequence on the handle itself).
// Every batch of every group is a complete, independent verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-525): its own
// transcript, challenge r, MSMs, pairing, boolean.
//
// On a handle over SEVERAL devices (capi_multi.hpp) every group goes to the device that owns its memory
// (hipPointerGetAttributes; a group whose three arrays lie on different devices is refused) and each device runs the pipeline
// above over its own groups on a host thread of its own: independent batches need no exchange, so a caller of the reference's
// one-process signature with a stream of 1 024-blob batches gets N devices' worth of the single-device rate from one call.

// The pipeline on ONE device: the groups idx[0 .. K) of the caller's arrays on handle c and its lanes.  The calling thread has
// set c's device and owns c (its lock, or the multi-device handle's).  ok_out / err: [all groups][B], written at idx[i] B.
static KzgRet groups_pipeline_locked(bool* ok_out, uint8_t* err, const void* const* d_blobs, const void* const* d_commitments,
                                     const void* const* d_proofs, const size_t* idx, size_t K, size_t n, size_t B, size_t in_flight,
                                     const KzgSettings* c) {
    if (K == 0) return KZG_OK;
    const size_t F = std::max<size_t>(1, std::min<size_t>(in_flight ? in_flight : 4, 8));
    const size_t d1 = F - 1, d3 = F > 1 ? 1 : 0, S = d1 + d3 + 1;  // S handles: the handle itself + S - 1 lanes
    KzgRet rc = pipeline_lanes(c, S - 1);
    if (rc != KZG_OK) return rc;
    auto lane = [&](size_t i) -> const KzgSettings* { return lane_of(c, i % S); };
    for (size_t i = 0; i < std::min(S, K); i++)
        if ((rc = ws_reserve(lane(i), n * B, B, STAGE_NONE)) != KZG_OK) return rc;
    // an error leaves groups in flight on the lanes: drain every stream before it goes back (the first message is kept)
    auto drained = [&](KzgRet code) {
        const std::string msg = g_err;
        for (size_t i = 0; i < S; i++) {
            const KzgSettings* h = lane(i);
            (void)hipStreamSynchronize(h->s1);
            if (h->s2) (void)hipStreamSynchronize(h->s2);
            if (h->s_sha) (void)hipStreamSynchronize(h->s_sha);
            h->ws.pending_n = h->ws.pending_b = h->ws.finish_b = 0;
        }
        (void)hipGetLastError();
        g_err = msg;
        return code;
    };
    for (size_t t = 0; t < K + d1 + d3; t++) {
        if (t < K) {
            const size_t g = idx[t];
            if (!d_blobs[g] || !d_commitments[g] || !d_proofs[g]) return drained(fail(KZG_BADARGS, "null group"));
            if ((rc = phase1_launch_locked(d_blobs[g], d_commitments[g], d_proofs[g], n, B, lane(t))) != KZG_OK) return drained(rc);
        }
        if (t >= d1 && t - d1 < K) {
            const size_t i = t - d1;
            const KzgSettings* h = lane(i);
            if ((rc = phase1_wait_locked(nullptr, err + idx[i] * B, h)) != KZG_OK) return drained(rc);
            if ((rc = phase2_launch_locked(nullptr, n, 0, h, 0, nullptr, false)) != KZG_OK) return drained(rc);
            if ((rc = finish_launch_locked(nullptr, 1, B, h)) != KZG_OK) return drained(rc);
        }
        if (t >= d1 + d3 && t - d1 - d3 < K) {
            const size_t k = t - d1 - d3, g = idx[k];
            if ((rc = finish_wait_locked(ok_out + g * B, lane(k))) != KZG_OK) return drained(rc);
            for (size_t b = 0; b < B; b++)
                if (err[g * B + b]) ok_out[g * B + b] = false;
        }
    }
    return KZG_OK;
}

// groups: n_groups launch groups of batches_per_group batches of n blobs each; group g at d_blobs[g] / d_commitments[g] /
// d_proofs[g] (device memory, batches contiguous) - the same pointers may repeat.  ok_out / err_out: [n_groups][batches_per_group].
extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_groups_device(bool* ok_out, uint8_t* err_out, const void* const* d_blobs,
                                                                const void* const* d_commitments, const void* const* d_proofs, size_t n,
                                                                size_t batches_per_group, size_t n_groups, size_t in_flight, const KzgSettings* s) try {
    KZG_ENTER(s && ok_out && d_blobs && d_commitments && d_proofs && n && batches_per_group);
    if (n_groups == 0) return KZG_OK;
    const size_t B = batches_per_group, K = n_groups;
    std::vector<uint8_t> err_local;
    if (!err_out) err_local.resize(K * B);
    uint8_t* const err = err_out ? err_out : err_local.data();
    KzgRet rc = KZG_OK;
    if (!s->multi) {
        std::vector<size_t> idx(K);
        for (size_t g = 0; g < K; g++) idx[g] = g;
        rc = groups_pipeline_locked(ok_out, err, d_blobs, d_commitments, d_proofs, idx.data(), K, n, B, in_flight, s);
    } else {
        // several devices: every group to the device that owns its memory, one pipeline and one host thread per device
        const size_t D = shard_count(s);
        std::vector<std::vector<size_t>> mine(D);
        std::vector<size_t> turn;
        for (size_t g = 0; g < K; g++) {
            if (!d_blobs[g] || !d_commitments[g] || !d_proofs[g]) return fail(KZG_BADARGS, "null group");
            if ((rc = multi_same_device(d_blobs[g], d_commitments[g], d_proofs[g])) != KZG_OK) return rc;
            size_t k = 0;
            if ((rc = multi_owner_shard(&k, d_blobs[g], s, turn)) != KZG_OK) return rc;
            mine[k].push_back(g);
        }
        std::vector<KzgRet> rcs(D, KZG_OK);
        std::vector<std::string> msgs(D);
        auto run = [&](size_t k) {
            const KzgSettings* c = shard_of(s, k);
            if (hipSetDevice(c->device) != hipSuccess) {
                (void)hipGetLastError();
                rcs[k] = KZG_ERROR;
                msgs[k] = "HIP: hipSetDevice";
                return;
            }
            try {
                rcs[k] = groups_pipeline_locked(ok_out, err, d_blobs, d_commitments, d_proofs, mine[k].data(), mine[k].size(), n, B, in_flight, c);
                if (rcs[k] != KZG_OK) msgs[k] = g_err;
            } catch (const std::bad_alloc&) {
                rcs[k] = KZG_MALLOC;
                msgs[k] = "host buffers of the pipeline";
            }
        };
        {
            std::vector<std::thread> pool;
            size_t inline_k = D;  // the calling thread takes the first busy shard
            for (size_t k = 0; k < D; k++) {
                if (mine[k].empty()) continue;
                if (inline_k == D) {
                    inline_k = k;
                    continue;
                }
                try {
                    pool.emplace_back(run, k);
                } catch (const std::system_error&) {
                    run(k);
                }
            }
            if (inline_k < D) run(inline_k);
            for (auto& th : pool) th.join();
        }
        HIPCHK(hipSetDevice(s->device));
        for (size_t k = 0; k < D; k++)
            if (rcs[k] != KZG_OK) return fail(rcs[k], msgs[k]);
    }
    if (rc != KZG_OK) return rc;
    if (!err_out)  // without an error array an invalid input anywhere fails the call, like the one-group form
        for (size_t i = 0; i < K * B; i++)
            if (err[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    return KZG_OK;
} catch (const std::bad_alloc&) {
    return fail(KZG_MALLOC, "host buffers of the pipeline");  // (nothing is thrown across the C ABI)
}

// test hook: the transcript records (160 B each: C || z LE || y LE || pi) of the last launch group that ran on lane `lane` of
// the handle (0 = the handle itself), entries [first, first + count) of the group's blobs in order - what lets a test compare
// the (z, y) the in-library pipeline computed with the oracle's (tests/test_gpu_baseline_sizes.py)
extern "C" KzgRet kzg_debug_lane_records(uint8_t* out, size_t lane, size_t first, size_t count, const KzgSettings* s) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (lane > s->lanes.size()) return fail(KZG_BADARGS, "no such lane");
    const KzgSettings* l = lane_of(s, lane);
    if (!l->ws.h_buf || first + count > l->ws.cap_n) return fail(KZG_BADARGS, "records out of range");
    memcpy(out, l->ws.h_buf + 160 * first, 160 * count);
    return KZG_OK;
}

// test hook (no GPU needed): the value KZG_OPTIONS gives `key` right now, as the library's parser reads it; -1 when unset,
// else the length of the value (copied, truncated, NUL-terminated into out[cap]); *ab_build = 1 in the A/B build
extern "C" int kzg_debug_option(const char* key, char* out, size_t cap, int* ab_build) {
    if (ab_build) *ab_build = KZG_AB_VARIANTS;
    const char* v = key ? opt_str(key) : nullptr;
    if (!v) return -1;
    if (out && cap) {
        strncpy(out, v, cap - 1);
        out[cap - 1] = 0;
    }
    return (int)strlen(v);
}

// test hook (no GPU needed): compute_challenge (src/kzg_proof.rs:46-72) as the HOST computes it for small host batches
// (host_only.hpp host_blob_challenge); z as 32 big-endian bytes
extern "C" void kzg_debug_host_blob_challenge(uint8_t z_be[32], const uint8_t* blob, const uint8_t* commitment48) {
    uint8_t le[32];
    host_blob_challenge(le, blob, commitment48);
    reverse32(z_be, le);
}
