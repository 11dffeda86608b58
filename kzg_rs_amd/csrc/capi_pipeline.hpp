// capi_pipeline.hpp - many independent batches through ONE call at the chip's rate: launch groups kept in flight inside the library.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.  Host code only.
//
// At n = 1 024 every phase of a batch is a latency-bound serial chain, so batches share kernel launches (a launch GROUP of B
// batches) and several groups overlap.  Rounds 1-2 kept the groups in flight from Python
// (kzg_rs_amd/distributed.py PipelinedVerifier) - not an entry point a caller of the reference has.  Here the same fixed-order
// software pipeline runs behind one C call: the handle grows private "lanes" (complete handles on the same device: their own
// streams and workspace), group t runs on lane t mod (F + 1), and iteration t of the host loop does, in this order,
//     phase1_launch(t);   phase1_wait(t - d1) + the group's transcript hashes + phase2_launch + finish_launch;   finish_wait(t - d1 - 1)
// with d1 = F - 1 groups between the first two steps (F = groups in flight; F = 1: everything in sequence on the handle itself).
// Every batch of every group is a complete, independent verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-525): its own
// transcript, challenge r, MSMs, pairing, boolean.

static KzgRet pipeline_lanes(const KzgSettings* s, size_t count) {  // at least `count` lanes beside the handle itself
    while (s->lanes.size() < count) {
        KzgSettings* l = nullptr;
        KzgRet rc = settings_common(&l, s->tau_g2_bytes);  // (the caller has set the device)
        if (rc != KZG_OK) return rc;
        s->lanes.push_back(l);
    }
    return KZG_OK;
}

// groups: n_groups launch groups of batches_per_group batches of n blobs each; group g at d_blobs[g] / d_commitments[g] /
// d_proofs[g] (device memory, batches contiguous) - the same pointers may repeat.  ok_out / err_out: [n_groups][batches_per_group].
extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_groups_device(bool* ok_out, uint8_t* err_out, const void* const* d_blobs,
                                                                const void* const* d_commitments, const void* const* d_proofs, size_t n,
                                                                size_t batches_per_group, size_t n_groups, size_t in_flight, const KzgSettings* s) {
    KZG_ENTER(s && ok_out && d_blobs && d_commitments && d_proofs && n && batches_per_group);
    if (n_groups == 0) return KZG_OK;
    const size_t B = batches_per_group, K = n_groups;
    const size_t F = std::max<size_t>(1, std::min<size_t>(in_flight ? in_flight : 3, 8));
    const size_t d1 = F - 1, d3 = F > 1 ? 1 : 0, S = d1 + d3 + 1;  // S handles: the handle itself + S - 1 lanes
    KzgRet rc = pipeline_lanes(s, S - 1);
    if (rc != KZG_OK) return rc;
    auto lane = [&](size_t i) -> const KzgSettings* { return i % S == 0 ? s : s->lanes[i % S - 1]; };
    for (size_t i = 0; i < std::min(S, K); i++)
        if ((rc = ws_reserve(lane(i), n * B, B, STAGE_NONE)) != KZG_OK) return rc;
    std::vector<uint8_t> err_local;
    try {
        if (!err_out) err_local.resize(K * B);
    } catch (const std::bad_alloc&) {
        return fail(KZG_MALLOC, "per-batch error flags");
    }
    uint8_t* const err = err_out ? err_out : err_local.data();
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
            if (!d_blobs[t] || !d_commitments[t] || !d_proofs[t]) return drained(fail(KZG_BADARGS, "null group"));
            if ((rc = phase1_launch_locked(d_blobs[t], d_commitments[t], d_proofs[t], n, B, lane(t))) != KZG_OK) return drained(rc);
        }
        if (t >= d1 && t - d1 < K) {
            const size_t i = t - d1;
            const KzgSettings* h = lane(i);
            if ((rc = phase1_wait_locked(nullptr, err + i * B, h)) != KZG_OK) return drained(rc);
            if ((rc = phase2_launch_locked(nullptr, n, 0, h, 0, nullptr, false)) != KZG_OK) return drained(rc);
            if ((rc = finish_launch_locked(nullptr, 1, B, h)) != KZG_OK) return drained(rc);
        }
        if (t >= d1 + d3 && t - d1 - d3 < K) {
            const size_t k = t - d1 - d3;
            if ((rc = finish_wait_locked(ok_out + k * B, lane(k))) != KZG_OK) return drained(rc);
            for (size_t b = 0; b < B; b++)
                if (err[k * B + b]) ok_out[k * B + b] = false;
        }
    }
    if (!err_out)  // without an error array an invalid input anywhere fails the call, like the one-group form
        for (size_t i = 0; i < K * B; i++)
            if (err[i]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    return KZG_OK;
}
