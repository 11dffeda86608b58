// capi_coalesce.hpp - ONE shared settings handle serves MANY host threads: the small calls of concurrent callers are coalesced into
// launches on pooled private lanes.  Part of the single translation unit kzg_capi.hip; not a stand-alone header.  Host code only.
//
// The reference's KzgSettings is three &'static slices, shared freely between threads (src/trusted_setup.rs:44-50; the static
// EnvKzgSettings::Default of :80-92), and its named caller is the revm point-evaluation precompile: many threads, one settings
// value, one small call each (verify_kzg_proof, src/kzg_proof.rs:353-397; a beacon node's verify_blob_kzg_proof[_batch] of the 6-9
// blobs of a block is the other).  On the GPU one such call is a latency-bound chain of ~1.7 ms that uses one CU per proof;
// rounds 1-4 ran them one at a time under the handle's mutex, so T threads got ~550 calls/s between them - while the same
// library verifies 130 k independent proofs/s when ONE caller hands them over together (kzg_verify_kzg_proofs).  This file
// makes the second rate available to the first kind of caller:
//
//   * every small call becomes a request in the handle's queue (SmallReq, capi_settings.hpp);
//   * a thread that finds a free LANE (a private handle on the device: own streams, workspace, pinned mirror; the handle's
//     tables) becomes a LEADER: it takes everything queued of the oldest request's kind - up to 1 024 proof tuples or 256
//     blobs - runs ONE launch of the independent-proofs path on that lane (proofs_independent_locked / blobs_parts_locked,
//     capi_verify.hpp: one pairing per item, side by side on CUs of their own), hands every request its own results, and wakes
//     its owners; an idle handle adds no wait - the first caller leads a launch of one, which is the round-4 single-call path;
//   * threads that find every lane busy wait - their requests accumulate and leave with the next free lane; up to max_lanes
//     (option small_lanes, default 2 per device) launches are in flight, on every device of a multi-device handle in turn.
//     Few lanes on purpose: a launch costs its 1.7-2 ms whether it carries 1 proof or 200 (one CU each), so what counts is that
//     the waiting calls travel together - measured at 256 threads: 99 k calls/s with 2 lanes, 80 k with 3, 64 k with 4, 55 k with 6;
//   * right after a launch that carried several calls, a new leader gives their callers a moment to come back (while requests
//     keep arriving, 250 us at most: small_submit) - a closed loop of T callers then travels as ONE launch of T, and its rate
//     is T / (the launch's 1.7-2 ms + the turn-around) instead of a stream of fragments that each wait for a lane;
//   * a waiting blob caller hashes its own blobs meanwhile (host_only.hpp hostpool), so the host-side SHA-256 of T callers runs
//     on T cores and the leader only collects the challenges.
// Results are per request: a wrong proof, a non-canonical scalar, an undecodable or off-subgroup point in one caller's input
// never changes another caller's answer (the instances share a launch, not a random linear combination).
// z = tau (the pairing's G2 point is the identity: only for who knows the setup's secret, i.e. test rigs) is flagged per item;
// its owner decides it through the general path under the handle's own lock, as before.

constexpr size_t SMALL_MAX_TUPLES = PROOFS_CHUNK;  // proof tuples per launch (SmallQueue::cap_proofs)
constexpr size_t SMALL_MAX_BLOBS = 256;            // blobs per launch (= the range of the one-pairing-per-blob form; SmallQueue::cap_blobs)

static_assert(SMALL_MAX_TUPLES == 1024 && SMALL_MAX_BLOBS == 256, "SmallQueue::cap_proofs / cap_blobs (small_queue.hpp) are these");
static void small_free(KzgSettings* s) {
    if (!s->small) return;
    for (size_t i = 0; i < s->small->n_lanes; i++) {
        SmallLane* l = s->small->lanes[i];
        if (l && l->h) {
            (void)hipSetDevice(l->h->device);
            kzg_settings_free(l->h);
        }
        delete l;
    }
    delete s->small;
    s->small = nullptr;
}

// the launch of one leader: `batch` = requests of one kind with `m` items in all, on lane L (not shared with anybody meanwhile)
static KzgRet small_run_proofs(SmallLane& L, std::vector<SmallReq*>& batch, size_t m) {
    const KzgSettings* l = L.h;
    if (m == 1) {  // nobody else was waiting: the one-proof path as it was (z, y through the pinned mirror, no gather)
        SmallReq& r = *batch[0];
        bool general = false, ok = false;
        if (be_geq_r(r.z) || be_geq_r(r.y)) {  // (:360-371; kzg_verify_kzg_proof has refused these before they are queued, kzg_verify_kzg_proofs has not)
            r.err[0] = 1;
            r.ok[0] = false;
            r.general[0] = 0;
            return KZG_OK;
        }
        const KzgRet rc = proof_single_locked(&ok, &general, r.c, r.z, r.y, r.p, l);
        if (rc == KZG_BADARGS) {
            r.err[0] = 1;
            r.ok[0] = false;
            r.general[0] = 0;
            return KZG_OK;
        }
        if (rc != KZG_OK) return rc;
        r.err[0] = 0;
        r.general[0] = general;
        r.ok[0] = !general && ok;
        return KZG_OK;
    }
    L.c.resize(48 * m);
    L.p.resize(48 * m);
    L.z.resize(32 * m);
    L.y.resize(32 * m);
    L.okerr.resize(2 * m);
    size_t off = 0;
    for (SmallReq* r : batch) {
        memcpy(L.c.data() + 48 * off, r->c, 48 * r->n);
        memcpy(L.p.data() + 48 * off, r->p, 48 * r->n);
        memcpy(L.z.data() + 32 * off, r->z, 32 * r->n);
        memcpy(L.y.data() + 32 * off, r->y, 32 * r->n);
        off += r->n;
    }
    bool* const ok = reinterpret_cast<bool*>(L.okerr.data());
    uint8_t* const err = L.okerr.data() + m;
    std::vector<size_t> general;
    const KzgRet rc = proofs_independent_locked(ok, err, L.c.data(), L.z.data(), L.y.data(), L.p.data(), m, l, general);
    if (rc != KZG_OK) return rc;
    off = 0;
    for (SmallReq* r : batch) {
        for (size_t i = 0; i < r->n; i++) {
            r->ok[i] = ok[off + i];
            r->err[i] = err[off + i];
            r->general[i] = 0;
        }
        off += r->n;
    }
    for (size_t g : general) {  // (indices into the launch, ascending)
        off = 0;
        for (SmallReq* r : batch) {
            if (g < off + r->n) {
                r->general[g - off] = 1;
                break;
            }
            off += r->n;
        }
    }
    return KZG_OK;
}

static KzgRet small_run_blobs(SmallLane& L, std::vector<SmallReq*>& batch, size_t m) {
    const KzgSettings* l = L.h;
    if (m == 1) {  // the one-blob path as it was: the hash runs while the blob crosses PCIe
        SmallReq& r = *batch[0];
        bool general = false, ok = false;
        const KzgRet rc = blob_single_locked(&ok, &general, r.blobs, r.c, r.p, l, r.hash.get());
        if (rc == KZG_BADARGS) {
            r.err[0] = 1;
            r.ok[0] = false;
            r.general[0] = 0;
            return KZG_OK;
        }
        if (rc != KZG_OK) return rc;
        r.err[0] = 0;
        r.general[0] = general;
        r.ok[0] = !general && ok;
        return KZG_OK;
    }
    std::vector<BlobsPart> parts(batch.size());
    for (size_t k = 0; k < batch.size(); k++) parts[k] = BlobsPart{batch[k]->blobs, batch[k]->c, batch[k]->p, batch[k]->n, batch[k]->hash.get(), false, false, false};
    const KzgRet rc = blobs_parts_locked(parts.data(), parts.size(), m, l);
    if (rc != KZG_OK) return rc;
    for (size_t k = 0; k < batch.size(); k++) {
        batch[k]->err[0] = parts[k].bad;
        batch[k]->general[0] = !parts[k].bad && parts[k].general;
        batch[k]->ok[0] = !parts[k].bad && !parts[k].general && parts[k].ok;
    }
    return KZG_OK;
}

// a new lane of the queue: sized once for the largest launch (1 024 tuples, 256 blobs: ~60 MB of device memory) - a workspace
// that grows with the launches would free and allocate device memory in the middle of the traffic, and hipFree waits for the
// whole device (measured: 100-200 ms stalls of every caller while the lanes grew)
static KzgRet small_lane_make(SmallLane& L, const SmallQueue& Q, const KzgSettings* home) {
    KzgRet rc = settings_lane(&L.h, home, Q.lane_priority);
    if (rc != KZG_OK) return rc;
    L.h->proof_two_streams = Q.lane_two_streams;
    ProofStreams ps{};
    ProofsLaunch pl{};
    if ((rc = proofs_reserve(ps, pl, SMALL_MAX_TUPLES, STAGE_CP, L.h)) != KZG_OK) return rc;
    if ((rc = ws_reserve(L.h, SMALL_MAX_TUPLES, 1, STAGE_NONE)) != KZG_OK) return rc;
    return ws_reserve(L.h, SMALL_MAX_BLOBS, 1, STAGE_BLOBS);
}

// Submit a request and return when it is done (small_queue.hpp small_submit_core); the launch a leader runs on its lane:
static KzgRet small_submit(const KzgSettings* s, SmallReq& r) {
    SmallQueue& Q = *s->small;
    auto run = [&](int li, SmallLane& L, std::vector<SmallReq*>& batch, size_t m, SmallReq::Kind kind, std::string& msg) -> KzgRet {
        const size_t shard = (size_t)li % shard_count(s);  // the lanes of a multi-device handle are dealt to its devices in turn
        KzgRet rc = KZG_OK;
        const KzgSettings* const home = shard_of(s, shard);
        if (hipSetDevice(home->device) != hipSuccess) {
            (void)hipGetLastError();
            rc = KZG_ERROR;
            msg = "HIP: hipSetDevice";
        }
        if (rc == KZG_OK && !L.h) {
            try {
                rc = small_lane_make(L, Q, home);
                if (rc != KZG_OK) msg = g_err;
            } catch (const std::bad_alloc&) {
                rc = KZG_MALLOC;
                msg = "host memory of a lane";
            }
            if (rc != KZG_OK) {
                if (L.h) kzg_settings_free(L.h);
                L.h = nullptr;
            }
        }
        if (rc == KZG_OK) {
            try {
                rc = kind == SmallReq::PROOFS ? small_run_proofs(L, batch, m) : small_run_blobs(L, batch, m);
                if (rc != KZG_OK) msg = g_err;
            } catch (const std::bad_alloc&) {
                rc = KZG_MALLOC;
                msg = "host buffers of the launch";
            }
            if (rc != KZG_OK) proof_drain(L.h);  // nothing of the launch stays in flight behind an error
            else {  // kzg_last_timings: the last launch's intervals - under the handle's lock like every other writer and the reader;
                    // a diagnostic: when a large call holds the lock (or the other lane's leader is writing) this launch's are skipped
                std::unique_lock<std::mutex> tl(s->mu, std::try_to_lock);
                if (tl.owns_lock()) memcpy(s->timings, L.h->timings, sizeof s->timings);
            }
        }
        if (shard != 0) (void)hipSetDevice(s->device);
        // no pool worker may still read a caller's blobs once its call has returned (an error path may not have come by the join)
        for (SmallReq* x : batch)
            if (x->hash) hostpool::finish(*x->hash);
        return rc;
    };
    const KzgRet rc = small_submit_core(Q, r, run);
    if (rc != KZG_OK) g_err = r.msg;  // (thread-local: the message of the launch that carried this caller's request)
    return rc;
}

static bool small_enabled(const KzgSettings* s) {
    static const bool msm_path = opt_is("proof_path", "msm");  // (round 3's path for single proofs: A/B, cross-check)
    return s->small && !msm_path;
}

// ---- the entry points' small branches -------------------------------------------------------------------------------------
// n tuples with a result each (kzg_verify_kzg_proof: n = 1; kzg_verify_kzg_proofs; kzg_verify_kzg_proof_batch's conjunction form)
static KzgRet small_proofs(bool* ok, uint8_t* err, uint8_t* general, const uint8_t* c, const uint8_t* z, const uint8_t* y, const uint8_t* p, size_t n,
                           const KzgSettings* s) {
    SmallReq r;
    r.kind = SmallReq::PROOFS;
    r.n = n;
    r.c = c;
    r.z = z;
    r.y = y;
    r.p = p;
    r.ok = ok;
    r.err = err;
    r.general = general;
    return small_submit(s, r);
}
// n host blobs of ONE call: *ok = the conjunction of their verdicts, *err = the reference's Err(BadArgs), *general = some z_i = tau
static KzgRet small_blobs(bool* ok, uint8_t* err, uint8_t* general, const uint8_t* blobs, const uint8_t* c, const uint8_t* p, size_t n, const KzgSettings* s) {
    SmallReq r;
    r.kind = SmallReq::BLOBS;
    r.n = n;
    r.blobs = blobs;
    r.c = c;
    r.p = p;
    r.ok = ok;
    r.err = err;
    r.general = general;
    std::vector<uint8_t> z_le(32 * n);
    r.hash = hostpool::make(z_le.data(), blobs, c, n);
    static const size_t threads = (size_t)std::max(1L, std::min(16L, opt_int("host_threads", 16)));
    hostpool::post(r.hash, threads);  // (n >= 2: the pool's workers start now; whoever needs the challenges finishes them)
    const KzgRet rc = small_submit(s, r);
    hostpool::finish(*r.hash);  // (z_le dies with this frame)
    return rc;
}

// diagnostic: launches | requests | items | the largest launch of the handle's small-call queue since the last reset, and its lanes
extern "C" KzgRet kzg_debug_small_queue_stats(const KzgSettings* s, uint64_t out[5], int reset) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    memset(out, 0, 5 * sizeof(uint64_t));
    if (!s->small) return KZG_OK;
    std::lock_guard<SmallSpinLock> lk(s->small->mu);
    out[0] = s->small->launches;
    out[1] = s->small->requests;
    out[2] = s->small->items;
    out[3] = s->small->max_items;
    out[4] = s->small->n_lanes;
    if (reset) s->small->launches = s->small->requests = s->small->items = s->small->max_items = 0;
    return KZG_OK;
}

// measurement + test hook: T host threads (plain std::threads inside the library: no interpreter lock, no ctypes) calling the
// PUBLIC small entry points on ONE shared handle for `seconds`, each checking every answer it gets.
//   kind 0: kzg_verify_kzg_proof(c[i], z[i], y[i], p[i])                               expect[i]: 0 false | 1 true | 2 Err(BadArgs)
//   kind 1: kzg_verify_blob_kzg_proof_batch(blobs / c / p [i per_call, (i + 1) per_call))   expect[i] for call i, i < n_items / per_call
//   kind 2: kzg_verify_kzg_proofs over tuples [i per_call, (i + 1) per_call) with err_out   expect[j] per tuple
// thread t takes calls t, t + T, ... (mod the number of distinct calls).  out: [0] calls completed, [1] elapsed seconds,
// [2] answers that differ from `expect` (or calls that failed with another code), [3] mean latency of a call in ms, [4] the longest one.
extern "C" KzgRet kzg_debug_concurrent_callers(double out[5], int kind, size_t threads, double seconds, const uint8_t* blobs, const uint8_t* c,
                                               const uint8_t* z, const uint8_t* y, const uint8_t* p, const uint8_t* expect, size_t n_items, size_t per_call,
                                               const KzgSettings* s) try {
    if (!out || !s || !c || !p || !expect || !threads || !n_items || kind < 0 || kind > 2) return fail(KZG_BADARGS, "bad argument");
    if (kind == 0) per_call = 1;
    if (!per_call || n_items < per_call || (kind == 1 && !blobs) || (kind != 1 && (!z || !y))) return fail(KZG_BADARGS, "bad argument");
    const size_t n_calls = n_items / per_call;
    std::atomic<uint64_t> calls{0}, wrong{0};
    std::atomic<bool> stop{false};
    // the threads wait for the start ASLEEP: 256 threads yielding in a loop while the rest are still being made use up the
    // cgroup's CPU quota of the period (16 cores on this project's boxes) and the measurement would begin throttled
    std::mutex go_mu;
    std::condition_variable go_cv;
    bool go = false;
    std::vector<double> lat_sum(threads, 0.0), lat_max(threads, 0.0);
    auto body = [&](size_t t) {
        {
            std::unique_lock<std::mutex> lk(go_mu);
            go_cv.wait(lk, [&] { return go; });
        }
        // independent callers do not arrive in lock-step: each thread's first call starts at its own moment within one launch
        // time (a fixed pseudo-random offset below 2.5 ms).  Released all at once, the T callers travel as ONE cohort for the
        // whole run - one launch of ~T in flight, the second lane idle - which is the closed loop's artefact, not the queue's
        // behaviour under arrivals spread over time (measured at T = 256: 86-90 k calls/s in lock-step, 94-107 k spread).
        std::this_thread::sleep_for(std::chrono::microseconds((uint32_t)(t * 2654435761u) % 2500u));
        std::vector<uint8_t> oks(per_call), errs(per_call);
        for (size_t i = t % n_calls; !stop.load(std::memory_order_relaxed); i = (i + threads) % n_calls) {
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t bad = 0;
            if (kind == 0) {
                bool ok = false;
                const KzgRet rc = kzg_verify_kzg_proof(&ok, c + 48 * i, z + 32 * i, y + 32 * i, p + 48 * i, s);
                const int got = rc == KZG_BADARGS ? 2 : rc == KZG_OK ? (ok ? 1 : 0) : 3;
                bad = got != expect[i];
            } else if (kind == 1) {
                bool ok = false;
                const size_t f = i * per_call;
                const KzgRet rc = kzg_verify_blob_kzg_proof_batch(&ok, blobs + (size_t)BLOB_BYTES * f, c + 48 * f, p + 48 * f, per_call, s);
                const int got = rc == KZG_BADARGS ? 2 : rc == KZG_OK ? (ok ? 1 : 0) : 3;
                bad = got != expect[i];
            } else {
                const size_t f = i * per_call;
                const KzgRet rc = kzg_verify_kzg_proofs(reinterpret_cast<bool*>(oks.data()), errs.data(), c + 48 * f, z + 32 * f, y + 32 * f, p + 48 * f, per_call, s);
                if (rc != KZG_OK) bad = 1;
                else
                    for (size_t k = 0; k < per_call; k++) bad += (errs[k] ? 2 : oks[k] ? 1 : 0) != expect[f + k];
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            lat_sum[t] += ms;
            lat_max[t] = std::max(lat_max[t], ms);
            if (bad) wrong.fetch_add(bad, std::memory_order_relaxed);
            calls.fetch_add(1, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(threads);
        for (size_t t = 0; t < threads; t++) pool.emplace_back(body, t);
    } catch (...) {  // (no more threads to be had: the ones made leave at once - a joinable thread must not meet its destructor)
        stop.store(true);
        {
            std::lock_guard<std::mutex> lk(go_mu);
            go = true;
        }
        go_cv.notify_all();
        for (auto& th : pool) th.join();
        return fail(KZG_ERROR, "kzg_debug_concurrent_callers: could not start the threads");
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));  // (every thread has reached its wait)
    {
        std::lock_guard<std::mutex> lk(go_mu);
        go = true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    go_cv.notify_all();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t counted = calls.load();  // (calls completed inside the interval; the ones in flight at its end are not counted)
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto& th : pool) th.join();
    double ls = 0, lm = 0;
    for (size_t t = 0; t < threads; t++) {
        ls += lat_sum[t];
        lm = std::max(lm, lat_max[t]);
    }
    out[0] = (double)counted;
    out[1] = elapsed;
    out[2] = (double)wrong.load();
    out[3] = calls.load() ? ls / (double)calls.load() : 0.0;
    out[4] = lm;
    return KZG_OK;
} catch (const std::exception& e) {
    return fail(KZG_ERROR, std::string("kzg_debug_concurrent_callers: ") + e.what());
}
