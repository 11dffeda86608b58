// capi_multi.hpp - multi-GPU behind the reference's own signature: ONE process, one settings handle over a device list.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.  Host code only.
//
// KzgProof::verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-477) takes one Vec<Blob> and returns one bool; a caller of it
// is one process.  A handle made over D devices (kzg_settings_*_devices, or KZG_DEVICES in the environment of an unchanged
// caller) shards the batch by blob in contiguous index ranges - the per-blob loop of :261-273 is the data-parallel axis -
// and runs the three phases of capi_verify.hpp on every device from this one process:
//   phase 1 per device (its slice crosses its OWN PCIe link on a thread of its own: 8 links instead of 1)
//   -> the 160-byte transcript records come back (pinned mirrors), ONE host hash over all of them gives r (:291-334)
//   -> phase 2 per device with r and its power offset r^offset (:279-289)
//   -> the "G1 all-reduce" of the north star: ncclAllGather of the 288-byte partial sums (A_k, B_k) over the in-process RCCL
//      communicators (xGMI), every device receives all D of them; point addition is not an RCCL reduction op, so the
//      reduction itself is k_fold_partials on the first device
//   -> one pairing there.
// Exchange fallback: host staging (the partials are 288 B per device and already have pinned mirrors) when librccl is not
// loadable, when the list names a device twice (a test rig: ncclCommInitAll refuses duplicates) or KZG_MULTI_EXCHANGE=host.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>

struct RcclApi {
    void* dl = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;  // why it is unusable
};

// librccl is bound at run time, on the first multi-device handle: a single-device caller never loads its 570 MB, and a
// process that already holds one (PyTorch brings its own librccl.so.1) shares that copy - two RCCLs on one HIP runtime
// would each keep their own view of the devices.
static RcclApi* rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* nm : names)
            if ((a.dl = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!a.dl)
            for (const char* nm : names)
                if ((a.dl = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!a.dl) {
            const char* e = dlerror();
            a.why = std::string("librccl not loadable: ") + (e ? e : "?");
            return a;
        }
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(a.dl, "ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.dl, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.dl, "ncclAllGather"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.dl, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.dl, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.dl, "ncclGetErrorString"));
        if (!a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
            a.why = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGroupStart / ncclGroupEnd";
            a.CommInitAll = nullptr;
        }
        return a;
    }();
    return &api;
}

enum { MULTI_EXCHANGE_NONE = 0, MULTI_EXCHANGE_HOST = 1, MULTI_EXCHANGE_RCCL = 2 };
struct MultiState {
    std::vector<int> devices;        // shard k runs on devices[k]; shard 0 is the handle itself
    std::vector<ncclComm_t> comms;   // one communicator per shard (exchange == RCCL)
    int exchange = MULTI_EXCHANGE_HOST;
    std::string exchange_note;       // why host staging was chosen, if it was
    size_t min_blobs = 256;          // host / primary-resident batches below this stay on shard 0 (KZG_MULTI_MIN_BLOBS)
};

static const KzgSettings* shard_of(const KzgSettings* s, size_t k) { return k == 0 ? s : s->peers[k - 1]; }
static size_t shard_count(const KzgSettings* s) { return s->multi ? s->multi->devices.size() : 1; }

#define NCCLCHK(api, expr)                                                                                     \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(KZG_ERROR, std::string("RCCL: ") + (api)->GetErrorString(r_) + " at " #expr); \
    } while (0)

// all-gather of the B x 288-byte partial sets of every shard into every shard's ws.d_parts ([world][B][2] G1Jac), each on
// its own stream s1: the data-path collective of the multi-GPU verification (SURVEY 8e "exchange 2")
static KzgRet multi_allgather_partials(const KzgSettings* s, size_t B) {
    MultiState* m = s->multi;
    RcclApi* api = rccl_api();
    const size_t D = m->devices.size();
    NCCLCHK(api, api->GroupStart());
    for (size_t k = 0; k < D; k++) {
        const KzgSettings* c = shard_of(s, k);
        HIPCHK(hipSetDevice(c->device));
        ncclResult_t r = api->AllGather(c->ws.d_ab, c->ws.d_parts, 288 * B, ncclUint8, m->comms[k], c->s1);
        if (r != ncclSuccess) {
            (void)api->GroupEnd();
            return fail(KZG_ERROR, std::string("RCCL: ") + api->GetErrorString(r) + " at ncclAllGather");
        }
    }
    NCCLCHK(api, api->GroupEnd());
    return KZG_OK;
}

static void multi_free(KzgSettings* s) {
    if (s->multi) {
        RcclApi* api = rccl_api();
        for (size_t k = 0; k < s->multi->comms.size(); k++)
            if (s->multi->comms[k] && api->CommDestroy) {
                (void)hipSetDevice(s->multi->devices[k]);
                (void)api->CommDestroy(s->multi->comms[k]);
            }
        delete s->multi;
        s->multi = nullptr;
    }
    for (KzgSettings* p : s->peers) {
        (void)hipSetDevice(p->device);
        kzg_settings_free(p);
    }
    s->peers.clear();
}

// Peers of a freshly built shard 0 (s->device == devices[0]) and the exchange.  On failure the caller frees the handle.
static KzgRet multi_build(KzgSettings* s, const uint8_t tau_g2[96], const std::vector<int>& devices) {
    static const bool force = getenv("KZG_MULTI_FORCE") && getenv("KZG_MULTI_FORCE")[0] == '1';  // test rig: a list of ONE device still takes the sharded path
    if (devices.size() < 2 && !force) return KZG_OK;  // a plain single-device handle
    MultiState* m = new MultiState();
    s->multi = m;
    m->devices = devices;
    if (const char* e = getenv("KZG_MULTI_MIN_BLOBS")) m->min_blobs = (size_t)std::max(2L, atol(e));
    const size_t D = devices.size();
    KzgRet rc;
    for (size_t k = 1; k < D; k++) {
        HIPCHK(hipSetDevice(devices[k]));
        KzgSettings* p = nullptr;
        if ((rc = settings_common(&p, tau_g2)) != KZG_OK) return rc;
        s->peers.push_back(p);
    }
    // every shard owns the group-sized buffers from the start (the warm-up collective below uses them)
    for (size_t k = 0; k < D; k++) {
        const KzgSettings* c = shard_of(s, k);
        HIPCHK(hipSetDevice(c->device));
        if ((rc = ws_reserve(c, 16, 1, STAGE_NONE)) != KZG_OK) return rc;
    }
    // exchange: RCCL when the devices are distinct
    const char* ex = getenv("KZG_MULTI_EXCHANGE");
    bool distinct = true;
    for (size_t a = 0; a < D; a++)
        for (size_t b = a + 1; b < D; b++) distinct &= devices[a] != devices[b];
    m->exchange = MULTI_EXCHANGE_HOST;
    if (ex && strcmp(ex, "host") == 0) m->exchange_note = "KZG_MULTI_EXCHANGE=host";
    else if (!distinct) m->exchange_note = "the device list names a device twice (ncclCommInitAll needs distinct devices)";
    else {
        RcclApi* api = rccl_api();
        if (!api->CommInitAll) m->exchange_note = api->why;
        else {
            m->comms.assign(D, nullptr);
            ncclResult_t r = api->CommInitAll(m->comms.data(), (int)D, devices.data());
            if (r != ncclSuccess) {
                m->exchange_note = std::string("ncclCommInitAll: ") + api->GetErrorString(r);
                m->comms.clear();
                (void)hipGetLastError();
            } else {
                m->exchange = MULTI_EXCHANGE_RCCL;
                // warm-up: the first collective of a communicator sets up its channels (hundreds of ms) - not inside a verification
                for (size_t k = 0; k < D; k++) {
                    const KzgSettings* c = shard_of(s, k);
                    HIPCHK(hipSetDevice(c->device));
                    HIPCHK(hipMemsetAsync(c->ws.d_ab, 0, 288, c->s1));
                }
                if ((rc = multi_allgather_partials(s, 1)) != KZG_OK) return rc;
                for (size_t k = 0; k < D; k++) {
                    const KzgSettings* c = shard_of(s, k);
                    HIPCHK(hipSetDevice(c->device));
                    HIPCHK(hipStreamSynchronize(c->s1));
                }
            }
        }
        if (ex && strcmp(ex, "rccl") == 0 && m->exchange != MULTI_EXCHANGE_RCCL)
            return fail(KZG_ERROR, "KZG_MULTI_EXCHANGE=rccl but RCCL is unusable: " + m->exchange_note);
    }
    HIPCHK(hipSetDevice(s->device));
    return KZG_OK;
}

// where the inputs of a sharded call lie
enum class MultiSrc { Host, Primary, PerDevice };
struct ShardIn {
    const uint8_t *blobs = nullptr, *c = nullptr, *p = nullptr;
    size_t n = 0;
};

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ONE batch of n_total = sum in[k].n blobs over the shards of the handle (in[k] = shard k's contiguous slice, global blob
// order = shard order).  The caller holds s->mu; peers are private to the handle, so nothing else can touch them.
static KzgRet multi_batch_locked(bool* ok, const std::vector<ShardIn>& in, MultiSrc kind, const KzgSettings* s) {
    MultiState* m = s->multi;
    const size_t D = in.size();
    const auto t_call = std::chrono::steady_clock::now();
    size_t n_total = 0, active = 0;
    std::vector<size_t> off(D, 0);
    for (size_t k = 0; k < D; k++) {
        off[k] = n_total;
        n_total += in[k].n;
        active += in[k].n != 0;
    }
    std::vector<uint8_t> records(160 * n_total);
    std::vector<KzgRet> rcs(D, KZG_OK);
    std::vector<std::string> msgs(D);
    std::vector<uint8_t> bad(D, 0);
    // ---- stage A, one host thread per shard: inputs onto the shard's device, phase 1, records back
    auto stage_a = [&](size_t k) {
        const KzgSettings* c = shard_of(s, k);
        const size_t nk = in[k].n;
        auto body = [&]() -> KzgRet {
            HIPCHK(hipSetDevice(c->device));
            const bool copy = kind == MultiSrc::Host || (kind == MultiSrc::Primary && c->device != s->device);
            KzgRet rc = ws_reserve(c, nk ? nk : 16, 1, copy && nk ? STAGE_BLOBS : STAGE_NONE);
            if (rc != KZG_OK || nk == 0) return rc;
            Workspace& w = c->ws;
            select_streams(c, nk);
            const void *db = in[k].blobs, *dc = in[k].c, *dp = in[k].p;
            const HostBatch hb{in[k].blobs, in[k].c, in[k].p};
            if (copy) {
                if (kind == MultiSrc::Host) {
                    // phase 1 brings the shard over itself, in slices across its blobs (capi_verify.hpp host_slices)
                } else {  // resident on the first device: the slice crosses xGMI (a caller that can should hand over per-device shards)
                    HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp, c->device, in[k].c, s->device, 48 * nk, c->s1));
                    HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp + 48 * nk, c->device, in[k].p, s->device, 48 * nk, c->s1));
                    HIPCHK(hipMemcpyPeerAsync(w.d_stage_blobs, c->device, in[k].blobs, s->device, (size_t)BLOB_BYTES * nk, c->s1));
                }
                db = w.d_stage_blobs;
                dc = w.d_stage_cp;
                dp = w.d_stage_cp + 48 * nk;
            }
            if ((rc = phase1_launch_locked(db, dc, dp, nk, 1, c, kind == MultiSrc::Host ? &hb : nullptr)) != KZG_OK) return rc;
            return phase1_wait_locked(records.data() + 160 * off[k], &bad[k], c);
        };
        rcs[k] = body();
        if (rcs[k] != KZG_OK) {
            msgs[k] = g_err;
            (void)hipStreamSynchronize(c->s1);  // nothing of this shard stays in flight behind an error
            if (c->s_copy) (void)hipStreamSynchronize(c->s_copy);
            (void)hipGetLastError();
        }
    };
    if (active > 1) {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < D; k++) {
            try {
                pool.emplace_back(stage_a, k);
            } catch (const std::system_error&) {  // no thread to be had: this shard runs on the calling thread
                stage_a(k);
            }
        }
        stage_a(0);
        for (auto& th : pool) th.join();
    } else {
        for (size_t k = 0; k < D; k++) stage_a(k);
    }
    HIPCHK(hipSetDevice(s->device));
    auto clear_groups = [&] {  // no shard keeps a group "in flight" behind a call that has ended
        for (size_t k = 0; k < D; k++) shard_of(s, k)->ws.pending_n = shard_of(s, k)->ws.pending_b = shard_of(s, k)->ws.finish_b = 0;
    };
    for (size_t k = 0; k < D; k++)
        if (rcs[k] != KZG_OK) {
            clear_groups();
            return fail(rcs[k], msgs[k]);
        }
    for (size_t k = 0; k < D; k++)  // the reference's Err for an undecodable point / non-canonical element, whichever shard holds it
        if (bad[k]) {
            clear_groups();
            return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
        }
    s->multi_ms[1] = (float)ms_since(t_call);
    // ---- r: the whole transcript hashed once, here
    auto t0 = std::chrono::steady_clock::now();
    uint8_t r_le[32];
    if (!host_batch_challenges(r_le, records.data(), 1, n_total, n_total, 0)) return fail(KZG_MALLOC, "batch transcript buffer");
    s->multi_ms[2] = (float)ms_since(t0);
    // ---- phase 2 on every shard (launches only), then the exchange
    t0 = std::chrono::steady_clock::now();
    const bool rccl = m->exchange == MULTI_EXCHANGE_RCCL;
    KzgRet rc = KZG_OK;
    auto drain = [&](KzgRet code) {  // an error after phase 2 was launched: leave nothing in flight
        const std::string msg = g_err;
        for (size_t k = 0; k < D; k++) {
            (void)hipSetDevice(shard_of(s, k)->device);
            (void)hipStreamSynchronize(shard_of(s, k)->s1);
        }
        (void)hipGetLastError();
        (void)hipSetDevice(s->device);
        clear_groups();
        g_err = msg;
        return code;
    };
    for (size_t k = 0; k < D; k++) {
        const KzgSettings* c = shard_of(s, k);
        if (hipSetDevice(c->device) != hipSuccess) return drain(fail(KZG_ERROR, "HIP: hipSetDevice"));
        if (in[k].n == 0) {
            // an empty shard contributes the identity: all-zero Jacobian coordinates (Z = 0)
            if (rccl && hipMemsetAsync(c->ws.d_ab, 0, 288, c->s1) != hipSuccess) return drain(fail(KZG_ERROR, "HIP: hipMemsetAsync"));
            continue;
        }
        if ((rc = phase2_launch_locked(nullptr, n_total, off[k], c, 0, r_le, /*want_partials=*/!rccl)) != KZG_OK) return drain(rc);
    }
    s->multi_ms[3] = (float)ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (rccl) {
        if ((rc = multi_allgather_partials(s, 1)) != KZG_OK) return drain(rc);
        if (hipSetDevice(s->device) != hipSuccess) return drain(fail(KZG_ERROR, "HIP: hipSetDevice"));
        s->multi_ms[4] = (float)ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if ((rc = finish_launch_locked(nullptr, D, 1, s, /*parts_on_device=*/true)) != KZG_OK) return drain(rc);
    } else {
        std::vector<uint8_t> parts;
        parts.reserve(288 * D);
        for (size_t k = 0; k < D; k++) {
            if (in[k].n == 0) continue;
            const KzgSettings* c = shard_of(s, k);
            uint8_t part[288];
            if (hipSetDevice(c->device) != hipSuccess) return drain(fail(KZG_ERROR, "HIP: hipSetDevice"));
            if ((rc = phase2_wait_locked(part, c)) != KZG_OK) return drain(rc);
            parts.insert(parts.end(), part, part + 288);
        }
        if (hipSetDevice(s->device) != hipSuccess) return drain(fail(KZG_ERROR, "HIP: hipSetDevice"));
        s->multi_ms[4] = (float)ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if ((rc = finish_launch_locked(parts.data(), parts.size() / 288, 1, s)) != KZG_OK) return drain(rc);
    }
    if ((rc = finish_wait_locked(ok, s)) != KZG_OK) return drain(rc);
    // the peers' groups are complete as well (their streams were waited for, or are behind the collective the fold has consumed)
    for (size_t k = 1; k < D; k++) {
        const KzgSettings* c = shard_of(s, k);
        if (rccl) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->s1);
        }
        c->ws.pending_n = c->ws.pending_b = c->ws.finish_b = 0;
    }
    (void)hipSetDevice(s->device);
    s->multi_ms[5] = (float)ms_since(t0);
    s->multi_ms[0] = (float)ms_since(t_call);
    return KZG_OK;
}

// contiguous split of [0, n) over D shards: sizes differ by at most one, the first n % D shards take the extra blob
static void multi_split(std::vector<ShardIn>& in, const uint8_t* blobs, const uint8_t* c, const uint8_t* p, size_t n, size_t D) {
    in.assign(D, ShardIn());
    size_t o = 0;
    for (size_t k = 0; k < D; k++) {
        const size_t nk = n / D + (k < n % D ? 1 : 0);
        in[k].n = nk;
        in[k].blobs = blobs + o * (size_t)BLOB_BYTES;
        in[k].c = c + 48 * o;
        in[k].p = p + 48 * o;
        o += nk;
    }
}

// does this call of n blobs (one array, host or first-device resident) go through the shards?
static bool multi_takes(const KzgSettings* s, size_t n) { return s->multi && n >= 2 && n >= s->multi->min_blobs; }

// one array of n blobs (host memory, or device memory of the first device) through the shards; the caller holds s->mu
static KzgRet multi_array_locked(bool* ok, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n, bool host,
                                 const KzgSettings* s) {
    try {  // (host buffers of the call: 160 bytes per blob of transcript records; nothing may be thrown across the C ABI)
        std::vector<ShardIn> in;
        multi_split(in, blobs, commitments, proofs, n, shard_count(s));
        return multi_batch_locked(ok, in, host ? MultiSrc::Host : MultiSrc::Primary, s);
    } catch (const std::bad_alloc&) {
        return fail(KZG_MALLOC, "host buffers of the sharded call");
    }
}

// A STREAM of host batches over the shards: batches [k B / D, (k + 1) B / D) go to shard k, which runs the single-device stream
// (chunked copies on its own copy stream overlapped with verification) on a host thread of its own.  Batches are independent,
// so there is no exchange at all.  The caller holds s->mu.
static KzgRet multi_host_stream_locked(bool* ok_out, uint8_t* err_out, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs,
                                       size_t n, size_t n_batches, const KzgSettings* s) {
    const size_t D = shard_count(s);
    std::vector<KzgRet> rcs(D, KZG_OK);
    std::vector<std::string> msgs(D);
    auto run = [&](size_t k) {
        const size_t b0 = n_batches * k / D, b1 = n_batches * (k + 1) / D;
        if (b0 == b1) return;
        const KzgSettings* c = shard_of(s, k);
        if (hipSetDevice(c->device) != hipSuccess) {
            rcs[k] = KZG_ERROR;
            msgs[k] = "HIP: hipSetDevice";
            return;
        }
        rcs[k] = host_stream_locked(ok_out + b0, err_out ? err_out + b0 : nullptr, blobs + b0 * n * (size_t)BLOB_BYTES, commitments + 48 * b0 * n,
                                    proofs + 48 * b0 * n, n, b1 - b0, c);
        if (rcs[k] != KZG_OK) msgs[k] = g_err;
    };
    {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < D; k++) {
            try {
                pool.emplace_back(run, k);
            } catch (const std::system_error&) {
                run(k);
            }
        }
        run(0);
        for (auto& th : pool) th.join();
    }
    HIPCHK(hipSetDevice(s->device));
    for (size_t k = 0; k < D; k++)
        if (rcs[k] != KZG_OK) return fail(rcs[k], msgs[k]);
    return KZG_OK;
}

// ---- entry points (include/kzg_rs_amd.h)
// One batch whose shards are ALREADY resident on the devices of the handle: shard k = n_local[k] blobs on devices[k], global
// blob order = shard order (BASELINE configs[4]: 8 x 32 768).  The per-device form of kzg_verify_blob_kzg_proof_batch_device.
extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_sharded(bool* ok, const void* const* d_blobs, const void* const* d_commitments,
                                                          const void* const* d_proofs, const size_t* n_local, size_t n_shards,
                                                          const KzgSettings* s) {
    if (!ok || !s || !n_local || !d_blobs || !d_commitments || !d_proofs) return fail(KZG_BADARGS, "null argument");
    if (n_shards != shard_count(s)) return fail(KZG_BADARGS, "kzg_verify_blob_kzg_proof_batch_sharded: one shard per device of the handle");
    size_t n = 0;
    for (size_t k = 0; k < n_shards; k++) {
        if (n_local[k] && (!d_blobs[k] || !d_commitments[k] || !d_proofs[k])) return fail(KZG_BADARGS, "null shard");
        n += n_local[k];
    }
    if (n == 0) {  // src/kzg_proof.rs:478-480
        *ok = true;
        return KZG_OK;
    }
    if (!s->multi || n == 1) {  // a single-device handle, or the single-blob branch (:482-489): the one non-empty shard
        size_t k = 0;
        while (n_local[k] == 0) k++;
        if (s->multi && k != 0) {  // (the lone blob lies on another device of the list: bring it to the first)
            std::lock_guard<std::mutex> lk(s->mu);
            HIPCHK(hipSetDevice(s->device));
            KzgRet rc = ws_reserve(s, 1, 1, STAGE_BLOBS);
            if (rc != KZG_OK) return rc;
            Workspace& w = s->ws;
            select_streams(s, 1);
            const int src = s->multi->devices[k];
            HIPCHK(hipMemcpyPeerAsync(w.d_stage_blobs, s->device, d_blobs[k], src, BLOB_BYTES, s->s1));
            HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp, s->device, d_commitments[k], src, 48, s->s1));
            HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp + 48, s->device, d_proofs[k], src, 48, s->s1));
            return batch_device_locked(ok, w.d_stage_blobs, w.d_stage_cp, w.d_stage_cp + 48, 1, s);
        }
        return kzg_verify_blob_kzg_proof_batch_device(ok, d_blobs[k], d_commitments[k], d_proofs[k], n_local[k], s);
    }
    std::lock_guard<std::mutex> lk(s->mu);
    try {
        std::vector<ShardIn> in(n_shards);
        for (size_t k = 0; k < n_shards; k++) {
            in[k].blobs = (const uint8_t*)d_blobs[k];
            in[k].c = (const uint8_t*)d_commitments[k];
            in[k].p = (const uint8_t*)d_proofs[k];
            in[k].n = n_local[k];
        }
        return multi_batch_locked(ok, in, MultiSrc::PerDevice, s);
    } catch (const std::bad_alloc&) {
        return fail(KZG_MALLOC, "host buffers of the sharded call");
    }
}

// the shape of a handle: its devices (shard k on devices_out[k]), and how the partial sums travel (0 single device, 1 host
// staging, 2 in-process RCCL all-gather)
extern "C" KzgRet kzg_settings_devices(const KzgSettings* s, size_t* n_devices, int* devices_out, size_t cap, int* exchange) {
    if (!s || !n_devices) return fail(KZG_BADARGS, "null argument");
    const size_t D = shard_count(s);
    *n_devices = D;
    if (devices_out)
        for (size_t k = 0; k < D && k < cap; k++) devices_out[k] = s->multi ? s->multi->devices[k] : s->device;
    if (exchange) *exchange = s->multi ? s->multi->exchange : MULTI_EXCHANGE_NONE;
    if (s->multi && s->multi->exchange == MULTI_EXCHANGE_HOST) g_err = s->multi->exchange_note;  // readable through kzg_last_error()
    return KZG_OK;
}

// host wall-clock stages of the last sharded call on this handle, milliseconds: [0] whole call, [1] inputs onto the devices +
// phase 1 (all shards), [2] the transcript hash, [3] phase-2 launches, [4] the exchange (RCCL: enqueue; host: waits for the
// partial sums), [5] fold + pairing
extern "C" KzgRet kzg_multi_last_timings(const KzgSettings* s, float out_ms[8]) {
    if (!s || !out_ms) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_ms, s->multi_ms, sizeof(float) * 8);
    return KZG_OK;
}
