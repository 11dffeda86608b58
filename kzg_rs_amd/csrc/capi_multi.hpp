// capi_multi.hpp - multi-GPU behind the reference's own signature: ONE process, one settings handle over a device list.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.  Host code only.
//
// KzgProof::verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-477) takes one Vec<Blob> and returns one bool; a caller of it
// is one process.  A handle made over D devices (kzg_settings_*_devices, or KZG_DEVICES in the environment of an unchanged
// caller) shards the batch by blob - the per-blob loop of :261-273 is the data-parallel axis - and runs the three phases of
// capi_verify.hpp on every device from this one process.  A sharded batch is a list of PIECES, each a contiguous range of
// the global blob order on one device, on a private lane (handle) of that device:
//   * shards ALREADY RESIDENT on their devices (kzg_verify_blob_kzg_proof_batch_sharded, BASELINE configs[4]): one piece per
//     device, contiguous ranges, global order = shard order;
//   * ONE array in host memory (the reference's Vec<Blob>) or in the memory of one device: cut into chunks that are dealt
//     to the devices INTERLEAVED (chunk c -> device c mod D, several chunks per device, each crossing that device's own
//     PCIe link), so that at any moment every device has finished about the same prefix of the global order;
//   phase 1 per piece  ->  the 160-byte transcript records come back piece by piece and ONE streaming SHA-256 context
//   (host_only.hpp BatchTranscript) consumes them in global order WHILE later pieces are still copied and computed - the
//   42 MB hash of a 262 144-blob batch (:291-334, ~20 ms on a SHA-NI core) overlaps the copies instead of following them
//   ->  phase 2 per piece with r and its power offset r^offset (:279-289)
//   ->  the "G1 all-reduce" of the north star: the 288-byte partial sums (A, B) of every piece are gathered - through pinned
//       host memory by default, or by ncclAllGather over in-process RCCL communicators (option multi_exchange=rccl; point
//       addition is not an RCCL reduction op, so the reduction itself is k_fold_partials on the first device)
//   ->  one pairing there.
// A STREAM of such batches (kzg_verify_blob_kzg_proof_batch_sharded_stream) keeps several of them in flight on private lane
// sets, so that the transcript hash of batch j runs beside phase 1 of batches j + 1 ...; and INDEPENDENT batches need no
// exchange at all: the many-batch entry points route every launch group to the device that owns its memory
// (capi_pipeline.hpp) or deal whole batches to the devices (multi_host_stream_locked).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>

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

// librccl is bound at run time, on the first handle that asks for the RCCL exchange: nobody else loads its 570 MB, and a
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

enum { MULTI_EXCHANGE_NONE = 0, MULTI_EXCHANGE_HOST = 1, MULTI_EXCHANGE_RCCL = 2,
       MULTI_EXCHANGE_BOTH = 3 };  // BOTH: the self-test only - the same partial sums through RCCL and through host memory, folded and paired twice
constexpr size_t MULTI_MAX_PIECES = MAX_WORLD;  // pieces of one sharded batch: each contributes one 288-byte partial to the fold
struct MultiState {
    std::vector<int> devices;        // shard k runs on devices[k]; shard 0 is the handle itself
    std::vector<ncclComm_t> comms;   // one communicator per shard (exchange == RCCL)
    int exchange = MULTI_EXCHANGE_HOST;
    std::string exchange_note;       // why the partial sums travel through host memory
    size_t min_blobs = 256;          // host / single-device-resident batches below this stay on shard 0 (option multi_min_blobs)
    size_t chunks_per_device = 8;    // a host (or single-device) array is dealt in about this many chunks per device (option multi_chunks)
    size_t min_chunk = 128;          // ... of at least this many blobs (option multi_min_chunk)
    std::mutex rccl_mu;              // collectives of concurrent lane sets go out one at a time: the same order on every communicator
    std::mutex last_r_mu;            // the test hook's copy of the last batch challenge (batches of a sharded stream finish on several threads)
};

static const KzgSettings* shard_of(const KzgSettings* s, size_t k) { return k == 0 ? s : s->peers[k - 1]; }
static size_t shard_count(const KzgSettings* s) { return s->multi ? s->multi->devices.size() : 1; }
// lane `i` of a handle: the handle itself for i == 0, else its (i - 1)-th private lane (pipeline_lanes has made it)
static const KzgSettings* lane_of(const KzgSettings* c, size_t i) { return i == 0 ? c : c->lanes[i - 1]; }
// at least `count` lanes beside the handle itself; the caller has set the handle's device and no other thread uses the handle
static KzgRet pipeline_lanes(const KzgSettings* s, size_t count) {
    while (s->lanes.size() < count) {
        KzgSettings* l = nullptr;
        KzgRet rc = settings_lane(&l, s);
        if (rc != KZG_OK) return rc;
        s->lanes.push_back(l);
    }
    return KZG_OK;
}

// all-gather of `count` x 288 bytes per shard - the partial sums of that shard's pieces, collected in its send buffer -
// into every shard's ws.d_parts ([D][count] x 288 B) on the given handle of every shard, each on that handle's stream s1:
// the data-path collective of the multi-GPU verification (SURVEY 8e "exchange 2").  A failing call still closes the group.
static KzgRet multi_allgather_partials(const KzgSettings* s, const std::vector<const KzgSettings*>& h, size_t count) {
    MultiState* m = s->multi;
    RcclApi* api = rccl_api();
    std::lock_guard<std::mutex> lk(m->rccl_mu);
    ncclResult_t r = api->GroupStart();
    if (r != ncclSuccess) return fail(KZG_ERROR, std::string("RCCL: ") + api->GetErrorString(r) + " at ncclGroupStart");
    std::string first;
    for (size_t k = 0; k < h.size() && first.empty(); k++) {
        if (hipSetDevice(h[k]->device) != hipSuccess) {
            (void)hipGetLastError();
            first = "HIP: hipSetDevice inside the RCCL group";
            break;
        }
        r = api->AllGather(h[k]->ws.d_send, h[k]->ws.d_parts, 288 * count, ncclUint8, m->comms[k], h[k]->s1);
        if (r != ncclSuccess) first = std::string("RCCL: ") + api->GetErrorString(r) + " at ncclAllGather";
    }
    r = api->GroupEnd();  // always: an open group would swallow every later collective of the process
    if (first.empty() && r != ncclSuccess) first = std::string("RCCL: ") + api->GetErrorString(r) + " at ncclGroupEnd";
    (void)hipSetDevice(s->device);
    return first.empty() ? KZG_OK : fail(KZG_ERROR, first);
}

static void multi_free(KzgSettings* s) {
    if (s->multi) {
        RcclApi* api = s->multi->comms.empty() ? nullptr : rccl_api();
        for (size_t k = 0; k < s->multi->comms.size(); k++)
            if (s->multi->comms[k] && api && api->CommDestroy) {
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

static KzgRet multi_exchange_selftest(KzgSettings* s, bool& equal, std::string& verdict);
// Peers of a freshly built shard 0 (s->device == devices[0]) and the exchange (chosen by a self-test, see below).  On failure
// the caller frees the handle.
static KzgRet multi_build(KzgSettings* s, const uint8_t tau_g2[96], const std::vector<int>& devices) {
    const bool force = opt_flag("multi_force", false);  // test rig: a list of ONE device still takes the sharded path
    if (devices.size() < 2 && !force) return KZG_OK;  // a plain single-device handle
    MultiState* m = new MultiState();
    s->multi = m;
    m->devices = devices;
    m->min_blobs = (size_t)std::max(2L, opt_int("multi_min_blobs", 256));  // (the options are read when the handle is made)
    m->chunks_per_device = (size_t)std::min(64L, std::max(1L, opt_int("multi_chunks", 8)));
    m->min_chunk = (size_t)std::max(1L, opt_int("multi_min_chunk", 128));
    const size_t D = devices.size();
    KzgRet rc;
    for (size_t k = 1; k < D; k++) {
        HIPCHK(hipSetDevice(devices[k]));
        KzgSettings* p = nullptr;
        if ((rc = settings_common(&p, tau_g2)) != KZG_OK) return rc;
        delete p->small;  // (the handle the caller holds queues the small calls of every device)
        p->small = nullptr;
        s->peers.push_back(p);
    }
    // every shard owns the group-sized buffers from the start (the fold and the warm-up collective below use them)
    for (size_t k = 0; k < D; k++) {
        const KzgSettings* c = shard_of(s, k);
        HIPCHK(hipSetDevice(c->device));
        if ((rc = ws_reserve(c, 16, 1, STAGE_NONE)) != KZG_OK) return rc;
    }
    // How the 288-byte partial sums travel.  KZG_OPTIONS multi_exchange=host | rccl forces one; otherwise the handle asks RCCL
    // for in-process communicators and PROVES the leg on this very set of devices before trusting it: multi_exchange_selftest
    // runs synthetic sharded batches through both exchanges and compares the gathered buffers bit for bit and the verdicts.
    // Equal -> the north star's "RCCL all-gather over xGMI" is the exchange; RCCL unusable, or any difference -> pinned host
    // memory, and the handle's note (kzg_settings_note) says why.  Rounds 3-4 kept RCCL an opt-in because no box this project
    // ran on had two GPUs; the self-test is what lets the first box that has them switch it on safely.
    const char* ex = opt_str("multi_exchange");
    const bool want_rccl = ex && strcmp(ex, "rccl") == 0, want_host = ex && strcmp(ex, "host") == 0;
    m->exchange = MULTI_EXCHANGE_HOST;
    if (want_host) m->exchange_note = "partial sums through pinned host memory (forced: multi_exchange=host)";
    else {
        bool distinct = true;
        for (size_t a = 0; a < D; a++)
            for (size_t b = a + 1; b < D; b++) distinct &= devices[a] != devices[b];
        RcclApi* api = distinct ? rccl_api() : nullptr;
        bool usable = false;
        if (!distinct) m->exchange_note = "the device list names a device twice (ncclCommInitAll needs distinct devices)";
        else if (!api->CommInitAll) m->exchange_note = api->why;
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
                std::vector<const KzgSettings*> h(D);
                for (size_t k = 0; k < D; k++) {
                    h[k] = shard_of(s, k);
                    HIPCHK(hipSetDevice(h[k]->device));
                    HIPCHK(hipMemsetAsync(h[k]->ws.d_send, 0, 288, h[k]->s1));
                }
                rc = multi_allgather_partials(s, h, 1);
                for (size_t k = 0; k < D && rc == KZG_OK; k++) {
                    HIPCHK(hipSetDevice(h[k]->device));
                    if (hipStreamSynchronize(h[k]->s1) != hipSuccess) {
                        (void)hipGetLastError();
                        rc = fail(KZG_ERROR, "HIP: hipStreamSynchronize behind the warm-up collective");
                    }
                }
                if (rc != KZG_OK) m->exchange_note = "the warm-up ncclAllGather failed: " + g_err;
                else usable = true;
            }
        }
        if (usable) {
            std::string verdict;
            bool equal = false;
            rc = multi_exchange_selftest(s, equal, verdict);
            if (rc != KZG_OK) {
                verdict = "exchange self-test could not run: " + g_err;
                equal = false;
            }
            m->exchange_note = verdict;
            if (!equal) {
                usable = false;
                fprintf(stderr, "kzg_rs_amd: %s - the partial sums of this handle travel through pinned host memory\n", verdict.c_str());
            }
        }
        m->exchange = usable ? MULTI_EXCHANGE_RCCL : MULTI_EXCHANGE_HOST;
        if (!usable) {
            for (size_t k = 0; k < m->comms.size(); k++)
                if (m->comms[k] && rccl_api()->CommDestroy) {
                    (void)hipSetDevice(devices[k]);
                    (void)rccl_api()->CommDestroy(m->comms[k]);
                }
            m->comms.clear();
            if (want_rccl) return fail(KZG_ERROR, "multi_exchange=rccl but RCCL is unusable: " + m->exchange_note);
            m->exchange_note = "partial sums through pinned host memory: " + m->exchange_note;
        }
    }
    s->note = "exchange: " + m->exchange_note;
    if (s->small) s->small->max_lanes = std::min<size_t>(SMALL_LANES_MAX, s->small->max_lanes * D);  // small calls: the same number of lanes on every device (lane i on shard i mod D)
    HIPCHK(hipSetDevice(s->device));
    return KZG_OK;
}

// where the inputs of a sharded call lie
enum class MultiSrc { Host, OneDevice, PerDevice };
struct ShardIn {
    const uint8_t *blobs = nullptr, *c = nullptr, *p = nullptr;
    size_t n = 0;
};

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// one contiguous range [off, off + n) of a sharded batch's global blob order, on lane `lane` of shard k
struct Piece {
    size_t k = 0, lane = 0, off = 0, n = 0;
    const uint8_t *blobs = nullptr, *c = nullptr, *p = nullptr;
    const KzgSettings* h = nullptr;  // the handle it runs on
};
struct MultiTimes {
    // [0] whole call [1] inputs onto the devices + phase 1, all pieces [2] what the transcript hash added after the last piece
    // [3] phase-2 launches [4] exchange [5] fold + pairing [6] the hash's own busy time [7] pieces
    float ms[8] = {};
    // (the exchange self-test) the gathered partial sums as the fold sees them: [D][slots] x 288 B, slot j of shard k = its j-th
    // piece's (A, B), unused slots all-zero = the identity - the same layout whichever way they travelled
    std::vector<uint8_t>* capture = nullptr;
    std::vector<uint8_t>* capture_rccl = nullptr;  // (exchange BOTH: capture = the host leg's buffer, capture_rccl = ncclAllGather's)
    bool ok_rccl = false;                          // (exchange BOTH: the verdict of the fold + pairing fed by ncclAllGather; *ok = the host leg's)
};

// ONE batch as pieces, listed in global blob order; src_dev: the device the inputs lie on (MultiSrc::OneDevice).  The caller
// holds s->mu or owns the lanes the pieces run on; lane `lane0` of the first shard folds and pairs (and lane0 of every shard
// is its handle in the RCCL collective).  Pieces of
// one shard run in order with at most two in flight (the next one's copy behind the current one's kernels); with a host
// source, or several pieces per shard, every shard gets a host thread of its own (a pageable copy holds its thread) and
// the calling thread hashes.
static KzgRet multi_pieces_locked(bool* ok, std::vector<Piece>& pieces, size_t n_total, MultiSrc kind, int src_dev, const KzgSettings* s,
                                  size_t lane0, MultiTimes& tm) {
    MultiState* m = s->multi;
    const KzgSettings* const fold = lane_of(s, lane0);
    const size_t D = shard_count(s), NP = pieces.size();
    const auto t_call = std::chrono::steady_clock::now();
    std::vector<uint8_t> records(160 * n_total);
    std::vector<KzgRet> rcs(NP, KZG_OK);
    std::vector<std::string> msgs(NP);
    std::vector<uint8_t> bad(NP, 0), done(NP, 0), ran(NP, 0);
    std::mutex mu;
    std::condition_variable cv;
    auto mark = [&](size_t i) {
        {
            std::lock_guard<std::mutex> lk(mu);
            done[i] = 1;
        }
        cv.notify_all();
    };
    auto drain_handle = [](const KzgSettings* h) {  // nothing of this handle stays in flight, nothing may still read the caller's memory
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->s1);
        if (h->s2) (void)hipStreamSynchronize(h->s2);
        if (h->s_sha) (void)hipStreamSynchronize(h->s_sha);
        if (h->s_copy) (void)hipStreamSynchronize(h->s_copy);
        (void)hipGetLastError();
        h->ws.pending_n = h->ws.pending_b = h->ws.finish_b = 0;
    };
    // ---- stage A: inputs onto the piece's device, phase 1, records back
    auto launch = [&](size_t i) -> KzgRet {
        Piece& pc = pieces[i];
        const KzgSettings* c = pc.h;
        HIPCHK(hipSetDevice(c->device));
        const bool copy = kind == MultiSrc::Host || (kind == MultiSrc::OneDevice && c->device != src_dev);
        KzgRet rc = ws_reserve(c, pc.n, 1, copy ? STAGE_BLOBS : STAGE_NONE);
        if (rc != KZG_OK) return rc;
        Workspace& w = c->ws;
        select_streams(c, pc.n);
        const void *db = pc.blobs, *dc = pc.c, *dp = pc.p;
        const HostBatch hb{pc.blobs, pc.c, pc.p};
        if (copy) {
            if (kind == MultiSrc::OneDevice) {  // resident on one device: the piece crosses xGMI (a caller that can should hand over per-device shards)
                HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp, c->device, pc.c, src_dev, 48 * pc.n, c->s1));
                HIPCHK(hipMemcpyPeerAsync(w.d_stage_cp + 48 * pc.n, c->device, pc.p, src_dev, 48 * pc.n, c->s1));
                HIPCHK(hipMemcpyPeerAsync(w.d_stage_blobs, c->device, pc.blobs, src_dev, (size_t)BLOB_BYTES * pc.n, c->s1));
            }  // (a host piece: phase 1 brings it over itself, in slices across its blobs - capi_verify.hpp host_slices)
            db = w.d_stage_blobs;
            dc = w.d_stage_cp;
            dp = w.d_stage_cp + 48 * pc.n;
        }
        ran[i] = 1;  // (from here on the handle may have work in flight)
        return phase1_launch_locked(db, dc, dp, pc.n, 1, c, kind == MultiSrc::Host ? &hb : nullptr);
    };
    auto wait = [&](size_t i) -> KzgRet {
        Piece& pc = pieces[i];
        HIPCHK(hipSetDevice(pc.h->device));
        return phase1_wait_locked(records.data() + 160 * pc.off, &bad[i], pc.h);
    };
    auto settle = [&](size_t i, KzgRet rc) {  // called on the thread that ran the piece (g_err is thread-local)
        rcs[i] = rc;
        if (rc != KZG_OK) {
            msgs[i] = g_err;
            drain_handle(pieces[i].h);
        }
        mark(i);
    };
    auto skip = [&](size_t i) {  // an earlier piece failed: this one is not run (or not waited for)
        if (ran[i]) drain_handle(pieces[i].h);
        rcs[i] = KZG_ERROR;
        msgs[i] = "not run: an earlier piece failed";
        mark(i);
    };
    std::vector<std::vector<size_t>> of_shard(D);
    for (size_t i = 0; i < NP; i++) of_shard[pieces[i].k].push_back(i);
    auto run_shard = [&](size_t k) {
        const std::vector<size_t>& q = of_shard[k];
        bool stop = false;
        for (size_t j = 0; j < q.size() && !stop; j++) {
            KzgRet rc = launch(q[j]);
            if (rc != KZG_OK) {
                settle(q[j], rc);
                stop = true;
            }
            if (j >= 1 && !done[q[j - 1]]) {
                rc = wait(q[j - 1]);
                settle(q[j - 1], rc);
                stop |= rc != KZG_OK;
            }
        }
        for (size_t j = 0; j < q.size(); j++)  // the last piece, or what a failure has left behind
            if (!done[q[j]]) {
                if (ran[q[j]] && !stop) {
                    const KzgRet rc = wait(q[j]);
                    settle(q[j], rc);
                    stop |= rc != KZG_OK;
                } else skip(q[j]);
            }
    };
    size_t busy_shards = 0, max_per_shard = 0;
    for (size_t k = 0; k < D; k++) {
        busy_shards += !of_shard[k].empty();
        max_per_shard = std::max(max_per_shard, of_shard[k].size());
    }
    const bool threads = busy_shards > 1 && (kind != MultiSrc::PerDevice || max_per_shard > 1);
    double hash_busy = 0.0;
    BatchTranscript transcript(n_total);
    auto t_last_piece = t_call;
    bool failed = false;
    auto consume = [&](size_t i) {  // the hasher: piece i's records, once they are there (pieces are listed in global order)
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[i] != 0; });
        }
        t_last_piece = std::chrono::steady_clock::now();
        if (rcs[i] != KZG_OK || bad[i]) failed = true;
        if (failed || n_total == 1) return;
        const auto t0 = std::chrono::steady_clock::now();
        transcript.records(records.data() + 160 * pieces[i].off, pieces[i].n);
        hash_busy += ms_since(t0);
    };
    if (threads) {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < D; k++) {
            if (of_shard[k].empty()) continue;
            try {
                pool.emplace_back(run_shard, k);
            } catch (const std::system_error&) {  // no thread to be had: this shard runs on the calling thread
                run_shard(k);
            }
        }
        for (size_t i = 0; i < NP; i++) consume(i);
        for (auto& th : pool) th.join();
    } else if (max_per_shard > 1 || kind != MultiSrc::PerDevice) {
        for (size_t k = 0; k < D; k++)
            if (!of_shard[k].empty()) run_shard(k);
        for (size_t i = 0; i < NP; i++) consume(i);
    } else {
        // resident shards, one piece each: every launch first, then wait + hash shard by shard - the hash of shard k runs
        // while the shards behind it finish
        bool stop = false;
        for (size_t i = 0; i < NP && !stop; i++) {
            const KzgRet rc = launch(i);
            if (rc != KZG_OK) {
                settle(i, rc);
                stop = true;
            }
        }
        for (size_t i = 0; i < NP; i++) {
            if (!done[i]) {
                if (ran[i] && !stop) {
                    const KzgRet rc = wait(i);
                    settle(i, rc);
                    stop |= rc != KZG_OK;
                } else skip(i);
            }
            consume(i);
        }
    }
    (void)hipSetDevice(s->device);
    auto drain_all = [&](KzgRet code) {  // an error: leave nothing in flight, no handle with a group pending
        const std::string msg = g_err;
        for (size_t i = 0; i < NP; i++) drain_handle(pieces[i].h);
        drain_handle(fold);
        (void)hipSetDevice(s->device);
        g_err = msg;
        return code;
    };
    for (size_t i = 0; i < NP; i++)  // the first real failure, not what it made us skip
        if (rcs[i] != KZG_OK && msgs[i].compare(0, 8, "not run:") != 0) return drain_all(fail(rcs[i], msgs[i]));
    for (size_t i = 0; i < NP; i++)
        if (rcs[i] != KZG_OK) return drain_all(fail(rcs[i], msgs[i]));
    for (size_t i = 0; i < NP; i++)  // the reference's Err for an undecodable point / non-canonical element, whichever piece holds it
        if (bad[i]) return drain_all(fail(KZG_BADARGS, "Failed to parse G1Affine from bytes"));
    tm.ms[1] = (float)std::chrono::duration<double, std::milli>(t_last_piece - t_call).count();
    // ---- r: the transcript was hashed as the pieces came in; close it
    uint8_t r_le[32] = {0};
    if (n_total > 1) transcript.r(r_le);
    {
        std::lock_guard<std::mutex> lk_r(m->last_r_mu);
        memcpy(s->multi_last_r, r_le, 32);
    }
    tm.ms[2] = (float)ms_since(t_last_piece);
    tm.ms[6] = (float)hash_busy;
    tm.ms[7] = (float)NP;
    // ---- phase 2 on every piece (launches only), then the exchange
    auto t0 = std::chrono::steady_clock::now();
    const bool both = m->exchange == MULTI_EXCHANGE_BOTH;  // (the self-test: one set of partial sums, both ways)
    const bool rccl = m->exchange == MULTI_EXCHANGE_RCCL || both;
    KzgRet rc = KZG_OK;
    for (size_t i = 0; i < NP; i++) {
        if (hipSetDevice(pieces[i].h->device) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipSetDevice"));
        if ((rc = phase2_launch_locked(nullptr, n_total, pieces[i].off, pieces[i].h, 0, r_le, /*want_partials=*/!rccl || both)) != KZG_OK) return drain_all(rc);
    }
    tm.ms[3] = (float)ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (rccl) {
        // every shard collects the partial sums of its pieces in the send buffer of one of its handles - `cnt` slots, unused
        // ones all-zero = the identity (Z = 0) - and the all-gather leaves [D][cnt] x 288 B on every shard
        const size_t cnt = std::max<size_t>(1, max_per_shard);
        std::vector<const KzgSettings*> h(D, nullptr);
        for (size_t k = 0; k < D; k++) h[k] = lane_of(shard_of(s, k), lane0);
        for (size_t k = 0; k < D; k++) {
            if (hipSetDevice(h[k]->device) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipSetDevice"));
            if (hipMemsetAsync(h[k]->ws.d_send, 0, 288 * cnt, h[k]->s1) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipMemsetAsync"));
            for (size_t j = 0; j < of_shard[k].size(); j++) {
                const KzgSettings* ph = pieces[of_shard[k][j]].h;
                if (ph != h[k]) {  // the piece's MSM runs on its own stream: the collective's stream waits for it
                    if (hipEventRecord(ph->ev[3], ph->s1) != hipSuccess || hipStreamWaitEvent(h[k]->s1, ph->ev[3], 0) != hipSuccess)
                        return drain_all(fail(KZG_ERROR, "HIP: event between a piece and the collective"));
                }
                if (hipMemcpyAsync((uint8_t*)h[k]->ws.d_send + 288 * j, ph->ws.d_ab, 288, hipMemcpyDeviceToDevice, h[k]->s1) != hipSuccess)
                    return drain_all(fail(KZG_ERROR, "HIP: hipMemcpyAsync"));
            }
        }
        if ((rc = multi_allgather_partials(s, h, cnt)) != KZG_OK) return drain_all(rc);
        if (hipSetDevice(fold->device) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipSetDevice"));
        std::vector<uint8_t>* const cap = both ? tm.capture_rccl : tm.capture;
        if (cap) {  // (behind the collective on the fold's stream; read after finish_wait_locked has waited for that stream)
            cap->assign(288 * D * cnt, 0);
            if (hipMemcpyAsync(cap->data(), fold->ws.d_parts, 288 * D * cnt, hipMemcpyDeviceToHost, fold->s1) != hipSuccess)
                return drain_all(fail(KZG_ERROR, "HIP: hipMemcpyAsync"));
        }
        tm.ms[4] = (float)ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if ((rc = finish_launch_locked(nullptr, D * cnt, 1, fold, /*parts_on_device=*/true)) != KZG_OK) return drain_all(rc);
    }
    if (!rccl || both) {
        std::vector<uint8_t> parts(288 * NP);
        for (size_t i = 0; i < NP; i++) {
            if (hipSetDevice(pieces[i].h->device) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipSetDevice"));
            if ((rc = phase2_wait_locked(parts.data() + 288 * i, pieces[i].h)) != KZG_OK) return drain_all(rc);
        }
        if (hipSetDevice(fold->device) != hipSuccess) return drain_all(fail(KZG_ERROR, "HIP: hipSetDevice"));
        // (the self-test: the pairing fed by ncclAllGather is in flight on the fold's stream - its verdict first; only now, because
        // finish_wait_locked closes the fold handle's group and phase2_wait_locked above reads the group's size)
        if (both && (rc = finish_wait_locked(&tm.ok_rccl, fold)) != KZG_OK) return drain_all(rc);
        if (tm.capture) {
            const size_t cnt = std::max<size_t>(1, max_per_shard);
            tm.capture->assign(288 * D * cnt, 0);
            for (size_t k = 0; k < D; k++)
                for (size_t j = 0; j < of_shard[k].size(); j++) memcpy(tm.capture->data() + 288 * (k * cnt + j), parts.data() + 288 * of_shard[k][j], 288);
        }
        if (!both) tm.ms[4] = (float)ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if ((rc = finish_launch_locked(parts.data(), NP, 1, fold)) != KZG_OK) return drain_all(rc);
    }
    if ((rc = finish_wait_locked(ok, fold)) != KZG_OK) return drain_all(rc);
    // the pieces' groups are complete as well (their streams were waited for, or are behind the collective the fold has consumed)
    for (size_t i = 0; i < NP; i++) {
        const KzgSettings* c = pieces[i].h;
        if (rccl && c != fold) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->s1);
        }
        c->ws.pending_n = c->ws.pending_b = c->ws.finish_b = 0;
    }
    (void)hipSetDevice(s->device);
    tm.ms[5] = (float)ms_since(t0);
    tm.ms[0] = (float)ms_since(t_call);
    return KZG_OK;
}

// the lanes a sharded call needs on every shard (lane 0 is the shard itself); made on the calling thread
static KzgRet multi_ensure_lanes(const KzgSettings* s, size_t lanes_total) {
    for (size_t k = 0; k < shard_count(s); k++) {
        const KzgSettings* c = shard_of(s, k);
        HIPCHK(hipSetDevice(c->device));
        KzgRet rc = pipeline_lanes(c, lanes_total ? lanes_total - 1 : 0);
        if (rc != KZG_OK) return rc;
        for (size_t i = 0; i < lanes_total; i++)  // (the group-sized buffers: the RCCL send / receive buffers of a shard without a piece)
            if (lane_of(c, i)->ws.cap_b < 1 && (rc = ws_reserve(lane_of(c, i), 16, 1, STAGE_NONE)) != KZG_OK) return rc;
    }
    HIPCHK(hipSetDevice(s->device));
    return KZG_OK;
}

// ONE batch whose shards are resident on their devices (in[k] = shard k's contiguous slice, global blob order = shard
// order), on lane `lane` of every shard.  The caller holds s->mu (or owns the lane) and the lanes exist.
static KzgRet multi_batch_resident(bool* ok, const std::vector<ShardIn>& in, const KzgSettings* s, size_t lane, MultiTimes& tm) {
    std::vector<Piece> pieces;
    size_t off = 0;
    for (size_t k = 0; k < in.size(); k++) {
        if (in[k].n) {
            Piece pc;
            pc.k = k;
            pc.lane = lane;
            pc.off = off;
            pc.n = in[k].n;
            pc.blobs = in[k].blobs;
            pc.c = in[k].c;
            pc.p = in[k].p;
            pc.h = lane_of(shard_of(s, k), lane);
            pieces.push_back(pc);
        }
        off += in[k].n;
    }
    return multi_pieces_locked(ok, pieces, off, MultiSrc::PerDevice, -1, s, lane, tm);
}

// First contact with this set of devices (multi_build): is the in-process RCCL all-gather of the partial sums THE SAME as
// carrying them through host memory?  Two synthetic batches of `nb` blobs per shard, resident on their devices; the
// partial sums of each batch travel BOTH ways (the very same device buffers: a partial sum is a Jacobian triple, and a second
// MSM run may reach the same point through another order of additions and so another triple - the first form of this test
// compared two runs and saw "DIFFER" with equal verdicts) - ncclAllGather into the fold's device buffer, and through the pinned
// mirrors - with the gathered [D][slots] x 288-byte buffer captured as the fold reads it, and fold + pairing run once from each:
//   batch 1: pseudo-random canonical blobs, commitments (3 + 2 i) G and proofs (5 + 7 i) G: not a valid batch under any setup
//            (the verdict must be false both ways), but every shard's partial sums are non-trivial points that depend on every
//            blob of the batch through r - a byte that goes missing or lands in the wrong slot changes the buffer;
//   batch 2: all-zero blobs with commitment and proof at infinity: VALID under every setup (p = 0: C = O, y = 0, pi = O) -
//            the verdict must be true both ways, and every slot is the identity.
// equal = all four runs succeeded, the buffers are bit-identical per batch and the verdicts are false / true as expected.
// option multi_selftest_blobs = blobs per shard (default 128; 0 skips the test and trusts RCCL as rounds 3-4's opt-in did).
static KzgRet multi_exchange_selftest(KzgSettings* s, bool& equal, std::string& verdict) {
    MultiState* m = s->multi;
    const size_t D = shard_count(s);
    const size_t nb = (size_t)std::max(0L, std::min(4096L, opt_int("multi_selftest_blobs", 128)));
    equal = true;
    if (nb == 0) {
        verdict = "RCCL all-gather (in-process communicators), self-test skipped (multi_selftest_blobs=0)";
        return KZG_OK;
    }
    const auto t_start = std::chrono::steady_clock::now();
    KzgRet rc = multi_ensure_lanes(s, 1);
    if (rc != KZG_OK) return rc;
    std::vector<DevTmp> d_blobs(D), d_cp(D), d_sc(D);
    std::vector<uint8_t> hb(nb * (size_t)BLOB_BYTES), hs(64 * nb);
    std::vector<ShardIn> in(D);
    const int saved_exchange = m->exchange;
    bool verdicts[2][2] = {{false, false}, {false, false}};
    std::vector<uint8_t> cap[2][2];
    for (int batch = 0; batch < 2 && rc == KZG_OK; batch++) {
        for (size_t k = 0; k < D && rc == KZG_OK; k++) {
            const KzgSettings* c = shard_of(s, k);
            HIPCHK(hipSetDevice(c->device));
            if (!d_blobs[k].p) {
                HIPCHK(hipMalloc(&d_blobs[k].p, nb * (size_t)BLOB_BYTES));
                HIPCHK(hipMalloc(&d_cp[k].p, 96 * nb));
                HIPCHK(hipMalloc(&d_sc[k].p, 64 * nb));
            }
            uint8_t* cp = d_cp[k].as<uint8_t>();
            if (batch == 0) {
                uint64_t x = 0x9E3779B97F4A7C15ull * (k + 1) + 0x1234567ull;  // xorshift64*: reproducible, different per shard
                uint64_t* w = reinterpret_cast<uint64_t*>(hb.data());
                for (size_t i = 0; i < hb.size() / 8; i++) {
                    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
                    w[i] = x * 0x2545F4914F6CDD1Dull;
                }
                for (size_t i = 0; i < hb.size(); i += 32) hb[i] &= 0x3F;  // every field element below 2^254 < r (big-endian: byte 0 leads)
                memset(hs.data(), 0, hs.size());
                for (size_t i = 0; i < nb; i++) {  // little-endian scalars: commitments (3 + 2 g) G | proofs (5 + 7 g) G, g = the global index
                    const uint64_t g = k * nb + i, a = 3 + 2 * g, b = 5 + 7 * g;
                    memcpy(hs.data() + 32 * i, &a, 8);
                    memcpy(hs.data() + 32 * (nb + i), &b, 8);
                }
                HIPCHK(hipMemcpyAsync(d_blobs[k].p, hb.data(), hb.size(), hipMemcpyHostToDevice, c->s1));
                HIPCHK(hipMemcpyAsync(d_sc[k].p, hs.data(), hs.size(), hipMemcpyHostToDevice, c->s1));
                hipLaunchKernelGGL(k_g1_mul_generator, dim3((unsigned)((2 * nb + 63) / 64)), dim3(64), 0, c->s1, d_sc[k].as<Fr>(), cp, (int)(2 * nb));
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(c->s1));  // (hb / hs are refilled for the next shard)
            } else {
                HIPCHK(hipMemsetAsync(d_blobs[k].p, 0, nb * (size_t)BLOB_BYTES, c->s1));
                std::vector<uint8_t> inf(96 * nb, 0);
                for (size_t i = 0; i < 2 * nb; i++) inf[48 * i] = 0xC0;
                HIPCHK(hipMemcpyAsync(cp, inf.data(), inf.size(), hipMemcpyHostToDevice, c->s1));
                HIPCHK(hipStreamSynchronize(c->s1));
            }
            in[k] = ShardIn{d_blobs[k].as<uint8_t>(), cp, cp + 48 * nb, nb};
        }
        if (rc == KZG_OK) {  // ONE run: the same partial sums through both exchanges, folded and paired once from each
            m->exchange = MULTI_EXCHANGE_BOTH;
            MultiTimes tm;
            tm.capture = &cap[batch][0];
            tm.capture_rccl = &cap[batch][1];
            HIPCHK(hipSetDevice(s->device));
            rc = multi_batch_resident(&verdicts[batch][0], in, s, 0, tm);
            verdicts[batch][1] = tm.ok_rccl;
        }
    }
    m->exchange = saved_exchange;
    (void)hipSetDevice(s->device);
    if (rc != KZG_OK) return rc;
    // (A/B build only, for the test of the rejection path: one bit of what ncclAllGather delivered is flipped)
    if (ab_int("multi_selftest_corrupt", 0) && cap[0][1].size() > 5) cap[0][1][5] ^= 1;
    const bool same0 = cap[0][0] == cap[0][1] && !cap[0][0].empty(), same1 = cap[1][0] == cap[1][1] && !cap[1][0].empty();
    bool nontrivial = false;
    for (uint8_t b : cap[0][0]) nontrivial |= b != 0;
    const bool v_ok = !verdicts[0][0] && !verdicts[0][1] && verdicts[1][0] && verdicts[1][1];
    equal = same0 && same1 && nontrivial && v_ok;
    char buf[320];
    snprintf(buf, sizeof buf, "exchange self-test on %zu device(s), 2 synthetic batches of %zu x %zu blobs, host vs ncclAllGather: gathered %zu-byte buffers %s, "
                              "verdicts host %d/%d rccl %d/%d (want 0/1) - %s (%.0f ms)",
             D, D, nb, cap[0][0].size(), same0 && same1 ? (nontrivial ? "bit-identical" : "identical but all-zero") : "DIFFER", (int)verdicts[0][0],
             (int)verdicts[1][0], (int)verdicts[0][1], (int)verdicts[1][1], equal ? "RCCL all-gather selected" : "RCCL REJECTED", ms_since(t_start));
    verdict = buf;
    return KZG_OK;
}

// does this call of n blobs (one array, host or single-device resident) go through the shards?
static bool multi_takes(const KzgSettings* s, size_t n) { return s->multi && n >= 2 && n >= s->multi->min_blobs; }

// the chunk size of an array of n blobs dealt over D devices: about chunks_per_device chunks per device, none below
// min_chunk blobs, at most MULTI_MAX_PIECES pieces in all (with every shard holding the same number of slots)
static size_t multi_chunk_blobs(const MultiState* m, size_t n, size_t D) {
    size_t cs = std::max(m->min_chunk, (n + D * m->chunks_per_device - 1) / (D * m->chunks_per_device));
    const size_t max_chunks = std::max<size_t>(1, MULTI_MAX_PIECES / D) * D;
    cs = std::max(cs, (n + max_chunks - 1) / max_chunks);
    return cs;
}

// one array of n blobs (host memory, or the memory of ONE device) through the shards, in chunks dealt to the devices
// interleaved; the caller holds s->mu
static KzgRet multi_array_locked(bool* ok, const uint8_t* blobs, const uint8_t* commitments, const uint8_t* proofs, size_t n, bool host,
                                 const KzgSettings* s) {
    try {  // (host buffers of the call: 160 bytes per blob of transcript records; nothing may be thrown across the C ABI)
        const size_t D = shard_count(s);
        int src_dev = -1;
        if (!host) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, blobs) == hipSuccess && at.type == hipMemoryTypeDevice) src_dev = at.device;
            else {
                (void)hipGetLastError();
                src_dev = s->device;
            }
        }
        const size_t cs = multi_chunk_blobs(s->multi, n, D), nc = (n + cs - 1) / cs, per = (nc + D - 1) / D;
        KzgRet rc = multi_ensure_lanes(s, per);
        if (rc != KZG_OK) return rc;
        std::vector<Piece> pieces(nc);
        for (size_t c = 0; c < nc; c++) {
            Piece& pc = pieces[c];
            pc.k = c % D;
            pc.lane = c / D;
            pc.off = c * cs;
            pc.n = std::min(cs, n - pc.off);
            pc.blobs = blobs + pc.off * (size_t)BLOB_BYTES;
            pc.c = commitments + 48 * pc.off;
            pc.p = proofs + 48 * pc.off;
            pc.h = lane_of(shard_of(s, pc.k), pc.lane);
        }
        MultiTimes tm;
        rc = multi_pieces_locked(ok, pieces, n, host ? MultiSrc::Host : MultiSrc::OneDevice, src_dev, s, 0, tm);
        if (rc == KZG_OK) memcpy(s->multi_ms, tm.ms, sizeof tm.ms);
        return rc;
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

// which shard of the handle owns a device pointer: the shards on the device the memory lies on, taken in turn (`turn`
// counts per device, so that a list naming a device several times - a test rig - spreads the work over its logical shards)
static KzgRet multi_owner_shard(size_t* out, const void* ptr, const KzgSettings* s, std::vector<size_t>& turn) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(KZG_BADARGS, "not a device pointer");
    }
    const std::vector<int>& devs = s->multi->devices;
    std::vector<size_t> on;
    for (size_t k = 0; k < devs.size(); k++)
        if (devs[k] == at.device) on.push_back(k);
    if (on.empty()) return fail(KZG_BADARGS, "device memory on a device that is not in the handle's list");
    if (turn.size() < devs.size()) turn.resize(devs.size(), 0);
    *out = on[turn[on[0]]++ % on.size()];
    return KZG_OK;
}
static KzgRet multi_same_device(const void* a, const void* b, const void* c) {
    hipPointerAttribute_t x, y, z;
    if (hipPointerGetAttributes(&x, a) != hipSuccess || hipPointerGetAttributes(&y, b) != hipSuccess || hipPointerGetAttributes(&z, c) != hipSuccess) {
        (void)hipGetLastError();
        return fail(KZG_BADARGS, "not a device pointer");
    }
    if (x.device != y.device || x.device != z.device) return fail(KZG_BADARGS, "blobs, commitments and proofs of a group lie on different devices");
    return KZG_OK;
}

// one launch group of independent batches on the shard whose device holds it; the caller holds s->mu
static KzgRet multi_batches_device_locked(bool* ok_out, uint8_t* err_out, const void* d_blobs, const void* d_commitments, const void* d_proofs,
                                          size_t n, size_t n_batches, const KzgSettings* s) {
    KzgRet rc = multi_same_device(d_blobs, d_commitments, d_proofs);
    if (rc != KZG_OK) return rc;
    size_t k = 0;
    std::vector<size_t> turn;
    if ((rc = multi_owner_shard(&k, d_blobs, s, turn)) != KZG_OK) return rc;
    const KzgSettings* c = shard_of(s, k);
    HIPCHK(hipSetDevice(c->device));
    rc = batches_device_locked(ok_out, err_out, d_blobs, d_commitments, d_proofs, n, n_batches, c);
    const std::string msg = g_err;
    (void)hipSetDevice(s->device);
    if (rc != KZG_OK) g_err = msg;
    return rc;
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
    if (!s->multi) {
        size_t k = 0;
        while (n_local[k] == 0) k++;
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
        // (n == 1, the single-blob branch :482-489: the one piece runs phase 2 with r^0 = 1 on the device that holds the blob)
        MultiTimes tm;
        KzgRet rc = multi_ensure_lanes(s, 1);
        if (rc == KZG_OK) rc = multi_batch_resident(ok, in, s, 0, tm);
        if (rc == KZG_OK) memcpy(s->multi_ms, tm.ms, sizeof tm.ms);
        return rc;
    } catch (const std::bad_alloc&) {
        return fail(KZG_MALLOC, "host buffers of the sharded call");
    }
}

// A STREAM of sharded batches: n_batches batches, batch j's shard k = n_local[j n_shards + k] blobs at d_blobs[j n_shards + k]
// (etc.) on the handle's k-th device.  `in_flight` of them (0: the default, 4; at most 8) run at the same time on private
// lane sets, each driven by a host thread of its own: the serial transcript hash of one batch (:291-334; 42 MB = ~20 ms at
// 262 144 blobs, against ~8 ms of phase 1 per 32 768-blob shard) runs beside the device phases of the others, so a stream of
// config-5 batches is bound by the GPUs, not by one host core.  ok_out[j] / err_out[j] (optional) as in the many-batch
// forms: without err_out an invalid input in any batch fails the call.
extern "C" KzgRet kzg_verify_blob_kzg_proof_batch_sharded_stream(bool* ok_out, uint8_t* err_out, const void* const* d_blobs,
                                                                 const void* const* d_commitments, const void* const* d_proofs,
                                                                 const size_t* n_local, size_t n_shards, size_t n_batches, size_t in_flight,
                                                                 const KzgSettings* s) {
    if (!ok_out || !s || !n_local || !d_blobs || !d_commitments || !d_proofs) return fail(KZG_BADARGS, "null argument");
    if (!s->multi) return fail(KZG_BADARGS, "kzg_verify_blob_kzg_proof_batch_sharded_stream needs a handle over a device list");
    if (n_shards != shard_count(s)) return fail(KZG_BADARGS, "kzg_verify_blob_kzg_proof_batch_sharded_stream: one shard per device of the handle");
    if (n_batches == 0) return KZG_OK;
    for (size_t i = 0; i < n_batches * n_shards; i++)
        if (n_local[i] && (!d_blobs[i] || !d_commitments[i] || !d_proofs[i])) return fail(KZG_BADARGS, "null shard");
    std::lock_guard<std::mutex> lk(s->mu);
    const auto t_call = std::chrono::steady_clock::now();
    const size_t F = std::min(n_batches, std::max<size_t>(1, std::min<size_t>(in_flight ? in_flight : 4, 8)));
    KzgRet rc = multi_ensure_lanes(s, F);
    if (rc != KZG_OK) return rc;
    std::atomic<size_t> next{0};
    std::atomic<bool> stop{false};
    std::vector<KzgRet> rcs(F, KZG_OK);
    std::vector<std::string> msgs(F);
    std::vector<MultiTimes> sums(F);
    std::vector<size_t> counts(F, 0);
    std::vector<uint8_t> err_local;
    try {
        if (!err_out) err_local.assign(n_batches, 0);
    } catch (const std::bad_alloc&) {
        return fail(KZG_MALLOC, "per-batch error flags");
    }
    uint8_t* const err = err_out ? err_out : err_local.data();
    auto worker = [&](size_t L) {
        try {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= n_batches || stop.load()) return;
                std::vector<ShardIn> in(n_shards);
                size_t n = 0;
                for (size_t k = 0; k < n_shards; k++) {
                    const size_t i = j * n_shards + k;
                    in[k].blobs = (const uint8_t*)d_blobs[i];
                    in[k].c = (const uint8_t*)d_commitments[i];
                    in[k].p = (const uint8_t*)d_proofs[i];
                    in[k].n = n_local[i];
                    n += n_local[i];
                }
                err[j] = 0;
                if (n == 0) {  // src/kzg_proof.rs:478-480
                    ok_out[j] = true;
                    continue;
                }
                bool ok = false;
                MultiTimes tm;
                const KzgRet rc = multi_batch_resident(&ok, in, s, L, tm);
                if (rc == KZG_BADARGS) {  // the reference's Err for this batch; the stream goes on
                    err[j] = 1;
                    ok_out[j] = false;
                    continue;
                }
                if (rc != KZG_OK) {
                    rcs[L] = rc;
                    msgs[L] = g_err;
                    stop = true;
                    return;
                }
                ok_out[j] = ok;
                for (int i = 0; i < 8; i++) sums[L].ms[i] += tm.ms[i];
                counts[L]++;
            }
        } catch (const std::bad_alloc&) {
            rcs[L] = KZG_MALLOC;
            msgs[L] = "host buffers of the sharded call";
            stop = true;
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t L = 1; L < F; L++) {
            try {
                pool.emplace_back(worker, L);
            } catch (const std::system_error&) {  // fewer threads: fewer batches in flight
                break;
            }
        }
        worker(0);
        for (auto& th : pool) th.join();
    }
    (void)hipSetDevice(s->device);
    for (size_t L = 0; L < F; L++)
        if (rcs[L] != KZG_OK) return fail(rcs[L], msgs[L]);
    // per-batch stage averages in [1..6]; [0] = the wall clock of the whole stream; [7] = batches
    size_t done = 0;
    float avg[8] = {};
    for (size_t L = 0; L < F; L++) {
        done += counts[L];
        for (int i = 0; i < 8; i++) avg[i] += sums[L].ms[i];
    }
    for (int i = 1; i < 7; i++) s->multi_ms[i] = done ? avg[i] / (float)done : 0.f;
    s->multi_ms[0] = (float)ms_since(t_call);
    s->multi_ms[7] = (float)n_batches;
    if (!err_out)
        for (size_t j = 0; j < n_batches; j++)
            if (err[j]) return fail(KZG_BADARGS, "Failed to parse G1Affine from bytes");
    return KZG_OK;
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
    return KZG_OK;
}

// host wall-clock stages of the last sharded call on this handle, milliseconds (include/kzg_rs_amd.h)
extern "C" KzgRet kzg_multi_last_timings(const KzgSettings* s, float out_ms[8]) {
    if (!s || !out_ms) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_ms, s->multi_ms, sizeof(float) * 8);
    return KZG_OK;
}

// test hook: the batch challenge r (32 bytes, big-endian) the last sharded call on a multi-device handle hashed from its
// streamed transcript - compared with the oracle's compute_r (tests/test_gpu_multidevice.py)
extern "C" KzgRet kzg_debug_multi_last_r(uint8_t out[32], const KzgSettings* s) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->multi) {
        std::lock_guard<std::mutex> lk_r(s->multi->last_r_mu);
        reverse32(out, s->multi_last_r);
    } else reverse32(out, s->multi_last_r);
    return KZG_OK;
}
