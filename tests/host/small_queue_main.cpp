// small_queue_main.cpp - the small-call queue's submit loop (csrc/small_queue.hpp small_submit_core: leader / follower coalescing,
// futex words, requeue, linger) on the CPU with a stand-in for the GPU launch, built with -fsanitize=thread (and once more with
// address,undefined) by tests/test_small_queue_host.py.  Checks, under T concurrent callers of mixed kinds:
//   * every caller gets exactly the results of ITS OWN request (computed from its own bytes), or - for a launch that failed -
//     the launch's error code and message; nobody else's;
//   * nothing hangs (a watchdog aborts after 120 s: a lost wake-up would leave a caller asleep for ever);
//   * requests really travel together (fewer launches than requests), never beyond a launch's capacity, never two launches on
//     one lane at a time, never more lanes than allowed;
//   * the blob callers' challenge hashes (the persistent pool of host_only.hpp) are complete when the launch reads them.
// argv[4] > 0: every fourth pass of a caller through the submit loop stalls that many microseconds between reading its request's
// lane and reading the futex word it is about to sleep on (SMALL_QUEUE_TEST_HOOK_BETWEEN_LOADS) - the window in which a leader can
// take the request and move the queue's sleepers; a caller that then slept on the queue's word would never be woken once traffic
// stops (round-5 advisor finding).  argv[5]: watchdog seconds.  argv[6] = 1: the callers meet at a barrier before every call,
// so the queue goes IDLE after every burst of T calls and a caller left asleep on the wrong word stays asleep (continuous
// traffic would wake it by accident).  Without the re-read of r.lane in small_submit_core this mode hangs within a few bursts.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <random>
#include <thread>
#include <vector>

static std::atomic<long> hook_delay_us{0};
static std::atomic<unsigned> hook_calls{0};
static void hook_between_loads();
#define SMALL_QUEUE_TEST_HOOK_BETWEEN_LOADS() hook_between_loads()
#define KZG_HOST_FE_PER_BLOB 64  // small "blobs" (2 KiB): the hashing pool's code paths without 128 KiB per request
#include "small_queue.hpp"

static void hook_between_loads() {
    const long d = hook_delay_us.load(std::memory_order_relaxed);
    if (d > 0 && (hook_calls.fetch_add(1, std::memory_order_relaxed) & 3) == 0) std::this_thread::sleep_for(std::chrono::microseconds(d));
}
static std::atomic<int> failures{0};
#define CHECK(x)                                                      \
    do {                                                              \
        if (!(x)) {                                                   \
            failures++;                                               \
            fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); \
        }                                                             \
    } while (0)

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 64, CALLS = argc > 2 ? atoi(argv[2]) : 200, LANES = argc > 3 ? atoi(argv[3]) : 2;
    hook_delay_us = argc > 4 ? atol(argv[4]) : 0;
    const int WATCHDOG_S = argc > 5 ? atoi(argv[5]) : 120;
    const bool BURSTS = argc > 6 && atoi(argv[6]) != 0;
    std::atomic<unsigned> bar_count{0}, bar_gen{0};
    auto barrier = [&] {
        const unsigned g = bar_gen.load(std::memory_order_acquire);
        if (bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)T) {
            bar_count.store(0, std::memory_order_relaxed);
            bar_gen.fetch_add(1, std::memory_order_release);
        } else {
            while (bar_gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
        }
    };
    const size_t BLOB = (size_t)32 * KZG_HOST_FE_PER_BLOB;
    SmallQueue Q;
    Q.max_lanes = (size_t)LANES;
    Q.cap_proofs = 48;  // small capacities: launches fill up and leave requests behind (the `more` path)
    Q.cap_blobs = 12;
    std::atomic<int> in_launch[SMALL_LANES_MAX];
    for (auto& x : in_launch) x = 0;
    std::atomic<uint64_t> launches{0}, carried{0}, failed_launches{0};
    std::atomic<bool> finished{false};
    std::thread watchdog([&] {
        for (int i = 0; i < 10 * WATCHDOG_S && !finished; i++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!finished) {
            fprintf(stderr, "WATCHDOG: callers still waiting after %d s - a lost wake-up\n", WATCHDOG_S);
            abort();
        }
    });
    // the stand-in launch: results are functions of the request's OWN bytes
    auto run = [&](int li, SmallLane& L, std::vector<SmallReq*>& batch, size_t m, SmallReq::Kind kind, std::string& msg) -> KzgRet {
        CHECK(li >= 0 && li < LANES);
        CHECK(in_launch[li].fetch_add(1) == 0);  // one launch per lane at a time
        (void)L;
        size_t items = 0;
        for (SmallReq* x : batch) {
            CHECK(x->kind == kind);
            items += x->n;
        }
        CHECK(items == m && m <= (kind == SmallReq::PROOFS ? Q.cap_proofs : Q.cap_blobs));
        const uint64_t nth = launches.fetch_add(1);
        carried += batch.size();
        std::this_thread::sleep_for(std::chrono::microseconds(150 + 2 * m));  // "1.7 ms whether it carries 1 or 200"
        KzgRet rc = KZG_OK;
        if (nth % 97 == 13) {  // a launch that fails: every request of it carries the error
            rc = KZG_ERROR;
            msg = "injected failure";
            failed_launches++;
        } else {
            for (SmallReq* x : batch) {
                if (kind == SmallReq::PROOFS) {
                    for (size_t i = 0; i < x->n; i++) {
                        x->ok[i] = ((x->c[48 * i] + x->z[32 * i]) & 1) != 0;
                        x->err[i] = x->p[48 * i] == 0xff;
                        x->general[i] = x->y[32 * i] == 0x7e;
                    }
                } else {
                    hostpool::finish(*x->hash);  // the challenges this launch needs
                    uint8_t acc = 0;
                    for (size_t i = 0; i < x->n; i++) acc ^= x->hash->z_le[32 * i];
                    x->ok[0] = (acc & 1) != 0;
                    x->err[0] = x->blobs[0] == 0xee;
                    x->general[0] = 0;
                }
            }
        }
        for (SmallReq* x : batch)
            if (x->hash) hostpool::finish(*x->hash);
        CHECK(in_launch[li].fetch_sub(1) == 1);
        return rc;
    };
    std::atomic<uint64_t> done_calls{0}, error_calls{0};
    auto caller = [&](int t) {
        std::mt19937_64 rng(1234 + t);
        for (int k = 0; k < CALLS; k++) {
            if (BURSTS) barrier();
            const bool blobs = t % 8 == 0;
            const size_t n = blobs ? 1 + rng() % 4 : 1 + rng() % 5;
            std::vector<uint8_t> c(48 * n), p(48 * n), z(32 * n), y(32 * n), bl(blobs ? BLOB * n : 0), zle(32 * n);
            for (auto* v : {&c, &p, &z, &y, &bl})
                for (auto& b : *v) b = (uint8_t)rng();
            std::vector<uint8_t> ok(n, 2), err(n, 2), gen(n, 2);
            SmallReq r;
            r.kind = blobs ? SmallReq::BLOBS : SmallReq::PROOFS;
            r.n = n;
            r.c = c.data();
            r.p = p.data();
            r.z = z.data();
            r.y = y.data();
            r.ok = reinterpret_cast<bool*>(ok.data());
            r.err = err.data();
            r.general = gen.data();
            if (blobs) {
                r.blobs = bl.data();
                r.hash = hostpool::make(zle.data(), bl.data(), c.data(), n);
                hostpool::post(r.hash, 4);
            }
            const KzgRet rc = small_submit_core(Q, r, run);
            if (blobs) hostpool::finish(*r.hash);
            if (rc != KZG_OK) {
                CHECK(rc == KZG_ERROR && strcmp(r.msg, "injected failure") == 0);
                error_calls++;
            } else if (blobs) {
                uint8_t acc = 0;
                for (size_t i = 0; i < n; i++) {
                    uint8_t want[32];
                    host_blob_challenge(want, bl.data() + BLOB * i, c.data() + 48 * i);
                    CHECK(memcmp(want, zle.data() + 32 * i, 32) == 0);
                    acc ^= want[0];
                }
                CHECK(ok[0] == (acc & 1) && err[0] == (bl[0] == 0xee) && gen[0] == 0);
            } else {
                for (size_t i = 0; i < n; i++) {
                    CHECK(ok[i] == ((c[48 * i] + z[32 * i]) & 1));
                    CHECK(err[i] == (p[48 * i] == 0xff));
                    CHECK(gen[i] == (y[32 * i] == 0x7e));
                }
            }
            done_calls++;
            if (!BURSTS && (rng() & 7) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));  // (not a pure closed loop)
        }
    };
    std::vector<std::thread> ths;
    for (int t = 0; t < T; t++) ths.emplace_back(caller, t);
    for (auto& th : ths) th.join();
    finished = true;
    watchdog.join();
    CHECK(done_calls.load() == (uint64_t)T * CALLS);
    CHECK(Q.q.empty());
    CHECK(Q.n_lanes >= 1 && Q.n_lanes <= (size_t)LANES);
    for (size_t i = 0; i < Q.n_lanes; i++) CHECK(!Q.lanes[i]->busy);
    CHECK(carried.load() == done_calls.load());                 // every request was carried by exactly one launch
    if (T >= 8) CHECK(launches.load() < done_calls.load());     // ... and they travelled together
    CHECK(Q.launches == launches.load() && Q.requests == done_calls.load());
    printf("threads %d calls %llu launches %llu (%.1f requests per launch, largest %llu items) failed launches %llu -> %llu calls saw the error; failures %d\n", T,
           (unsigned long long)done_calls.load(), (unsigned long long)launches.load(), (double)carried.load() / (double)launches.load(),
           (unsigned long long)Q.max_items, (unsigned long long)failed_launches.load(), (unsigned long long)error_calls.load(), failures.load());
    for (size_t i = 0; i < Q.n_lanes; i++) delete Q.lanes[i];
    return failures ? 1 : 0;
}
