// small_queue.hpp - the small-call queue of a settings handle: requests, lanes, waiting and waking, and the submit loop with the
// launch itself left to the caller (capi_coalesce.hpp supplies the GPU launch).  Plain C++17 + Linux futexes, no HIP: part of the
// translation unit kzg_capi.hip, and compiled on its own with g++ -fsanitize=thread by tests/test_small_queue_host.py
// (tests/host/small_queue_main.cpp: hundreds of threads, a stand-in launch, every result checked, no lost wake-up, no race).
#pragma once
#include <linux/futex.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_rs_amd.h"
#include "host_only.hpp"

// One request = the small call of one host thread: n (commitment, z, y, proof) tuples, or n host blobs with their commitments
// and proofs; every item gets its own pairing and the request gets its own results - what the entry point makes of them
// (one verdict, a conjunction, a verdict per item) is the submitter's business.
struct SmallReq {
    enum Kind { PROOFS = 0, BLOBS = 1 };
    Kind kind = PROOFS;
    size_t n = 0;
    const uint8_t *c = nullptr, *p = nullptr;  // n x 48 bytes each
    const uint8_t *z = nullptr, *y = nullptr;  // PROOFS: n x 32 big-endian bytes each
    const uint8_t* blobs = nullptr;            // BLOBS: n x 131072 bytes
    hostpool::JobRef hash;                     // BLOBS: the challenges (the submitter's buffer behind hash->z_le), claimed blob by blob by whoever has time
    // results.  PROOFS: per item.  BLOBS: [0] only - the conjunction over the request's blobs, any parse failure among them, any z = tau
    bool* ok = nullptr;
    uint8_t *err = nullptr, *general = nullptr;
    KzgRet rc = KZG_OK;  // a failure of the launch that carried the request (every request of that launch gets it)
    char msg[192] = {0};  // (a fixed buffer: completing a request must not allocate - a leader that threw half-way through would leave callers asleep)
    std::atomic<bool> taken{false}, done{false};  // taken: written under the queue's lock; done: the leader's LAST access to the request
    std::atomic<int> lane{-1};                    // the lane whose launch carries the request (its owner then sleeps on that lane's word)
};
constexpr size_t SMALL_LANES_MAX = 16;
struct SmallLane {
    KzgSettings* h = nullptr;  // a private lane (settings_lane) on one device of the handle
    bool busy = false;
    std::atomic<uint32_t> word{0};  // the futex word the callers of this lane's launch sleep on
    std::vector<uint8_t> c, z, y, p, okerr;  // the gathered tuples of a launch
};
// The queue's lock: held for a push, a look at the lanes, the taking of a batch - tens of nanoseconds to a few microseconds - by up
// to hundreds of threads that arrive within ~100 us of each other when a launch completes.  A pthread mutex sends every
// contended acquirer through futex_wait / futex_wake (a context switch each); this one spins on the flag (test-and-test-and-set,
// `pause`), yields the core after a while, and never sleeps in the kernel.  Lockable: works with std::unique_lock / lock_guard.
struct SmallSpinLock {
    std::atomic<bool> held{false};
    void lock() {
        for (unsigned spins = 0;; spins++) {
            if (!held.load(std::memory_order_relaxed) && !held.exchange(true, std::memory_order_acquire)) return;
            if (spins < 256) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            } else {
                std::this_thread::yield();
            }
        }
    }
    bool try_lock() { return !held.load(std::memory_order_relaxed) && !held.exchange(true, std::memory_order_acquire); }
    void unlock() { held.store(false, std::memory_order_release); }
};
struct SmallQueue {
    SmallSpinLock mu;
    std::deque<SmallReq*> q;        // waiting requests, oldest first
    SmallLane* lanes[SMALL_LANES_MAX] = {};  // made on demand, up to max_lanes (a slot, once set, never changes: read without the lock by who knows its index)
    size_t n_lanes = 0;
    size_t max_lanes = 2;
    bool lane_two_streams = false;  // option small_streams=2: chain C of the one-proof path behind chain B, two streams per lane (A/B measurement)
    // The lanes' streams are made at the device's highest priority (option small_priority=0: normal).  Not for the priority
    // itself: the HIP runtime keeps a separate pool of hardware queues per priority, so the lanes' streams do not share queues
    // with each other's or with the launch-group pipeline's normal-priority streams.  Streams that share a hardware queue run
    // one behind the other: with normal-priority lanes and GPU_MAX_HW_QUEUES=8, two threads calling verify_kzg_proof at once
    // took 2.9 ms each instead of 1.7 (1.84 with 16 queues, 1.75 with priority lanes: profiles/r5_small_call_queues.txt).
    int lane_priority = 1;
    long linger_us = 250, linger_gap_us = 40;  // options small_linger_us / small_linger_gap_us (capi_coalesce.hpp small_submit); 0: never wait
    std::atomic<uint32_t> epoch{0};    // the futex word the callers whose request is still in the queue sleep on
    std::atomic<uint64_t> arrivals{0};
    uint64_t last_done_us = 0;         // when the last launch finished, and how many calls it carried
    size_t last_done_items = 0;
    uint64_t launches = 0, requests = 0, items = 0, max_items = 0;  // since the last kzg_debug_small_queue_stats(reset)
    size_t cap_proofs = 1024, cap_blobs = 256;  // items per launch
};


// Waiting and waking.  Waiters sleep on 32-bit futex words - the queue's while their request is still in the queue, their
// lane's once a launch carries it - and a state change bumps the word and wakes its sleepers with one system call; a request's
// `done` flag is read without the lock.  (The first form had a condition variable per request, and a leader woke its 255
// followers one system call at a time under the queue's lock: 0.5-1 ms of the 2 ms a launch takes.)
static void small_sleep(std::atomic<uint32_t>& word, uint32_t seen) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word), FUTEX_WAIT_PRIVATE, (unsigned long)seen, nullptr, nullptr, 0UL);
}
static void small_wake(std::atomic<uint32_t>& word, int how_many) {
    word.fetch_add(1, std::memory_order_release);
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word), FUTEX_WAKE_PRIVATE, (unsigned long)how_many, nullptr, nullptr, 0UL);
}
static void small_wake_all(std::atomic<uint32_t>& word) { small_wake(word, 0x7fffffff); }
// Everybody asleep on `from` goes to sleep on `to` instead, without waking (FUTEX_CMP_REQUEUE): the callers a leader has just
// taken into its launch move from the queue's word to the lane's, so that the launch's completion wakes exactly them - with one
// word for everybody every completion woke the callers of the OTHER launch in flight as well, 256 threads of which half went
// straight back to sleep (10 host cores at 256 threads; the cgroups of this project's boxes give a job 16).  `from` is bumped
// first: a caller that was about to sleep on it (it read the old value) does not, and finds its lane on the way round.
// Returns false when the word moved under us (then the caller wakes everybody instead: they sort themselves out).
static bool small_requeue_all(std::atomic<uint32_t>& from, std::atomic<uint32_t>& to) {
    const uint32_t now = from.fetch_add(1, std::memory_order_acq_rel) + 1;
    // (uaddr, op, nr_wake = 0, nr_requeue in the timeout slot, uaddr2, the value uaddr must still hold)
    const long rc = syscall(SYS_futex, reinterpret_cast<uint32_t*>(&from), FUTEX_CMP_REQUEUE_PRIVATE, 0UL, (unsigned long)0x7fffffff, reinterpret_cast<uint32_t*>(&to),
                            (unsigned long)now);
    return rc >= 0;
}
static uint64_t small_now_us() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// a lane for a new leader: a free one, or a new one while the handle has fewer than max_lanes (the slot is taken under the
// lock, the lane itself - two streams, a dozen events - is made by the leader outside it); -1: all busy
static int small_take_lane(SmallQueue& Q) {
    for (size_t i = 0; i < Q.n_lanes; i++)
        if (!Q.lanes[i]->busy) {
            Q.lanes[i]->busy = true;
            return (int)i;
        }
    if (Q.n_lanes < Q.max_lanes && Q.n_lanes < SMALL_LANES_MAX) {
        SmallLane* L = new (std::nothrow) SmallLane();  // (no memory for a lane: as if every lane were busy)
        if (!L) return -1;
        L->busy = true;
        Q.lanes[Q.n_lanes] = L;
        return (int)Q.n_lanes++;
    }
    return -1;
}

// (tests/host/small_queue_main.cpp defines this to stall a caller between its read of r.lane and its read of the word it will
// sleep on - the window in which a leader can take the request)
#ifndef SMALL_QUEUE_TEST_HOOK_BETWEEN_LOADS
#define SMALL_QUEUE_TEST_HOOK_BETWEEN_LOADS() ((void)0)
#endif

// Submit a request and return when it is done (r.rc, r.msg; the per-item results where the request points).  The calling thread
// may lead launches meanwhile - its own request's, or one that only carries older requests.
//   run(lane index, lane, batch, items, kind, msg) -> KzgRet : the launch itself, called WITHOUT the queue's lock by the leader that
//   owns the lane: it writes every request's per-item results (or returns a failure that every request of the batch then carries)
//   and leaves nothing in flight.  The GPU form is capi_coalesce.hpp's; tests/host/small_queue_main.cpp runs this very function
//   with a stand-in under ThreadSanitizer (tests/test_small_queue_host.py).
template <class Run>
static KzgRet small_submit_core(SmallQueue& Q, SmallReq& r, Run&& run) {
    bool queued = false;
    for (;;) {
        const int my_lane = r.lane.load(std::memory_order_acquire);
        std::atomic<uint32_t>& word = my_lane >= 0 ? Q.lanes[my_lane]->word : Q.epoch;  // where this caller sleeps: its launch's lane, or the queue
        SMALL_QUEUE_TEST_HOOK_BETWEEN_LOADS();
        const uint32_t seen = word.load(std::memory_order_acquire);
        if (r.done.load(std::memory_order_acquire)) break;
        // A leader may have taken the request between the two loads above: it stores r.lane, THEN bumps the queue's word and
        // moves that word's sleepers to its lane's.  A `seen` read after that bump would let this caller sleep on the queue's
        // word at its current value while the launch's completion only wakes the lane's word - asleep until unrelated traffic
        // happens by, for ever on a handle that goes idle.  The lane store is ordered before the bump, so a post-bump `seen`
        // implies the new lane is visible here: go round and sleep on the right word.
        if (r.lane.load(std::memory_order_acquire) != my_lane) continue;
        int li = -1;
        std::unique_lock<SmallSpinLock> lk(Q.mu, std::defer_lock);
        // (ONE visit to the queue's lock for a caller that ends up a follower: it queues its request and looks for a lane in the
        // same critical section, and once its request has been taken it never touches the lock again - 256 threads on one
        // mutex, two visits per call, cost more host time than everything else in the call)
        if (!r.taken.load(std::memory_order_relaxed)) {
            lk.lock();
            if (!queued) {
                Q.q.push_back(&r);
                Q.requests++;
                queued = true;
                Q.arrivals.fetch_add(1, std::memory_order_relaxed);
            }
            if (!r.taken.load(std::memory_order_relaxed)) li = small_take_lane(Q);
            if (li < 0) lk.unlock();
        }
        if (li < 0) {
            if (r.hash && r.hash->unclaimed()) hostpool::help(*r.hash);  // nothing to lead: hash the own blobs instead of sleeping
            else small_sleep(word, seen);
            continue;
        }
        // ---- leader (holds the lock and lane li)
        SmallLane& L = *Q.lanes[(size_t)li];
        // The callers of a launch that has just finished come back within ~100 us of each other.  A leader that takes the lane
        // the moment it is free would leave with the first of them and the rest would wait a whole launch for the next lane;
        // so while requests keep arriving (no gap of linger_gap_us) it waits, linger_us at most - but only right after a launch
        // that carried several calls: a handle with one caller at a time never waits.  Measured, 64 threads of verify_kzg_proof:
        // 6-16 k calls/s without (launches of 6-12), 27 k with a fixed 150 us.
        if (Q.linger_us > 0 && Q.last_done_items >= 2 && small_now_us() - Q.last_done_us < 400) {
            lk.unlock();
            const uint64_t t0 = small_now_us();
            uint64_t last_change = t0, seen_arrivals = Q.arrivals.load(std::memory_order_relaxed);
            for (;;) {
                std::this_thread::yield();
                const uint64_t now = small_now_us(), a = Q.arrivals.load(std::memory_order_relaxed);
                if (a != seen_arrivals) {
                    seen_arrivals = a;
                    last_change = now;
                }
                if (now - last_change >= (uint64_t)Q.linger_gap_us || now - t0 >= (uint64_t)Q.linger_us) break;
            }
            lk.lock();
        }
        // everything queued of the oldest request's kind, in order, up to the launch's capacity (our own request may have left
        // with another leader meanwhile: then this launch only carries others)
        std::vector<SmallReq*> batch;
        size_t m = 0;
        SmallReq::Kind kind = SmallReq::PROOFS;
        if (!Q.q.empty()) {
            try {
                batch.reserve(Q.q.size());  // (nothing below may throw between taking requests off the queue and completing them)
            } catch (const std::bad_alloc&) {  // no memory for the list: the lane goes back, this call fails, the queue is as it was
                L.busy = false;
                for (auto it = Q.q.begin(); it != Q.q.end(); ++it)
                    if (*it == &r) {
                        Q.q.erase(it);
                        break;
                    }
                const bool was_taken = r.taken.load(std::memory_order_relaxed);
                lk.unlock();
                small_wake(Q.epoch, 1);
                if (!was_taken) {
                    r.rc = KZG_MALLOC;
                    snprintf(r.msg, sizeof r.msg, "%s", "host buffers of the launch");
                    return r.rc;
                }
                continue;  // (another leader carries our request: wait for it)
            }
            kind = Q.q.front()->kind;
            const size_t cap = kind == SmallReq::PROOFS ? Q.cap_proofs : Q.cap_blobs;
            for (auto it = Q.q.begin(); it != Q.q.end();) {
                SmallReq* x = *it;
                if (x->kind == kind && m + x->n <= cap) {
                    x->lane.store(li, std::memory_order_release);
                    x->taken.store(true, std::memory_order_relaxed);
                    m += x->n;
                    batch.push_back(x);
                    it = Q.q.erase(it);
                    if (m == cap) break;
                } else ++it;
            }
        }
        if (batch.empty()) {  // (everything left while this thread lingered)
            L.busy = false;
            lk.unlock();
            continue;
        }
        Q.launches++;
        Q.items += m;
        Q.max_items = std::max<uint64_t>(Q.max_items, m);
        const bool more = !Q.q.empty();
        // the callers of this launch: from the queue's word to the lane's (everybody asleep on the queue's word is in the launch
        // when the queue is empty now; with requests of the other kind, or beyond the launch's capacity, left in the queue -
        // another lane may be free for them - everybody is woken and finds its place)
        lk.unlock();
        if (more || !small_requeue_all(Q.epoch, L.word)) small_wake_all(Q.epoch);  // (outside the lock: moving 200 sleepers takes the kernel a while)
        KzgRet rc = KZG_OK;
        char msg_buf[sizeof r.msg] = {0};
        try {
            std::string msg;
            rc = run(li, L, batch, m, kind, msg);
            snprintf(msg_buf, sizeof msg_buf, "%s", msg.c_str());
        } catch (...) {  // (whatever the launch threw: its callers are completed, with an error)
            rc = KZG_MALLOC;
            snprintf(msg_buf, sizeof msg_buf, "%s", "host buffers of the launch");
        }
        lk.lock();
        L.busy = false;
        Q.last_done_us = small_now_us();
        Q.last_done_items = batch.size();
        const bool waiting = !Q.q.empty();
        lk.unlock();
        for (SmallReq* x : batch) {
            x->rc = rc;
            if (rc != KZG_OK) memcpy(x->msg, msg_buf, sizeof x->msg);
            x->done.store(true, std::memory_order_release);  // (the owner may return, and its request die, from here on)
        }
        small_wake_all(L.word);                // the launch's callers
        if (waiting) small_wake(Q.epoch, 1);   // the lane is free again: ONE of the callers still in the queue leads (it takes the others along)
    }
    return r.rc;
}

