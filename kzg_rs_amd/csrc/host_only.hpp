// host_only.hpp - the library's host-side data handling that touches no HIP API: SHA-256 (portable + SHA-NI), helpers on
// 32-byte big-endian scalars, the trusted-setup text parser (build.rs:23-105 of the reference) and the batch-transcript hash
// (src/kzg_proof.rs:291-348).  Part of the translation unit kzg_capi.hip, and - being plain C++17 - also compiled on its
// own with g++ -fsanitize=address,undefined by tests/test_host_only_sanitized.py (tests/host/host_only_main.cpp), where the
// parser is fuzzed with truncated files, bad hex and wrong counts.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#ifndef KZG_HOST_FE_PER_BLOB
#define KZG_HOST_FE_PER_BLOB 4096
#endif

// SHA-256 (FIPS 180-4) for the batch transcript - host code, independent of the device kernel
namespace hostsha {
alignas(16) static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void block(uint32_t st[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace hostsha
#include <immintrin.h>
namespace hostsha {
// x86 SHA extensions (runtime-detected): the batch transcript is one serial chain, so per-block latency is
// what matters; sha256rnds2 does it at ~1.5 GB/s per core.
__attribute__((target("sha,sse4.1,ssse3"))) static void blocks_ni(uint32_t st[8], const uint8_t* data, size_t nblocks) {
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i TMP = _mm_loadu_si128((const __m128i*)&st[0]);
    __m128i STATE1 = _mm_loadu_si128((const __m128i*)&st[4]);
    TMP = _mm_shuffle_epi32(TMP, 0xB1);
    STATE1 = _mm_shuffle_epi32(STATE1, 0x1B);
    __m128i STATE0 = _mm_alignr_epi8(TMP, STATE1, 8);
    STATE1 = _mm_blend_epi16(STATE1, TMP, 0xF0);
    while (nblocks--) {
        const __m128i ABEF = STATE0, CDGH = STATE1;
        __m128i M[4];
        for (int g = 0; g < 16; g++) {
            if (g < 4) M[g] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * g)), MASK);
            __m128i msg = _mm_add_epi32(M[g & 3], _mm_loadu_si128((const __m128i*)&K[4 * g]));
            STATE1 = _mm_sha256rnds2_epu32(STATE1, STATE0, msg);
            if (g >= 3 && g < 15) {
                __m128i t = _mm_alignr_epi8(M[g & 3], M[(g - 1) & 3], 4);
                M[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(M[(g + 1) & 3], t), M[g & 3]);
            }
            msg = _mm_shuffle_epi32(msg, 0x0E);
            STATE0 = _mm_sha256rnds2_epu32(STATE0, STATE1, msg);
            if (g >= 1 && g < 13) M[(g - 1) & 3] = _mm_sha256msg1_epu32(M[(g - 1) & 3], M[g & 3]);
        }
        STATE0 = _mm_add_epi32(STATE0, ABEF);
        STATE1 = _mm_add_epi32(STATE1, CDGH);
        data += 64;
    }
    TMP = _mm_shuffle_epi32(STATE0, 0x1B);
    STATE1 = _mm_shuffle_epi32(STATE1, 0xB1);
    STATE0 = _mm_blend_epi16(TMP, STATE1, 0xF0);
    STATE1 = _mm_alignr_epi8(STATE1, TMP, 8);
    _mm_storeu_si128((__m128i*)&st[0], STATE0);
    _mm_storeu_si128((__m128i*)&st[4], STATE1);
}
static bool have_ni() {
    static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return v;
}
#else
static bool have_ni() { return false; }
static void blocks_ni(uint32_t*, const uint8_t*, size_t) {}
#endif
// The hash as a STREAM: update() takes the message in pieces of any length as they become available (the transcript of a
// sharded batch is hashed while the shards still produce records, capi_multi.hpp), finish() pads and writes the digest.
struct Stream {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64] = {0};  // the bytes of an incomplete block
    size_t fill = 0;
    uint64_t total = 0;
    void blocks(const uint8_t* p, size_t nb) {
        if (!nb) return;
        if (have_ni()) blocks_ni(st, p, nb);
        else
            for (size_t i = 0; i < nb; i++) block(st, p + 64 * i);
    }
    void update(const uint8_t* data, size_t len) {
        total += len;
        if (fill) {
            const size_t take = std::min(len, (size_t)64 - fill);
            memcpy(buf + fill, data, take);
            fill += take;
            data += take;
            len -= take;
            if (fill < 64) return;
            blocks(buf, 1);
            fill = 0;
        }
        const size_t full = len / 64;
        blocks(data, full);
        fill = len - 64 * full;
        if (fill) memcpy(buf, data + 64 * full, fill);
    }
    void finish(uint8_t out[32]) {
        uint8_t tail[128] = {0};
        memcpy(tail, buf, fill);
        tail[fill] = 0x80;
        const size_t tl = fill + 9 <= 64 ? 64 : 128;
        const uint64_t bits = total * 8;
        for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
        block(st, tail);
        if (tl == 128) block(st, tail + 64);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
            out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
        }
    }
};
static void digest(uint8_t out[32], const uint8_t* data, size_t len) {
    Stream s;
    s.update(data, len);
    s.finish(out);
}
}  // namespace hostsha

// r (big-endian) and helpers on 32-byte big-endian integers
static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                 0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
static bool be_geq_r(const uint8_t v[32]) { return memcmp(v, R_BE, 32) >= 0; }
static void be_sub_r(uint8_t v[32]) {
    int borrow = 0;
    for (int i = 31; i >= 0; i--) {
        int d = (int)v[i] - R_BE[i] - borrow;
        borrow = d < 0;
        v[i] = (uint8_t)(d + (borrow << 8));
    }
}
static void reverse32(uint8_t* dst, const uint8_t* src) {
    for (int i = 0; i < 32; i++) dst[i] = src[31 - i];
}


// ---------------------------------------------------------------- trusted-setup text (build.rs:23-56)
namespace hostparse {
static inline int hexnib(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
// "<n_g1>\n<n_g2>\n" + n_g1 lines of 96 hex characters + n_g2 lines of 192 hex characters (\r\n tolerated).
// g1b: n1 x 48 bytes stored bit-reversal permuted (build.rs:79,89-105: file line i -> slot brp(i)); g2b: n2 x 96 bytes in
// file order; first: g1 points 0 and 1 of the FILE order (the monomial-form check of build.rs:107-129).
// Returns false with a message for anything the reference's build script would reject (InvalidTrustedSetup /
// InvalidHexFormat): missing header, n_g1 != 4096, n_g2 < 2, truncated file, a line of the wrong length, a non-hex digit.
static bool trusted_setup_text(const char* txt, size_t len, std::vector<uint8_t>& g1b, std::vector<uint8_t>& g2b, uint8_t (&first)[2][48], long& n1,
                               long& n2, std::string& err) {
    std::vector<std::pair<const char*, size_t>> lines;
    const char *p = txt, *end = txt + len;
    while (p < end) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!q) q = end;
        size_t l = (size_t)(q - p);
        if (l && p[l - 1] == '\r') l--;
        lines.emplace_back(p, l);
        p = q + 1;
    }
    if (lines.size() < 2) return err = "trusted setup: missing header lines", false;
    auto number = [](const std::pair<const char*, size_t>& ln, long& v) {  // decimal digits only, at most 9 of them
        if (ln.second == 0 || ln.second > 9) return false;
        v = 0;
        for (size_t i = 0; i < ln.second; i++) {
            if (ln.first[i] < '0' || ln.first[i] > '9') return false;
            v = v * 10 + (ln.first[i] - '0');
        }
        return true;
    };
    if (!number(lines[0], n1) || !number(lines[1], n2)) return err = "trusted setup: bad header line", false;
    if (n1 != KZG_HOST_FE_PER_BLOB) return err = "trusted setup: expected 4096 G1 points", false;
    if (n2 < 2 || n2 > 65536 || (long)lines.size() < 2 + n1 + n2) return err = "trusted setup: truncated file", false;
    // hex -> bytes for every point line (hex_to_bytes, build.rs:15-21: KzgError::InvalidHexFormat)
    auto unhex = [&](uint8_t* dst, const std::pair<const char*, size_t>& ln, size_t nbytes) {
        if (ln.second != 2 * nbytes) return false;
        for (size_t i = 0; i < nbytes; i++) {
            int a = hexnib((unsigned char)ln.first[2 * i]), b = hexnib((unsigned char)ln.first[2 * i + 1]);
            if (a < 0 || b < 0) return false;
            dst[i] = (uint8_t)(a << 4 | b);
        }
        return true;
    };
    g1b.assign(48 * (size_t)n1, 0);
    g2b.assign(96 * (size_t)n2, 0);
    for (long i = 0; i < n1; i++) {
        uint8_t tmp[48];
        if (!unhex(tmp, lines[2 + i], 48)) return err = "trusted setup: bad G1 line", false;
        if (i < 2) memcpy(first[i], tmp, 48);
        uint32_t r = 0;
        for (int k = 0; k < 12; k++) r |= ((uint32_t)(i >> k) & 1u) << (11 - k);
        memcpy(g1b.data() + 48 * (size_t)r, tmp, 48);
    }
    for (long i = 0; i < n2; i++)
        if (!unhex(g2b.data() + 96 * (size_t)i, lines[2 + n1 + i], 96)) return err = "trusted setup: bad G2 line", false;
    return true;
}
}  // namespace hostparse

// ---------------------------------------------------------------- the batch transcript hash
// compute_r_powers' hash (src/kzg_proof.rs:291-348) for B batches: r_b = SHA-256(domain || degree || n_total || records of batch b) mod r,
// written as 32 little-endian bytes (= Scalar::to_bytes(), the device limb layout) to r_out + 32 b.  Pure host code (SHA-NI):
// one serial chain of 160 n_total + 32 bytes per batch - hopeless on a GPU lane, ~2 GB/s on a CPU core - and the batches are
// independent, so they are spread over a few host threads (option host_threads, default 16).  Record layouts:
//   world == 0 : [B][n_total]       every batch's records in global blob order
//   world  > 0 : [world][B][n]      as an all-gather / all-to-all of equal shards leaves them (n_total = world n)
// Returns false when a transcript buffer could not be allocated (nothing is thrown, in a worker thread or out of it).
// One batch transcript as a stream: the 32-byte header of :298-311 at construction, then the 160-byte records in global blob
// order, in pieces of any number of records, as they arrive; r() closes the hash: digest mod r as 32 little-endian bytes.
struct BatchTranscript {
    hostsha::Stream sha;
    explicit BatchTranscript(size_t n_total) {
        uint8_t h[32];
        memcpy(h, "RCKZGBATCH___V1_", 16);
        memset(h + 16, 0, 16);
        h[22] = (uint8_t)(KZG_HOST_FE_PER_BLOB >> 8);
        h[23] = (uint8_t)(KZG_HOST_FE_PER_BLOB & 0xff);
        for (int k = 0; k < 8; k++) h[24 + k] = (uint8_t)((uint64_t)n_total >> (56 - 8 * k));
        sha.update(h, 32);
    }
    void records(const uint8_t* rec, size_t count) { sha.update(rec, 160 * count); }
    void r(uint8_t r_le[32]) {
        uint8_t dg[32];
        sha.finish(dg);
        while (be_geq_r(dg)) be_sub_r(dg);  // digest mod r: at most two subtractions (2^256 < 3r)
        reverse32(r_le, dg);
    }
};
// compute_challenge (src/kzg_proof.rs:46-72) on the HOST for blobs that lie in host memory: z = SHA-256("FSBLOBVERIFY_V1_" ||
// u64_be(0) || u64_be(4096) || blob || commitment) mod r, as 32 little-endian bytes (the device limb layout).  The same
// argument as for the batch transcript: a 131 KB serial chain is 65 us on a SHA-NI core and 2.8 ms on GPU lanes, and a call
// with a handful of blobs has no parallelism across blobs to give the GPU (capi_verify.hpp: host batches of up to
// host_challenge_max_blobs blobs; larger ones, and everything device-resident, hash on the GPU).
static void host_blob_challenge(uint8_t z_le[32], const uint8_t* blob, const uint8_t* commitment48) {
    uint8_t h[32];
    memcpy(h, "FSBLOBVERIFY_V1_", 16);
    memset(h + 16, 0, 16);
    h[30] = (uint8_t)(KZG_HOST_FE_PER_BLOB >> 8);
    h[31] = (uint8_t)(KZG_HOST_FE_PER_BLOB & 0xff);
    hostsha::Stream sha;
    sha.update(h, 32);
    sha.update(blob, (size_t)32 * KZG_HOST_FE_PER_BLOB);
    sha.update(commitment48, 48);
    uint8_t dg[32];
    sha.finish(dg);
    while (be_geq_r(dg)) be_sub_r(dg);  // digest mod r (:74-91)
    reverse32(z_le, dg);
}
// The per-blob challenge hashes of host batches run on ONE persistent pool of host threads per process (option host_threads,
// default 16, started by the first job with more than one blob): rounds 3-4 spawned a thread plus up to 16 hashing threads in
// every call, which T concurrent callers turned into 17 T threads.  A job is n independent chains; whoever holds it claims
// blob indices from an atomic counter - the pool's workers, the thread that posted it, the leader of the launch that needs
// its challenges (capi_coalesce.hpp) - so a job never waits for a worker to become free, and a caller that has to wait for
// the GPU anyway hashes its own blobs meanwhile.
namespace hostpool {
struct Job {
    const uint8_t *blobs = nullptr, *commitments = nullptr;
    uint8_t* z_le = nullptr;  // n x 32 bytes, little-endian (the device limb layout)
    size_t n = 0;
    std::atomic<size_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
    bool unclaimed() const { return next.load(std::memory_order_relaxed) < n; }
};
using JobRef = std::shared_ptr<Job>;
// claim and hash blobs of the job until none is left unclaimed (does not wait for the ones other threads hold)
static void help(Job& j) {
    for (;;) {
        const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
        if (i >= j.n) return;
        host_blob_challenge(j.z_le + 32 * i, j.blobs + (size_t)32 * KZG_HOST_FE_PER_BLOB * i, j.commitments + 48 * i);
        if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.n) {
            std::lock_guard<std::mutex> lk(j.mu);
            j.cv.notify_all();
        }
    }
}
// help, then wait until every blob of the job has its challenge
static void finish(Job& j) {
    help(j);
    if (j.done.load(std::memory_order_acquire) == j.n) return;
    std::unique_lock<std::mutex> lk(j.mu);
    j.cv.wait(lk, [&] { return j.done.load(std::memory_order_acquire) == j.n; });
}
struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<JobRef> jobs;
    size_t workers = 0, idle = 0, max_workers = 16;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            while (!jobs.empty() && !jobs.front()->unclaimed()) jobs.pop_front();
            if (jobs.empty()) {
                idle++;
                cv.wait(lk);
                idle--;
                continue;
            }
            JobRef j = jobs.front();
            lk.unlock();
            help(*j);
            lk.lock();
        }
    }
};
static Pool& pool() {  // (never destroyed: its detached workers may be parked in it when the process exits)
    static Pool* p = [] {
        Pool* q = new Pool();
#ifdef KZG_HOST_THREADS_OPTION
        const long v = KZG_HOST_THREADS_OPTION;
#else
        const long v = 16;
#endif
        const unsigned hc = std::thread::hardware_concurrency();
        q->max_workers = (size_t)std::max(1L, std::min(v < 1 ? 1 : v, hc ? (long)hc : 16L));
        return q;
    }();
    return *p;
}
// hand the job to the pool's workers (a job of one blob stays with its poster: waking a worker costs more than the chain);
// max_threads caps how many workers are woken for it
static void post(const JobRef& j, size_t max_threads = (size_t)-1) {
    if (j->n < 2 || max_threads < 2) return;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    P.jobs.push_back(j);
    size_t want = std::min(std::min(j->n, max_threads), P.max_workers);
    // new workers only for what the idle ones do not cover: idle + started < wanted (round 5's condition compared against a count
    // that did not move while spawning, so the first job of two blobs started all max_workers threads)
    for (size_t started = 0; P.workers < P.max_workers && P.idle + started < want; started++) {
        try {
            std::thread(&Pool::worker, &P).detach();
            P.workers++;
        } catch (const std::system_error&) {
            break;  // (the poster and whoever needs the result hash what no worker takes)
        }
    }
    if (want >= P.idle) P.cv.notify_all();
    else
        for (size_t k = 0; k < want; k++) P.cv.notify_one();
}
// Whoever posts a job over memory it does not own (the caller's blobs, a buffer of its own frame) joins it on EVERY way out -
// an error return, an exception - before that memory goes away: the pool's workers read the blobs and write z_le.
struct JoinOnExit {
    std::vector<JobRef> jobs;
    JoinOnExit() = default;
    JoinOnExit(const JoinOnExit&) = delete;
    JoinOnExit& operator=(const JoinOnExit&) = delete;
    void add(const JobRef& j) {
        if (j) jobs.push_back(j);
    }
    ~JoinOnExit() {
        for (const JobRef& j : jobs) finish(*j);
    }
};
static JobRef make(uint8_t* z_le, const uint8_t* blobs, const uint8_t* commitments, size_t n) {
    JobRef j = std::make_shared<Job>();
    j->blobs = blobs;
    j->commitments = commitments;
    j->z_le = z_le;
    j->n = n;
    return j;
}
}  // namespace hostpool
// n challenges, on the calling thread and up to max_threads - 1 workers of the pool; returns when all are written
static void host_blob_challenges(uint8_t* z_le, const uint8_t* blobs, const uint8_t* commitments, size_t n, size_t max_threads) {
    if (n == 0) return;
    hostpool::JobRef j = hostpool::make(z_le, blobs, commitments, n);
    hostpool::post(j, max_threads);
    hostpool::finish(*j);
}

static bool host_batch_challenges(uint8_t* r_out, const uint8_t* all_records, size_t B, size_t n, size_t n_total, size_t world) {
    std::atomic<bool> failed{false};
    auto digest_range = [&](size_t b0, size_t b1) {  // (the records are hashed where they lie: no transcript copy)
        for (size_t b = b0; b < b1; b++) {
            BatchTranscript t(n_total);
            if (world == 0) t.records(all_records + 160 * n_total * b, n_total);
            else
                for (size_t k = 0; k < world; k++) t.records(all_records + 160 * n * (k * B + b), n);
            t.r(r_out + 32 * b);
        }
    };
    static const size_t host_threads = [] {  // option host_threads (capi_host_util.hpp; 16 in the stand-alone sanitizer build)
#ifdef KZG_HOST_THREADS_OPTION
        long v = KZG_HOST_THREADS_OPTION;
#else
        long v = 16;
#endif
        return (size_t)(v < 1 ? 1 : v > 64 ? 64 : v);
    }();
    const size_t nthr = std::min(host_threads, std::min(B, (B * 160 * n_total) / (512 * 1024) + 1));
    if (nthr <= 1) {
        digest_range(0, B);
    } else {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < nthr; k++) pool.emplace_back(digest_range, B * k / nthr, B * (k + 1) / nthr);
        digest_range(0, B / nthr);
        for (auto& th : pool) th.join();
    }
    return !failed;
}

