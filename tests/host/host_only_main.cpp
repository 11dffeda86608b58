// Sanitizer harness for kzg_rs_amd/csrc/host_only.hpp (the library's HIP-free host code), built by
// tests/test_host_only_sanitized.py with g++ -fsanitize=address,undefined -fno-sanitize-recover=all:
//   host_only_main <trusted_setup.txt> <seed> <n_mutations>
// 1. SHA-256: known answers, portable path against the SHA-NI path on every length 0..300 and on 1 MiB.
// 2. the trusted-setup parser on the shipped file (counts, bit-reversal placement) and on n_mutations mutated copies
//    (truncation, non-hex digits, dropped / doubled characters, header edits, CRLF): never a crash or a sanitizer
//    report, a clean rejection for every syntactic break, acceptance of the CRLF form.
// 3. the batch-transcript hash in both record layouts and thread counts; prints r for the Python side (oracle check).
#include <stdio.h>

#include <fstream>
#include <random>
#include <sstream>

#include "host_only.hpp"

static int failures = 0;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            printf("CHECK FAILED line %d: %s\n", __LINE__, #c);    \
            failures++;                                            \
        }                                                          \
    } while (0)

static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) {
        s += d[p[i] >> 4];
        s += d[p[i] & 15];
    }
    return s;
}
static void sha_portable(uint8_t out[32], const uint8_t* data, size_t len) {  // hostsha::digest with the NI path switched off
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) hostsha::block(st, data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    hostsha::block(st, tail);
    if (tl == 128) hostsha::block(st, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string good = ss.str();
    const unsigned seed = (unsigned)atoi(argv[2]);
    const int n_mut = atoi(argv[3]);
    // ---- 1. SHA-256
    uint8_t d[32], d2[32];
    hostsha::digest(d, (const uint8_t*)"abc", 3);
    CHECK(hex(d, 32) == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
    hostsha::digest(d, (const uint8_t*)"", 0);
    CHECK(hex(d, 32) == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855");
    std::vector<uint8_t> buf(1 << 20);
    std::mt19937 rng(seed);
    for (auto& b : buf) b = (uint8_t)rng();
    for (size_t len = 0; len <= 300; len++) {
        hostsha::digest(d, buf.data(), len);
        sha_portable(d2, buf.data(), len);
        CHECK(memcmp(d, d2, 32) == 0);
    }
    hostsha::digest(d, buf.data(), buf.size());
    sha_portable(d2, buf.data(), buf.size());
    CHECK(memcmp(d, d2, 32) == 0);
    printf("sha_ni %d\n", (int)hostsha::have_ni());
    // the streaming form: the same message in pieces of random length (0 included) gives the one-shot digest
    for (int it = 0; it < 200; it++) {
        const size_t len = it < 100 ? rng() % 700 : rng() % buf.size();
        sha_portable(d2, buf.data(), len);
        hostsha::Stream st;
        size_t pos = 0;
        while (pos < len) {
            const size_t take = std::min(len - pos, (size_t)(rng() % (it % 3 == 0 ? 5 : it % 3 == 1 ? 130 : 100000)));
            st.update(buf.data() + pos, take);
            pos += take;
        }
        st.finish(d);
        CHECK(memcmp(d, d2, 32) == 0);
    }
    // ---- 2. parser
    std::vector<uint8_t> g1b, g2b;
    uint8_t first[2][48];
    long n1 = 0, n2 = 0;
    std::string err;
    CHECK(hostparse::trusted_setup_text(good.data(), good.size(), g1b, g2b, first, n1, n2, err));
    CHECK(n1 == 4096 && n2 == 65 && g1b.size() == 48 * 4096 && g2b.size() == 96 * 65);
    {   // file line 1 (the second G1 point) lands in slot brp(1) = 2048
        size_t pos = good.find('\n', good.find('\n') + 1) + 1;  // start of G1 line 0
        std::string l0 = good.substr(pos, 96), l1 = good.substr(pos + 97, 96);
        CHECK(hex(g1b.data(), 48) == l0 && hex(g1b.data() + 48 * 2048, 48) == l1);
        CHECK(hex(first[0], 48) == l0 && hex(first[1], 48) == l1);
    }
    int rejected = 0, accepted = 0;
    for (int it = 0; it < n_mut; it++) {
        std::string m = good;
        const int kind = it % 10;
        bool must_reject = true;
        switch (kind) {
            case 0: m.resize(rng() % (m.size() - 200)); break;                                 // truncation (at least the last two lines go)
            case 1: m[200 + rng() % (m.size() - 200)] = "gzGZ xX-"[rng() % 8]; must_reject = false; break;  // may hit a newline: either outcome, no crash
            case 2: m.erase(200 + rng() % (m.size() - 200), 1); must_reject = false; break;    // a line loses a character (or two lines merge)
            case 3: m.insert(200 + rng() % (m.size() - 200), 1, 'a'); must_reject = false; break;
            case 4: m.replace(0, 4, std::string("409") + "578"[rng() % 3]); break;               // wrong G1 count
            case 5: m.replace(5, 2, rng() % 2 ? "1" : "99999999"); break;                      // G2 count too small / absurd
            case 6: m = rng() % 2 ? std::string() : m.substr(0, 5); break;                     // empty / header only
            case 7: {                                                                          // CRLF line ends: accepted
                std::string c;
                for (char ch : m) {
                    if (ch == '\n') c += '\r';
                    c += ch;
                }
                m = c;
                must_reject = false;
                break;
            }
            case 8: m[rng() % 4] = "-+ x"[rng() % 4]; break;                                   // header with a sign / blank / letter
            case 9: m += std::string(rng() % 3, '\n') + "trailing"; must_reject = false; break;  // extra lines after the last point: ignored like the reference
        }
        std::vector<uint8_t> a, b;
        uint8_t fr[2][48];
        long x1 = 0, x2 = 0;
        std::string e;
        const bool ok = hostparse::trusted_setup_text(m.data(), m.size(), a, b, fr, x1, x2, e);
        if (ok) accepted++;
        else {
            rejected++;
            CHECK(!e.empty());
        }
        if (must_reject) CHECK(!ok);
        if (kind == 7 || kind == 9) CHECK(ok && a == g1b && b == g2b);
        if (ok) CHECK(a.size() == 48 * 4096 && b.size() == 96 * (size_t)x2 && x1 == 4096);
    }
    printf("parser mutations %d rejected %d accepted %d\n", n_mut, rejected, accepted);
    // ---- 3. transcript hash: [B][n_total] against the same records laid out [world][B][n]
    const size_t B = 5, n = 7, world = 3, n_total = n * world;
    std::vector<uint8_t> flat(160 * n_total * B), gathered(160 * n_total * B);
    for (auto& x : flat) x = (uint8_t)rng();
    for (size_t i = 0; i < n_total * B; i++) flat[160 * i + 79] = flat[160 * i + 111] = 0;  // z, y (little-endian) canonical: below 2^248
    for (size_t k = 0; k < world; k++)
        for (size_t b = 0; b < B; b++) memcpy(gathered.data() + 160 * n * (k * B + b), flat.data() + 160 * (n_total * b + n * k), 160 * n);
    std::vector<uint8_t> r1(32 * B), r2(32 * B);
    host_batch_challenges(r1.data(), flat.data(), B, n_total, n_total, 0);
    host_batch_challenges(r2.data(), gathered.data(), B, n, n_total, world);
    CHECK(r1 == r2);
    for (size_t b = 0; b < B; b++) {
        uint8_t be[32];
        reverse32(be, r1.data() + 32 * b);
        CHECK(!be_geq_r(be));
    }
    {   // the transcript as a stream of record pieces (capi_multi.hpp feeds it chunk by chunk)
        BatchTranscript t(n_total);
        for (size_t i = 0; i < n_total;) {
            const size_t take = std::min(n_total - i, (size_t)(1 + rng() % 5));
            t.records(flat.data() + 160 * i, take);
            i += take;
        }
        uint8_t rr[32];
        t.r(rr);
        CHECK(memcmp(rr, r1.data(), 32) == 0);
    }
    printf("records %s\n", hex(flat.data(), 160 * n_total).c_str());  // batch 0's transcript records
    printf("r0 %s\n", hex(r1.data(), 32).c_str());                     // little-endian
    // a launch-group-sized call takes the threaded path (16 threads by default)
    std::vector<uint8_t> big(160 * 1024 * 40), rb(32 * 40), rs(32 * 40);
    for (auto& x : big) x = (uint8_t)rng();
    host_batch_challenges(rb.data(), big.data(), 40, 1024, 1024, 0);
    for (size_t b = 0; b < 40; b++) host_batch_challenges(rs.data() + 32 * b, big.data() + 160 * 1024 * b, 1, 1024, 1024, 0);
    CHECK(rb == rs);
    {   // the pool's FIRST job (round-5 advisor finding (a), below): at most the workers it asked for
        hostpool::Pool& P = hostpool::pool();
        size_t before;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            before = P.workers;
        }
        const size_t BLOB = (size_t)32 * KZG_HOST_FE_PER_BLOB;
        std::vector<uint8_t> blobs(BLOB * 2), cs(48 * 2), z(64), want(64);
        for (auto& x : blobs) x = (uint8_t)rng();
        for (size_t i = 0; i < 2; i++) host_blob_challenge(want.data() + 32 * i, blobs.data() + BLOB * i, cs.data() + 48 * i);
        hostpool::JobRef j = hostpool::make(z.data(), blobs.data(), cs.data(), 2);
        hostpool::post(j, 2);
        hostpool::finish(*j);
        {
            std::lock_guard<std::mutex> lk(P.mu);
            CHECK(before == 0 && P.workers <= 2);   // the pool's first job: two blobs, two threads wanted
            CHECK(P.workers <= P.max_workers);
        }
        CHECK(z == want);
    }
    {   // the per-blob challenges on the persistent pool (hostpool): 8 caller threads at once, each with jobs of 1..5 blobs, some of
        // them posted and finished by different threads (the small-call queue's leader finishes jobs its followers posted)
        const size_t BLOB = (size_t)32 * KZG_HOST_FE_PER_BLOB, NB = 5;
        std::vector<uint8_t> blobs(BLOB * NB), cs(48 * NB), want(32 * NB);
        for (auto& x : blobs) x = (uint8_t)rng();
        for (auto& x : cs) x = (uint8_t)rng();
        for (size_t i = 0; i < NB; i++) host_blob_challenge(want.data() + 32 * i, blobs.data() + BLOB * i, cs.data() + 48 * i);
        std::atomic<int> wrong{0};
        std::vector<std::thread> callers;
        for (int t = 0; t < 8; t++)
            callers.emplace_back([&, t] {
                for (int rep = 0; rep < 6; rep++) {
                    const size_t n = 1 + (size_t)((t + rep) % NB);
                    std::vector<uint8_t> z(32 * n, 0xee);
                    if (rep & 1) host_blob_challenges(z.data(), blobs.data(), cs.data(), n, 16);
                    else {
                        hostpool::JobRef j = hostpool::make(z.data(), blobs.data(), cs.data(), n);
                        hostpool::post(j);
                        std::thread other([j] { hostpool::finish(*j); });
                        other.join();
                        hostpool::finish(*j);  // (a second finish of a complete job returns at once)
                    }
                    if (memcmp(z.data(), want.data(), 32 * n) != 0) wrong++;
                }
            });
        for (auto& th : callers) th.join();
        CHECK(wrong.load() == 0);
    }
    {   // round-5 advisor findings on the pool.  (a) post() started all max_workers threads for the first job of two blobs, whatever it
        // wanted (the spawn loop compared against a count that did not move): a job that wants k workers starts at most k beyond the
        // idle ones.  (b) JoinOnExit: a scope that posted jobs over its own buffers leaves - by return or by exception - only after
        // the workers are done with them.
        const size_t BLOB = (size_t)32 * KZG_HOST_FE_PER_BLOB;
        for (int thrown = 0; thrown < 2; thrown++) {
            std::vector<uint8_t> zz(32 * 4, 0xee), bl(BLOB * 4), cc(48 * 4), w4(32 * 4);
            for (auto& x : bl) x = (uint8_t)rng();
            for (size_t i = 0; i < 4; i++) host_blob_challenge(w4.data() + 32 * i, bl.data() + BLOB * i, cc.data() + 48 * i);
            bool done_at_exit = false;
            try {
                hostpool::JobRef jj;
                struct Probe {  // destroyed AFTER the guard below (declared before it): sees the state the guard leaves
                    hostpool::JobRef* j;
                    bool* out;
                    ~Probe() { *out = *j && (*j)->done.load() == (*j)->n; }
                } probe{&jj, &done_at_exit};
                hostpool::JoinOnExit joined;
                jj = hostpool::make(zz.data(), bl.data(), cc.data(), 4);
                joined.add(jj);
                hostpool::post(jj, 3);
                if (thrown) throw std::bad_alloc();   // an error path that never reaches a finish() of its own
            } catch (const std::bad_alloc&) {
            }
            CHECK(done_at_exit);
            CHECK(zz == w4);
        }
    }
    printf("failures %d\n", failures);
    return failures ? 1 : 0;
}
