// capi_host_util.hpp - host helpers of the C ABI: error strings, HIP checks, host SHA-256 (SHA-NI), big-endian scalar helpers.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// ---------------------------------------------------------------- host helpers
// (SHA-256, big-endian scalar helpers, the trusted-setup text parser and the transcript hash live in host_only.hpp: plain C++
// without HIP, so that they also build with g++ -fsanitize=address,undefined - tests/test_host_only_sanitized.py)
static thread_local std::string g_err;
static KzgRet fail(KzgRet rc, const std::string& msg) {
    g_err = msg;
    return rc;
}
extern "C" const char* kzg_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------- options
// ONE environment variable, KZG_OPTIONS = "key=value;key=value;..." (';' or blanks between entries), carries every tuning
// and test switch of the library; a switch is named by its key in the comment beside the code that reads it, and
// INTEGRATION.md lists them.  The string is parsed when it is first needed and again whenever its value has changed
// (tests change it between handles); most switches are latched by their reader on first use.  Switches that select a kernel
// variant compiled only into the A/B build (KZG_AB_VARIANTS=1: fp29=0, evaluate_kernel=32, proofs_chunks=16) are ignored by
// the product build - ab_variants_built() tells which one is loaded.
// The device list of an UNCHANGED caller is the one deployment knob with a variable of its own: KZG_DEVICES.
//
// The library does not touch the process environment (rounds 1-3 set GPU_MAX_HW_QUEUES from a load-time constructor): the
// pipeline of launch groups wants 8 hardware queues (measured through bench.py with GPU_MAX_HW_QUEUES = 2 / 4 / 8 / 16:
// 4.34 / 4.68 / 4.92 / 4.85 M blobs/s; ROCm's default is 4), which the HOST sets in its environment before the HIP runtime
// initialises; hw_queues_note() is what a constructor leaves in kzg_last_error() when it sees fewer.
#ifndef KZG_AB_VARIANTS
#define KZG_AB_VARIANTS 0
#endif
#include <map>
static const char* opt_str(const char* key) {  // nullptr when unset; the pointer stays valid until the process ends
    static std::mutex mu;
    static std::string parsed_from;
    static std::map<std::string, std::string*> kv;  // (values handed out earlier stay alive across a re-parse: a few bytes per change of the string)
    std::lock_guard<std::mutex> lk(mu);
    const char* e = getenv("KZG_OPTIONS");
    const std::string cur = e ? e : "";
    if (cur != parsed_from) {
        parsed_from = cur;
        kv.clear();
        size_t i = 0;
        while (i < cur.size()) {
            size_t j = cur.find_first_of("; \t\n", i);
            if (j == std::string::npos) j = cur.size();
            const std::string item = cur.substr(i, j - i);
            const size_t eq = item.find('=');
            if (!item.empty()) kv[eq == std::string::npos ? item : item.substr(0, eq)] = new std::string(eq == std::string::npos ? "1" : item.substr(eq + 1));
            i = j + 1;
        }
    }
    auto it = kv.find(key);
    return it == kv.end() ? nullptr : it->second->c_str();
}
static bool opt_flag(const char* key, bool dflt) {
    const char* v = opt_str(key);
    return v ? !(v[0] == '0' || v[0] == 'n' || v[0] == 'f') : dflt;
}
static long opt_int(const char* key, long dflt) {
    const char* v = opt_str(key);
    return v && *v ? atol(v) : dflt;
}
static bool opt_is(const char* key, const char* value) {
    const char* v = opt_str(key);
    return v && strcmp(v, value) == 0;
}
// ab_flag / ab_int: switches whose only use is to select a form kept for A/B MEASUREMENT (a superseded kernel form, a tuning
// value that was swept once): compiled into the A/B build only - the product build takes the default at compile time, so its
// KZG_OPTIONS surface is the deployment and test switches, not the lab's.
#if KZG_AB_VARIANTS
static bool ab_flag(const char* key, bool dflt) { return opt_flag(key, dflt); }
static long ab_int(const char* key, long dflt) { return opt_int(key, dflt); }
#else
static constexpr bool ab_flag(const char*, bool dflt) { return dflt; }
static constexpr long ab_int(const char*, long dflt) { return dflt; }
#endif
#define KZG_HOST_THREADS_OPTION opt_int("host_threads", 16)
#include "host_only.hpp"
static const char* hw_queues_note() {
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    if (q && atoi(q) >= 8) return nullptr;
    return "note: GPU_MAX_HW_QUEUES is unset or below 8 - set it to 8 in the environment before the HIP runtime starts; with "
           "ROCm's default of 4 hardware queues the launch-group pipeline runs ~5 % below its rate";
}

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (void)hipGetLastError();                                                                       \
            /* c-kzg-4844's C_KZG_MALLOC: the device (or pinned host) allocation did not fit */           \
            return fail(e_ == hipErrorOutOfMemory ? KZG_MALLOC : KZG_ERROR,                                \
                        std::string("HIP: ") + hipGetErrorString(e_) + " at " #expr);                      \
        }                                                                                                  \
    } while (0)

// (dyn_lds_ensure / DYN_LDS: dyn_lds.hpp)

// event timing that never leaves a sticky HIP error behind (an event may not have been recorded on this path)
static void elapsed(float* out, hipEvent_t a, hipEvent_t b) {
    if (hipEventElapsedTime(out, a, b) != hipSuccess) {
        (void)hipGetLastError();
        *out = 0.f;
    }
}

// a scratch device allocation of one call: released on every path out of the function (hipFree waits for the work that
// may still use it)
struct DevTmp {
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Point arithmetic of the decode and MSM kernels: the radix-2^29 field (fp29.hpp) unless the A/B build's option fp29=0 selects the 12x32
// field (measurement, cross-check).  Decides the table format (G1Jac29Mem / G1Jac) for the whole process.
static bool fp29_enabled() {
#if KZG_AB_VARIANTS
    static const bool v = opt_flag("fp29", true);
    return v;
#else
    return true;
#endif
}
// Throughput layout of the verification MSM (MSM_CHUNKS tables) with AFFINE entries and mixed additions (msm.hpp
// k_mult_to_affine29); option msm_affine=0 keeps Jacobian entries (A/B measurement).  Radix-2^29 field only.
static bool msm_affine_enabled() {
    static const bool v = fp29_enabled() && ab_flag("msm_affine", true);
    return v;
}
constexpr size_t MULT_ENTRY_BYTES = sizeof(G1Jac29Mem) > sizeof(G1Jac) ? sizeof(G1Jac29Mem) : sizeof(G1Jac);
