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
#include "host_only.hpp"

// ROCm gives a process 4 hardware queues by default and multiplexes its streams onto them; the pipeline of launch groups
// (capi_pipeline.hpp) keeps ~4 handles x (2 plain + 2 CU-masked + 1 copy) streams busy.  Measured through bench.py
// (GPU_MAX_HW_QUEUES = 2 / 4 / 8 / 16): 4.34 / 4.68 / 4.92 / 4.85 M blobs/s.  The runtime reads the variable when it
// initialises (its first API call), so the library asks for 8 when it is loaded - unless the caller has chosen a value, and
// without effect if the process has already used HIP.
__attribute__((constructor)) static void kzg_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }

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

// Point arithmetic of the decode and MSM kernels: the radix-2^29 field (fp29.hpp) unless KZG_FP29=0 selects the 12x32
// field (A/B measurement, cross-check).  Decides the table format (G1Jac29Mem / G1Jac) for the whole process.
static bool fp29_enabled() {
    static const bool v = [] {
        const char* e = getenv("KZG_FP29");
        return !(e && e[0] == '0');
    }();
    return v;
}
// Throughput layout of the verification MSM (MSM_CHUNKS tables) with AFFINE entries and mixed additions (msm.hpp
// k_mult_to_affine29); KZG_MSM_AFFINE=0 keeps Jacobian entries (A/B measurement).  Radix-2^29 field only.
static bool msm_affine_enabled() {
    static const bool v = [] {
        const char* e = getenv("KZG_MSM_AFFINE");
        return fp29_enabled() && !(e && e[0] == '0');
    }();
    return v;
}
constexpr size_t MULT_ENTRY_BYTES = sizeof(G1Jac29Mem) > sizeof(G1Jac) ? sizeof(G1Jac29Mem) : sizeof(G1Jac);
