// fieldbench.hip - dev microbenchmark: latency/throughput of the Montgomery multiply variants
// in kzg_rs_amd/csrc/field.hpp on gfx950, each checked against a host __int128 computation.
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/fieldbench.hip -o /tmp/fieldbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "field.hpp"
using namespace kzg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F, int V>
__global__ void k_mul(const typename F::E* a, const typename F::E* b, typename F::E* o, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    typename F::E x = a[i], y = b[i];
    for (int k = 0; k < iters; k++) {
        if (V == 0) x = F::mul_cios(x, y);
        if (V == 1) x = F::mul_ps(x, y);
        if (V == 2) x = F::mul(x, y);
    }
    o[i] = x;
}

// raw instruction-rate probes: dependent chains of one opcode
__global__ void k_mad_chain(uint64_t* o, uint32_t a, uint32_t b, int iters) {
    uint64_t acc = threadIdx.x;
    uint32_t x = a + threadIdx.x, y = b;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 32; u++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_mad_indep(uint64_t* o, uint32_t a, uint32_t b, int iters) {
    uint64_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    uint32_t x = a + threadIdx.x, y = b;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void k_add_chain(uint32_t* o, uint32_t a, int iters) {
    uint32_t acc = threadIdx.x, x = a;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 32; u++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(x));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_macpair_chain(uint64_t* o, uint32_t a, uint32_t b, int iters) {
    uint64_t acc = threadIdx.x;
    uint32_t hi = 0, x = a + threadIdx.x, y = b;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 32; u++)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc");
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc + hi;
}

typedef unsigned __int128 u128;
// host reference Montgomery product with 32-bit limbs (independent of the device code path)
template <int N>
static void host_mont(uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* m, uint32_t inv) {
    uint32_t t[N + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        for (int j = 0; j < N; j++) { uint64_t x = (uint64_t)a[j] * b[i] + t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
        uint64_t x = (uint64_t)t[N] + c; t[N] = (uint32_t)x; t[N + 1] = (uint32_t)(x >> 32);
        uint32_t q = t[0] * inv;
        x = (uint64_t)q * m[0] + t[0]; c = x >> 32;
        for (int j = 1; j < N; j++) { x = (uint64_t)q * m[j] + t[j] + c; t[j - 1] = (uint32_t)x; c = x >> 32; }
        x = (uint64_t)t[N] + c; t[N - 1] = (uint32_t)x; t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    // conditional subtract
    uint32_t d[N]; uint64_t br = 0;
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)t[i] - m[i] - br; d[i] = (uint32_t)x; br = (x >> 63) & 1; }
    bool take = t[N] || !br;
    for (int i = 0; i < N; i++) r[i] = take ? d[i] : t[i];
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <class F, int V>
static void bench_mul(const char* name, const uint32_t* mod, uint32_t inv, int blocks, int threads, int iters) {
    constexpr int N = F::N;
    size_t n = (size_t)blocks * threads;
    std::vector<typename F::E> ha(n), hb(n), ho(n);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < N; k++) { ha[i].l[k] = (uint32_t)splitmix(); hb[i].l[k] = (uint32_t)splitmix(); }
        ha[i].l[N - 1] &= 0x0fffffff; hb[i].l[N - 1] &= 0x0fffffff;  // < modulus
    }
    typename F::E *da, *db, *dout;
    CK(hipMalloc(&da, n * sizeof(typename F::E))); CK(hipMalloc(&db, n * sizeof(typename F::E))); CK(hipMalloc(&dout, n * sizeof(typename F::E)));
    CK(hipMemcpy(da, ha.data(), n * sizeof(typename F::E), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * sizeof(typename F::E), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_mul<F, V><<<blocks, threads>>>(da, db, dout, 4);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_mul<F, V><<<blocks, threads>>>(da, db, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(ho.data(), dout, n * sizeof(typename F::E), hipMemcpyDeviceToHost));
    // check a sample of lanes
    int bad = 0;
    for (size_t i = 0; i < n; i += (n / 64 ? n / 64 : 1)) {
        uint32_t x[N]; memcpy(x, ha[i].l, sizeof x);
        for (int k = 0; k < iters; k++) host_mont<N>(x, x, hb[i].l, mod, inv);
        if (memcmp(x, ho[i].l, sizeof x)) bad++;
    }
    double per_mul_ns = (double)ms * 1e6 / iters;
    double tput = (double)n * iters / (ms * 1e-3) / 1e9;
    printf("%-22s blocks=%5d thr=%4d  %8.3f ms  latency/mul %8.1f ns (~%6.0f cyc@2.4GHz)  %8.2f Gmul/s  %s\n", name, blocks, threads, ms,
           per_mul_ns, per_mul_ns * 2.4, tput, bad ? "MISMATCH" : "ok");
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int ITERS = 2000;
    // one wave per SIMD: 256 CUs x 4 SIMDs = 1024 waves
    for (int cfg = 0; cfg < 3; cfg++) {
        int blocks = cfg == 0 ? 1024 : cfg == 1 ? 2048 : 4096, threads = cfg == 0 ? 64 : cfg == 1 ? 256 : 256;
        printf("--- config %d: %d blocks x %d threads (%.1f waves/SIMD)\n", cfg, blocks, threads, blocks * (threads / 64) / 1024.0);
        bench_mul<FrF, 0>("Fr mul (CIOS, hipcc)", consts::FR_MOD, FR_INV32, blocks, threads, ITERS);
        bench_mul<FrF, 1>("Fr mul_ps (asm mac)", consts::FR_MOD, FR_INV32, blocks, threads, ITERS);
        bench_mul<FrF, 2>("Fr mul_pc (asm chain)", consts::FR_MOD, FR_INV32, blocks, threads, ITERS);
        bench_mul<FpF, 0>("Fp mul (CIOS, hipcc)", consts::FP_MOD, FP_INV32, blocks, threads, ITERS);
        bench_mul<FpF, 1>("Fp mul_ps (asm mac)", consts::FP_MOD, FP_INV32, blocks, threads, ITERS);
        bench_mul<FpF, 2>("Fp mul_pc (asm chain)", consts::FP_MOD, FP_INV32, blocks, threads, ITERS);
    }
    // raw opcode probes
    for (int cfg = 0; cfg < 2; cfg++) {
        int blocks = cfg == 0 ? 1024 : 4096, threads = cfg == 0 ? 64 : 256;
        size_t n = (size_t)blocks * threads;
        uint64_t* d64; uint32_t* d32; CK(hipMalloc(&d64, n * 8)); CK(hipMalloc(&d32, n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int it = 20000; float ms;
        auto report = [&](const char* nm, int ops) {
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            double per_op_ns = ms * 1e6 / ((double)it * ops);
            printf("%-28s %5d x %4d: %7.3f ms  %6.2f ns/op per wave (~%5.1f cyc)  %8.1f Gop/s (lanes)\n", nm, blocks, threads, ms, per_op_ns, per_op_ns * 2.4,
                   (double)n * it * ops / (ms * 1e-3) / 1e9);
        };
        k_mad_chain<<<blocks, threads>>>(d64, 3, 5, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_mad_chain<<<blocks, threads>>>(d64, 3, 5, it); report("v_mad_u64_u32 dependent", 32);
        CK(hipEventRecord(e0)); k_mad_indep<<<blocks, threads>>>(d64, 3, 5, it); report("v_mad_u64_u32 4-indep", 32);
        CK(hipEventRecord(e0)); k_add_chain<<<blocks, threads>>>(d32, 3, it); report("v_add_u32 dependent", 32);
        CK(hipEventRecord(e0)); k_macpair_chain<<<blocks, threads>>>(d64, 3, 5, it); report("mad+addc pair dependent", 32);
        CK(hipFree(d64)); CK(hipFree(d32));
    }
    return 0;
}
