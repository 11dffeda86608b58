// shabench.hip - dev microbenchmark: pure-register SHA-256 compression (no loads, no LDS, no barriers) at several
// occupancies: SIMD cycles per 64-byte block per wave, for the whole compression and for its two halves as the
// producer/consumer challenge kernel splits them (message schedule + K  |  the 64 rounds).
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/shabench.hip -o tools/microbench/shabench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "sha256.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_full(uint32_t* o, int iters) {
    Sha256State st; sha256_init(st);
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = threadIdx.x * 31 + i;
    for (int k = 0; k < iters; k++) {
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = w[i] + st.h[i & 7];
        sha256_compress(st, m);
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = st.h[0] ^ st.h[5];
}
__global__ void k_schedule(uint32_t* o, int iters) {
    uint32_t w[16], acc = 0;
    for (int i = 0; i < 16; i++) w[i] = threadIdx.x * 31 + i;
    for (int k = 0; k < iters; k++) {
        uint32_t kw[64];
        sha256_schedule_kw(kw, w);
        for (int i = 0; i < 64; i++) acc ^= kw[i];
        w[0] += acc;
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_rounds(uint32_t* o, int iters) {
    uint32_t a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    uint4 kw[16];
    for (int i = 0; i < 16; i++) kw[i] = make_uint4(i, threadIdx.x, 3 * i, 7);
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            asm volatile("" : "+v"(kw[t].x), "+v"(kw[t].y), "+v"(kw[t].z), "+v"(kw[t].w));
            sha256_rounds4(a, b, c, d, e, f, g, h, kw[t]);
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a ^ e;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 8192 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* n; void (*f)(uint32_t*, int); } ks[] = {{"compress (schedule+rounds)", k_full}, {"schedule + K only", k_schedule}, {"64 rounds only", k_rounds}};
    int cfgs[][2] = {{1024, 64}, {2048, 64}, {4096, 64}, {2048, 256}, {4096, 256}};
    for (auto& c : cfgs) {
        double wps = (double)c[0] * (c[1] / 64) / 1024.0;
        printf("--- %d blocks x %d threads = %.0f waves/SIMD\n", c[0], c[1], wps);
        for (auto& k : ks) {
            const int it = 4000;
            k.f<<<c[0], c[1]>>>(d, 10); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k.f<<<c[0], c[1]>>>(d, it); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-28s %8.3f ms   %7.1f SIMD-cycles per wave-block   (%.2f M blocks/s/lane-wave -> %.1f M blobs/s if alone)\n", k.n, ms,
                   ms * 1e-3 * 2.4e9 / ((double)it * wps), 0.0, (double)c[0] * c[1] * it / (ms * 1e-3) / 2050 / 1e6);
        }
    }
    return 0;
}
