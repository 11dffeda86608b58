// fr29bench.hip - dev microbenchmark: radix-2^29 Fr product (fr29.hpp) against the 8x32 product (field.hpp),
// dependent chains x = x * y, at several occupancies.
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/fr29bench.hip -o tools/microbench/fr29bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "field.hpp"
#include "fr29.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_mul32(uint32_t* o, int iters) {
    Fr x, y;
    for (int i = 0; i < 8; i++) { x.l[i] = threadIdx.x * 7 + i; y.l[i] = blockIdx.x + 3 * i + 1; }
    for (int k = 0; k < iters; k++) x = FrF::mul(x, y);
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= x.l[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul29(uint32_t* o, int iters) {
    Fr29 x, y;
    for (int i = 0; i < 9; i++) { x.l[i] = (threadIdx.x * 7 + i) & FR29_MASK; y.l[i] = (blockIdx.x + 3 * i + 1) & FR29_MASK; }
    for (int k = 0; k < iters; k++) x = fr29_mul(x, y);
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s ^= x.l[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two independent products per iteration (what a tree merge offers)
__global__ void k_mul29x2(uint32_t* o, int iters) {
    Fr29 x, y, z;
    for (int i = 0; i < 9; i++) { x.l[i] = (threadIdx.x * 7 + i) & FR29_MASK; y.l[i] = (blockIdx.x + 3 * i + 1) & FR29_MASK; z.l[i] = (threadIdx.x + i) & FR29_MASK; }
    for (int k = 0; k < iters; k++) { x = fr29_mul(x, y); z = fr29_mul(z, y); }
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s ^= x.l[i] ^ z.l[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul32x2(uint32_t* o, int iters) {
    Fr x, y, z;
    for (int i = 0; i < 8; i++) { x.l[i] = threadIdx.x * 7 + i; y.l[i] = blockIdx.x + 3 * i + 1; z.l[i] = threadIdx.x + i; }
    for (int k = 0; k < iters; k++) { x = FrF::mul(x, y); z = FrF::mul(z, y); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= x.l[i] ^ z.l[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* n; void (*f)(uint32_t*, int); int per; } ks[] = {{"Fr 8x32 mul", k_mul32, 1}, {"Fr 9x29 mul", k_mul29, 1}, {"Fr 8x32 mul x2 indep", k_mul32x2, 2}, {"Fr 9x29 mul x2 indep", k_mul29x2, 2}};
    int cfgs[][2] = {{1024, 64}, {2048, 64}, {3072, 64}, {4096, 64}, {2048, 256}, {4096, 256}};
    for (auto& c : cfgs) {
        double wps = (double)c[0] * (c[1] / 64) / 1024.0;
        printf("--- %d blocks x %d threads = %.0f waves/SIMD\n", c[0], c[1], wps);
        for (auto& k : ks) {
            const int it = 2000;
            k.f<<<c[0], c[1]>>>(d, 10); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k.f<<<c[0], c[1]>>>(d, it); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double cyc_per_mul_simd = ms * 1e-3 * 2.4e9 / ((double)it * k.per * wps);
            printf("%-24s %8.3f ms   %7.1f SIMD-cycles per wave-product   %6.1f Gmul/s\n", k.n, ms, cyc_per_mul_simd,
                   (double)c[0] * c[1] * it * k.per / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
