// stridebench.hip - dev microbenchmark: one lane per "blob", every lane streams its own 128 KiB region line by line with
// eight 16-byte loads per 128-byte line (the access pattern of k_blob_challenge).  Regions 2^17 bytes apart (the
// caller's blob layout) against 2^17 + 128 / + 4096 / + 8320: does the power-of-two lane stride cost bandwidth?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/stridebench.hip -o tools/microbench/stridebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_stream(const uint8_t* base, size_t stride, uint32_t* out, int n, int spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)i * stride);
    uint32_t acc = 0;
    uint4 L[8];
#pragma unroll
    for (int j = 0; j < 8; j++) L[j] = p[j];
    for (int m = 0; m < 1024; m++) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) x ^= L[j].x ^ L[j].y ^ L[j].z ^ L[j].w;
        if (m + 1 < 1024) {
            const uint4* q = p + 8 * (m + 1);
#pragma unroll
            for (int j = 0; j < 8; j++) L[j] = q[j];
        }
        for (int s = 0; s < spin; s++) x = x * 1664525u + 1013904223u;  // stands for the two compressions
        acc ^= x;
    }
    out[i] = acc;
}

int main() {
    const int n = 131072;
    size_t strides[] = {131072, 131072 + 128, 131072 + 4096, 131072 + 8320};
    uint8_t* d; CK(hipMalloc(&d, (size_t)n * (131072 + 8320)));
    CK(hipMemset(d, 1, (size_t)n * (131072 + 8320)));
    uint32_t* o; CK(hipMalloc(&o, 4 * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int spin : {0, 200, 1400}) {
        for (size_t st : strides) {
            k_stream<<<n / 256, 256>>>(d, st, o, n, spin); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k_stream<<<n / 256, 256>>>(d, st, o, n, spin); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("spin %4d  lane stride %7zu B: %8.3f ms  %7.1f GB/s\n", spin, st, ms, (double)n * 131072 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
