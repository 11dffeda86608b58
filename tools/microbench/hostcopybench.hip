// Host-buffer hand-over costs on this box: pageable vs registered vs pinned 128 MiB host -> device copies, the price of
// hipHostRegister / hipHostUnregister, whether a pageable hipMemcpyAsync returns before the copy is done, and a kernel
// that reads registered host memory directly (zero-copy) at the access pattern of the challenge kernel's producer.
//   hipcc -O3 --offload-arch=gfx950 -o hostcopybench hostcopybench.hip && ./hostcopybench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_sum(const uint4* __restrict__ src, size_t n_per_lane, size_t stride_u4, unsigned* out) {
    // lane i streams its own 128 KiB region, 128 bytes at a time (8 x 16 B back to back), like one blob per lane
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4* p = src + i * stride_u4;
    unsigned acc = 0;
    for (size_t k = 0; k < n_per_lane; k += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[k + j];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    out[i] = acc;
}
int main() {
    const size_t N = 128ull << 20;
    void *d, *dout;
    CK(hipMalloc(&d, N));
    CK(hipMalloc(&dout, 4 * 1024));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* pageable = (char*)malloc(N + 4096);
    memset(pageable, 1, N + 4096);
    char* pinned;
    CK(hipHostMalloc((void**)&pinned, N, hipHostMallocDefault));
    memset(pinned, 2, N);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, pageable + 64, N, hipMemcpyHostToDevice, st));
        double t1 = now();
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("pageable  H2D 128 MiB: call returns after %.3f ms, done after %.3f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, st));
        double t1 = now();
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("pinned    H2D 128 MiB: call returns after %.3f ms, done after %.3f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; rep++) {
        char* fresh = (char*)malloc(N + 4096);
        memset(fresh, rep, N + 4096);
        double t0 = now();
        CK(hipHostRegister(fresh + 64, N, hipHostRegisterDefault));
        double t1 = now();
        void* dp;
        CK(hipHostGetDevicePointer(&dp, fresh + 64, 0));
        CK(hipMemcpyAsync(d, fresh + 64, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        hipLaunchKernelGGL(k_sum, dim3(16), dim3(64), 0, st, (const uint4*)dp, (size_t)8192, (size_t)8192, (unsigned*)dout);
        CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(fresh + 64));
        double t4 = now();
        printf("fresh malloc: hipHostRegister %.3f ms, registered H2D %.3f ms (%.1f GB/s), zero-copy kernel (1024 lanes x 128 KiB, line at a time) %.3f ms (%.1f GB/s), unregister %.3f ms\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, N / (t2 - t1) / 1e9, (t3 - t2) * 1e3, N / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
        free(fresh);
    }
    // the same kernel on device memory, for scale
    double t0 = now();
    hipLaunchKernelGGL(k_sum, dim3(16), dim3(64), 0, st, (const uint4*)d, (size_t)8192, (size_t)8192, (unsigned*)dout);
    CK(hipStreamSynchronize(st));
    printf("same kernel on HBM: %.3f ms\n", (now() - t0) * 1e3);
    return 0;
}
