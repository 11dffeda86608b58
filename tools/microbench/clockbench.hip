// clockbench.hip - dev microbenchmark: the shader clock a multiply-add-dense kernel actually runs at, alone on one CU
// and on the whole chip (s_memtime counts shader cycles, s_memrealtime the constant 100 MHz reference).
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/clockbench.hip -o tools/microbench/clockbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "fr29.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_mul29(uint32_t* o, unsigned long long* clk, int iters) {
    Fr29 x, y;
    for (int i = 0; i < 9; i++) { x.l[i] = (threadIdx.x * 7 + i) & FR29_MASK; y.l[i] = (blockIdx.x + 3 * i + 1) & FR29_MASK; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; k++) x = fr29_mul(x, y);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s ^= x.l[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 8192 * 256 * 4));
    unsigned long long* c; CK(hipMalloc(&c, 8192 * 16));
    unsigned long long* h = (unsigned long long*)malloc(8192 * 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int cfgs[][3] = {{1, 64, 20000}, {256, 64, 20000}, {1024, 256, 4000}, {4096, 256, 2000}, {4096, 256, 20000}, {8192, 256, 40000}};
    for (auto& g : cfgs) {
        k_mul29<<<g[0], g[1]>>>(d, c, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_mul29<<<g[0], g[1]>>>(d, c, g[2]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, c, g[0] * 16, hipMemcpyDeviceToHost));
        double st = 0, sr = 0;
        for (int i = 0; i < g[0]; i++) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
        double waves_per_simd = (double)g[0] * (g[1] / 64) / 1024.0;
        double wall_cyc = ms * 1e-3 * 2.4e9 / ((double)g[2] * (waves_per_simd < 1 ? 1 : waves_per_simd));
        printf("%5d blocks x %3d threads, %6d products: %8.2f ms  s_memtime/s_memrealtime = %.4f -> %.0f MHz if s_memtime is the shader clock;"
               " %.0f wall-clock cycles @2.4 GHz per wave-product\n", g[0], g[1], g[2], ms, st / sr, st / sr * 100.0, wall_cyc);
    }
    return 0;
}
