// lonewavebench.hip - dev microbenchmark: ONE wavefront running a chain of dependent radix-2^29 squarings (the decode pass's
// instruction mix), launched again and again; per launch: where it ran (XCC, SE, CU, SIMD), shader cycles, wall time.
// Question: why does a lone-wave kernel take 1.9 ... 2.5 ms from one call to the next for identical work?
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/lonewavebench.hip -o tools/microbench/lonewavebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "fp29.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace kzg;

__global__ void k_chain(unsigned long long* out, int iters, int lds_every) {
    extern __shared__ uint4 dyn[];
    Fp29 a;
    for (int i = 0; i < 14; i++) a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & FP29_MASK;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
#ifdef BIG_BODY
        if (k % BIG_BODY == 0) {  // BIG_BODY squarings of straight-line code per trip (instruction-fetch footprint)
#pragma unroll
            for (int u = 0; u < BIG_BODY - 1; u++) a = fp29_sqr(a);
            k += BIG_BODY - 1;
        }
#endif
        a = fp29_sqr(a);
        if (lds_every && k % lds_every == 0) {  // park and unpark the value (conflict-free b128, like g1_29.hpp's LdsPark)
            dyn[threadIdx.x] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
            dyn[64 + threadIdx.x] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
            __builtin_amdgcn_s_waitcnt(0);
            const uint4 u = dyn[threadIdx.x], v = dyn[64 + threadIdx.x];
            a.l[0] = u.x; a.l[1] = u.y; a.l[2] = u.z; a.l[3] = u.w; a.l[4] = v.x; a.l[5] = v.y; a.l[6] = v.z; a.l[7] = v.w;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = t1 - t0;
        out[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        out[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        out[5] = __builtin_amdgcn_s_getreg((31 << 11) | 6);    // LDS_ALLOC: base [7:0], size [20:12] (64-dword granules)
    }
    if (a.l[0] == 0x12345) out[4] = a.l[1];
}
__global__ void k_nop(unsigned long long* out) { if (threadIdx.x == 999) out[5] = 1; }

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 32, nops = argc > 2 ? atoi(argv[2]) : 0, lds_kb = argc > 3 ? atoi(argv[3]) : 0, lds_every = argc > 4 ? atoi(argv[4]) : 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned long long *d, h[8];
    CK(hipMalloc(&d, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("launch  xcc se sh cu simd |  shader cycles   wall us   event ms   (nops between launches: %d)\n", nops);
    for (int l = 0; l < launches; l++) {
        for (int k = 0; k < nops; k++) k_nop<<<1, 64, 0, st>>>(d);
        CK(hipEventRecord(e0, st));
        k_chain<<<1, 64, lds_kb * 1024, st>>>(d, 2000, lds_every);
        CK(hipEventRecord(e1, st));
        CK(hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const unsigned hw = (unsigned)h[2], xcc = (unsigned)h[3] & 0xF;
        printf("%5d   %3u %2u %2u %2u %4u | %14llu %9.1f %9.3f   lds base %3u size %3u\n", l, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, h[0], h[1] / 100.0, ms,
               (unsigned)h[5] & 0xFF, ((unsigned)h[5] >> 12) & 0x1FF);
    }
    return 0;
}
