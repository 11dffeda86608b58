// depbench.hip - does a DEPENDENT chain of v_mad_u64_u32 (a column of a big-integer product accumulating into one 64-bit
// register) issue slower than independent ones, and at how many wavefronts per SIMD does either saturate the SIMD?
// 256 blocks (one per CU) of 256 N threads = N wavefronts per SIMD, N = 1..4; per probe: cycles per wave-instruction per
// SIMD and per wavefront.   hipcc -O3 --offload-arch=gfx950 tools/microbench/depbench.hip -o tools/microbench/depbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define PROBE(name, ASM)                                                                                \
    __global__ void name(uint32_t* o, uint32_t a, uint32_t b, int iters) {                              \
        uint32_t x = a + threadIdx.x, y = b | 1;                                                        \
        uint64_t q0 = threadIdx.x, q1 = a, q2 = b, q3 = 7;                                              \
        for (int k = 0; k < iters; k++) {                                                               \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                               \
                asm volatile(ASM : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(x), "v"(y) : "vcc");   \
        }                                                                                               \
        o[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(q0 + q1 + q2 + q3);                      \
    }
PROBE(p_dep1, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0")
PROBE(p_dep2, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1")
PROBE(p_dep4, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3")
// a mad whose result feeds a 64-bit shift and an and (the carry step of a radix-2^29 column), dependent
PROBE(p_mad_shift, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_lshrrev_b64 %1, 29, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_lshrrev_b64 %0, 29, %1")
template <class K>
static void run(const char* name, K kern, uint32_t* d, int waves_per_simd) {
    const int iters = 4000, blocks = 256, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 32, cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s N=%d  %7.3f ms   %5.2f cyc per instr per wavefront   %5.2f cyc per instr per SIMD\n", name, waves_per_simd, ms, cyc / instr_per_wave,
           cyc / (instr_per_wave * waves_per_simd));
}
int main() {
    uint32_t* d;
    CK(hipMalloc(&d, 4 * 256 * 1024));
    for (int n = 1; n <= 4; n++) {
        run("mad chain, 1 accumulator", p_dep1, d, n);
        run("mad chains, 2 accumulators", p_dep2, d, n);
        run("mad chains, 4 accumulators", p_dep4, d, n);
        run("mad -> shift64 -> mad -> shift64", p_mad_shift, d, n);
    }
    return 0;
}
