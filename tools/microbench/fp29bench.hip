// fp29bench.hip - feasibility probe: BLS12-381 Fp product and square in radix 2^29 (14 limbs, Montgomery radix 2^406,
// single 64-bit column accumulator, no carry instructions) against the 12x32 product of field.hpp.
//   hipcc -O3 --offload-arch=gfx950 -I kzg_rs_amd/csrc tools/microbench/fp29bench.hip -o tools/microbench/fp29bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "field.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr uint32_t M29 = 0x1FFFFFFFu;
struct Fp29 { uint32_t l[14]; };
__constant__ uint32_t P29[14];
__constant__ uint32_t PINV29;

__device__ __forceinline__ uint32_t sreg(uint32_t v) { asm("" : "+s"(v)); return v; }

template <bool SQR>
__device__ __forceinline__ Fp29 fp29_mul(const Fp29& a, const Fp29& b) {
    uint64_t acc = 0;
    uint32_t m[14], mod[14], a2[14];
    Fp29 out;
#pragma unroll
    for (int i = 0; i < 14; i++) { mod[i] = sreg(P29[i]); if (SQR) a2[i] = a.l[i] << 1; }
    const uint32_t pinv = sreg(PINV29);
#pragma unroll
    for (int k = 0; k < 28; k++) {
        const int lo = k < 14 ? 0 : k - 13, hi = k < 14 ? k : 13;
        if (SQR) {
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                const int j = k - i;
                if (i < j) acc += (uint64_t)a.l[i] * a2[j];
                else if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
            }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k < 14 ? i < k : true) acc += (uint64_t)m[i] * mod[k - i];
        if (k < 14) {
            m[k] = ((uint32_t)acc * pinv) & M29;
            acc += (uint64_t)m[k] * mod[0];
        } else {
            out.l[k - 14] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    // column 27 leaves the top carry in acc: fold into the last limb
    out.l[13] = out.l[13] | ((uint32_t)acc << 29);
    return out;
}

template <int V>
__global__ void k_chain(uint32_t* o, int iters) {
    uint32_t s = 0;
    if (V == 0) {
        Fp x, y;
        for (int i = 0; i < 12; i++) { x.l[i] = threadIdx.x * 7 + i; y.l[i] = blockIdx.x + 3 * i + 1; }
        for (int k = 0; k < iters; k++) x = FpF::mul(x, y);
        for (int i = 0; i < 12; i++) s ^= x.l[i];
    } else {
        Fp29 x, y;
        for (int i = 0; i < 14; i++) { x.l[i] = (threadIdx.x * 7 + i) & M29; y.l[i] = (blockIdx.x + 3 * i + 1) & M29; }
        for (int k = 0; k < iters; k++) x = V == 1 ? fp29_mul<false>(x, y) : fp29_mul<true>(x, x);
        for (int i = 0; i < 14; i++) s ^= x.l[i];
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    // p in 29-bit limbs and -p^-1 mod 2^29 (values only matter for timing; computed properly anyway)
    const uint32_t p32[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    uint32_t p29[14];
    for (int i = 0; i < 14; i++) {
        int bit = 29 * i; uint64_t w = 0;
        for (int k = 0; k < 3; k++) { int idx = bit / 32 + k; if (idx < 12) w |= (idx == bit / 32 + 2) ? 0 : ((uint64_t)p32[idx] << (32 * k)); }
        p29[i] = (uint32_t)(w >> (bit % 32)) & M29;
    }
    uint32_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - p29[0] * inv;  // p^-1 mod 2^32
    uint32_t pinv = (0u - inv) & M29;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(P29), p29, sizeof p29)); CK(hipMemcpyToSymbol(HIP_SYMBOL(PINV29), &pinv, 4));
    uint32_t* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* n; void (*f)(uint32_t*, int); } ks[] = {{"Fp 12x32 mul", k_chain<0>}, {"Fp 14x29 mul", k_chain<1>}, {"Fp 14x29 sqr", k_chain<2>}};
    int cfgs[][2] = {{1024, 64}, {2048, 64}, {4096, 256}};
    for (auto& c : cfgs) {
        double wps = (double)c[0] * (c[1] / 64) / 1024.0;
        printf("--- %d blocks x %d threads = %.0f waves/SIMD\n", c[0], c[1], wps);
        for (auto& k : ks) {
            const int it = 1000;
            k.f<<<c[0], c[1]>>>(d, 10); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k.f<<<c[0], c[1]>>>(d, it); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-16s %8.3f ms   %7.1f SIMD-cycles per wave-product\n", k.n, ms, ms * 1e-3 * 2.4e9 / ((double)it * wps));
        }
    }
    return 0;
}
