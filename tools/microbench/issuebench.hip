// issuebench.hip - dev microbenchmark: VALU issue cost per wave-instruction on gfx950, saturated SIMDs
// (4096 blocks x 256 threads = 16 waves per SIMD) and a lone wave per SIMD (1024 x 64).  Each probe is 4 independent
// chains of one opcode, 32 instructions per loop iteration.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/issuebench.hip -o /tmp/issuebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define PROBE(name, ASM4, ...)                                                                          \
    __global__ void name(uint32_t* o, uint32_t a, uint32_t b, int iters) {                              \
        uint32_t r0 = threadIdx.x, r1 = a, r2 = b, r3 = a ^ b, x = a + threadIdx.x, y = b | 1;          \
        uint64_t q0 = threadIdx.x, q1 = a, q2 = b, q3 = 7;                                              \
        for (int k = 0; k < iters; k++) {                                                               \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                               \
                asm volatile(ASM4 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) \
                             : "v"(x), "v"(y) : __VA_ARGS__);                                                \
        }                                                                                               \
        o[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + (uint32_t)(q0 + q1 + q2 + q3);  \
    }
// operands: %0-%3 32-bit regs, %4-%7 64-bit regs, %8 x, %9 y
#define R4(op) op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8"
#define R4_3(op) op " %0, %0, %8, %9\n\t" op " %1, %1, %8, %9\n\t" op " %2, %2, %8, %9\n\t" op " %3, %3, %8, %9"
PROBE(p_add_u32, R4("v_add_u32"), "memory")
PROBE(p_xor_b32, R4("v_xor_b32"), "memory")
PROBE(p_and_b32, R4("v_and_b32"), "memory")
PROBE(p_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %1, 3, %1\n\tv_lshrrev_b32 %2, 3, %2\n\tv_lshrrev_b32 %3, 3, %3", "memory")
PROBE(p_add3_u32, R4_3("v_add3_u32"), "memory")
PROBE(p_alignbit, "v_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %1, %1, %1, 7\n\tv_alignbit_b32 %2, %2, %2, 7\n\tv_alignbit_b32 %3, %3, %3, 7", "memory")
PROBE(p_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96", "memory")
PROBE(p_xad_u32, R4_3("v_xad_u32"), "memory")
PROBE(p_and_or, R4_3("v_and_or_b32"), "memory")
PROBE(p_lshl_add, "v_lshl_add_u32 %0, %0, 3, %8\n\tv_lshl_add_u32 %1, %1, 3, %8\n\tv_lshl_add_u32 %2, %2, 3, %8\n\tv_lshl_add_u32 %3, %3, 3, %8", "memory")
PROBE(p_perm, R4_3("v_perm_b32"), "memory")
PROBE(p_mul_lo, R4("v_mul_lo_u32"), "memory")
PROBE(p_mul_hi, R4("v_mul_hi_u32"), "memory")
PROBE(p_mad_u32_u24, R4_3("v_mad_u32_u24"), "memory")
PROBE(p_mul_u32_u24, R4("v_mul_u32_u24"), "memory")
PROBE(p_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7", "vcc")
PROBE(p_mad_u64_sgprcarry, "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[22:23], %8, %9, %5\n\tv_mad_u64_u32 %6, s[24:25], %8, %9, %6\n\tv_mad_u64_u32 %7, s[26:27], %8, %9, %7", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
PROBE(p_add_co, "v_add_co_u32 %0, vcc, %0, %8\n\tv_add_co_u32 %1, vcc, %1, %8\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_add_co_u32 %3, vcc, %3, %8", "vcc")
PROBE(p_addc_co, "v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc", "vcc")
PROBE(p_lshrrev_b64, "v_lshrrev_b64 %4, 29, %4\n\tv_lshrrev_b64 %5, 29, %5\n\tv_lshrrev_b64 %6, 29, %6\n\tv_lshrrev_b64 %7, 29, %7", "memory")
PROBE(p_lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n\tv_lshlrev_b64 %5, 3, %5\n\tv_lshlrev_b64 %6, 3, %6\n\tv_lshlrev_b64 %7, 3, %7", "memory")
PROBE(p_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc", "memory")
PROBE(p_mov, "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %9\n\tv_mov_b32 %3, %9", "memory")
PROBE(p_bfe, "v_bfe_u32 %0, %0, 3, 29\n\tv_bfe_u32 %1, %1, 3, 29\n\tv_bfe_u32 %2, %2, 3, 29\n\tv_bfe_u32 %3, %3, 3, 29", "memory")
PROBE(p_alignbit64, "v_alignbit_b32 %0, %1, %0, 29\n\tv_alignbit_b32 %1, %2, %1, 29\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_alignbit_b32 %3, %0, %3, 29", "memory")
PROBE(p_pk_add_u16, R4("v_pk_add_u16"), "memory")
PROBE(p_fma_f64, "v_fma_f64 %4, %4, %5, %6\n\tv_fma_f64 %5, %5, %6, %7\n\tv_fma_f64 %6, %6, %7, %4\n\tv_fma_f64 %7, %7, %4, %5", "memory")
PROBE(p_fma_f32, R4_3("v_fma_f32"), "memory")
PROBE(p_mad_i32_i24, R4_3("v_mad_i32_i24"), "memory")
PROBE(p_dot4_u8, "v_dot4_u32_u8 %0, %8, %9, %0\n\tv_dot4_u32_u8 %1, %8, %9, %1\n\tv_dot4_u32_u8 %2, %8, %9, %2\n\tv_dot4_u32_u8 %3, %8, %9, %3", "memory")
PROBE(p_add_dpp_quad, "v_add_u32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
PROBE(p_add_dpp_shr4, "v_add_u32_dpp %0, %8, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %1, %8, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %2, %8, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_u32_dpp %3, %8, %9 row_shl:4 row_mask:0xf bank_mask:0x5", "memory")
PROBE(p_add_dpp_ident, "v_add_u32_dpp %0, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %1, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %2, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %3, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa", "memory")
// dependent chains: every instruction consumes the previous result (what one SHA-256 round looks like)
PROBE(p_chain_add, "v_add_u32 %0, %0, %8\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %0, %0, %9", "memory")
PROBE(p_chain_alignbit_bitop, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_add3_u32 %0, %0, %1, %2", "memory")
// the 10-instruction two-lane SHA-256 round of sha256.hpp (x0 = %0, temporaries %1 %2 %3; the other state words fixed)
PROBE(p_sha_round10, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32_dpp %3, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %0, %9, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5", "memory")
// the 11-instruction form before it (select + one DPP add, two filler instructions for the hazard)
PROBE(p_sha_round11, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_cndmask_b32 %2, %1, %9, vcc\n\tv_and_b32 %3, %8, %9\n\tv_add_u32 %3, %3, %9\n\tv_add_u32_dpp %0, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
// variants of the round's tail, to see what the DPP additions cost on the dependent chain
PROBE(p_sha_tail_plain, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32 %3, %8, %9\n\tv_add_u32 %0, %9, %1\n\tv_add_u32 %0, %1, %0", "memory")
PROBE(p_sha_tail_none, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32 %0, %9, %1", "memory")
PROBE(p_sha_tail_onedpp, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32 %3, %8, %9\n\tv_add_u32 %2, %9, %1\n\tv_add_u32_dpp %0, %1, %2 row_shl:4 row_mask:0xf bank_mask:0xf", "memory")
PROBE(p_sha_tail_twodpp_indep, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32 %3, %8, %9\n\tv_add_u32_dpp %2, %9, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf", "memory")
// the committed order: the two bank-masked writes of the next x0 with the hk addition between them
PROBE(p_sha_round10_split, "v_alignbit_b32 %1, %0, %0, %8\n\tv_alignbit_b32 %2, %0, %0, %9\n\tv_alignbit_b32 %3, %0, %0, %8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n\t"
      "v_bitop3_b32 %2, %0, %8, %9 bitop3:0xd2\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n\tv_add3_u32 %1, %1, %2, %8\n\t"
      "v_add_u32_dpp %0, %9, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_add_u32_dpp %3, %8, %9 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\t"
      "v_add_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5", "memory")

typedef void (*probe_t)(uint32_t*, uint32_t, uint32_t, int);
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    struct { const char* n; probe_t f; } probes[] = {
        {"v_add_u32", p_add_u32}, {"v_xor_b32", p_xor_b32}, {"v_and_b32", p_and_b32}, {"v_lshrrev_b32", p_lshrrev_b32}, {"v_mov_b32", p_mov},
        {"v_cndmask_b32", p_cndmask}, {"v_add3_u32", p_add3_u32}, {"v_alignbit_b32", p_alignbit}, {"v_alignbit_b32 (64b shift)", p_alignbit64},
        {"v_bitop3_b32", p_bitop3}, {"v_xad_u32", p_xad_u32}, {"v_and_or_b32", p_and_or}, {"v_lshl_add_u32", p_lshl_add}, {"v_bfe_u32", p_bfe},
        {"v_perm_b32", p_perm}, {"v_mul_lo_u32", p_mul_lo}, {"v_mul_hi_u32", p_mul_hi}, {"v_mul_u32_u24", p_mul_u32_u24}, {"v_mad_u32_u24", p_mad_u32_u24},
        {"v_mad_i32_i24", p_mad_i32_i24}, {"v_mad_u64_u32 (vcc)", p_mad_u64_u32}, {"v_mad_u64_u32 (sgpr pair)", p_mad_u64_sgprcarry},
        {"v_add_co_u32", p_add_co}, {"v_addc_co_u32", p_addc_co}, {"v_lshrrev_b64", p_lshrrev_b64}, {"v_lshlrev_b64", p_lshlrev_b64},
        {"v_pk_add_u16", p_pk_add_u16}, {"v_fma_f32", p_fma_f32}, {"v_fma_f64", p_fma_f64}, {"v_dot4_u32_u8", p_dot4_u8},
        {"v_add_u32_dpp quad_perm", p_add_dpp_quad}, {"v_add_u32_dpp row_shr/shl:4 bank-masked", p_add_dpp_shr4}, {"v_add_u32_dpp identity bank-masked", p_add_dpp_ident},
        {"dependent v_add_u32 chain", p_chain_add}, {"dependent alignbit/bitop3/add3 chain", p_chain_alignbit_bitop},
        {"SHA round, 10 instr, masked writes adjacent", p_sha_round10}, {"SHA round, 11 instr", p_sha_round11}, {"SHA round, 10 instr, masked writes apart", p_sha_round10_split},
        {"SHA round, tail = 3 plain adds", p_sha_tail_plain}, {"SHA round, tail = 1 add (8 instr)", p_sha_tail_none}, {"SHA round, tail = 2 adds + 1 DPP add", p_sha_tail_onedpp},
        {"SHA round, tail = add + 2 DPP adds, separate dst", p_sha_tail_twodpp_indep},
    };
    for (int cfg = 0; cfg < 2; cfg++) {
        int blocks = cfg == 0 ? 4096 : 1024, threads = cfg == 0 ? 256 : 64;
        double waves_per_simd = (double)blocks * (threads / 64) / 1024.0;
        printf("--- %d blocks x %d threads = %.0f waves/SIMD: cycles per wave-instruction per SIMD (2.4 GHz)\n", blocks, threads, waves_per_simd);
        size_t n = (size_t)blocks * threads;
        uint32_t* d; CK(hipMalloc(&d, n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int it = cfg == 0 ? 4000 : 20000;
        for (auto& pr : probes) {
            pr.f<<<blocks, threads>>>(d, 3, 5, 10); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); pr.f<<<blocks, threads>>>(d, 3, 5, it); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double cyc = ms * 1e-3 * 2.4e9 / ((double)it * 32 * waves_per_simd);
            printf("%-28s %8.3f ms  %6.2f cyc\n", pr.n, ms, cyc);
        }
        CK(hipFree(d));
    }
    return 0;
}
