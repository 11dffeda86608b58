// capi_debug.hpp - diagnostic hooks and timings.
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// diagnostic / test hook: the host-side SHA-256 used for the batch transcript (force_portable skips SHA-NI)
extern "C" int kzg_debug_host_sha256(uint8_t out[32], const uint8_t* data, size_t len, int force_portable) {
    if (force_portable) {
        uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        if (len % 64) return -1;
        for (size_t i = 0; i < len / 64; i++) hostsha::block(st, data + 64 * i);
        for (int i = 0; i < 8; i++) { out[4*i] = (uint8_t)(st[i] >> 24); out[4*i+1] = (uint8_t)(st[i] >> 16); out[4*i+2] = (uint8_t)(st[i] >> 8); out[4*i+3] = (uint8_t)st[i]; }
        return 0;
    }
    hostsha::digest(out, data, len);
    return hostsha::have_ni() ? 1 : 0;
}

// diagnostic: in-kernel shader clock (MHz) = delta s_memtime / delta s_memrealtime * 100 MHz
__global__ void k_clock_probe(unsigned long long* out, int spin) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0 + (x == 12345u);
    }
}

// diagnostic (tools/ only): run the VERIFY program on `instances` copies of zero inputs, `reps` times;
// returns the average kernel time and the in-kernel shader clock seen by a 1-block probe launched alone.
extern "C" KzgRet kzg_debug_slp_bench(float* ms_out, float* mhz_out, int instances, int reps, const KzgSettings* s) {
    if (!s || !ms_out || instances < 1) return fail(KZG_BADARGS, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    DevTmp t_in, t_out, t_clk;
    HIPCHK(hipMalloc(&t_in.p, sizeof(Fp) * 6 * instances));
    HIPCHK(hipMalloc(&t_out.p, sizeof(Fp) * 6 * instances));
    HIPCHK(hipMalloc(&t_clk.p, 16));
    Fp *d_in = t_in.as<Fp>(), *d_out = t_out.as<Fp>();
    unsigned long long* d_clk = t_clk.as<unsigned long long>();
    HIPCHK(hipMemset(d_in, 0, sizeof(Fp) * 6 * instances));
    KzgRet rc = run_program(s->verify, d_in, s->d_prep, d_out, instances, s->s1);
    if (rc != KZG_OK) return rc;
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipEventRecord(s->ev[2], s->s1));
    for (int i = 0; i < reps; i++)
        if ((rc = run_program(s->verify, d_in, s->d_prep, d_out, instances, s->s1)) != KZG_OK) return rc;
    HIPCHK(hipEventRecord(s->ev[3], s->s1));
    HIPCHK(hipEventSynchronize(s->ev[3]));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[2], s->ev[3]));
    *ms_out = ms / reps;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s->s1, d_clk, 200000);
    unsigned long long h[2];
    HIPCHK(hipMemcpyAsync(h, d_clk, 16, hipMemcpyDeviceToHost, s->s1));
    HIPCHK(hipStreamSynchronize(s->s1));
    if (mhz_out) *mhz_out = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;
    return KZG_OK;
}

// test hook (tests/test_gpu_parity.py): the sum of npts (a power of two, 2..128) Jacobian points through k_msm_sum_quads.
// points / out: the library's in-memory G1Jac - x | y | z, each 12 little-endian 32-bit words in Montgomery form (R = 2^384),
// z = 0 for the identity.
extern "C" KzgRet kzg_debug_msm_sum_quads(uint8_t out[144], const uint8_t* points, int npts, const KzgSettings* s) {
    if (!s || !out || !points || npts < 2 || npts > SUMQ_MAX_POINTS || (npts & (npts - 1))) return fail(KZG_BADARGS, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    DevTmp t_in, t_out;
    HIPCHK(hipMalloc(&t_in.p, sizeof(G1Jac) * npts));
    HIPCHK(hipMalloc(&t_out.p, sizeof(G1Jac)));
    HIPCHK(hipMemcpy(t_in.p, points, sizeof(G1Jac) * npts, hipMemcpyHostToDevice));
    HIPCHK(DYN_LDS(k_msm_sum_quads, SUMQ_LDS_BYTES));
    hipLaunchKernelGGL(k_msm_sum_quads, dim3(1), dim3(256), SUMQ_LDS_BYTES, s->s1, t_in.as<G1Jac>(), t_out.as<G1Jac>(), npts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->s1));
    HIPCHK(hipMemcpy(out, t_out.p, sizeof(G1Jac), hipMemcpyDeviceToHost));
    return KZG_OK;
}

// diagnostic: placement and duration of the first wavefront of the last latency-layout decode kernel (g_decode_dbg)
extern "C" KzgRet kzg_debug_decode_placement(unsigned long long out[4], const KzgSettings* s) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_decode_dbg), 32));
    return KZG_OK;
}

extern "C" KzgRet kzg_last_timings(const KzgSettings* s, float out_ms[8]) {
    if (!s || !out_ms) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_ms, s->timings, sizeof(float) * 8);
    return KZG_OK;
}

// The same intervals summed over every launch group finished on this handle since the last reset (count = groups):
// what bench.py averages over its timed region.  reset != 0 clears the sums after reading.
extern "C" KzgRet kzg_timing_totals(const KzgSettings* s, double out_sum_ms[8], uint64_t* count, int reset) {
    if (!s || !out_sum_ms || !count) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_sum_ms, s->tsum, sizeof(double) * 8);
    *count = s->tcount;
    for (const KzgSettings* l : s->lanes) {  // the groups that ran on the handle's pipeline lanes (capi_pipeline.hpp) count as its own
        for (int i = 0; i < 8; i++) out_sum_ms[i] += l->tsum[i];
        *count += l->tcount;
    }
    if (reset) {
        memset(s->tsum, 0, sizeof s->tsum);
        s->tcount = 0;
        for (const KzgSettings* l : s->lanes) {
            memset(l->tsum, 0, sizeof l->tsum);
            l->tcount = 0;
        }
    }
    return KZG_OK;
}

// The kernels' OWN execution intervals (in-kernel s_memrealtime stamps: first wavefront in, last wavefront out - no queueing
// behind other launch groups in them), summed over the launch groups finished on this handle and its pipeline lanes since the
// last reset, milliseconds: [0] k_blob_challenge (throughput form) [1] k_blob_evaluate [2] k_g1_decode_multiples29 (affine
// layout) [3] k_msm_window; *count = groups.  last_ms (optional): the same four of the handle's last group.
extern "C" KzgRet kzg_kernel_stamp_totals(const KzgSettings* s, double out_sum_ms[4], uint64_t* count, float last_ms[4], int reset) {
    if (!s || !out_sum_ms || !count) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    memcpy(out_sum_ms, s->kstamp_sum, sizeof(double) * 4);
    *count = s->kstamp_count;
    if (last_ms) memcpy(last_ms, s->kstamp_ms, sizeof(float) * 4);
    for (const KzgSettings* l : s->lanes) {
        for (int i = 0; i < 4; i++) out_sum_ms[i] += l->kstamp_sum[i];
        *count += l->kstamp_count;
    }
    if (reset) {
        memset(s->kstamp_sum, 0, sizeof s->kstamp_sum);
        s->kstamp_count = 0;
        for (const KzgSettings* l : s->lanes) {
            memset(l->kstamp_sum, 0, sizeof l->kstamp_sum);
            l->kstamp_count = 0;
        }
    }
    return KZG_OK;
}

// diagnostic (bench.py `valu`): the shader clock the throughput-form challenge kernel ran at since the last reset - every wave
// of it reads s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end; out = { sum of cycles, sum of ticks }
// over this handle and its pipeline lanes: MHz = 100 * out[0] / out[1].  The chip's DVFS decides this figure, and every
// cycles-per-instruction statement about the path depends on it.
extern "C" KzgRet kzg_debug_shader_clock(const KzgSettings* s, double out[2], int reset) {
    if (!s || !out) return fail(KZG_BADARGS, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    out[0] = s->clk_sum[0];
    out[1] = s->clk_sum[1];
    for (const KzgSettings* l : s->lanes) {
        out[0] += l->clk_sum[0];
        out[1] += l->clk_sum[1];
    }
    if (reset) {
        s->clk_sum[0] = s->clk_sum[1] = 0;
        for (const KzgSettings* l : s->lanes) l->clk_sum[0] = l->clk_sum[1] = 0;
    }
    return KZG_OK;
}
