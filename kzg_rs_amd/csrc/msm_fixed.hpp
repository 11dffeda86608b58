// msm_fixed.hpp - the FIXED-BASE form of G1Projective::msm_variable_base for sums over the settings' own points
// (kzg_g1_msm_setup, capi_pieces.hpp): term i = scalar_i x g1_points[i mod N], N = 4 096 Lagrange points the handle holds
// decoded (src/trusted_setup.rs:20-26; call sites src/kzg_proof.rs:419,429,430; BASELINE.json configs[3]).
//
// The variable-base kernels (msm.hpp) build 4 table rows per point per call and spend 32 bucket additions per term (GLV, 8-bit
// windows: 4 chunks x 8 windows).  With the base points fixed, the table is made ONCE per handle and can afford a row for every
// 16-bit window: rows[v][j] = 2^(16 v) P_j, v < 16, affine, 128 B each = 8 MB for 4 096 points (+ as much again for the doubled rows of the rare digit 2^15; MALL resident).  Then
//   * no GLV, no per-call decode, no subgroup requirement;
//   * a term is 16 signed 16-bit digits d_v in [-(2^15 - 1), 2^15] -> 16 bucket additions per term, HALF of the variable-base form;
//   * ALL windows share ONE set of 2^15 buckets (bucket b collects +-2^(16 v) P_j for every (term, v) with |d_v| = b), so there is
//     no Horner chain over windows and one bucket reduction per call instead of one per window.
// 32 768 buckets do not fit a workgroup's lanes, so the (term, window) entries are first PARTITIONED in HBM by the high bits of
// the bucket number (128 partitions of 256 buckets: a counting sort, 4 B per entry), and a workgroup then takes a slice of at most
// L entries of ONE partition, sorts it by the low 8 bits in LDS and accumulates one bucket per lane in registers with mixed
// additions - the LDS-sorted-list scheme of k_msm_window (msm.hpp), fed from the partition instead of from the scalars' bytes.
//   sum_b b B_b  with  b = 256 p + l :   sum_l l C_l + 256 sum_p p R_p,   C_l = sum over all workgroups of bucket l,  R_p = sum of
//   everything partition p's workgroups accumulated - two 256-bucket reductions, run by the large-sum tail of msm.hpp
//   (k_msm_bucket_fold, k_msm_bucket_sum_quads, k_msm_reduce_quads) as TWO "windows" 8 bits apart, joined by k_msm_combine_quad.
// Part of the single translation unit kzg_capi.hip (after msm.hpp).
#pragma once

namespace kzg {

constexpr int FBM_WINDOWS = 16;           // 16-bit windows of a 255-bit scalar
constexpr int FBM_PARTS = 128;            // |digit| >> 8 for |digit| <= 2^15 - 1 (a digit of exactly 2^15 travels as two entries of 2^14, below)
constexpr int FBM_SLICE_ENTRIES = 12288;  // entries a workgroup sorts in LDS (48 KB: three workgroups per CU, like k_msm_window)
constexpr uint32_t FBM_ENTRY_ROW_MASK = 0x1ffffu, FBM_ENTRY_NEG = 0x20000u;  // entry = low byte of |digit| << 24 | negative << 17 | row (v N + j; doubled rows from 16 N on)
// device-side plan of one call: entries per partition | their exclusive scan (+ total) | first workgroup of
// each partition (+ total workgroups) | then the scatter's cursors
constexpr int FBM_PLAN_COUNT = 0, FBM_PLAN_OFF = FBM_PARTS, FBM_PLAN_BLK = 2 * FBM_PARTS + 1, FBM_PLAN_CUR = 3 * FBM_PARTS + 2, FBM_PLAN_WORDS = 4 * FBM_PARTS + 2;

// big-endian scalars of any value < 2^256 -> canonical little-endian limbs (Scalar::from_raw semantics: reduced mod r; 2^256 < 3 r)
__global__ __launch_bounds__(256) void k_scalars_reduce_be(const uint8_t* __restrict__ be, Fr* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(be) + 2 * (size_t)i;
    Fr v = fr_from_be_words(src[0], src[1]);
    v = FrF::reduce_once(v);
    v = FrF::reduce_once(v);
    out[i] = v;
}

// the 16 signed digits d_v in [-(2^15 - 1), 2^15] of a canonical scalar, least significant first: f(v, |d_v|, d_v < 0, doubled row) for every non-zero digit
template <class F>
__device__ __forceinline__ void fb_digits(const Fr& k, F&& f) {
    uint32_t carry = 0;
#pragma unroll
    for (int v = 0; v < FBM_WINDOWS; v++) {
        const uint32_t x = ((k.l[v >> 1] >> (16 * (v & 1))) & 0xffffu) + carry;
        carry = x > 32768u ? 1u : 0u;
        const uint32_t mag = carry ? 65536u - x : x;
        // |d| = 2^15 (one digit in 65 536) would be the ONLY magnitude of a 129th partition: ~n / 4 096 entries in one bucket, added by
        // one lane one after the other - at 2^20 terms a 4 ms straggler behind a 4 ms kernel (profiles/r6_fb_window_timeline.txt).
        // It travels as 2^14 x the DOUBLED row 2^(16 v + 1) P_j instead (rows 16 N ..: one more row per window and point).
        if (mag == 32768u) f(v, 16384u, false, true);
        else if (mag) f(v, mag, carry != 0, false);
    }
    // (k < r < 2^255: the top window is below 2^15, so no carry leaves it)
}

constexpr int FBM_TERMS_PER_BLOCK = 1024;
// pass 1a: entries per partition.  pflag[j] != 0: point j is the identity (unchecked decode, build.rs:66-70) - its terms add nothing
__global__ __launch_bounds__(256) void k_fb_count(const Fr* __restrict__ scalars, const uint32_t* __restrict__ pflag, int n, int npoints,
                                                  uint32_t* __restrict__ plan) {
    __shared__ uint32_t h[FBM_PARTS];
    for (int i = threadIdx.x; i < FBM_PARTS; i += 256) h[i] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * FBM_TERMS_PER_BLOCK;
    for (int t = t0 + threadIdx.x; t < min(n, t0 + FBM_TERMS_PER_BLOCK); t += 256) {
        if (pflag[t % npoints]) continue;
        const Fr k = scalars[t];
        fb_digits(k, [&](int, uint32_t mag, bool, bool) { atomicAdd(&h[mag >> 8], 1u); });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FBM_PARTS; i += 256)
        if (h[i]) atomicAdd(&plan[FBM_PLAN_COUNT + i], h[i]);
}
// pass 1b (one workgroup): offsets of the partitions, their workgroups (slices of at most L entries), the scatter's cursors
__global__ __launch_bounds__(64) void k_fb_plan(uint32_t* __restrict__ plan, int L) {
    if (threadIdx.x) return;
    uint32_t off = 0, blk = 0;
    for (int p = 0; p < FBM_PARTS; p++) {
        const uint32_t c = plan[FBM_PLAN_COUNT + p];
        plan[FBM_PLAN_OFF + p] = off;
        plan[FBM_PLAN_CUR + p] = off;
        plan[FBM_PLAN_BLK + p] = blk;
        off += c;
        blk += (c + (uint32_t)L - 1) / (uint32_t)L;
    }
    plan[FBM_PLAN_OFF + FBM_PARTS] = off;
    plan[FBM_PLAN_BLK + FBM_PARTS] = blk;
}
// pass 1c: the entries, partition by partition (order inside a partition: whatever the workgroups' reservations make it - a sum
// does not care).  A workgroup counts its terms' entries per partition in LDS, reserves one run per partition, and fills the runs.
__global__ __launch_bounds__(256) void k_fb_scatter(const Fr* __restrict__ scalars, const uint32_t* __restrict__ pflag, int n, int npoints,
                                                    uint32_t* __restrict__ plan, uint32_t* __restrict__ entries) {
    __shared__ uint32_t h[FBM_PARTS], base[FBM_PARTS];
    for (int i = threadIdx.x; i < FBM_PARTS; i += 256) h[i] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * FBM_TERMS_PER_BLOCK, t1 = min(n, t0 + FBM_TERMS_PER_BLOCK);
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        if (pflag[t % npoints]) continue;
        const Fr k = scalars[t];
        fb_digits(k, [&](int, uint32_t mag, bool, bool) { atomicAdd(&h[mag >> 8], 1u); });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FBM_PARTS; i += 256) {
        base[i] = h[i] ? atomicAdd(&plan[FBM_PLAN_CUR + i], h[i]) : 0u;
        h[i] = 0;
    }
    __syncthreads();
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        const int j = t % npoints;
        if (pflag[j]) continue;
        const Fr k = scalars[t];
        fb_digits(k, [&](int v, uint32_t mag, bool neg, bool doubled) {
            const uint32_t p = mag >> 8, pos = base[p] + atomicAdd(&h[p], 1u);
            entries[pos] = (mag & 255u) << 24 | (neg ? FBM_ENTRY_NEG : 0u) | (uint32_t)(((doubled ? FBM_WINDOWS : 0) + v) * npoints + j);
        });
    }
}

// table build, once per handle: jac[(v - 1) N + j] = 2^(16 v) P_j for v = 1 .. 15, jac[(15 + v) N + j] = 2^(16 v + 1) P_j for v = 0 .. 15
// (k_jac29_to_aff29 turns them into rows 1 .. 31; row 0 = P_j is the setup's affine table row)
__global__ __launch_bounds__(64) void k_fb_build_rows(const G1Aff29Mem* __restrict__ row0, const uint32_t* __restrict__ pflag, G1Jac29Mem* __restrict__ jac, int npoints) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npoints) return;
    G1Jac29 p;
    if (pflag[j]) {
        p = g1j29_identity();
        p.z = fp29_const(cp29::FP29_ONE);  // (never read: a finite stand-in keeps the conversion's inversion defined)
    } else {
        const G1Aff29 a = g1a29_load(row0[j]);
        p.x = a.x;
        p.y = a.y;
        p.z = fp29_const(cp29::FP29_ONE);
    }
#pragma unroll 1
    for (int v = 0; v < FBM_WINDOWS; v++) {
        p = g1j29_dbl(p);
        g1j29_store(jac[(size_t)(FBM_WINDOWS - 1 + v) * npoints + j], p);  // 2^(16 v + 1) P
        if (v + 1 == FBM_WINDOWS) break;
#pragma unroll 1
        for (int k = 1; k < 16; k++) p = g1j29_dbl(p);
        g1j29_store(jac[(size_t)v * npoints + j], p);  // 2^(16 (v + 1)) P
    }
}

// pass 2: workgroup blk = slice s of partition p: at most L entries, sorted by the low byte of the bucket number in LDS, one bucket
// per lane (handed out by decreasing size), mixed additions of table rows with the digit's sign.  Leaves its 256 bucket sums in
// save slot blk (point-major 48-word records: the format k_msm_bucket_fold reads).  Workgroups beyond the plan's total leave identities.
__global__ __launch_bounds__(256, KZG_MSM_OCC) void k_fb_window(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ plan,
                                                                const G1Aff29Mem* __restrict__ rows, uint32_t* __restrict__ save, int L,
                                                                unsigned long long* __restrict__ ktime) {
    using CV = Curve29Aff;
    using Pt = typename CV::Pt;
    kstamp_in(ktime);
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    __shared__ uint32_t cnt[MSM_BUCKETS], off[MSM_BUCKETS + 1], cur[MSM_BUCKETS];
    extern __shared__ __attribute__((aligned(16))) uint32_t fb_lst[];
    uint32_t* const out = save + (size_t)blk * MSM_SAVE2_WORDS * 256;
    auto put = [&](int col, const Pt& p) {
        uint4* const o4 = reinterpret_cast<uint4*>(out + (size_t)col * MSM_SAVE2_WORDS);
        const Fp29* const c[3] = {&p.x, &p.y, &p.z};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            o4[4 * j] = make_uint4(c[j]->l[0], c[j]->l[1], c[j]->l[2], c[j]->l[3]);
            o4[4 * j + 1] = make_uint4(c[j]->l[4], c[j]->l[5], c[j]->l[6], c[j]->l[7]);
            o4[4 * j + 2] = make_uint4(c[j]->l[8], c[j]->l[9], c[j]->l[10], c[j]->l[11]);
            o4[4 * j + 3] = make_uint4(c[j]->l[12], c[j]->l[13], 0u, 0u);
        }
    };
    const uint32_t* const bstart = plan + FBM_PLAN_BLK;
    if (blk >= bstart[FBM_PARTS]) {
        put(tid, CV::identity());
        kstamp_out(ktime);
        return;
    }
    int p = 0;  // the partition whose workgroups include blk: the last p with bstart[p] <= blk (uniform over the workgroup)
    for (int step = 128; step >= 1; step >>= 1)
        if (p + step < FBM_PARTS && bstart[p + step] <= blk) p += step;
    // the partition's entries in EQUAL slices (its workgroup count was fixed by k_fb_plan: ceil(entries / L)): no short last slice
    const uint32_t c = plan[FBM_PLAN_OFF + p + 1] - plan[FBM_PLAN_OFF + p], ns = bstart[p + 1] - bstart[p], len = (c + ns - 1) / ns, sl = blk - bstart[p];
    const uint32_t e0 = plan[FBM_PLAN_OFF + p] + sl * len;
    const uint32_t m = sl * len < c ? min(len, c - sl * len) : 0u;
    const uint32_t* const src = entries + e0;
    cnt[tid] = 0;
    cur[tid] = 0;
    __syncthreads();
    for (uint32_t t = tid; t < m; t += 256) atomicAdd(&cnt[src[t] >> 24], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t s = 0;
        for (int b = 0; b < MSM_BUCKETS; b++) {
            off[b] = s;
            s += cnt[b];
        }
        off[MSM_BUCKETS] = s;
    }
    __syncthreads();
    for (uint32_t t = tid; t < m; t += 256) {
        const uint32_t e = src[t], dig = e >> 24;
        fb_lst[off[dig] + atomicAdd(&cur[dig], 1u)] = e & (FBM_ENTRY_NEG | FBM_ENTRY_ROW_MASK);
    }
    __threadfence_block();
    __syncthreads();
    {  // buckets by decreasing size: wavefront 0 takes the 64 fullest (a wavefront runs as long as its fullest bucket)
        const uint32_t mine = cnt[tid];
        uint32_t rank = 0;
        for (int b = 0; b < MSM_BUCKETS; b++) {
            const uint32_t c = cnt[b];
            rank += (c > mine) | ((c == mine) & (b < tid));
        }
        cur[rank] = tid;
    }
    __syncthreads();
    // Which SIMD of the CU gets the wavefront with the fullest buckets rotates from workgroup to workgroup: a workgroup's four
    // wavefronts go to the CU's four SIMDs in order, and without the rotation SIMD 0 of every CU collects every resident workgroup's
    // longest lists (68 entries against 42 on SIMD 3).  Workgroups reach an XCD round-robin by id and a CU in turn, so neither
    // blk & 3 nor (blk >> 3) & 3 varies between the workgroups that share a CU: a multiplicative hash of the id does.
    const int bucket = cur[(tid + 64 * (int)((blk * 0x9E3779B1u) >> 30)) & 255];
    auto row_of = [&](uint32_t e) -> typename CV::Entry {
        typename CV::Entry q = CV::load(rows[e & FBM_ENTRY_ROW_MASK]);
        const Fp29 ny = fp29_neg<3>(q.y);  // rows hold y < 4p
        const bool neg = (e & FBM_ENTRY_NEG) != 0;
#pragma unroll
        for (int i = 0; i < 14; i++) q.y.l[i] = neg ? ny.l[i] : q.y.l[i];
        return q;
    };
    uint32_t k = off[bucket];
    const uint32_t kend = off[bucket + 1];
    Pt a = CV::identity();
    const uint32_t first = k + 1;
    uint32_t wr = k;
    if (k < kend) {  // the first entry is a copy, not an addition to the identity
        a = CV::from_entry(row_of(fb_lst[k++]));
        wr = k;
    }
    for (; k < kend; k++) {
        const uint32_t e = fb_lst[k];
        const typename CV::Entry q = row_of(e);
        const typename CV::EntryHead h = CV::entry_head(a, q);
        if (CV::entry_special(h)) fb_lst[wr++] = e;  // same x (P + P, P - P) or an accumulator at infinity: the complete formula, below
        else a = CV::entry_tail(a, q, h);
    }
    for (uint32_t j = first; j < wr; j++) a = CV::add_entry(a, row_of(fb_lst[j]));  // rare
    put(bucket, a);
    kstamp_out(ktime);
}

// R_p = the sum of everything partition p's workgroups accumulated (all 256 buckets of all its slices), as bucket p of a second
// 256-bucket "window" (48-word records, out[256]); p >= 128: the identity.  One workgroup per p: lane l sums bucket l over the
// partition's slices, then a tree over the lanes.  (Complete additions: empty buckets and equal points are ordinary here.)
__global__ __launch_bounds__(256) void k_fb_rowsum(const uint32_t* __restrict__ save, const uint32_t* __restrict__ plan, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t pts[256 * SUMQ_POINT_WORDS];
    const int p = blockIdx.x, tid = threadIdx.x;
    if (p >= FBM_PARTS) {
        if (tid == 0) save2_store(out + (size_t)p * SUMQ_POINT_WORDS, g1j29_identity());
        return;
    }
    const uint32_t b0 = plan[FBM_PLAN_BLK + p], b1 = plan[FBM_PLAN_BLK + p + 1];
    G1Jac29 acc = g1j29_identity();
#pragma unroll 1
    for (uint32_t b = b0; b < b1; b++) acc = g1j29_add(acc, save2_load(save + ((size_t)b * 256 + tid) * MSM_SAVE2_WORDS));
    save2_store(pts + tid * SUMQ_POINT_WORDS, acc);
    __syncthreads();
#pragma unroll 1
    for (int half = 128; half >= 1; half >>= 1) {
        if (tid < half) {
            const G1Jac29 x = save2_load(pts + tid * SUMQ_POINT_WORDS), y = save2_load(pts + (tid + half) * SUMQ_POINT_WORDS);
            save2_store(pts + tid * SUMQ_POINT_WORDS, g1j29_add(x, y));
        }
        __syncthreads();
    }
    if (tid < SUMQ_POINT_WORDS / 4) reinterpret_cast<uint4*>(out + (size_t)p * SUMQ_POINT_WORDS)[tid] = reinterpret_cast<const uint4*>(pts)[tid];
}

// device memory the tail wants beside the save area: flags [gp] | partial sums [256][gp] | buckets [2][256] | window sums [2] (G1Jac)
constexpr size_t fb_tail_bytes(int gp) { return 4 * ((size_t)MSM_FOLD_MAX_GROUPS + (size_t)256 * gp * MSM_SAVE2_WORDS + 2 * 256 * MSM_SAVE2_WORDS) + 2 * sizeof(G1Jac) + 256; }
// workgroups of pass 2 for n_entries entries at most (every partition may end in a part-filled slice)
inline unsigned fb_max_blocks(size_t n_entries, int L) { return (unsigned)(FBM_PARTS + (n_entries + (size_t)L - 1) / (size_t)L); }

// Host side: scalars (canonical limbs, n of them) -> out_ab[0] = the sum (Jacobian, 12x32 form).  `entries`: 16 n words; `save`:
// fb_max_blocks x 48 KB; `tmp`: fb_tail_bytes(gp); plan: FBM_PLAN_WORDS words.  Everything on `st`.
inline hipError_t fb_msm_launch(const Fr* scalars, const uint32_t* pflag, int n, int npoints, const G1Aff29Mem* rows, uint32_t* plan, uint32_t* entries,
                                uint32_t* save, uint8_t* tmp, G1Jac* out_ab, int L, int fold_per_min, unsigned long long* ktime, hipStream_t st,
                                hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr) {
    const unsigned nb = (unsigned)((n + FBM_TERMS_PER_BLOCK - 1) / FBM_TERMS_PER_BLOCK);
    hipError_t e = hipMemsetAsync(plan, 0, 4 * FBM_PLAN_WORDS, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fb_count, dim3(nb), dim3(256), 0, st, scalars, pflag, n, npoints, plan);
    hipLaunchKernelGGL(k_fb_plan, dim3(1), dim3(64), 0, st, plan, L);
    hipLaunchKernelGGL(k_fb_scatter, dim3(nb), dim3(256), 0, st, scalars, pflag, n, npoints, plan, entries);
    const unsigned Z = fb_max_blocks((size_t)FBM_WINDOWS * n, L);
    if ((e = DYN_LDS(k_fb_window, 4 * (size_t)L + 16)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fb_window, dim3(Z), dim3(256), 4 * (size_t)L + 16, st, (const uint32_t*)entries, (const uint32_t*)plan, rows, save, L, ktime);
    // C_l: every workgroup's bucket l, folded layer by layer (groups of `per` layers per thread, then a tree with four lanes per addition)
    const int per = std::max(fold_per_min, (int)((Z + MSM_FOLD_MAX_GROUPS - 1) / MSM_FOLD_MAX_GROUPS));
    int gp;
    const int groups = msm_large_tail_groups(Z, per, &gp);
    uint32_t* const flags = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* const part = flags + MSM_FOLD_MAX_GROUPS;
    uint32_t* const buckets = part + (size_t)256 * gp * MSM_SAVE2_WORDS;  // [2][256] records: C | R
    G1Jac* const wsums = reinterpret_cast<G1Jac*>(reinterpret_cast<uint8_t*>(buckets + (size_t)2 * 256 * MSM_SAVE2_WORDS) + 64);
    // R_p (k_fb_rowsum: a chain of ~19 additions per lane, 128 workgroups) beside the fold of C on a second stream when the caller has one
    const bool forked = side && side != st && ev_fork && ev_join;
    if (forked) {
        if ((e = hipEventRecord(ev_fork, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side, ev_fork, 0)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_fb_rowsum, dim3(256), dim3(256), 0, forked ? side : st, (const uint32_t*)save, (const uint32_t*)plan, buckets + (size_t)256 * MSM_SAVE2_WORDS);
    if (forked && (e = hipEventRecord(ev_join, side)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(flags, 0, 4 * (size_t)gp, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_bucket_fold<false>, dim3(gp), dim3(256), 0, st, (const uint32_t*)save, part, flags, 1, (int)Z, per, groups, gp);
    hipLaunchKernelGGL(k_msm_bucket_fold<true>, dim3(gp), dim3(256), 0, st, (const uint32_t*)save, part, flags, 1, (int)Z, per, groups, gp);
    if ((e = DYN_LDS(k_msm_bucket_sum_quads, msm_bucket_sum_lds_bytes(MSM_FOLD_MAX_GROUPS))) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_bucket_sum_quads, dim3(256), dim3(64), msm_bucket_sum_lds_bytes(gp), st, (const uint32_t*)part, buckets, gp);
    if (forked && (e = hipStreamWaitEvent(st, ev_join, 0)) != hipSuccess) return e;
    if ((e = DYN_LDS(k_msm_reduce_quads, MSM_REDQ_LDS_BYTES)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_msm_reduce_quads, dim3(2), dim3(256), MSM_REDQ_LDS_BYTES, st, (const uint32_t*)buckets, wsums);
    hipLaunchKernelGGL(k_msm_combine_quad, dim3(1), dim3(64), 0, st, (const G1Jac*)wsums, out_ab, 2);  // sum_l l C_l + 2^8 sum_p p R_p
    return hipGetLastError();
}

}  // namespace kzg
