// capi_glue_kernels.hpp - glue kernels of the C ABI: byte <-> point conversions, term tables, record packing (included by kzg_capi.hip).
// Part of the single translation unit kzg_capi.hip; not a stand-alone header.

// ---------------------------------------------------------------- small kernels
// points [0, n0) come from bytes0, [n0, n) from bytes1 (commitments then proofs in one launch)
__global__ __launch_bounds__(64) void k_g1_decode(const uint8_t* __restrict__ bytes0, const uint8_t* __restrict__ bytes1, int n0,
                                                  G1Aff* __restrict__ out, uint32_t* __restrict__ flag, int n, int check_subgroup) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = i < n0 ? bytes0 + (size_t)i * 48 : bytes1 + (size_t)(i - n0) * 48;
    G1Aff a;
    uint32_t st = g1_decompress(a, src, check_subgroup != 0);
    out[i] = a;
    flag[i] = st;
}

__global__ void k_g2_decompress(const uint8_t* __restrict__ bytes, Fp* __restrict__ out4, uint32_t* __restrict__ flag) {
    if (threadIdx.x || blockIdx.x) return;
    G2Aff q;
    uint32_t st = g2_decompress(q, bytes);
    out4[0] = q.x.c0;
    out4[1] = q.x.c1;
    out4[2] = q.y.c0;
    out4[3] = q.y.c1;
    *flag = st;
}

// n G2 points, one per workgroup (trusted-setup load: build.rs:73, from_compressed_unchecked)
__global__ void k_g2_decompress_n(const uint8_t* __restrict__ bytes, Fp* __restrict__ out4, uint32_t* __restrict__ flag) {
    if (threadIdx.x) return;
    const int i = blockIdx.x;
    G2Aff q;
    q.x.c0 = q.x.c1 = q.y.c0 = q.y.c1 = FpF::zero();
    uint32_t st = g2_decompress(q, bytes + 96 * (size_t)i);
    out4[4 * i] = q.x.c0;
    out4[4 * i + 1] = q.x.c1;
    out4[4 * i + 2] = q.y.c0;
    out4[4 * i + 3] = q.y.c1;
    flag[i] = st;
}

// affine points -> 48 compressed bytes (flag != 0: the identity encoding)
__global__ void k_aff_compress(const G1Aff* __restrict__ pts, const uint32_t* __restrict__ flag, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_compress(out + 48 * (size_t)i, pts[i], flag[i] != 0);
}

// blob bytes -> MSM scalars: element i of blob b (32 big-endian bytes) as plain little-endian limbs; status[b] |= 1
// when an element is >= r (src/dtypes.rs:48-57)
__global__ void k_blob_scalars(const uint8_t* __restrict__ blobs, Fr* __restrict__ scalars, uint32_t* __restrict__ status, int total) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint4* src = reinterpret_cast<const uint4*>(blobs) + 2 * (size_t)t;
    Fr v = fr_from_be_words(src[0], src[1]);
    if (FrF::geq_mod(v)) atomicOr(&status[t / FE_PER_BLOB], 1u);
    scalars[t] = v;
}

// term tables of n_out independent MSMs over the same 4096 points: term t of output b = (point t, scalar b * 4096 + t)
__global__ void k_commit_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int total) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    term_point[t] = t % FE_PER_BLOB;
    term_scalar[t] = t;
}

// Jacobian -> compressed, one point per thread
__global__ void k_jac_compress_n(const G1Jac* __restrict__ p, uint8_t* __restrict__ out, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    G1Aff a;
    // up to one wavefront of points (a prover chunk's commitments or proofs): the binary Euclidean inversion - 0.1 ms for a lone lane, and
    // still ahead of the Fermat chain's 1.0 ms when 64 lanes run it side by side (every lane waits for the longest run of halvings among
    // them: measured 5.54 against 5.68 ms for a 64-blob commitment call, 2.5 against 3.65 ms for one blob)
    bool finite = count <= 64 ? g1_to_affine<true>(a, p[i]) : g1_to_affine<false>(a, p[i]);
    g1_compress(out + 48 * (size_t)i, a, !finite);
}

__global__ void k_g2_generator(Fp* __restrict__ out4) {
    if (threadIdx.x || blockIdx.x) return;
    out4[0] = fp_const(consts::G2_GEN_X0_MONT);
    out4[1] = fp_const(consts::G2_GEN_X1_MONT);
    out4[2] = fp_const(consts::G2_GEN_Y0_MONT);
    out4[3] = fp_const(consts::G2_GEN_Y1_MONT);
}

// re-compress an affine G2 point (x.c1 || x.c0 with flags) - settings round-trip check
__global__ void k_g2_compress(const Fp* __restrict__ in4, uint8_t* __restrict__ out96) {
    if (threadIdx.x || blockIdx.x) return;
    Fp x0 = FpF::from_mont(in4[0]), x1 = FpF::from_mont(in4[1]);
    FpF::to_be_bytes(out96, x1);
    FpF::to_be_bytes(out96 + 48, x0);
    out96[0] |= 0x80;
    bool largest = FpF::is_zero(in4[3]) ? fp_is_lex_largest(in4[2]) : fp_is_lex_largest(in4[3]);
    if (largest) out96[0] |= 0x20;
}

// term tables of the batch equation (msm.hpp) for a launch group of B batches of n blobs (T = B n):
// points: C of all batches [0, T), pi of all batches [T, 2T), generator at 2T; scalars of batch b at b(2n+1).
// blockIdx.y = batch.  Tables are [2B][max_terms], row 2b = output A, row 2b+1 = output B.
__global__ void k_batch_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int n, int T, int max_terms) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, bt = blockIdx.y;
    uint32_t* tpA = term_point + (size_t)(2 * bt) * max_terms;
    uint32_t* tsA = term_scalar + (size_t)(2 * bt) * max_terms;
    uint32_t* tpB = tpA + max_terms;
    uint32_t* tsB = tsA + max_terms;
    const uint32_t sb = (uint32_t)bt * (2 * n + 1);
    if (t < n) {
        tpA[t] = T + bt * n + t;  // (pi_t, a_t) -> A
        tsA[t] = sb + t;
        tpB[t] = T + bt * n + t;  // (pi_t, b_t) -> B
        tsB[t] = sb + n + t;
        tpB[n + t] = bt * n + t;  // (C_t, a_t) -> B
        tsB[n + t] = sb + t;
    }
    if (t == 0) {
        tpB[2 * n] = 2 * T;       // (G, g) -> B
        tsB[2 * n] = sb + 2 * n;
    }
}

// the G1 generator as point `idx` (the -(sum r^i y_i) G term of the batch equation)
__global__ void k_set_generator(G1Aff* __restrict__ points, uint32_t* __restrict__ pflag, int idx) {
    if (threadIdx.x || blockIdx.x) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    points[idx] = g;
    pflag[idx] = 0;
}

// transcript records on the device: out[i] = C_i (48) | z_i (32, LE) | y_i (32, LE) | pi_i (48) as 40 little words
__global__ void k_pack_records(const uint32_t* __restrict__ c, const uint32_t* __restrict__ p, const uint32_t* __restrict__ z,
                               const uint32_t* __restrict__ y, uint32_t* __restrict__ out, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T * 40) return;
    int i = t / 40, k = t % 40;
    out[t] = k < 12 ? c[12 * i + k] : k < 20 ? z[8 * i + k - 12] : k < 28 ? y[8 * i + k - 20] : p[12 * i + k - 28];
}

// The same for a small launch, with the host mirror written by the kernel itself (pinned, device-visible memory): records,
// then the per-blob status words and the 2 T point flags - one dispatch instead of one kernel and three copies.
__global__ void k_pack_records_mirror(const uint32_t* __restrict__ c, const uint32_t* __restrict__ p, const uint32_t* __restrict__ z,
                                      const uint32_t* __restrict__ y, uint32_t* __restrict__ out, uint32_t* __restrict__ host,
                                      const uint32_t* __restrict__ status, const uint32_t* __restrict__ pflag, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T * 40) {
        int i = t / 40, k = t % 40;
        const uint32_t v = k < 12 ? c[12 * i + k] : k < 20 ? z[8 * i + k - 12] : k < 28 ? y[8 * i + k - 20] : p[12 * i + k - 28];
        out[t] = v;
        host[t] = v;
    } else if (t < T * 41) {
        host[t] = status[t - T * 40];
    } else if (t < T * 43) {
        host[t] = pflag[t - T * 41];
    }
}

// the generator and its precomputed multiples (gen_mult[4], made once per settings) as point `idx` of a workspace
template <class Mem>
__global__ void k_set_generator_multiples(G1Aff* __restrict__ points, uint32_t* __restrict__ pflag, Mem* __restrict__ mult,
                                          const Mem* __restrict__ gen_mult, int idx, int stride, int chunks) {
    if (threadIdx.x || blockIdx.x) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    points[idx] = g;
    pflag[idx] = 0;
    for (int j = 0; j < chunks; j++) mult[(size_t)j * stride + idx] = gen_mult[j];
}

// plain msm: output 0 over terms (point t, scalar t)
__global__ void k_plain_terms(uint32_t* __restrict__ term_point, uint32_t* __restrict__ term_scalar, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        term_point[t] = t;
        term_scalar[t] = t;
    }
}

// n==1 / verify_kzg_proof scalars: a_0 = 1, b_0 = z, g = -y   (r^0 = 1, so no transcript hash is needed); block = batch
__global__ void k_single_scalars(const Fr* __restrict__ z, const Fr* __restrict__ y, Fr* __restrict__ scalars) {
    if (threadIdx.x) return;
    const int bt = blockIdx.x;
    Fr one = FrF::zero();
    one.l[0] = 1;
    scalars[3 * bt] = one;
    scalars[3 * bt + 1] = z[bt];
    scalars[3 * bt + 2] = FrF::from_mont(FrF::neg(FrF::to_mont(y[bt])));
}

// MSM results (Jacobian) -> SLP inputs; the identity is canonicalised to (0, 1, 0)
__global__ void k_jac_to_slp(const G1Jac* __restrict__ ab, Fp* __restrict__ slp_in) {
    int o = threadIdx.x;
    if (o >= 2) return;
    ab += 2 * blockIdx.x;       // block = batch
    slp_in += 6 * blockIdx.x;
    G1Jac p = ab[o];
    if (g1_is_identity(p)) p = g1_identity();
    slp_in[3 * o] = p.x;
    slp_in[3 * o + 1] = p.y;
    slp_in[3 * o + 2] = p.z;
}

// sum `world` partial (A_k, B_k) pairs (multi-GPU fold; src/kzg_proof.rs:433 generalised)
// partials: [world][B][2]; block = batch
__global__ void k_fold_partials(const G1Jac* __restrict__ partials, int world, int B, G1Jac* __restrict__ ab) {
    int o = threadIdx.x;
    if (o >= 2) return;
    const int bt = blockIdx.x;
    G1Jac acc = partials[2 * bt + o];
    for (int k = 1; k < world; k++) acc = g1_add(acc, partials[(size_t)2 * (k * B + bt) + o]);
    ab[2 * bt + o] = acc;
}

// Jacobian -> 48-byte compressed
__global__ void k_jac_compress(const G1Jac* __restrict__ p, uint8_t* __restrict__ out, int count) {
    int i = threadIdx.x;
    if (i >= count || blockIdx.x) return;
    G1Aff a;
    bool finite = g1_to_affine<true>(a, p[i]);  // (one or two lanes: the Euclidean inversion, 0.1 ms instead of 1.0)
    g1_compress(out + 48 * i, a, !finite);
}

// affine (decoded) -> Jacobian inputs of the pairing program (kzg_pairing_check)
__global__ void k_aff_to_slp(const G1Aff* __restrict__ pts, const uint32_t* __restrict__ flag, Fp* __restrict__ slp_in) {
    int o = threadIdx.x;
    if (o >= 2 || blockIdx.x) return;
    G1Jac p = flag[o] == G1_INFINITY ? g1_identity() : g1_from_affine(pts[o]);
    slp_in[3 * o] = p.x;
    slp_in[3 * o + 1] = p.y;
    slp_in[3 * o + 2] = p.z;
}

// affine -> x || y big-endian (plain)
__global__ void k_aff_to_bytes(const G1Aff* __restrict__ pts, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FpF::to_be_bytes(out + 96 * (size_t)i, FpF::from_mont(pts[i].x));
    FpF::to_be_bytes(out + 96 * (size_t)i + 48, FpF::from_mont(pts[i].y));
}

// out[i] = compress(scalars[i] * G1 generator)  - prover-side helper used to build synthetic
// (commitment, proof) pairs under a known-tau test setup; not on the verification path.
__global__ __launch_bounds__(64) void k_g1_mul_generator(const Fr* __restrict__ scalars, uint8_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Aff g;
    g.x = fp_const(consts::G1_GEN_X_MONT);
    g.y = fp_const(consts::G1_GEN_Y_MONT);
    Fr k = scalars[i];
    G1Jac acc = g1_identity();
    for (int b = 254; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((k.l[b >> 5] >> (b & 31)) & 1) acc = g1_add_affine(acc, g);
    }
    G1Aff a;
    bool finite = g1_to_affine(a, acc);
    g1_compress(out + 48 * (size_t)i, a, !finite);
}
