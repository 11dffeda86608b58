/*
 * kzg_rs_amd.h - C ABI of the MI355X-native KZG blob-proof verifier (libkzg_rs_amd.so).
 *
 * Drop-in boundary for the verification path of succinctlabs/kzg-rs v0.2.8.  The reference has
 * no FFI today (its boundary is the Rust API re-exported at src/lib.rs:12-18); each entry point
 * below replaces one Rust function and is what a `kzg-rs`-compatible shim crate binds
 * (INTEGRATION.md shows the Rust `extern "C"` block and the shim).  Shapes follow
 * c-kzg-4844's C API so C callers can switch too.
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are borrowed for the duration of the call.
 *   - return value: KzgRet.  KZG_OK means "*ok is valid" (Ok(true)/Ok(false) in the reference);
 *     every other value is the reference's Err(KzgError::...) (src/enums.rs:6-18):
 *         KZG_BADARGS        <-> KzgError::BadArgs            (undecodable / non-canonical input,
 *                                                              src/kzg_proof.rs:17-43)
 *         KZG_INVALID_LENGTH <-> KzgError::InvalidBytesLength (src/dtypes.rs:20-25,
 *                                                              src/kzg_proof.rs:491-501)
 *         KZG_ERROR          <-> KzgError::InternalError      (HIP failure, no GPU, ...)
 *         KZG_MALLOC         <-> (allocation failure; c-kzg-4844's C_KZG_MALLOC)
 *         KZG_BAD_SETUP      <-> KzgError::InvalidTrustedSetup
 *     kzg_last_error() returns a thread-local message for the last non-OK return.
 *   - there is NO CPU fallback: without a usable gfx950 device every call returns KZG_ERROR.
 *   - thread safety: a settings handle is shared freely between host threads, like the reference's &KzgSettings
 *     (src/trusted_setup.rs:44-50,80-92).  The SMALL calls of concurrent callers - kzg_verify_kzg_proof, kzg_verify_blob_kzg_proof,
 *     kzg_verify_blob_kzg_proof_batch / kzg_verify_kzg_proof_batch of up to 256 items, kzg_verify_kzg_proofs of up to 256 tuples -
 *     are coalesced inside the library: whoever finds a free lane of the handle leads ONE launch that carries every call
 *     waiting at that moment (up to 1 024 proofs / 256 blobs, one pairing per item, up to KZG_OPTIONS small_lanes = 2
 *     launches in flight per device), and every caller gets its own verdict and its own Err (csrc/capi_coalesce.hpp); a call on an
 *     idle handle starts at once.  Large calls (batches the GPU fills by itself, the many-batch and prover entry points)
 *     take the handle's own lock one at a time and run beside the small launches.  Handles are immutable after creation.
 *   - environment: the library reads KZG_DEVICES (below) and KZG_OPTIONS ("key=value;key=value": tuning and test switches,
 *     listed in INTEGRATION.md) and writes nothing.  The launch-group pipeline wants 8 HIP hardware queues: set
 *     GPU_MAX_HW_QUEUES=8 in the host's environment before the HIP runtime initialises (ROCm's default is 4, ~5 % slower); a
 *     constructor that sees fewer returns KZG_OK and says so in kzg_settings_note() (kzg_last_error() is empty after a success).
 */
#ifndef KZG_RS_AMD_H
#define KZG_RS_AMD_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZG_BYTES_PER_FIELD_ELEMENT 32      /* src/consts.rs:3  */
#define KZG_FIELD_ELEMENTS_PER_BLOB 4096    /* src/consts.rs:7  */
#define KZG_BYTES_PER_BLOB 131072           /* src/consts.rs:8  */
#define KZG_BYTES_PER_COMMITMENT 48         /* src/consts.rs:9  */
#define KZG_BYTES_PER_PROOF 48              /* src/consts.rs:10 */
#define KZG_BYTES_PER_G2 96                 /* src/consts.rs:2  */

typedef enum {
    KZG_OK = 0,
    KZG_BADARGS = 1,
    KZG_ERROR = 2,
    KZG_MALLOC = 3,
    KZG_INVALID_LENGTH = 4,
    KZG_BAD_SETUP = 5
} KzgRet;

/* Opaque settings handle: replaces KzgSettings / EnvKzgSettings (src/trusted_setup.rs:44-98).
 * Owns the per-device tables (roots of unity in bit-reversed order, prepared lines of [tau]G2
 * and of the G2 generator, pairing programs) and the call workspace. */
typedef struct KzgSettings KzgSettings;

/* KzgSettings::load_trusted_setup_file (src/trusted_setup.rs:94-98) for a caller-supplied copy of
 * the text format of src/trusted_setup.txt / build.rs:23-87:
 *   "<n_g1>\n<n_g2>\n" + n_g1 lines of 96 hex chars (G1, Lagrange) + n_g2 lines of 192 hex chars (G2).
 * Verification reads only roots_of_unity (recomputed from SCALE2_ROOT_OF_UNITY[12],
 * build.rs:131-170) and g2_points[1] (src/kzg_proof.rs:211,386,438). */
KzgRet kzg_settings_load_trusted_setup(KzgSettings **out, const char *txt, size_t len);
/* EnvKzgSettings::Custom (src/trusted_setup.rs:52-57): settings from g2_points[1] = [tau]G2 alone
 * (96-byte compressed).  Used by the synthetic known-tau workloads of the benchmark. */
KzgRet kzg_settings_from_tau_g2(KzgSettings **out, const uint8_t tau_g2[96]);
/* The same two constructors over a DEVICE LIST: one process, one handle, several GPUs.  A Rust caller of
 * KzgProof::verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-477) is one process and passes one &KzgSettings; a handle
 * made here puts the whole device list (HIP ordinals; shard k on devices[k]; devices == NULL or n_devices == 0: every
 * visible device) behind the UNCHANGED signatures:
 *   - ONE batch (kzg_verify_blob_kzg_proof_batch / _device, at least multi_min_blobs = 256 blobs) is sharded by blob: the
 *     array is cut into chunks dealt to the devices interleaved (chunk c -> device c mod D, ~8 chunks per device, each
 *     crossing that device's own PCIe link); phase 1 per chunk; ONE streaming host hash consumes the transcript records
 *     in global order while later chunks are still on their way (:291-334); phase 2 per chunk from r^offset; then the north
 *     star's "G1 all-reduce" - the 288-byte partial sums gathered and folded on the first device - and ONE pairing.
 *     kzg_verify_blob_kzg_proof_batch_sharded takes shards that are already resident per device (contiguous ranges), and
 *     kzg_verify_blob_kzg_proof_batch_sharded_stream keeps several such batches in flight so that one batch's hash runs
 *     beside the device phases of the next.
 *   - MANY INDEPENDENT batches need no exchange: kzg_verify_blob_kzg_proof_batch_groups_device and _batches_device run
 *     every launch group on the device that owns its memory (a pipeline and a host thread per device; a group whose
 *     arrays lie on different devices, or on a device outside the list, is KZG_BADARGS), and
 *     kzg_verify_blob_kzg_proof_batches (host-fed) deals contiguous ranges of whole batches to the devices, each streamed
 *     over its own PCIe link.
 * The handle is also a complete single-device handle on devices[0]: the single-proof, prover-side and kzg_shard_* entry
 * points (the one-process-per-GPU form) run there.  The plain constructors above read KZG_DEVICES ("all" or "0,1,2,...")
 * from the environment, so an unchanged caller gets the same handle without a source change.
 * Exchange of the partial sums (288 bytes per chunk): the constructor asks RCCL for in-process communicators over the device
 * list and PROVES that leg before using it - two synthetic sharded batches (KZG_OPTIONS multi_selftest_blobs = 128 blobs per
 * device) are verified once with the sums carried through pinned host memory and once with ncclAllGather over xGMI, and the
 * gathered [devices][slots] x 288-byte buffers and the verdicts must agree bit for bit.  Equal: exchange = RCCL.  RCCL
 * unusable (librccl missing, a device named twice) or any difference: exchange = host memory.  kzg_settings_note() carries the
 * outcome either way.  KZG_OPTIONS multi_exchange=host | rccl forces one; with =rccl a handle on which RCCL is unusable or
 * fails its self-test does not construct. */
KzgRet kzg_settings_load_trusted_setup_devices(KzgSettings **out, const char *txt, size_t len, const int *devices,
                                               size_t n_devices);
KzgRet kzg_settings_from_tau_g2_devices(KzgSettings **out, const uint8_t tau_g2[96], const int *devices, size_t n_devices);
/* The shape of a handle: *n_devices shards, shard k on devices_out[k] (optional, `cap` entries); *exchange (optional) =
 * 0 single device, 1 partial sums through host memory (kzg_settings_note() says why), 2 in-process RCCL all-gather. */
KzgRet kzg_settings_devices(const KzgSettings *s, size_t *n_devices, int *devices_out, size_t cap, int *exchange);
void kzg_settings_free(KzgSettings *s);
/* What a successful constructor wants its caller to know about the handle ("" = nothing): fewer than 8 HIP hardware queues,
 * the outcome of a multi-device handle's exchange self-test.  The string lives as long as the handle. */
const char *kzg_settings_note(const KzgSettings *s);
/* roots_of_unity[i] as 32 big-endian bytes (i < 4096), for parity tests of the settings tables. */
KzgRet kzg_settings_root_of_unity(const KzgSettings *s, size_t i, uint8_t out[32]);
/* The rest of the trusted setup (handles made by kzg_settings_load_trusted_setup only; verification does not read it):
 * g1_points[i] after the bit-reversal permutation of build.rs:79,89-105 (i < 4096, 48 compressed bytes) and
 * g2_points[i] (i < the file's G2 count, 96 bytes), both re-compressed from the device-side decoded tables; and
 * is_trusted_setup_in_lagrange_form (build.rs:107-129): e(g1[1], g2[0]) == e(g1[0], g2[1]) on the FILE order -
 * false for the Lagrange-form file the crate ships (the reference computes and discards it). */
KzgRet kzg_settings_g1_point(const KzgSettings *s, size_t i, uint8_t out[48]);
KzgRet kzg_settings_g2_point(const KzgSettings *s, size_t i, uint8_t out[96]);
KzgRet kzg_settings_is_monomial_form(bool *ok, const KzgSettings *s);
/* g2_points[1] re-compressed from the device-side decompressed point (round-trip check). */
KzgRet kzg_settings_tau_g2(const KzgSettings *s, uint8_t out[96]);

/* KzgProof::verify_kzg_proof (src/kzg_proof.rs:353-397).  One proof at a time runs the reference's own equation,
 * e(C - [y]G, G2) == e(pi, [tau]G2 - [z]G2): both scalar multiplications and the Miller-loop lines of the per-call G2 point
 * beside the two square roots, the subgroup test beside the pairing (csrc/proof_kernels.hpp): 1.6-1.7 ms on MI355X. */
KzgRet kzg_verify_kzg_proof(bool *ok, const uint8_t commitment[48], const uint8_t z[32], const uint8_t y[32],
                            const uint8_t proof[48], const KzgSettings *s);
/* n INDEPENDENT verify_kzg_proof calls (src/kzg_proof.rs:353-397) through one call, each with its own pairing and its own
 * verdict - SURVEY 8f rank 3's "verify_kzg_proof x N, each with its own pairing": the revm precompile's workload when every
 * proof needs its own result (kzg_verify_kzg_proof_batch below gives ONE boolean for all).  commitments / proofs: n x 48 bytes,
 * zs / ys: n x 32 big-endian bytes, host memory.  ok_out[i] = the result of proof i; err_out[i] (optional) = 1 where the
 * reference would return Err for proof i (then ok_out[i] = false); without err_out any such proof fails the whole call with
 * KZG_BADARGS.  Every proof is one instance of the one-proof programs (one workgroup each), 1 024 per launch. */
KzgRet kzg_verify_kzg_proofs(bool *ok_out, uint8_t *err_out, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys,
                             const uint8_t *proofs, size_t n, const KzgSettings *s);
/* KzgProof::verify_kzg_proof_batch (src/kzg_proof.rs:399-444): n (commitment, z, y, proof) tuples checked with one
 * random linear combination (r from compute_r_powers, :291-348) and ONE pairing.  The reference takes decoded
 * &[G1Affine] / &[Scalar]; across the C ABI they are n*48 compressed bytes and n*32 big-endian canonical bytes in
 * HOST memory, decoded on the device with the same checks as Bytes48/Bytes32 decoding (:17-43) - an undecodable
 * point or a non-canonical scalar is KZG_BADARGS.  n == 0 -> *ok = true (both sides are the identity).
 * DEFAULT FOR SMALL BATCHES - a different algorithm from the reference's, same answer: 2 <= n <= 256 (KZG_OPTIONS
 * small_batch_pairings_max, default 256; small_batch_pairings_max=0 runs the reference's random linear combination :399-444 at
 * every size) returns the CONJUNCTION of n one-proof checks, one pairing
 * per tuple on a CU of its own (1.7-2.6 ms against 2.9 ms for decode -> MSM -> pairing).  The combination is a probabilistic
 * test of exactly that conjunction: it holds whenever the conjunction does, and could hold without it only if the
 * hash-derived r were a root of a fixed non-zero polynomial of degree < n over Fr (probability < 2^-246). */
KzgRet kzg_verify_kzg_proof_batch(bool *ok, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys,
                                  const uint8_t *proofs, size_t n, const KzgSettings *s);
/* KzgProof::verify_blob_kzg_proof (src/kzg_proof.rs:446-470).  Host memory: the blob's Fiat-Shamir hash (:46-72) runs on the
 * calling host core (SHA-NI, 65 us) while the blob crosses PCIe; evaluation, pairing and every point operation on the GPU. */
KzgRet kzg_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t commitment[48], const uint8_t proof[48],
                                 const KzgSettings *s);
/* KzgProof::verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-525).  blobs: n * 131072 bytes,
 * commitments / proofs: n * 48 bytes, HOST memory (a Rust Vec<Blob> is exactly this layout).
 * n == 0 -> *ok = true (:478-480).  The Vec-length-mismatch errors (:491-501) are raised by the
 * caller-side shim, which is the only place that knows three separate lengths.
 * Where the per-blob SHA-256 chains (:46-72) run: a HOST batch of up to 256 blobs (KZG_OPTIONS host_challenge_max_blobs) has
 * them hashed on up to 16 host threads beside the GPU's point decode - a chain is 2.8 ms on GPU lanes however few blobs
 * there are and 65 us on a SHA-NI core; larger host batches cross PCIe in slices with the chains running on the GPU behind
 * them, and device-resident input always hashes on the GPU.  Field and curve arithmetic is never done on the host.
 * DEFAULT FOR SMALL BATCHES - a different algorithm from the reference's, same answer: at the sizes a beacon node calls this
 * with (the 6-9 blobs of a block; 2 <= n <= KZG_OPTIONS small_batch_pairings_max = 256 host blobs) every blob gets its own pairing, side by side on CUs of their own, and *ok is the conjunction of the n
 * verify_blob_kzg_proof verdicts - 1.8-3.2 ms instead of 3.0-3.7 ms; see kzg_verify_kzg_proof_batch for why that is the
 * same answer.  small_batch_pairings_max=0 keeps the reference's combined form (:399-444, one pairing per batch) at every size;
 * tests/test_gpu_parity.py checks that the two forms agree on 2 000 seeded mixed batches. */
KzgRet kzg_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, const uint8_t *commitments,
                                       const uint8_t *proofs, size_t n, const KzgSettings *s);
/* Same, with all three arrays already resident in DEVICE memory (HBM) - the form the throughput
 * benchmark times.  Pointers are device pointers on the settings' device. */
KzgRet kzg_verify_blob_kzg_proof_batch_device(bool *ok, const void *d_blobs, const void *d_commitments,
                                              const void *d_proofs, size_t n, const KzgSettings *s);

/* One batch whose shards are ALREADY resident on the devices of a multi-device handle (BASELINE configs[4]: 8 x 32 768
 * blobs): shard k = n_local[k] blobs at d_blobs[k] / d_commitments[k] / d_proofs[k] in the memory of the handle's k-th
 * device, global blob order = shard order; n_shards = the handle's device count; empty shards allowed.  Same result as
 * kzg_verify_blob_kzg_proof_batch over the concatenation (n == 0 -> true, n == 1 -> the single-blob branch :482-489). */
KzgRet kzg_verify_blob_kzg_proof_batch_sharded(bool *ok, const void *const *d_blobs, const void *const *d_commitments,
                                               const void *const *d_proofs, const size_t *n_local, size_t n_shards,
                                               const KzgSettings *s);
/* A STREAM of such batches: batch j's shard k = n_local[j * n_shards + k] blobs at d_blobs[j * n_shards + k] (etc.) on the
 * handle's k-th device; `in_flight` batches (0 = the default, 4; at most 8) run at once on private lanes, each driven by
 * a host thread of its own, so the serial transcript hash of one batch (:291-334: 42 MB = ~20 ms at 262 144 blobs) runs
 * beside the device phases of the others.  ok_out[j] per batch; err_out[j] (optional) = 1 where the reference would return
 * Err (then ok_out[j] = false); without err_out an invalid input in any batch fails the call. */
KzgRet kzg_verify_blob_kzg_proof_batch_sharded_stream(bool *ok_out, uint8_t *err_out, const void *const *d_blobs,
                                                      const void *const *d_commitments, const void *const *d_proofs,
                                                      const size_t *n_local, size_t n_shards, size_t n_batches, size_t in_flight,
                                                      const KzgSettings *s);
/* Host wall-clock stages of the last sharded call on a multi-device handle, milliseconds: [0] whole call, [1] inputs onto
 * the devices + phase 1 (until the last piece's records are back), [2] what the transcript hash added after that (it runs
 * beside [1]), [3] phase-2 launches, [4] exchange, [5] fold + pairing, [6] the hash's own busy time, [7] pieces.  After
 * kzg_verify_blob_kzg_proof_batch_sharded_stream: [0] the whole stream, [1..6] per-batch averages, [7] batches. */
KzgRet kzg_multi_last_timings(const KzgSettings *s, float out_ms[8]);

/* ---- multi-GPU, one process per GPU (torch.distributed / any transport): the batch sharded by blob in contiguous index ranges ----
 * (the loop of src/kzg_proof.rs:261-273 is the data-parallel axis; the batch challenge r of :291-348
 * needs every (C, z, y, pi), and the three MSMs of :419-430 are sums that split by index range).
 *   1. kzg_shard_phase1 on every rank: decode + challenge + evaluate its n_local blobs (device pointers);
 *      returns its slice of the batch transcript: n_local records of 160 bytes C || z(LE) || y(LE) || pi.
 *   2. the caller all-gathers the records in rank order (RCCL / any transport).
 *   3. kzg_shard_phase2 on every rank: r = H(transcript) from all n_total records, then the rank's partial
 *      sums A_k = sum r^(offset+i) pi_i and B_k = sum r^(offset+i) (C_i + z_i pi_i) - (sum r^(offset+i) y_i) G
 *      as 2 x 144 bytes (Jacobian X,Y,Z, 12 x u32 little-endian Montgomery limbs each).
 *   4. the caller all-gathers the 288-byte partials (point addition is not an RCCL reduction op).
 *   5. kzg_shard_finish on any rank: fold the partials and run the single pairing check. */
KzgRet kzg_shard_phase1(uint8_t *records_out, const void *d_blobs, const void *d_commitments, const void *d_proofs,
                        size_t n_local, const KzgSettings *s);
KzgRet kzg_shard_phase2(uint8_t partial_out[288], const uint8_t *all_records, size_t n_total, size_t offset,
                        size_t n_local, const KzgSettings *s);
KzgRet kzg_shard_finish(bool *ok, const uint8_t *partials, size_t world, const KzgSettings *s);
/* The same three phases split into launch (enqueue on the handle's HIP streams, returns at once) and wait
 * (block on that handle only), and generalised to a launch GROUP of n_batches independent batches of n_local
 * blobs each (contiguous device arrays of n_batches * n_local entries; batch b = entries [b n, (b+1) n); every
 * batch has its own transcript, challenge r, MSM and pairing instance).  At n = 1024 every phase of a batch is
 * a latency-bound serial chain that occupies a sliver of the chip, so the batch dimension inside the kernels is
 * what fills the machine; several handles additionally let ONE host thread pipeline groups in a fixed,
 * collective-safe order (kzg_rs_amd/distributed.py).  A handle runs one group at a time: phase1 -> phase2 -> finish.
 *   records_out : (optional) [n_batches][n_local] x 160 B;  bad_out (optional): n_batches flags, 1 = batch holds an
 *                 invalid input (without bad_out such a group returns KZG_BADARGS).  The handle keeps its records.
 *   all_records : [n_batches][n_total] x 160 B, each batch's records of ALL ranks in global blob order;
 *                 NULL = the handle's own records (single rank: n_total = n_local, offset = 0)
 *   partial_out : [n_batches] x 288 B;  partials: [world][n_batches] x 288 B (NULL: single rank, no fold)
 * Bulk exchange without host copies (the multi-GPU throughput path): kzg_shard_records_device enqueues a copy of
 * the group's records, [n_batches][n_local] x 160 B, into caller-provided DEVICE memory (e.g. the send buffer of an
 * RCCL all-gather; complete once kzg_shard_phase1_wait has returned), and kzg_shard_phase2_launch_gathered takes
 * the all-gathered result as it lands, [world][n_batches][n_local] x 160 B in (pinned) host memory, equal shards. */
KzgRet kzg_shard_phase1_launch(const void *d_blobs, const void *d_commitments, const void *d_proofs, size_t n_local,
                               size_t n_batches, const KzgSettings *s);
KzgRet kzg_shard_phase1_wait(uint8_t *records_out, uint8_t *bad_out, const KzgSettings *s);
KzgRet kzg_shard_records_device(void *d_records_out, const KzgSettings *s);
KzgRet kzg_shard_phase2_launch(const uint8_t *all_records, size_t n_total, size_t offset, const KzgSettings *s);
KzgRet kzg_shard_phase2_launch_gathered(const uint8_t *gathered, size_t world, size_t rank, const KzgSettings *s);
/* Hash once, not on every rank: kzg_batch_challenges is the hash half of compute_r_powers (src/kzg_proof.rs:291-334) as
 * pure HOST code (no handle, no GPU): r_b = SHA-256(domain || degree || n_total || records of batch b) mod r for n_batches
 * batches, written as 32 little-endian bytes each (= Scalar::to_bytes()).  records: world == 0: [n_batches][n_local] x
 * 160 B (n_total = n_local); world > 0: [world][n_batches][n_local] x 160 B as an all-gather / all-to-all of equal
 * shards leaves them (n_total = world n_local).  kzg_shard_phase2_launch_r is phase 2 with the challenges supplied
 * (n_batches x 32 B, canonical) - so ONE rank hashes a batch's transcript and 32 bytes travel instead. */
KzgRet kzg_batch_challenges(uint8_t *r_le_out, const uint8_t *records, size_t world, size_t n_batches, size_t n_local);
KzgRet kzg_shard_phase2_launch_r(const uint8_t *r_le, size_t n_total, size_t offset, const KzgSettings *s);
KzgRet kzg_shard_phase2_wait(uint8_t *partial_out, const KzgSettings *s);
KzgRet kzg_shard_finish_launch(const uint8_t *partials, size_t world, size_t n_batches, const KzgSettings *s);
KzgRet kzg_shard_finish_wait(bool *ok /* n_batches */, const KzgSettings *s);
/* n_batches independent verify_blob_kzg_proof_batch calls (src/kzg_proof.rs:472-525) of n blobs each in one
 * launch group, device-resident inputs (on a multi-device handle: on whichever device of the list holds them).  ok_out[b] is the result of batch b; err_out[b] (optional) = 1 where the
 * reference would return Err (then ok_out[b] = false); without err_out any invalid input fails the whole call. */
KzgRet kzg_verify_blob_kzg_proof_batches_device(bool *ok_out, uint8_t *err_out, const void *d_blobs,
                                                const void *d_commitments, const void *d_proofs, size_t n,
                                                size_t n_batches, const KzgSettings *s);

/* MANY launch groups through one call, kept in flight inside the library (csrc/capi_pipeline.hpp): n_groups groups of
 * batches_per_group independent batches of n blobs each, group g at d_blobs[g] / d_commitments[g] / d_proofs[g] (device
 * memory, the group's batches contiguous; pointers may repeat); `in_flight` groups overlap on private per-group lanes of
 * the device (0 = the default, 4).  On a handle over several devices every group runs on the device that owns its memory,
 * `in_flight` groups per device, all devices at once.  ok_out / err_out (optional): [n_groups][batches_per_group], as in the
 * one-group form.  This is the entry point behind the benchmark's headline: 256 batches of 1 024 blobs per group, 4 groups
 * in flight (measured with 3 / 4 / 5 in flight: 4.91-4.94 / 4.94-4.96 / 4.95-4.97 M blobs/s). */
KzgRet kzg_verify_blob_kzg_proof_batch_groups_device(bool *ok_out, uint8_t *err_out, const void *const *d_blobs,
                                                     const void *const *d_commitments, const void *const *d_proofs, size_t n,
                                                     size_t batches_per_group, size_t n_groups, size_t in_flight,
                                                     const KzgSettings *s);

/* The same for HOST-resident inputs (n_batches Vec<Blob>s back to back): the batches cross PCIe in chunks on a copy stream
 * while the previous chunk is verified, so a stream of host batches runs at the link's rate (~56 GB/s = ~0.43 M blobs/s on
 * MI355X) instead of copy + compute per call.  Same result convention as the device form. */
KzgRet kzg_verify_blob_kzg_proof_batches(bool *ok_out, uint8_t *err_out, const uint8_t *blobs, const uint8_t *commitments,
                                         const uint8_t *proofs, size_t n, size_t n_batches, const KzgSettings *s);

/* Prover side (not in the reference - c-kzg-4844's blob_to_kzg_commitment): C_b = sum_i blob_b[i] *
 * g1_points[i], one 4096-term MSM per blob over the settings' Lagrange points.  blobs: n * 131072 bytes (host), out:
 * n * 48 bytes.  KZG_BADARGS for a non-canonical field element, or settings without G1 points.  1.1 ms for one blob (the fixed-base
 * form of kzg_g1_msm_setup per blob, for one or two blobs), 5.5 ms for 64 (mixed additions over the setup's affine table rows);
 * kzg_compute_blob_kzg_proof: 1.9 / 6.2 ms. */
KzgRet kzg_blob_to_kzg_commitment(uint8_t *out48, const uint8_t *blobs, size_t n, const KzgSettings *s);
/* c-kzg-4844's compute_kzg_proof for n (blob, z) pairs: ys_out[i] = p_i(z_i) (32 bytes big-endian), proofs_out[i] =
 * commitment to the quotient (p_i(X) - y_i) / (X - z_i) (48 bytes), z = a root of unity included.  zs: n * 32 bytes
 * big-endian canonical.  And compute_blob_kzg_proof: the same at z = compute_challenge(blob, commitment)
 * (src/kzg_proof.rs:46-72) - the proof verify_blob_kzg_proof accepts.  Host pointers.  The blobs' challenge hashes run on the
 * host's SHA-NI cores (KZG_OPTIONS host_challenge_max_blobs / host_threads, as for the verifier's small host batches); the
 * buffers of these three entry points stay on the handle after the first call (~30 MB). */
KzgRet kzg_compute_kzg_proof(uint8_t *proofs_out, uint8_t *ys_out, const uint8_t *blobs, const uint8_t *zs, size_t n,
                             const KzgSettings *s);
KzgRet kzg_compute_blob_kzg_proof(uint8_t *proofs_out, const uint8_t *blobs, const uint8_t *commitments, size_t n,
                                  const KzgSettings *s);

/* ---- pieces of the path, exposed for parity tests and the per-kernel benchmarks ---- */
/* compute_challenge (src/kzg_proof.rs:46-72) for n blobs: z_out = n * 32 bytes, big-endian canonical.
 * Host pointers.  commitments are used as bytes (to_compressed(from_compressed(b)) == b). */
KzgRet kzg_compute_challenges(uint8_t *z_out, const uint8_t *blobs, const uint8_t *commitments, size_t n,
                              const KzgSettings *s);
/* evaluate_polynomial_in_evaluation_form (src/kzg_proof.rs:94-133) for n blobs at n points:
 * zs = n * 32 bytes big-endian (reduced mod r like scalar_from_bytes_unchecked :74-81),
 * ys_out = n * 32 bytes big-endian canonical.  KZG_BADARGS if any blob element is >= r
 * (src/dtypes.rs:48-57).  Host pointers. */
KzgRet kzg_evaluate_polynomials(uint8_t *ys_out, const uint8_t *blobs, const uint8_t *zs, size_t n,
                                const KzgSettings *s);
/* Device-resident form of the above (BASELINE config 3): d_z / d_y are n * 32-byte little-endian
 * limb arrays (plain integers) in device memory. */
KzgRet kzg_evaluate_polynomials_device(void *d_y, const void *d_blobs, const void *d_z, size_t n,
                                       const KzgSettings *s);
/* G1Affine::from_compressed (src/kzg_proof.rs:17-25) for n points: status_out[i] = 0 valid,
 * 1 valid identity, 2 rejected; xy_out (optional, n * 96 bytes) = affine x || y big-endian. */
KzgRet kzg_g1_decompress(uint8_t *status_out, uint8_t *xy_out, const uint8_t *points48, size_t n,
                         const KzgSettings *s);
/* G1Projective::msm_variable_base (call sites src/kzg_proof.rs:419,429,430): out = sum scalars[i] * points[i];
 * points: n * 48 bytes compressed, must lie in G1 (checked: the MSM uses the GLV endomorphism); scalars: n * 32 bytes
 * big-endian (reduced mod r),
 * out: 48 bytes compressed.  Host pointers.  n <= 2^26 (KZG_BADARGS above).  2^20 terms: 7.8 ms + 24.5 ms of decompression,
 * subgroup tests and table rows for the 2^20 points (for sums over the setup's own points see kzg_g1_msm_setup). */
KzgRet kzg_g1_msm(uint8_t out[48], const uint8_t *points48, const uint8_t *scalars, size_t n, const KzgSettings *s);
/* The same sum over the HANDLE'S OWN G1 Lagrange points (KzgSettings::g1_points, src/trusted_setup.rs:20-26, bit-reversal
 * permuted as build.rs:79,89-105 leaves them): out = sum_i scalars[i] * g1_points[i mod 4096] - what msm_variable_base computes
 * at the reference's call sites (src/kzg_proof.rs:419,429,430) when the points are trusted-setup points, and the shape of
 * BASELINE.json configs[3] ("2^20 trusted-setup points x random Fr scalars").  scalars: n * 32 bytes big-endian (reduced mod r);
 * host pointers; n <= 2^26.  Needs a handle made from a trusted-setup file (KZG_BADARGS otherwise; KZG_BAD_SETUP when a point is
 * outside G1).  Nothing is decoded per call - the tables were made when the setup was loaded; the sum takes the
 * fixed-base form (csrc/msm_fixed.hpp: 16-bit signed windows over rows 2^(16 v) P_j, half the bucket additions of the
 * variable-base form).  KZG_OPTIONS g1_msm_setup_form = window | fixed forces a form (same result bit for bit).
 * 2^20 terms: 5.1-5.4 ms, 6.0 ms for the call.  Like kzg_g1_msm, the timing assumes scalars whose digits spread over the buckets
 * (random, or hash-derived as in the verifier): a million EQUAL scalars put every entry of a window into one bucket, which one
 * lane then adds one after the other - the sum is still exact, the call takes on the order of a second. */
KzgRet kzg_g1_msm_setup(uint8_t out[48], const uint8_t *scalars, size_t n, const KzgSettings *s);
/* out48[i] = compress(scalars[i] * G1::generator()); scalars n * 32 bytes big-endian (reduced mod r).
 * Prover-side helper used to build synthetic (commitment, proof) pairs under a
 * known-tau test setup (the `G1Affine::generator() * scalar` of src/kzg_proof.rs:388,423). */
KzgRet kzg_g1_mul_generator(uint8_t *out48, const uint8_t *scalars, size_t n, const KzgSettings *s);
/* pairings_verify (src/pairings.rs:5-9) specialised to the verifier's use (src/kzg_proof.rs:436-441):
 * *ok = ( e(a, g2_points[1]) == e(b, G2::generator()) ); a, b: 48-byte compressed G1 (unchecked). */
KzgRet kzg_pairing_check(bool *ok, const uint8_t a[48], const uint8_t b[48], const KzgSettings *s);
/* pairings_verify (src/pairings.rs:5-9, re-exported at src/lib.rs:15) with ARBITRARY G2 arguments:
 * *ok = ( e(a1, a2) == e(b1, b2) ).  a1, b1: 48-byte compressed G1; a2, b2: 96-byte compressed G2 (x.c1 || x.c0, the
 * flag bits of src/trusted_setup.txt's G2 lines).  The reference takes decoded G1Affine / G2Affine values; here the
 * bytes are decoded on the device like from_compressed_unchecked (on the curve; the subgroup invariant of a typed value
 * is the caller's) and an undecodable point is KZG_BADARGS.  Identity arguments (G1 or G2) make their pair contribute 1,
 * as the reference's multi_miller_loop skips them.  The handle supplies the device and the pairing programs only. */
KzgRet kzg_pairings_verify(bool *ok, const uint8_t a1[48], const uint8_t a2[96], const uint8_t b1[48], const uint8_t b2[96],
                           const KzgSettings *s);

/* Timing of the last batch call on this handle, in milliseconds, measured with HIP events on the
 * library's own stream: [0] whole call (device work), [1] per-blob phase (challenge + evaluate +
 * point decode), [2] MSM (split + window + combine), [3] pairing, [4] evaluate kernel, [5] challenge kernel,
 * [6] point decode + subgroup test + MSM multiples (one kernel), [7] the generator's table copy. */
KzgRet kzg_last_timings(const KzgSettings *s, float out_ms[8]);
/* The same intervals summed over every launch group finished on this handle since the last reset; *count = groups. */
KzgRet kzg_timing_totals(const KzgSettings *s, double out_sum_ms[8], uint64_t *count, int reset);
/* Diagnostic: the shader clock the throughput-form challenge kernel (k_blob_challenge) really ran at since the last reset:
 * out = { shader cycles (s_memtime), 100 MHz reference ticks (s_memrealtime) } summed over its waves on this handle and its
 * pipeline lanes; MHz = 100 * out[0] / out[1] (0 / 0 when that kernel has not run).  bench.py prices cycles per instruction
 * with it instead of the nominal 2.4 GHz. */
KzgRet kzg_debug_shader_clock(const KzgSettings *s, double out[2], int reset);

const char *kzg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
