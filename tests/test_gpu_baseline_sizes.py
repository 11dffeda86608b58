"""GPU parity at the FULL sizes of BASELINE.json's configs 3 and 4 (config 2 is test_gpu_parity.py::
test_device_resident_batch_full_size, config 5 is the sharded path of test_distributed_cpu.py + bench.py --gpus 8).

The oracle cannot run these sizes in seconds, so each test pins the whole result through a property that does not
depend on the size, and a sample of the per-item results directly against the oracle."""
import os
import random

import numpy as np
import pytest

import golden_data as G
import oracle_lib as O
from kzg_rs_amd import api
from kzg_rs_amd.api import KzgSettings

pytestmark = pytest.mark.gpu
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def settings():
    return KzgSettings.load_trusted_setup_file()


def test_config3_evaluate_16384_blobs(settings):
    """BASELINE config 3: evaluate_polynomial_in_evaluation_form (src/kzg_proof.rs:94-133) for 16384 device-resident
    blobs (2 GiB) in one call.
      - 48 sampled (blob, z) pairs, first and last included, against the oracle;
      - every 64th z is a root of unity: the result must be the blob's own element (the :104-108 early return), checked
        for all 256 of them from the blob bytes alone;
      - blobs [8192, 16384) repeat blobs [0, 8192) with the same z: both halves must agree bit for bit."""
    import torch
    n, half = 16384, 8192
    osettings = O.Settings.mainnet()
    g = torch.Generator(device="cuda").manual_seed(3)
    d_blobs = torch.empty((n, 131072), dtype=torch.uint8, device="cuda")
    d_blobs[:half] = torch.randint(0, 256, (half, 131072), dtype=torch.uint8, device="cuda", generator=g)
    d_blobs[:half, 0::32] &= 0x3F  # every element < 2^254 < r (canonical)
    d_blobs[half:] = d_blobs[:half]
    rng = random.Random(33)
    zs = [rng.randrange(R) for _ in range(half)]
    root_idx = {}
    for i in range(0, half, 64):
        k = rng.randrange(4096)
        root_idx[i] = k
        zs[i] = int.from_bytes(osettings.root(k), "big")
    z_le = np.frombuffer(b"".join(z.to_bytes(32, "little") for z in zs + zs), dtype=np.uint8).copy()
    d_z = torch.from_numpy(z_le).cuda()
    d_y = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    ms = settings.last_timings()[4]
    print("config 3: k_blob_evaluate over %d blobs: %.3f ms = %.0f blobs/s, %.0f GB/s of blob bytes" %
          (n, ms, n / ms * 1e3, n * 131072 / ms / 1e6))
    y = d_y.cpu().numpy().reshape(n, 32)
    assert (y[:half] == y[half:]).all()
    for i, k in root_idx.items():
        elem = d_blobs[i, 32 * k: 32 * k + 32].cpu().numpy().tobytes()
        assert y[i].tobytes()[::-1] == elem, (i, k)
    sample = [0, 1, 63, 65, half - 1, half + 7, n - 1] + [rng.randrange(n) for _ in range(41)]
    for i in sample:
        blob = d_blobs[i].cpu().numpy().tobytes()
        want = O.evaluate_polynomial_in_evaluation_form(blob, zs[i % half].to_bytes(32, "big"), osettings)
        assert y[i].tobytes()[::-1] == want, i


def test_config3_rejects_one_bad_element_among_16384(settings):
    """One non-canonical field element (== r) in the last blob of a 4096-blob call -> Err(BadArgs), src/dtypes.rs:48-57."""
    import torch
    n = 4096
    d_blobs = torch.zeros((n, 131072), dtype=torch.uint8, device="cuda")
    d_z = torch.ones(n * 32, dtype=torch.uint8, device="cuda")
    d_z.view(n, 32)[:, 31] = 0
    d_y = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    assert not d_y.any()  # the zero polynomial
    d_blobs[n - 1, 32 * 4095: 32 * 4096] = torch.tensor(list(R.to_bytes(32, "big")), dtype=torch.uint8)
    torch.cuda.synchronize()
    with pytest.raises(api.KzgError) as e:
        api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    assert e.value.kind == "BadArgs"


def test_config4_msm_2_pow_20_points(settings):
    """BASELINE config 4: msm_variable_base (src/kzg_proof.rs:419,429,430) over 2^20 (point, scalar) pairs - the 4096
    G1 Lagrange points of the mainnet setup tiled 256 times, uniform random scalars.  By bilinearity the result must
    equal the 4096-term MSM over the distinct points with the 256 scalars of each point summed mod r, which the
    oracle computes."""
    ts = open(os.path.join(O.ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    base = [bytes.fromhex(ts[2 + i]) for i in range(4096)]
    reps = 256
    n = 4096 * reps
    rng = np.random.Generator(np.random.PCG64(4))
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 0] &= 0x7F  # some scalars land in [r, 2^255): the entry point reduces them mod r like Scalar::from_raw
    got = api.g1_msm(base * reps, [sc[i].tobytes() for i in range(n)], settings)
    print("config 4: MSM kernels over 2^20 terms: %.1f ms" % settings.last_timings()[2])
    sums = [0] * 4096
    for i in range(n):
        sums[i & 4095] += int.from_bytes(sc[i].tobytes(), "big")
    want = O.g1_msm(b"".join(base), b"".join((s % R).to_bytes(32, "big") for s in sums), 4096)
    assert got == want


def test_config5_shard_size_single_batch():
    """BASELINE config 5 is ONE batch of 262 144 blobs, 32 768 per GPU.  A single batch of 20 000 blobs (2.6 GB) on one
    GPU runs every kernel at that kind of size - the one-lane-per-blob challenge kernel, MSM rows of 20 000 and
    40 001 terms in one bucket set, a 3.2 MB transcript - through size-independent properties: valid -> true, one
    corrupted proof anywhere -> false, one non-canonical element -> Err, and two uneven sub-batches -> true."""
    import torch
    from kzg_rs_amd import synth
    from kzg_rs_amd.api import KzgError, KzgProof
    n = 20000
    blobs, cs, ps, st = synth.make_valid_batch(n, seed=55, chunk=2000)
    d_blobs = torch.from_numpy(blobs).cuda()
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    run = lambda b, c, p, k: KzgProof.verify_blob_kzg_proof_batch_device(b, c, p, k, st)
    assert run(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    t = st.last_timings()
    print("config-5 shard size, one batch of %d blobs: %.1f ms device time (challenge %.1f, evaluate %.1f, decode %.1f, msm %.1f, pairing %.1f)"
          % (n, t[0], t[5], t[4], t[6], t[2], t[3]))
    lo = 7001
    assert run(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), lo) is True
    assert run(d_blobs.data_ptr() + lo * 131072, d_c.data_ptr() + 48 * lo, d_p.data_ptr() + 48 * lo, n - lo) is True
    bad = list(ps)
    bad[n - 3] = O.g1_add(ps[n - 3], bytes.fromhex(
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"))
    d_pb = torch.frombuffer(bytearray(b"".join(bad)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert run(d_blobs.data_ptr(), d_c.data_ptr(), d_pb.data_ptr(), n) is False
    d_blobs[12345, 32 * 4000: 32 * 4001] = torch.tensor(list(R.to_bytes(32, "big")), dtype=torch.uint8)
    torch.cuda.synchronize()
    with pytest.raises(KzgError):
        run(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)


def _partial_to_affine(part144):
    """144 bytes of a partial sum (Jacobian X, Y, Z as 12 x u32 little-endian Montgomery limbs, radix 2^384) -> affine
    (x, y) integers, or None for the identity."""
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    rinv = pow(1 << 384, -1, P)
    X, Y, Z = (int.from_bytes(part144[48 * i: 48 * i + 48], "little") * rinv % P for i in range(3))
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    return X * zi * zi % P, Y * zi * zi * zi % P


def test_config5_full_shard_of_262144_blob_batch():
    """BASELINE configs[4] at its real sizes, one rank at a time: ONE batch of 262 144 blobs sharded over 8 GPUs = 32 768
    blobs (4 GiB) per rank.  The test box has one GPU, so it plays every rank in turn on the same shard contents (1 024
    distinct valid tuples tiled x32 in a random order; the global transcript is that shard's records tiled x8):
      * phase 1 on the full 32 768-blob shard - its records equal the oracle's (z, y) on a sample of blobs;
      * the batch challenge of the 42 MB transcript (kzg_batch_challenges, hashed once) equals the oracle's compute_r;
      * phase 2 with n_total = 262 144 and offset = k 32 768 for k = 0 and k = 7: the partial sums (A_k, B_k) equal the
        oracle's MSMs with the scalars r^(offset + i) (collapsed onto the 1 024 distinct points);
      * all 8 partials folded + one pairing: true; with rank 5's B replaced by rank 4's B: false."""
    import numpy as np
    import torch
    from kzg_rs_amd import synth
    from kzg_rs_amd.distributed import HipBackend
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    m, tile, world = 1024, 32, 8
    n_local, n_total = m * tile, m * tile * world
    blobs, cs, ps, st = synth.make_valid_batch(m, seed=56, chunk=1024)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    rng = np.random.Generator(np.random.PCG64(5))
    order = np.concatenate([rng.permutation(m) for _ in range(tile)])  # shard position i holds distinct tuple order[i]
    idx = torch.from_numpy(order).cuda()
    d_blobs = torch.from_numpy(blobs).cuda()[idx].contiguous()  # 4 GiB
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda().view(m, 48)[idx].contiguous()
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda().view(m, 48)[idx].contiguous()
    torch.cuda.synchronize()
    be = HipBackend(st)
    recs = be.phase1((d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n_local))
    assert len(recs) == 160 * n_local
    for i in (0, 1, 777, n_local - 1):
        j = int(order[i])
        z = O.compute_challenge(blobs[j].tobytes(), cs[j])
        y = O.evaluate_polynomial_in_evaluation_form(blobs[j].tobytes(), z, ost)
        assert recs[160 * i: 160 * i + 160] == cs[j] + z[::-1] + y[::-1] + ps[j]
    transcript = recs * world  # every rank holds the same shard contents in this test
    r_le = be.batch_challenges(transcript, 0, 1, n_total)
    rec = [transcript[160 * i: 160 * i + 160] for i in range(n_local)] * world
    r_want = O.compute_r(b"".join(x[:48] for x in rec), b"".join(x[48:80][::-1] for x in rec), b"".join(x[80:112][::-1] for x in rec),
                         b"".join(x[112:] for x in rec), n_total)
    assert r_le[::-1] == r_want
    r = int.from_bytes(r_want, "big")
    zs = [int.from_bytes(recs[160 * i + 48: 160 * i + 80], "little") for i in range(n_local)]
    ys = [int.from_bytes(recs[160 * i + 80: 160 * i + 112], "little") for i in range(n_local)]
    parts = []
    for k in range(world):
        part = be.phase2_r(r_le, n_total, k * n_local, n_local)
        parts.append(part)
        if k not in (0, 7):
            continue
        sa, sb, g = [0] * m, [0] * m, 0  # scalars collapsed onto the distinct tuples
        rp = pow(r, k * n_local, R)
        for i in range(n_local):
            j = int(order[i])
            sa[j] = (sa[j] + rp) % R
            sb[j] = (sb[j] + rp * zs[i]) % R
            g = (g + rp * ys[i]) % R
            rp = rp * r % R
        A = O.g1_msm(b"".join(ps), b"".join(x.to_bytes(32, "big") for x in sa), m)
        Bc = O.g1_msm(b"".join(cs), b"".join(x.to_bytes(32, "big") for x in sa), m)
        Bp = O.g1_msm(b"".join(ps), b"".join(x.to_bytes(32, "big") for x in sb), m)
        Bw = O.g1_add(O.g1_add(Bc, Bp), O.g1_mul(G1_GEN, ((R - g) % R).to_bytes(32, "big")))
        for got, want in ((part[:144], A), (part[144:], Bw)):
            xy, inf = O.g1_decompress(want)
            aff = _partial_to_affine(got)
            assert (aff is None) == inf
            if aff:
                assert aff[0].to_bytes(48, "big") + aff[1].to_bytes(48, "big") == xy, k
    assert be.finish(b"".join(parts), world) is True
    # every partial is a valid relation on its own (an RLC of valid tuples), so swapping whole partials changes nothing:
    assert be.finish(b"".join(parts[:5] + [parts[4]] + parts[6:]), world) is True
    # ... but rank 5's A with rank 4's B does not pair up
    assert be.finish(b"".join(parts[:5] + [parts[5][:144] + parts[4][144:]] + parts[6:]), world) is False


@pytest.mark.parametrize("n,B", [(1024, 8), (512, 40), (3000, 3)])
def test_mid_size_launch_groups(n, B):
    """Launch groups between the single-batch latency path and the bench's large groups: the producer/consumer challenge
    kernel beyond one batch, the default MSM layout with separate (B < 16), paired (B < 32) and merged chunk blocks, term
    slicing at n = 3000.  One batch of each group carries a wrong proof."""
    import torch
    from kzg_rs_amd import synth
    blobs, cs, ps, st = synth.make_valid_batch(n, seed=77 + n)
    perm = lambda b: torch.roll(torch.arange(n), b * 17)
    idx = torch.cat([perm(b) for b in range(B)])
    d_blobs = torch.from_numpy(blobs).cuda()[idx.cuda()].contiguous()
    c_all = [cs[i] for i in idx.tolist()]
    p_all = [ps[i] for i in idx.tolist()]
    wrong = B // 2
    p_all[wrong * n + n - 1] = O.g1_add(p_all[wrong * n + n - 1], bytes.fromhex(
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"))
    d_c = torch.frombuffer(bytearray(b"".join(c_all)), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(b"".join(p_all)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    got = api.verify_blob_kzg_proof_batches_device(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, B, st)
    assert got == [b != wrong for b in range(B)]


def test_throughput_layout_special_batches():
    """A launch of 130 batches x 64 blobs (8 320 blobs: the throughput MSM layout with affine tables, beyond the
    latency layout's 4 096) in which single batches hold what the big synthetic runs never do: 64 copies of one
    tuple (equal points meet in the buckets: the doubling branch of the mixed addition), a zero blob whose commitment
    and proof are the point at infinity (flagged points are skipped, the batch stays valid), a commitment on the
    curve but outside G1 (Err), a junk proof encoding (Err) and a wrong proof (false).  Expected values from the
    oracle on each special batch alone."""
    import torch
    from kzg_rs_amd import synth
    n, B = 64, 130
    blobs, cs, ps, st = synth.make_valid_batch(n, seed=4242)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    inf = bytes([0xC0]) + bytes(47)
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    x = 5
    while True:  # a point of the curve: with overwhelming probability outside the subgroup (the oracle confirms below)
        y2 = (x * x * x + 4) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            break
        x += 1
    enc = bytearray(x.to_bytes(48, "big"))
    enc[0] |= 0x80 | (0x20 if y > P - y else 0)
    off_subgroup = bytes(enc)
    with pytest.raises(O.OracleError):
        O.g1_decompress(off_subgroup)
    idx = torch.cat([torch.roll(torch.arange(n), 5 * b) for b in range(B)])
    all_blobs = torch.from_numpy(blobs)[idx].contiguous()
    c_all = [cs[i] for i in idx.tolist()]
    p_all = [ps[i] for i in idx.tolist()]
    want = [True] * B

    def setb(b, k, blob=None, c=None, p=None):
        if blob is not None:
            all_blobs[b * n + k] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        if c is not None:
            c_all[b * n + k] = c
        if p is not None:
            p_all[b * n + k] = p

    for k in range(n):                                  # batch 3: one tuple 64 times
        setb(3, k, blobs[9].tobytes(), cs[9], ps[9])
    setb(40, 7, bytes(131072), inf, inf)               # batch 40: zero polynomial
    setb(41, 0, bytes(131072), inf, inf)
    setb(41, 63, bytes(131072), inf, inf)
    setb(77, 11, c=off_subgroup)                        # batch 77: commitment outside G1 -> Err
    setb(78, 12, p=bytes([0x81]) + bytes(range(1, 48)))  # batch 78: junk proof -> Err
    setb(129, 63, p=O.g1_add(p_all[129 * n + 63], cs[0]))  # last batch: wrong proof -> false
    for b in (3, 40, 41, 77, 78, 129):
        bl = [all_blobs[i].numpy().tobytes() for i in range(b * n, (b + 1) * n)]
        try:
            want[b] = O.verify_blob_kzg_proof_batch(bl, c_all[b * n:(b + 1) * n], p_all[b * n:(b + 1) * n], ost)
        except O.OracleError:
            want[b] = None
    assert [want[b] for b in (3, 40, 41, 77, 78, 129)] == [True, True, True, None, None, False]
    d_b = all_blobs.cuda()
    d_c = torch.frombuffer(bytearray(b"".join(c_all)), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(b"".join(p_all)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    got = api.verify_blob_kzg_proof_batches_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, B, st)
    assert got == want


def test_launch_group_of_65536_blobs_with_poisoned_batches():
    """The benchmarked shape, unforced: ONE launch group of 64 batches x 1 024 blobs = 65 536 blobs (8 GiB) - above the
    49 152-blob switch to the throughput-form challenge kernel (k_blob_challenge, one lane per blob), with the merged-chunk
    MSM blocks (>= 32 batches) and the one-wave pairing program (> 32 instances), none of them forced by an environment
    variable.  Every batch is a different permutation of 1 024 valid tuples; three batches are poisoned:
        batch 11: a valid G1 point that is not its blob's proof          -> false   (src/kzg_proof.rs:436-444)
        batch 29: a field element equal to r in one blob                  -> Err     (src/dtypes.rs:48-57)
        batch 47: a commitment on the curve but outside G1                -> Err     (src/kzg_proof.rs:17-25)
    all other 61 batches -> true.  16 (z, y) records spread over the group (first and last blob included, one from each
    poisoned batch that still has a record) are compared with the oracle, and the whole group again through the
    launch/wait phases gives the same results."""
    import torch
    from kzg_rs_amd import synth
    from kzg_rs_amd.distributed import HipBackend
    n, B = 1024, 64
    NOT_IN_G1 = bytes.fromhex("8123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef")
    blobs, cs, ps, st = synth.make_valid_batch(n, seed=99, chunk=1024)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    with pytest.raises(O.OracleError):
        O.g1_decompress(NOT_IN_G1)
    gen = torch.Generator(device="cpu").manual_seed(12)
    perms = [torch.randperm(n, generator=gen) for _ in range(B)]
    order = torch.cat(perms)
    idx = order.cuda()
    d_blobs = torch.from_numpy(blobs).cuda()[idx].contiguous()  # 8 GiB
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda().view(n, 48)[idx].contiguous()
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda().view(n, 48)[idx].contiguous()
    bf, be, bs = 11, 29, 47
    i_f, i_e, i_s = bf * n + 500, be * n + 1023, bs * n
    d_p[i_f] = d_p[i_f + 1]  # another blob's proof
    d_blobs[i_e, 4064:4096] = torch.tensor(list(R.to_bytes(32, "big")), dtype=torch.uint8, device="cuda")
    d_c[i_s] = torch.tensor(list(NOT_IN_G1), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = api.verify_blob_kzg_proof_batches_device(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, B, st)
    want = [True] * B
    want[bf], want[be], want[bs] = False, None, None
    assert res == want
    # the oracle agrees on the three poisoned batches (1 024 blobs each: a few seconds of CPU per batch on 8 threads)
    host_c, host_p = d_c.cpu().numpy(), d_p.cpu().numpy()
    for b in (bf, be, bs):
        o = order[b * n: (b + 1) * n].tolist()
        bl = [blobs[j].tobytes() for j in o]
        if b == be:
            x = bytearray(bl[1023])
            x[4064:4096] = R.to_bytes(32, "big")
            bl[1023] = bytes(x)
        cc = [host_c[b * n + k].tobytes() for k in range(n)]
        pp = [host_p[b * n + k].tobytes() for k in range(n)]
        try:
            ores = O.verify_blob_kzg_proof_batch(bl, cc, pp, ost, nthreads=8)
        except O.OracleError:
            ores = None
        assert ores == want[b], b
    # records of the same group through the launch / wait phases: (z, y) of 16 blobs against the oracle
    hb = HipBackend(st)
    hb.phase1_launch((d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n), B)
    recs = hb.phase1_wait()
    assert hb.bad == [w is None for w in want]
    sample = [0, 1, 63, 64, 1023, 1024, i_f, i_f + 1, be * n, i_s + 1, 31 * n + 17, 40000, 49151, 49152, 65000, n * B - 1]
    for i in sample:
        j = int(order[i])
        z = O.compute_challenge(blobs[j].tobytes(), host_c[i].tobytes())
        y = O.evaluate_polynomial_in_evaluation_form(blobs[j].tobytes(), z, ost)
        assert recs[160 * i: 160 * i + 160] == host_c[i].tobytes() + z[::-1] + y[::-1] + host_p[i].tobytes(), i
    hb.phase2_launch(None, n, 0)
    hb.finish_launch(None, 1)
    got = hb.finish_wait()
    assert [None if e else r for r, e in zip(got, hb.bad)] == want


def test_three_launch_groups_in_flight_with_poisoned_batches_through_the_pipeline_entry():
    """The entry point behind the headline at (a quarter of) its shape: kzg_verify_blob_kzg_proof_batch_groups_device with
    3 launch groups of 64 batches x 1 024 blobs (3 x 8 GiB) and in_flight = 3 - the in-library pipeline with every lane
    carrying a group, the throughput-form challenge kernel, merged-chunk MSM blocks and the one-wave pairing program.
    Every batch is a different permutation of 1 024 valid tuples; poisoned in DIFFERENT groups (= lanes):
        group 0, batch 11: a valid G1 point that is not its blob's proof   -> false   (src/kzg_proof.rs:436-444)
        group 1, batch 29: a field element equal to r                      -> Err     (src/dtypes.rs:48-57)
        group 2, batch 47: a commitment on the curve but outside G1        -> Err     (src/kzg_proof.rs:17-25)
        group 2, batch 63: a wrong proof in the very last batch            -> false
    all other 188 batches -> true; then the same three groups as SEVEN (pointers repeat: lanes are reused while groups are in
    flight) with in_flight 2 and 4.  8 (z, y) records read back from the lanes - first / last blob of a group, a neighbour
    of each poisoned blob - equal the oracle's compute_challenge / evaluate_polynomial_in_evaluation_form."""
    import ctypes as C
    import torch
    from kzg_rs_amd import synth
    n, B, K = 1024, 64, 3
    blobs, cs, ps, st = synth.make_valid_batch(n, seed=2024, chunk=1024)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    off = G.off_subgroup_g1()
    with pytest.raises(O.OracleError):
        O.g1_decompress(off)
    gen = torch.Generator(device="cpu").manual_seed(4)
    base_b = torch.from_numpy(blobs).cuda()
    base_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda().view(n, 48)
    base_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda().view(n, 48)
    orders, groups = [], []
    for g in range(K):
        order = torch.cat([torch.randperm(n, generator=gen) for _ in range(B)])
        idx = order.cuda()
        groups.append([base_b[idx].contiguous(), base_c[idx].contiguous(), base_p[idx].contiguous()])
        orders.append(order)
    r_t = torch.tensor(list(R.to_bytes(32, "big")), dtype=torch.uint8, device="cuda")
    groups[0][2][11 * n + 500] = groups[0][2][11 * n + 501].clone()
    groups[1][0][29 * n + 1023, 4064:4096] = r_t
    groups[2][1][47 * n] = torch.tensor(list(off), dtype=torch.uint8, device="cuda")
    groups[2][2][63 * n + 1023] = groups[2][2][63 * n + 1022].clone()
    torch.cuda.synchronize()
    want = [[True] * B for _ in range(K)]
    want[0][11], want[1][29], want[2][47], want[2][63] = False, None, None, False
    ptrs = [tuple(t.data_ptr() for t in g) for g in groups]
    assert api.verify_blob_kzg_proof_batch_groups_device(ptrs, n, B, st, in_flight=3) == want
    # (z, y) as the pipeline's lanes computed them: group g ran on lane g
    L = api.lib()
    L.kzg_debug_lane_records.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    sample = [(0, 0), (0, 11 * n + 499), (0, 11 * n + 500), (1, 29 * n + 1022), (1, n * B - 1), (2, 47 * n + 1), (2, 63 * n + 1023), (2, 40000)]
    for g, i in sample:
        rec = C.create_string_buffer(160)
        api._chk(L.kzg_debug_lane_records(rec, g, i, 1, st._h))
        j = int(orders[g][i])
        cbytes = bytes(groups[g][1][i].cpu().numpy().tobytes())
        pbytes = bytes(groups[g][2][i].cpu().numpy().tobytes())
        z = O.compute_challenge(blobs[j].tobytes(), cbytes)
        y = O.evaluate_polynomial_in_evaluation_form(blobs[j].tobytes(), z, ost)
        assert rec.raw == cbytes + z[::-1] + y[::-1] + pbytes, (g, i)
    # the oracle on two of the poisoned batches (1 024 blobs each, 8 threads)
    for g, b in ((0, 11), (2, 63)):
        o = orders[g][b * n:(b + 1) * n].tolist()
        hc = groups[g][1][b * n:(b + 1) * n].cpu().numpy()
        hp = groups[g][2][b * n:(b + 1) * n].cpu().numpy()
        assert O.verify_blob_kzg_proof_batch([blobs[j].tobytes() for j in o], [hc[k].tobytes() for k in range(n)], [hp[k].tobytes() for k in range(n)], ost, nthreads=8) is False
    # lanes reused while groups are in flight, other pipeline depths
    seven = [0, 1, 2, 2, 0, 1, 0]
    for in_flight in (2, 4):
        assert api.verify_blob_kzg_proof_batch_groups_device([ptrs[g] for g in seven], n, B, st, in_flight=in_flight) == [want[g] for g in seven]


def test_host_batch_of_8192_blobs_dealt_over_two_logical_devices_with_default_chunking():
    """The reference's call shape on a multi-device handle at a size where the DEFAULT dealing applies (no test options):
    one host Vec<Blob> of 8 192 blobs (1 GiB) over a handle on [0, 0] -> 16 interleaved chunks of 512 blobs, 8 lanes per
    logical device, every chunk crossing in slices behind its SHA-256 segments, the transcript hashed by the streaming
    context as the chunks' records come back: r equal to the oracle's compute_r over all 8 192 records (the oracle computes
    every z and y itself), valid -> True, one wrong proof in the last chunk -> False with the oracle's r for THAT transcript,
    one non-canonical element in chunk 9 -> Err."""
    import ctypes as C
    from kzg_rs_amd import synth
    n = 8192
    blobs, cs, ps, st1 = synth.make_valid_batch(n, seed=8192, chunk=1024)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    st2 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0])
    L = api.lib()
    L.kzg_debug_multi_last_r.argtypes = [C.c_char_p, C.c_void_p]
    ok = C.c_bool(False)
    hc, hp = b"".join(cs), b"".join(ps)
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st2._h))
    assert ok.value is True
    t = st2.multi_last_timings()
    assert int(t[7]) == 16, t
    got_r = C.create_string_buffer(32)
    api._chk(L.kzg_debug_multi_last_r(got_r, st2._h))
    zs = [O.compute_challenge(blobs[i].tobytes(), cs[i]) for i in range(n)]
    ys = [O.evaluate_polynomial_in_evaluation_form(blobs[i].tobytes(), zs[i], ost) for i in range(n)]
    assert got_r.raw == O.compute_r(hc, b"".join(zs), b"".join(ys), hp, n)
    bad = list(ps)
    bad[n - 3] = ps[n - 4]
    hb = b"".join(bad)
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hb, n, st2._h))
    assert ok.value is False
    api._chk(L.kzg_debug_multi_last_r(got_r, st2._h))
    assert got_r.raw == O.compute_r(hc, b"".join(zs), b"".join(ys), hb, n)
    bb = blobs.copy()
    bb[9 * 512 + 77, 32 * 4000: 32 * 4001] = list(R.to_bytes(32, "big"))
    assert L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), bb.ctypes.data_as(C.c_char_p), hc, hp, n, st2._h) == api.KZG_BADARGS
    # the single-device handle agrees on the corrupted batch
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hb, n, st1._h))
    assert ok.value is False
    st2.close()


def test_stream_of_sharded_batches_with_throughput_sized_shards():
    """kzg_verify_blob_kzg_proof_batch_sharded_stream with shards above the latency layouts' range: 5 batches of 2 x 5 000
    resident blobs on a handle over [0, 0] (throughput-form MSM tables, affine mixed additions, merged chunks off: one batch per
    launch), 3 in flight on private lane sets, each with its own 10 000-record transcript: valid / wrong proof in the second
    shard / valid / non-canonical element in the first shard / valid; the wrong-proof batch also through the oracle."""
    import torch
    from kzg_rs_amd import synth
    n = 5000
    blobs, cs, ps, st1 = synth.make_valid_batch(2 * n, seed=5000, chunk=1000)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    st2 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0])
    d_b = torch.from_numpy(blobs).cuda()
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda()
    wrong = list(ps)
    wrong[n + 4321] = ps[n + 4320]
    d_p, d_pw = (torch.frombuffer(bytearray(b"".join(x)), dtype=torch.uint8).cuda() for x in (ps, wrong))
    bb = blobs.copy()
    bb[1234, 0:32] = list(R.to_bytes(32, "big"))
    d_bb = torch.from_numpy(bb).cuda()
    torch.cuda.synchronize()

    def shards(db, dp):
        return [(db.data_ptr() + 131072 * lo, d_c.data_ptr() + 48 * lo, dp.data_ptr() + 48 * lo, n) for lo in (0, n)]

    batches = [shards(d_b, d_p), shards(d_b, d_pw), shards(d_b, d_p), shards(d_bb, d_p), shards(d_b, d_p)]
    want = [True, False, True, None, True]
    for in_flight in (3, 1):
        assert api.verify_blob_kzg_proof_batch_sharded_stream(batches, st2, in_flight=in_flight) == want
    assert O.verify_blob_kzg_proof_batch([blobs[i].tobytes() for i in range(2 * n)], cs, wrong, ost, nthreads=8) is False
    assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_pw.data_ptr(), 2 * n, st1) is False
    st2.close()
