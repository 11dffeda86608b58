"""Multi-GPU behind the reference's own signature (include/kzg_rs_amd.h: kzg_settings_*_devices, capi_multi.hpp).

One process, one settings handle over a device list; KzgProof::verify_blob_kzg_proof_batch (src/kzg_proof.rs:472-525)
through the UNCHANGED entry points shards the batch by blob over the list.  The test box has one GPU, so the list names
device 0 several times - logical shards with their own handles, streams, workspaces, power offsets r^offset, partial sums,
fold and pairing; the partial sums then travel through host memory (ncclCommInitAll refuses duplicate devices).  The
in-process RCCL exchange itself (dlopen, communicator, ncclAllGather on the library's stream, fold from the gathered
buffer) runs in a child process over a list of ONE device (KZG_OPTIONS multi_force=1).  Every result is compared with the oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")


@pytest.fixture(scope="module")
def env():
    import torch
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd import api

    with api.options(multi_min_blobs=2, multi_min_chunk=1):  # read when a handle is made: shard even the 7-blob mainnet batches
        st3 = api.KzgSettings.load_trusted_setup_file(devices=[0, 0, 0])
    st1 = api.KzgSettings.load_trusted_setup_file()
    return {"torch": torch, "G": G, "O": O, "api": api, "st3": st3, "st1": st1, "ost": O.Settings.mainnet()}


def _mainnet(G):
    return [list(x) for x in zip(*G.valid_blob_tuples())]


def _call(api, blobs, cs, ps, st):
    return api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in blobs], [api.Bytes48(c) for c in cs], [api.Bytes48(p) for p in ps], st)


def test_handle_shape(env):
    api = env["api"]
    devs, ex = env["st3"].devices()
    assert devs == [0, 0, 0] and ex == "host"
    assert env["st1"].devices() == ([0], "none")
    # the multi-device handle is a complete single-device handle as well: settings accessors, single proofs
    assert env["st3"].root_of_unity(1) == env["st1"].root_of_unity(1)
    assert env["st3"].g1_point(5) == env["st1"].g1_point(5)
    with pytest.raises(api.KzgError):
        api.KzgSettings.from_tau_g2(env["st1"].tau_g2(), devices=[0, 99])


def test_mainnet_batches_sharded_over_three_logical_devices(env):
    """7 blobs over 3 shards (3 + 2 + 2, an uneven split): valid, corrupted proof -> False, non-canonical element ->
    Err, off-curve commitment -> Err, each equal to the oracle and to the single-device handle."""
    api, O, G = env["api"], env["O"], env["G"]
    blobs, cs, ps = _mainnet(G)
    assert _call(api, blobs, cs, ps, env["st3"]) is True is O.verify_blob_kzg_proof_batch(blobs, cs, ps, env["ost"])
    t = env["st3"].multi_last_timings()
    assert t[0] > 0 and t[1] > 0  # the call went through the shards
    for where in (0, 3, 6):  # a corrupted proof in each shard
        p2 = list(ps)
        p2[where] = O.g1_add(ps[where], G1_GEN)
        assert _call(api, blobs, cs, p2, env["st3"]) is False is O.verify_blob_kzg_proof_batch(blobs, cs, p2, env["ost"])
        assert _call(api, blobs, cs, p2, env["st1"]) is False
    for where in (1, 4, 6):  # a non-canonical field element in each shard
        b2 = list(blobs)
        bb = bytearray(b2[where])
        bb[96:128] = R.to_bytes(32, "big")
        b2[where] = bytes(bb)
        with pytest.raises(api.KzgError) as e:
            _call(api, b2, cs, ps, env["st3"])
        assert e.value.kind == "BadArgs"
        with pytest.raises(O.OracleError):
            O.verify_blob_kzg_proof_batch(b2, cs, ps, env["ost"])
    c2 = list(cs)
    c2[5] = bytes([c2[5][0]]) + bytes(46) + b"\x01"  # x = 1 is not on the curve
    with pytest.raises(api.KzgError):
        _call(api, blobs, c2, ps, env["st3"])
    # rotations: every blob visits every shard and every power offset
    for rot in range(1, 7):
        b, c, p = (x[rot:] + x[:rot] for x in (blobs, cs, ps))
        assert _call(api, b, c, p, env["st3"]) is True


def test_all_committed_batch_vectors_through_the_sharded_handle(env):
    """The 24 c-kzg-4844 batch vectors the reference ships (tests/golden), strict null <=> Err, through three shards."""
    api, G = env["api"], env["G"]
    seen = 0
    for c in G.vectors()["verify_blob_kzg_proof_batch"]:
        try:
            blobs = [api.Blob.from_slice(G.blob(b)) for b in c["blobs"]]
            cs = [api.Bytes48.from_hex(x) for x in c["commitments"]]
            ps = [api.Bytes48.from_hex(x) for x in c["proofs"]]
            got = api.KzgProof.verify_blob_kzg_proof_batch(blobs, cs, ps, env["st3"])
        except api.KzgError:
            got = None
        assert got == c["output"], c["name"]
        seen += 1
    assert seen == 24


def test_fewer_blobs_than_shards_and_single_blob(env):
    api, O, G = env["api"], env["O"], env["G"]
    blobs, cs, ps = _mainnet(G)
    # 2 blobs over 3 shards: the last shard is empty
    assert _call(api, blobs[2:4], cs[2:4], ps[2:4], env["st3"]) is True
    p2 = [ps[2], O.g1_add(ps[3], G1_GEN)]
    assert _call(api, blobs[2:4], cs[2:4], p2, env["st3"]) is False is O.verify_blob_kzg_proof_batch(blobs[2:4], cs[2:4], p2, env["ost"])
    # 1 blob: the single-blob branch (src/kzg_proof.rs:482-489) on the first device
    assert _call(api, blobs[4:5], cs[4:5], ps[4:5], env["st3"]) is True
    assert _call(api, [], [], [], env["st3"]) is True


def test_synthetic_batch_host_device_and_per_device_shards(env):
    """300 synthetic blobs under the known-tau setup: the host form, the device form (arrays on the first device) and the
    per-device form (three resident shards of 128 + 100 + 72 blobs) against the oracle; a corrupted proof in the last
    shard; z / y of the records unaffected by sharding is implied by the equality of the boolean on the corrupted batch."""
    import ctypes as C
    api, O, torch = env["api"], env["O"], env["torch"]
    from kzg_rs_amd import synth

    n = 300
    blobs, cs, ps, sst = synth.make_valid_batch(n, seed=4242)
    with api.options(multi_min_blobs=2, multi_min_chunk=16):  # 300 blobs -> 19 chunks of 16 (the last one of 12), 6-7 per shard
        st3 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0, 0])
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    bl = [blobs[i].tobytes() for i in range(n)]
    assert O.verify_blob_kzg_proof_batch(bl, cs, ps, ost) is True
    ok = C.c_bool(False)
    L = api.lib()
    hc, hp = b"".join(cs), b"".join(ps)
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st3._h))
    assert ok.value is True
    bad = list(ps)
    bad[290] = O.g1_add(ps[290], G1_GEN)
    hb = b"".join(bad)
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hb, n, st3._h))
    assert ok.value is False and O.verify_blob_kzg_proof_batch(bl, cs, bad, ost) is False
    # device form
    d_b = torch.from_numpy(blobs).cuda()
    d_c = torch.frombuffer(bytearray(hc), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(hp), dtype=torch.uint8).cuda()
    d_pb = torch.frombuffer(bytearray(hb), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st3) is True
    assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_pb.data_ptr(), n, st3) is False
    assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_pb.data_ptr(), n, sst) is False
    # per-device resident shards of unequal size, one of them empty in a second call
    cuts = [0, 128, 228, 300]

    def shards(dp, cuts):
        return [(d_b.data_ptr() + 131072 * lo, d_c.data_ptr() + 48 * lo, dp.data_ptr() + 48 * lo, hi - lo) for lo, hi in zip(cuts, cuts[1:])]

    assert api.verify_blob_kzg_proof_batch_sharded(shards(d_p, cuts), st3) is True
    assert api.verify_blob_kzg_proof_batch_sharded(shards(d_pb, cuts), st3) is False
    assert api.verify_blob_kzg_proof_batch_sharded(shards(d_p, [0, 0, 150, 300]), st3) is True
    assert api.verify_blob_kzg_proof_batch_sharded(shards(d_pb, [0, 300, 300, 300]), st3) is False
    with pytest.raises(api.KzgError):  # one shard per device of the handle
        api.verify_blob_kzg_proof_batch_sharded(shards(d_p, [0, 150, 300]), st3)
    st3.close()


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import torch
import golden_data as G, oracle_lib as O
from kzg_rs_amd import api
st = api.KzgSettings.load_trusted_setup_file(devices=[0])
devs, ex = st.devices()
assert devs == [0] and ex == "rccl", (devs, ex)
blobs, cs, ps = [list(x) for x in zip(*G.valid_blob_tuples())]
call = lambda p: api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in blobs], [api.Bytes48(c) for c in cs], [api.Bytes48(x) for x in p], st)
assert call(ps) is True
assert st.multi_last_timings()[0] > 0
g = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
p2 = list(ps); p2[3] = O.g1_add(ps[3], g)
assert call(p2) is False
st.close()
print("rccl-exchange-ok")
"""


def test_in_process_rccl_exchange_world_of_one():
    """The RCCL leg of the exchange on the one-GPU box: a device list of one entry forced through the sharded path
    (KZG_OPTIONS multi_force=1; multi_exchange=rccl selects the RCCL leg and makes a fallback to host staging an error) - librccl bound at run time,
    ncclCommInitAll, the warm-up collective, ncclAllGather of the 288-byte partial sums on the library's stream, the fold
    from the gathered device buffer, the pairing."""
    e = dict(os.environ, KZG_OPTIONS="multi_force=1;multi_exchange=rccl;multi_min_blobs=2;multi_min_chunk=2")
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD % {"root": ROOT, "here": HERE}], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "rccl-exchange-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_REJECT_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import golden_data as G
from kzg_rs_amd import api
st = api.KzgSettings.load_trusted_setup_file(devices=[0])
devs, ex = st.devices()
print("NOTE", st.note())
assert ex == "host" and "DIFFER" in st.note() and "RCCL REJECTED" in st.note(), (ex, st.note())
blobs, cs, ps = [list(x) for x in zip(*G.valid_blob_tuples())]
assert api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in blobs], [api.Bytes48(c) for c in cs], [api.Bytes48(x) for x in ps], st) is True
st.close()
try:
    with api.options(multi_exchange="rccl"):
        api.KzgSettings.load_trusted_setup_file(devices=[0])
    print("constructed")
except api.KzgError as e:
    print("REFUSED", e.kind, "RCCL REJECTED" in e.msg)
"""


def test_exchange_self_test_rejects_a_collective_that_delivers_other_bytes():
    """The rejection path of the exchange self-test: with one bit of the all-gathered buffer flipped (KZG_OPTIONS
    multi_selftest_corrupt=1, a hook of the A/B build) the handle says so - on stderr and in its note - and carries its partial
    sums through host memory, verifying correctly; with multi_exchange=rccl forced the same handle does not construct."""
    from kzg_rs_amd import api
    e = dict(os.environ, KZG_OPTIONS="multi_force=1;multi_min_blobs=2;multi_min_chunk=2;multi_selftest_blobs=16;multi_selftest_corrupt=1",
             KZG_LIB_OVERRIDE=api.LIB_AB_PATH)
    r = subprocess.run([sys.executable, "-c", _REJECT_CHILD % {"root": ROOT, "here": HERE}], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "REFUSED InternalError True" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "kzg_rs_amd: exchange self-test" in r.stderr and "pinned host memory" in r.stderr, r.stderr[-2000:]


def test_env_selected_devices_for_an_unchanged_caller():
    """KZG_DEVICES in the environment turns the reference-shaped constructor's handle into a multi-device one."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch, golden_data as G\n"
        "from kzg_rs_amd import api\n"
        "st = api.KzgSettings.load_trusted_setup_file()\n"
        "assert st.devices() == ([0, 0], 'host'), st.devices()\n"
        "b, c, p = [list(x) for x in zip(*G.valid_blob_tuples())]\n"
        "assert api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(x) for x in b], [api.Bytes48(x) for x in c], [api.Bytes48(x) for x in p], st) is True\n"
        "assert st.multi_last_timings()[0] > 0\n"
        "print('env-devices-ok')\n" % (ROOT, HERE))
    e = dict(os.environ, KZG_DEVICES="0,0", KZG_OPTIONS="multi_min_blobs=2;multi_min_chunk=1")
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "env-devices-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_host_stream_of_batches_dealt_to_the_shards(env):
    """kzg_verify_blob_kzg_proof_batches on a multi-device handle: contiguous ranges of whole batches per shard, each shard
    running the single-device stream on its own thread - 7 batches of 6 mainnet blobs (2 + 2 + 3 over three shards) with a
    corrupted proof in one batch and a non-canonical element in another, against the oracle and the single-device handle;
    also fewer batches than shards."""
    api, O, G = env["api"], env["O"], env["G"]
    tuples = G.valid_blob_tuples()
    n, B = 6, 7
    hb, hc, hp, want = [], [], [], []
    for b in range(B):
        t = (tuples[b % 7:] + tuples[:b % 7])[:n]
        bl, c, p = [list(x) for x in zip(*t)]
        if b == 3:
            p[2] = O.g1_add(p[2], G1_GEN)
        if b == 5:
            x = bytearray(bl[4])
            x[0:32] = R.to_bytes(32, "big")
            bl[4] = bytes(x)
        try:
            want.append(O.verify_blob_kzg_proof_batch(bl, c, p, env["ost"]))
        except O.OracleError:
            want.append(None)
        hb += bl
        hc += c
        hp += p
    assert want == [True, True, True, False, True, None, True]
    args = (b"".join(hb), b"".join(hc), b"".join(hp))
    assert api.verify_blob_kzg_proof_batches(*args, n, B, env["st3"]) == want
    assert api.verify_blob_kzg_proof_batches(*args, n, B, env["st1"]) == want
    two = (args[0][: 131072 * n * 2], args[1][: 48 * n * 2], args[2][: 48 * n * 2])
    assert api.verify_blob_kzg_proof_batches(*two, n, 2, env["st3"]) == want[:2]


def _expected_r(O, ost, bl, cs, ps):
    """compute_r_powers' hash (src/kzg_proof.rs:291-334) from the oracle's own z = compute_challenge, y = p(z)"""
    zs = [O.compute_challenge(b, c) for b, c in zip(bl, cs)]
    ys = [O.evaluate_polynomial_in_evaluation_form(b, z, ost) for b, z in zip(bl, zs)]
    return O.compute_r(b"".join(cs), b"".join(zs), b"".join(ys), b"".join(ps), len(bl))


def _last_r(api, st):
    import ctypes as C
    L = api.lib()
    L.kzg_debug_multi_last_r.argtypes = [C.c_char_p, C.c_void_p]
    out = C.create_string_buffer(32)
    api._chk(L.kzg_debug_multi_last_r(out, st._h))
    return out.raw


def test_interleaved_chunks_and_the_streamed_transcript_hash(env):
    """One host Vec<Blob> of 118 blobs over three logical devices in 3 x 8 chunks of 5 blobs dealt interleaved (chunk c ->
    device c mod 3; the last chunk is ragged: 3 blobs), every chunk on a lane of its own with its own power offset: the batch
    challenge r that the streaming SHA-256 context produced from the records as the chunks came in equals the oracle's
    compute_r over the whole transcript (src/kzg_proof.rs:291-334); 24 partial sums folded; valid -> True, a wrong proof
    in the ragged chunk -> False, a non-canonical element in the middle of the stream -> Err (and the next call is clean);
    the same through the device form (one array on the first device) and with other chunkings of the same batch."""
    import ctypes as C
    api, O, torch = env["api"], env["O"], env["torch"]
    from kzg_rs_amd import synth

    n = 118
    blobs, cs, ps, sst = synth.make_valid_batch(n, seed=777)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    bl = [blobs[i].tobytes() for i in range(n)]
    want_r = _expected_r(O, ost, bl, cs, ps)
    hc, hp = b"".join(cs), b"".join(ps)
    bad = list(ps)
    bad[116] = O.g1_add(ps[116], G1_GEN)
    hb = b"".join(bad)
    want_r_bad = _expected_r(O, ost, bl, cs, bad)
    L = api.lib()
    ok = C.c_bool(False)
    for chunk, chunks_per_dev, pieces in ((5, 8, 24), (7, 8, 17), (1, 64, 59), (40, 8, 3), (118, 8, 1)):
        with api.options(multi_min_blobs=2, multi_min_chunk=chunk, multi_chunks=chunks_per_dev):
            st3 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0, 0])
        api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st3._h))
        assert ok.value is True
        t = st3.multi_last_timings()
        assert int(t[7]) == pieces, (chunk, t)
        assert _last_r(api, st3) == want_r, chunk
        api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hb, n, st3._h))
        assert ok.value is False and _last_r(api, st3) == want_r_bad
        if chunk == 5:
            bb = blobs.copy()
            bb[61, 32 * 100: 32 * 101] = list(R.to_bytes(32, "big"))
            assert L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), bb.ctypes.data_as(C.c_char_p), hc, hp, n, st3._h) == api.KZG_BADARGS
            api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st3._h))
            assert ok.value is True
            # the device form: one array on the first device, the same interleaved chunks
            d_b = torch.from_numpy(blobs).cuda()
            d_c = torch.frombuffer(bytearray(hc), dtype=torch.uint8).cuda()
            d_p = torch.frombuffer(bytearray(hp), dtype=torch.uint8).cuda()
            d_pb = torch.frombuffer(bytearray(hb), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st3) is True
            assert _last_r(api, st3) == want_r and int(st3.multi_last_timings()[7]) == 24
            assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_pb.data_ptr(), n, st3) is False
        st3.close()
    assert O.verify_blob_kzg_proof_batch(bl, cs, ps, ost) is True and O.verify_blob_kzg_proof_batch(bl, cs, bad, ost) is False


def _group_tensors(torch, blobs, cs, ps, lo, hi):
    d_b = torch.from_numpy(blobs[lo:hi]).cuda()
    d_c = torch.frombuffer(bytearray(b"".join(cs[lo:hi])), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(b"".join(ps[lo:hi])), dtype=torch.uint8).cuda()
    return d_b, d_c, d_p


def test_small_calls_of_many_threads_spread_over_the_devices_of_the_handle(env):
    """The small-call queue of a MULTI-device handle (csrc/capi_coalesce.hpp: 2 lanes per device, lane i on shard i mod D): 24
    threads of verify_kzg_proof and of 6-blob verify_blob_kzg_proof_batch calls on the handle over [0, 0, 0] - six lanes, two on
    every logical shard - through the in-library harness that checks every answer; the mainnet vectors as the inputs, true and
    false and Err cases among them."""
    api, O, G = env["api"], env["O"], env["G"]
    ost = env["ost"]
    st3 = api.KzgSettings.load_trusted_setup_file(devices=[0, 0, 0])   # (default options: batches below 256 blobs are small calls, not sharded ones)

    def well_sized(c):
        try:
            return all(len(bytes.fromhex(c[k])) == w for k, w in (("commitment", 48), ("z", 32), ("y", 32), ("proof", 48)))
        except ValueError:
            return False

    cases = [c for c in G.vectors()["verify_kzg_proof"] if well_sized(c)]   # 114 of the 122 vectors (the others fail from_slice in the caller)
    want = [c["output"] for c in cases]
    assert want.count(True) >= 5 and want.count(False) >= 5 and want.count(None) >= 5
    exp = bytes(2 if w is None else int(w) for w in want)
    c_, z_, y_, p_ = (b"".join(bytes.fromhex(c[k]) for c in cases) for k in ("commitment", "z", "y", "proof"))
    st3.small_queue_stats(reset=True)
    r = st3.concurrent_callers("proof", 24, 1.5, c_, p_, exp, z=z_, y=y_)
    q = st3.small_queue_stats()
    assert r["wrong"] == 0 and r["calls"] > 500, r
    assert 3 <= q["lanes"] <= 6 and q["launches"] < q["requests"], q   # lanes on more than one shard; calls shared launches
    tuples = G.valid_blob_tuples()
    blobs = b"".join(t[0] for t in tuples[:6]) * 2
    cs = b"".join(t[1] for t in tuples[:6]) * 2
    ps = bytearray(b"".join(t[2] for t in tuples[:6]) * 2)
    ps[48 * 8: 48 * 9] = tuples[1][2] if tuples[1][2] != tuples[2][2] else tuples[3][2]   # call 1: blob 2 gets another blob's proof
    wantb = [O.verify_blob_kzg_proof_batch([blobs[131072 * i: 131072 * (i + 1)] for i in range(6 * k, 6 * k + 6)],
                                           [cs[48 * i: 48 * i + 48] for i in range(6 * k, 6 * k + 6)],
                                           [bytes(ps[48 * i: 48 * i + 48]) for i in range(6 * k, 6 * k + 6)], ost) for k in range(2)]
    assert wantb == [True, False]
    rb = st3.concurrent_callers("blobs", 12, 1.5, cs, bytes(ps), bytes(int(w) for w in wantb), blobs=blobs, per_call=6)
    assert rb["wrong"] == 0 and rb["calls"] > 100, rb
    st3.close()


def test_launch_groups_routed_to_the_devices_of_the_handle(env):
    """kzg_verify_blob_kzg_proof_batch_groups_device and _batches_device on a handle over [0, 0, 0]: every launch group goes
    to a shard on the device that owns its memory (here: the three logical shards in turn - 7 groups = 3 + 2 + 2, an uneven
    deal), each shard runs its own pipeline (lanes, in_flight 2) on a host thread of its own.  7 groups x 3 batches x 6 blobs
    with a wrong proof, a non-canonical element and an off-subgroup commitment in groups of DIFFERENT shards: every batch
    equal to the oracle; a host pointer or a null group is refused."""
    api, O, torch = env["api"], env["O"], env["torch"]
    from kzg_rs_amd import synth

    n, B, K = 6, 3, 7
    blobs, cs, ps, sst = synth.make_valid_batch(n * B * K, seed=991)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    cs, ps = list(cs), list(ps)
    blobs = blobs.copy()
    # group 1 batch 2: wrong proof; group 3 batch 0: element = r; group 6 batch 1: commitment on the curve but outside G1
    ps[(1 * B + 2) * n + 4] = O.g1_add(ps[(1 * B + 2) * n + 4], G1_GEN)
    blobs[(3 * B + 0) * n + 5, 32 * 7: 32 * 8] = list(R.to_bytes(32, "big"))
    off = env["G"].off_subgroup_g1()
    with pytest.raises(O.OracleError):
        O.g1_decompress(off)
    cs[(6 * B + 1) * n + 0] = off
    want = []
    for g in range(K):
        row = []
        for b in range(B):
            lo = (g * B + b) * n
            try:
                row.append(O.verify_blob_kzg_proof_batch([blobs[i].tobytes() for i in range(lo, lo + n)], cs[lo:lo + n], ps[lo:lo + n], ost))
            except O.OracleError:
                row.append(None)
        want.append(row)
    assert want[1][2] is False and want[3][0] is None and want[6][1] is None and sum(x is True for r in want for x in r) == K * B - 3
    with api.options(multi_min_blobs=2):
        st3 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0, 0])
    keep = [_group_tensors(torch, blobs, cs, ps, g * B * n, (g + 1) * B * n) for g in range(K)]
    torch.cuda.synchronize()
    groups = [(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()) for t in keep]
    for in_flight in (2, 1, 3):
        assert api.verify_blob_kzg_proof_batch_groups_device(groups, n, B, st3, in_flight=in_flight) == want
    assert api.verify_blob_kzg_proof_batch_groups_device(groups, n, B, sst, in_flight=2) == want  # the single-device handle
    assert api.verify_blob_kzg_proof_batch_groups_device(groups[:2], n, B, st3) == want[:2]     # fewer groups than shards
    for g in range(K):  # one launch group at a time: routed to the shard that owns it
        assert api.verify_blob_kzg_proof_batches_device(*groups[g], n, B, st3) == want[g]
    with pytest.raises(api.KzgError) as e:  # host memory is not a device pointer
        host = np.zeros(16, dtype=np.uint8)
        api.verify_blob_kzg_proof_batch_groups_device([(host.ctypes.data, groups[0][1], groups[0][2])], n, B, st3)
    assert e.value.kind == "BadArgs"
    with pytest.raises(api.KzgError):
        api.verify_blob_kzg_proof_batch_groups_device([(0, groups[0][1], groups[0][2])], n, B, st3)
    assert api.verify_blob_kzg_proof_batch_groups_device(groups, n, B, st3) == want  # clean after the refusals
    st3.close()


def test_stream_of_sharded_batches_in_flight(env):
    """kzg_verify_blob_kzg_proof_batch_sharded_stream: 9 sharded batches (shards of unequal size resident on the three
    logical devices, one batch with an empty shard, one with a single blob, one empty) with 1, 2 and 4 batches in flight on
    private lane sets, each hashing its own transcript on its own host thread: valid / wrong proof / non-canonical element /
    off-curve proof per batch, all equal to the oracle; the stage averages are reported."""
    api, O, torch = env["api"], env["O"], env["torch"]
    from kzg_rs_amd import synth

    n = 40
    blobs, cs, ps, sst = synth.make_valid_batch(n, seed=5150)
    ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
    bl = [blobs[i].tobytes() for i in range(n)]
    with api.options(multi_min_blobs=2):
        st3 = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0, 0])
    d_b = torch.from_numpy(blobs).cuda()
    bad_blobs = blobs.copy()
    bad_blobs[17, 0:32] = list(R.to_bytes(32, "big"))
    d_bb = torch.from_numpy(bad_blobs).cuda()
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda()
    wrong = list(ps)
    wrong[33] = O.g1_add(ps[33], G1_GEN)
    offc = list(ps)
    offc[2] = bytes([ps[2][0]]) + bytes(46) + b"\x01"  # x = 1 is not on the curve
    d_p, d_pw, d_po = (torch.frombuffer(bytearray(b"".join(x)), dtype=torch.uint8).cuda() for x in (ps, wrong, offc))
    torch.cuda.synchronize()

    def shards(db, dp, cuts):
        return [(db.data_ptr() + 131072 * lo, d_c.data_ptr() + 48 * lo, dp.data_ptr() + 48 * lo, hi - lo) for lo, hi in zip(cuts, cuts[1:])]

    batches = [shards(d_b, d_p, [0, 13, 27, 40]), shards(d_b, d_pw, [0, 20, 30, 40]), shards(d_bb, d_p, [0, 10, 25, 40]),
               shards(d_b, d_p, [0, 0, 22, 40]), shards(d_b, d_po, [0, 14, 28, 40]), shards(d_b, d_p, [5, 5, 6, 6]),
               shards(d_b, d_p, [0, 0, 0, 0]), shards(d_b, d_pw, [30, 33, 36, 40]), shards(d_b, d_pw, [0, 11, 22, 33])]
    want = [True, False, None, True, None, True, True, False, True]
    assert O.verify_blob_kzg_proof_batch(bl, cs, wrong, ost) is False and O.verify_blob_kzg_proof_batch(bl[30:], cs[30:], wrong[30:], ost) is False
    assert O.verify_blob_kzg_proof_batch(bl[:33], cs[:33], wrong[:33], ost) is True
    for in_flight in (1, 2, 4, 0):
        assert api.verify_blob_kzg_proof_batch_sharded_stream(batches, st3, in_flight=in_flight) == want, in_flight
        t = st3.multi_last_timings()
        assert t[0] > 0 and t[1] > 0 and int(t[7]) == len(batches)
    # the one-batch form gives the same answers, one at a time
    for b, w in zip(batches, want):
        try:
            got = api.verify_blob_kzg_proof_batch_sharded(b, st3)
        except api.KzgError:
            got = None
        assert got == w
    with pytest.raises(api.KzgError):  # a single-device handle has no shards
        api.verify_blob_kzg_proof_batch_sharded_stream(batches[:1], sst)
    st3.close()


def test_rccl_exchange_is_never_a_silent_fallback(env):
    """KZG_OPTIONS multi_exchange=rccl on a handle on which RCCL cannot run (here: a list naming device 0 twice -
    ncclCommInitAll refuses duplicates) fails to construct instead of quietly using the host path; without the option the
    same list gives a host-exchange handle whose note says why; multi_exchange=host never touches RCCL."""
    api = env["api"]
    from kzg_rs_amd import synth
    with api.options(multi_exchange="rccl"):
        with pytest.raises(api.KzgError) as e:
            api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0])
    assert e.value.kind == "InternalError" and "RCCL is unusable" in e.value.msg
    st = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0, 0])
    assert st.devices() == ([0, 0], "host") and "names a device twice" in st.note(), st.note()
    st.close()
    with api.options(multi_exchange="host", multi_force=1):
        st = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=[0])
    assert st.devices() == ([0], "host") and "forced" in st.note(), st.note()
    st.close()


_SELFTEST_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import torch
import golden_data as G, oracle_lib as O
from kzg_rs_amd import api
st = api.KzgSettings.load_trusted_setup_file(devices=[0])
devs, ex = st.devices()
note = st.note()
print("NOTE", note)
assert devs == [0] and ex == "rccl", (devs, ex, note)
assert "bit-identical" in note and "RCCL all-gather selected" in note and "verdicts host 0/1 rccl 0/1" in note, note
assert api.lib().kzg_last_error() == b""
blobs, cs, ps = [list(x) for x in zip(*G.valid_blob_tuples())]
call = lambda p: api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in blobs], [api.Bytes48(c) for c in cs], [api.Bytes48(x) for x in p], st)
assert call(ps) is True
g = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
p2 = list(ps); p2[5] = O.g1_add(ps[5], g)
assert call(p2) is False
st.close()
print("selftest-ok")
"""


def test_exchange_self_test_selects_rccl_in_the_world_of_one():
    """First contact with a set of devices is self-validating (csrc/capi_multi.hpp multi_exchange_selftest): with NO
    multi_exchange option the constructor runs two synthetic sharded batches through the host exchange and through
    ncclAllGather, compares the gathered 288-byte slots bit for bit and the verdicts (false for the random batch, true for the
    all-zero one), and only then makes RCCL the exchange - here on the one-GPU rig (a device list of one entry forced through
    the sharded path, KZG_OPTIONS multi_force=1); the handle's note carries the outcome and the mainnet vectors verify."""
    e = dict(os.environ, KZG_OPTIONS="multi_force=1;multi_min_blobs=2;multi_min_chunk=2;multi_selftest_blobs=32")
    r = subprocess.run([sys.executable, "-c", _SELFTEST_CHILD % {"root": ROOT, "here": HERE}], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "selftest-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_RCCL_TWO = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import torch, ctypes as C
import oracle_lib as O
from kzg_rs_amd import api, synth
nd = torch.cuda.device_count()
devs = list(range(nd))
st = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=devs)
assert st.devices() == (devs, "rccl"), (st.devices(), st.note())
assert "bit-identical" in st.note() and "RCCL all-gather selected" in st.note(), st.note()
print("NOTE", st.note())
n = 64 * nd + 5
blobs, cs, ps, sst = synth.make_valid_batch(n, seed=31337)
ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
L = api.lib(); ok = C.c_bool(False)
hc = b"".join(cs); bad = list(ps); bad[n - 2] = O.g1_add(ps[n - 2], bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"))
for p, want in ((ps, True), (bad, False)):
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, b"".join(p), n, st._h))
    assert ok.value is want
    assert O.verify_blob_kzg_proof_batch([blobs[i].tobytes() for i in range(n)], cs, p, ost) is want
# resident shards, one per device, unequal sizes
cuts = [0] + [n * (k + 1) // nd for k in range(nd)]
keep = []
def shards(pp):
    out = []
    for k in range(nd):
        lo, hi = cuts[k], cuts[k + 1]
        with torch.cuda.device(k):
            t = (torch.from_numpy(blobs[lo:hi]).cuda(), torch.frombuffer(bytearray(b"".join(cs[lo:hi])), dtype=torch.uint8).cuda(),
                 torch.frombuffer(bytearray(b"".join(pp[lo:hi])), dtype=torch.uint8).cuda())
            torch.cuda.synchronize()
        keep.append(t)
        out.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), hi - lo))
    return out
assert api.verify_blob_kzg_proof_batch_sharded(shards(ps), st) is True
assert api.verify_blob_kzg_proof_batch_sharded(shards(bad), st) is False
assert api.verify_blob_kzg_proof_batch_sharded_stream([shards(ps), shards(bad), shards(ps)], st, in_flight=2) == [True, False, True]
st.close()
print("rccl-two-ok", nd)
"""


def test_in_process_rccl_exchange_between_distinct_devices(env):
    """The RCCL all-gather of the partial sums between at least TWO GPUs (ncclCommInitAll over distinct devices, grouped
    ncclAllGather on per-shard streams, the [world][slots] layout fed to the fold, identity slots of shards with fewer
    pieces) against the oracle: host array, per-device resident shards, a stream of sharded batches.  Skipped on a box with
    one GPU - which is every box this project has run on so far.  No multi_exchange option: the constructor's self-test
    (host exchange vs ncclAllGather, bit for bit) is what selects RCCL, and the test asserts that it did."""
    if env["torch"].cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs")
    e = dict(os.environ, KZG_OPTIONS="multi_min_blobs=2;multi_min_chunk=16")
    r = subprocess.run([sys.executable, "-c", _RCCL_TWO % {"root": ROOT, "here": HERE}], env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "rccl-two-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
