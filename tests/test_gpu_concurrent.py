"""ONE shared settings handle, MANY host threads (SURVEY 8b "Threading"; the reference's KzgSettings is three &'static slices
shared freely between threads, src/trusted_setup.rs:44-50,80-92, and its named caller is the revm precompile: one small call per
thread).  The library coalesces concurrent small calls into launches on pooled lanes (csrc/capi_coalesce.hpp); these tests pin
that every caller still gets exactly its own answer - the oracle's - whatever shares its launch."""
import threading

import pytest

import golden_data as G
import oracle_lib as O
from kzg_rs_amd import api, synth
from kzg_rs_amd.api import Blob, Bytes32, Bytes48, KzgError, KzgProof

pytestmark = pytest.mark.gpu
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G1_INF = bytes([0xC0]) + bytes(47)


@pytest.fixture(scope="module")
def rig():
    tau, tau_g2 = synth.synthetic_setup()
    return {"tau": tau, "st": api.KzgSettings.from_tau_g2(tau_g2), "ost": O.Settings.from_tau_g2(tau_g2)}


def _oracle(fn):
    try:
        return fn()
    except O.OracleError:
        return None


def _call(fn):
    try:
        return fn()
    except KzgError as e:
        assert e.kind == "BadArgs", e
        return None


def _proof_cases(rig, n_valid=24):
    """(commitment, z, y, proof) tuples of every outcome: valid, wrong y, wrong proof, non-canonical z / y, off-subgroup and
    off-curve points on either side, points at infinity, and z = tau (the general path) both ways."""
    st, tau = rig["st"], rig["tau"]
    cs, zs, ys, ps, _ = synth.make_valid_proofs(n_valid, seed=77, settings=st)
    off = G.off_subgroup_g1()
    notcurve = bytes([0x80]) + bytes(46) + b"\x01"
    zt = tau.to_bytes(32, "big")
    y7 = (7777).to_bytes(32, "big")
    yG = api.g1_mul_generator([y7], st)[0]
    cases = [(cs[i], zs[i], ys[i], ps[i]) for i in range(n_valid)]
    cases += [(cs[0], zs[0], ys[1], ps[0]), (cs[1], zs[1], ys[1], ps[2]), (cs[2], zs[3], ys[2], ps[2]),
              (cs[3], R.to_bytes(32, "big"), ys[3], ps[3]), (cs[4], zs[4], (R + 9).to_bytes(32, "big"), ps[4]),
              (off, zs[5], ys[5], ps[5]), (cs[6], zs[6], ys[6], off), (notcurve, zs[7], ys[7], ps[7]), (cs[8], zs[8], ys[8], notcurve),
              (G1_INF, zs[9], bytes(32), G1_INF), (G1_INF, zs[9], ys[9], G1_INF),
              (yG, zt, y7, ps[10]), (cs[10], zt, y7, ps[10]), (yG, zt, y7, G1_INF)]
    return cases


def test_32_threads_mixed_verify_kzg_proof_on_one_handle(rig):
    """32 Python threads (ctypes releases the interpreter lock) x 12 verify_kzg_proof calls each on ONE handle, valid / wrong /
    non-canonical / off-subgroup / off-curve / identity / z = tau inputs interleaved so that every launch of the queue mixes
    them: every call returns the oracle's answer for ITS tuple (true / false / Err), and the queue really coalesced."""
    st, ost = rig["st"], rig["ost"]
    cases = _proof_cases(rig)
    want = [_oracle(lambda c=c: O.verify_kzg_proof(*c, ost)) for c in cases]
    assert want.count(True) >= 20 and want.count(False) >= 4 and want.count(None) >= 6
    T, PER = 32, 12
    got, errors = {}, []
    st.small_queue_stats(reset=True)
    start = threading.Barrier(T)

    def work(t):
        try:
            start.wait()
            for k in range(PER):
                i = (t * 5 + k * 7) % len(cases)
                c_, z_, y_, p_ = cases[i]
                got[(t, k)] = (i, _call(lambda: KzgProof.verify_kzg_proof(Bytes48(c_), Bytes32(z_), Bytes32(y_), Bytes48(p_), st)))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert len(got) == T * PER
    bad = [(k, i, r, want[i]) for k, (i, r) in got.items() if r is not want[i]]
    assert not bad, bad[:5]
    stats = st.small_queue_stats()
    # (a non-canonical z or y is refused before it is queued; everything else is a request) coalesced: fewer launches than requests
    assert T * PER * 0.8 < stats["requests"] <= T * PER and stats["launches"] < stats["requests"] and stats["max_items"] >= 2, stats
    assert 1 <= stats["lanes"] <= 8


def test_32_threads_mixed_blob_calls_on_one_handle(rig):
    """The blob entry points from 32 threads on one handle: verify_blob_kzg_proof (one blob) and verify_blob_kzg_proof_batch of
    2..6 host blobs, with a wrong proof, a non-canonical field element, an off-subgroup commitment in SOME callers' inputs:
    every caller gets the oracle's answer for its own batch - a bad blob in one caller's batch never leaks into another's
    that shares its launch."""
    st, ost = rig["st"], rig["ost"]
    n = 12
    blobs, cs, ps, _ = synth.make_valid_batch(n, seed=91, settings=st)
    bl = [blobs[i].tobytes() for i in range(n)]
    wrong_p = O.g1_add(ps[3], G1_GEN)
    noncanon = bytearray(bl[5])
    noncanon[32 * 100: 32 * 101] = R.to_bytes(32, "big")
    noncanon = bytes(noncanon)
    off = G.off_subgroup_g1()
    jobs = []
    for k in range(40):
        m = 1 + k % 6
        idx = [(k + j) % n for j in range(m)]
        b_, c_, p_ = [bl[i] for i in idx], [cs[i] for i in idx], [ps[i] for i in idx]
        if k % 5 == 1:
            p_[-1] = wrong_p if idx[-1] == 3 else O.g1_add(p_[-1], G1_GEN)
        if k % 7 == 3:
            b_[0] = noncanon
        if k % 11 == 6:
            c_[m // 2] = off
        jobs.append((b_, c_, p_))
    want = [_oracle(lambda j=j: O.verify_blob_kzg_proof_batch(j[0], j[1], j[2], ost)) for j in jobs]
    assert want.count(True) >= 10 and want.count(False) >= 4 and want.count(None) >= 5
    T = 32
    got, errors = {}, []
    start = threading.Barrier(T)

    def work(t):
        try:
            start.wait()
            for k in range(t, len(jobs) * 2, T):
                b_, c_, p_ = jobs[k % len(jobs)]
                if len(b_) == 1 and k % 2:
                    r = _call(lambda: KzgProof.verify_blob_kzg_proof(Blob(b_[0]), Bytes48(c_[0]), Bytes48(p_[0]), st))
                else:
                    r = _call(lambda: KzgProof.verify_blob_kzg_proof_batch([Blob(x) for x in b_], [Bytes48(x) for x in c_], [Bytes48(x) for x in p_], st))
                got[k] = r
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    bad = [(k, r, want[k % len(jobs)]) for k, r in got.items() if r is not want[k % len(jobs)]]
    assert len(got) == 2 * len(jobs) and not bad, bad[:5]


def test_native_threads_on_one_handle_all_entry_points(rig):
    """The same through the in-library harness (kzg_debug_concurrent_callers: T std::threads, no interpreter in the way), which
    checks every answer against the expected table: verify_kzg_proof at T = 64, kzg_verify_kzg_proofs of 5 tuples per call, and
    6-blob verify_blob_kzg_proof_batch calls; no wrong answer, and the shared handle's rate is far above one call at a time."""
    st, ost = rig["st"], rig["ost"]
    cases = _proof_cases(rig)
    want = [_oracle(lambda c=c: O.verify_kzg_proof(*c, ost)) for c in cases]
    expect = bytes(2 if w is None else int(w) for w in want)
    c_, z_, y_, p_ = (b"".join(x[k] for x in cases) for k in range(4))
    one = st.concurrent_callers("proof", 1, 0.5, c_, p_, expect, z=z_, y=y_)
    many = st.concurrent_callers("proof", 64, 1.5, c_, p_, expect, z=z_, y=y_)
    assert one["wrong"] == 0 and many["wrong"] == 0 and one["calls"] > 50, (one, many)
    assert many["calls_per_s"] > 8 * one["calls_per_s"], (one, many)
    n5 = len(cases) // 5 * 5
    five = st.concurrent_callers("proofs", 16, 1.0, c_[:48 * n5], p_[:48 * n5], expect[:n5], z=z_[:32 * n5], y=y_[:32 * n5], per_call=5)
    assert five["wrong"] == 0 and five["calls"] > 100, five
    # 6-blob batches: 4 distinct calls (24 blobs), the third with a wrong proof, the fourth with a non-canonical element
    n = 24
    blobs, cs, ps, _ = synth.make_valid_batch(n, seed=93, settings=st)
    bl = [blobs[i].tobytes() for i in range(n)]
    ps = list(ps)
    ps[14] = O.g1_add(ps[14], G1_GEN)
    nc = bytearray(bl[20])
    nc[32 * 4095: 32 * 4096] = (R + 1).to_bytes(32, "big")
    bl[20] = bytes(nc)
    wantb = [_oracle(lambda k=k: O.verify_blob_kzg_proof_batch(bl[6 * k: 6 * k + 6], cs[6 * k: 6 * k + 6], ps[6 * k: 6 * k + 6], ost)) for k in range(4)]
    assert wantb == [True, True, False, None]
    expb = bytes(2 if w is None else int(w) for w in wantb)
    res = st.concurrent_callers("blobs", 16, 1.5, b"".join(cs), b"".join(ps), expb, blobs=b"".join(bl), per_call=6)
    assert res["wrong"] == 0 and res["calls"] > 100, res


def test_queue_off_is_the_old_path(rig):
    """KZG_OPTIONS coalesce=0: a handle without the queue runs every small call under its own lock, as rounds 1-4 did - same
    answers (the differential check of the queue against the path it replaced)."""
    tau, tau_g2 = synth.synthetic_setup()
    with api.options(coalesce=0):
        st0 = api.KzgSettings.from_tau_g2(tau_g2)
    cases = _proof_cases(rig, n_valid=11)
    for c_, z_, y_, p_ in cases:
        a = _call(lambda: KzgProof.verify_kzg_proof(Bytes48(c_), Bytes32(z_), Bytes32(y_), Bytes48(p_), st0))
        b = _call(lambda: KzgProof.verify_kzg_proof(Bytes48(c_), Bytes32(z_), Bytes32(y_), Bytes48(p_), rig["st"]))
        assert a is b, (c_[:3].hex(), z_[:3].hex())
    assert st0.small_queue_stats()["requests"] == 0 and rig["st"].small_queue_stats()["requests"] > 0
    st0.close()
