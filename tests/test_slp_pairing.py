"""CPU validation of the generated pairing SLP programs (kzg_rs_amd/slp): the binary programs
are run by the reference interpreter in schedule.py and compared with the independent
big-int model oracle/pymodel.py (pairings_verify = reference src/pairings.rs:5-9)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as m  # noqa: E402
from kzg_rs_amd.slp import gen_pairing, schedule, schedule2  # noqa: E402

P, R = m.P, m.R


@pytest.fixture(scope="module")
def programs():
    prep, _ = schedule.schedule(gen_pairing.build_prep(), lanes=64, n_instance_inputs=4)
    ver, _ = schedule.schedule(gen_pairing.build_verify(), lanes=64, n_instance_inputs=6)
    return prep, ver


@pytest.fixture(scope="module")
def latency_program():
    blob, stats = schedule2.schedule2(gen_pairing.build_verify_latency(), lanes=gen_pairing.LATENCY_LANES, n_instance_inputs=6)
    return blob, stats


def run_prep(prep, Q):
    (x0, x1), (y0, y1) = Q
    return schedule.run_reference(prep, [x0, x1, y0, y1])


def jac(pt, z):
    if pt is None:
        return [0, 1, 0]
    x, y = pt
    return [x * z * z % P, y * z * z * z % P, z]


def test_pairing_programs(programs):
    prep, ver = programs
    rnd = random.Random(1234)
    tau = rnd.randrange(1, R)
    tau_g2 = m.g2_mul(m.G2_GEN, tau)
    lines = run_prep(prep, tau_g2) + run_prep(prep, m.G2_GEN)
    assert len(lines) == 2 * 68 * 6
    cases = []
    a = rnd.randrange(1, R)
    cases.append((a, a * tau % R, True))
    cases.append((a, (a * tau + 1) % R, False))
    cases.append((0, 0, True))       # both identity (n = 2 vector e61aafba051ddf79: A = B = infinity)
    cases.append((0, 5, False))
    cases.append((7, 0, False))
    for (ka, kb, expect) in cases:
        A = m.g1_mul(m.G1_GEN, ka) if ka else None
        B = m.g1_mul(m.G1_GEN, kb) if kb else None
        out = schedule.run_reference(ver, jac(A, rnd.randrange(1, P)) + jac(B, rnd.randrange(1, P)), lines)
        got = all(v == 0 for v in out)
        assert got == expect == m.pairings_verify(A, tau_g2, B, m.G2_GEN)


def test_latency_program(programs, latency_program):
    """The multi-wave latency form of VERIFY (schoolbook towers, 4-ary lazy linear steps, pre-added product operands;
    schedule2.py) through its reference interpreter - which asserts the kernel's range pre-conditions at every step -
    against the independent model, on the same cases plus Jacobian inputs with Z = 1 and a second setup point."""
    prep, _ = programs
    blob, stats = latency_program
    assert stats["lds_bytes"] + 4096 <= 160 * 1024 and stats["lanes"] % 64 == 0
    assert stats["mul_steps"] <= 500 and stats["lin_steps"] <= 1100  # the point of the exercise: ~3 steps per product level
    rnd = random.Random(4321)
    for trial in range(2):
        tau = rnd.randrange(1, R)
        tau_g2 = m.g2_mul(m.G2_GEN, tau)
        lines = run_prep(prep, tau_g2) + run_prep(prep, m.G2_GEN)
        a = rnd.randrange(1, R)
        cases = [(a, a * tau % R, True), (a, (a * tau + 1) % R, False), (0, 0, True), (0, 5, False), (7, 0, False)]
        for i, (ka, kb, expect) in enumerate(cases if trial == 0 else cases[:2]):
            A = m.g1_mul(m.G1_GEN, ka) if ka else None
            B = m.g1_mul(m.G1_GEN, kb) if kb else None
            za, zb = (1, 1) if i == 1 else (rnd.randrange(1, P), rnd.randrange(1, P))
            out = schedule2.run_reference2(blob, jac(A, za) + jac(B, zb), lines)
            assert all(v == 0 for v in out) == expect == m.pairings_verify(A, tau_g2, B, m.G2_GEN)


def test_committed_programs_match_generator(programs, latency_program):
    path = os.path.join(ROOT, "kzg_rs_amd", "data", "slp_verify2.bin")
    if os.path.exists(path):
        assert open(path, "rb").read() == latency_program[0]
    """kzg_rs_amd/data/slp_*.bin (embedded into the library at build time) are what the generator emits."""
    for name, blob in zip(("prep", "verify"), programs):
        path = os.path.join(ROOT, "kzg_rs_amd", "data", "slp_%s.bin" % name)
        if os.path.exists(path):
            assert open(path, "rb").read() == blob


# ---------------------------------------------------------------- the one-proof path: SCALARS and VERIFY3
def _proj1(pt):
    return [0, 1, 0] if pt is None else [pt[0], pt[1], 1]


def _proj2(pt):
    return [0, 0, 1, 0, 0, 0] if pt is None else list(pt[0]) + list(pt[1]) + [1, 0]


def _table_entries(k, gen, mul):
    """what k_proof_select copies for the scalar k: entry w = [digit_w 2^(8w)] gen, the identity for a zero digit"""
    return [(mul(gen, ((k >> (8 * w)) & 255) << (8 * w)) if (k >> (8 * w)) & 255 else None) for w in range(32)]


def scalars_inputs(z, y, tau_g2):
    ti = []
    for e in _table_entries(z, m.G2_GEN, m.g2_mul):
        ti += _proj2(e)
    for e in _table_entries(y, m.G1_GEN, m.g1_mul):
        ti += _proj1(e)
    return ti + list(tau_g2[0]) + list(tau_g2[1])


@pytest.fixture(scope="module")
def proof_programs():
    sc, s1 = schedule2.schedule2(gen_pairing.build_scalars(), lanes=gen_pairing.LATENCY_LANES, n_instance_inputs=32 * 9 + 4, out_values=True)
    v3, s2 = schedule2.schedule2(gen_pairing.build_verify3(), lanes=gen_pairing.LATENCY_LANES, n_instance_inputs=9 + 68 * 6 + 2)
    return sc, s1, v3, s2


def test_one_proof_programs(programs, proof_programs):
    """KzgProof::verify_kzg_proof in the reference's own form (src/kzg_proof.rs:384-396), e(C - [y]G, G2) == e(pi, [tau]G2 -
    [z]G2), as the two latency programs of the one-proof path: SCALARS (fixed-base sums by complete additions over the table
    entries the digits of z and y select, the lines of the per-call G2 point from projective coordinates) feeding VERIFY3 (the
    pairing, points homogeneous, C - [y]G formed in the graph) - through the reference interpreter, which asserts the kernel's
    range pre-conditions at every step, against the independent model: valid, wrong y, wrong z, pi = O with C = [y]G, the zero
    polynomial (C = O, y = 0, pi = O) and its wrong y, z = 0 (every window digit zero), y = 0, z = r - 1."""
    prep, _ = programs
    sc, st_sc, v3, st_v3 = proof_programs
    assert st_sc["lds_bytes"] + 4096 <= 160 * 1024 and st_v3["lds_bytes"] + 4096 <= 160 * 1024
    assert st_sc["steps"] <= 560 and st_v3["steps"] <= 1320  # SCALARS must stay in the shadow of the two square roots
    rnd = random.Random(77)
    tau = rnd.randrange(1, R)
    tau_g2 = m.g2_mul(m.G2_GEN, tau)
    setlines = run_prep(prep, tau_g2) + run_prep(prep, m.G2_GEN)

    def check(C, z, y, Pi, expect):
        so = schedule2.run_reference2(sc, scalars_inputs(z, y, tau_g2))
        assert len(so) == 3 + 408 + 2 and so[411:] != [0, 0]
        yg = so[:3]
        want_yg = m.g1_mul(m.G1_GEN, y) if y else None
        if want_yg is None:
            assert yg[2] == 0 and yg[1] != 0
        else:
            zi = pow(yg[2], -1, P)
            assert (yg[0] * zi % P, yg[1] * zi % P) == want_yg
        out = schedule2.run_reference2(v3, _proj1(Pi) + _proj1(C) + so, setlines)
        assert out[6:] == so[411:]
        assert all(v == 0 for v in out[:6]) == expect == m.verify_kzg_proof_impl(C, z, y, Pi, tau_g2)

    a, z, y = rnd.randrange(1, R), rnd.randrange(R), rnd.randrange(R)
    C = m.g1_mul(m.G1_GEN, a)
    proof = lambda a_, z_, y_: m.g1_mul(m.G1_GEN, (a_ - y_) * pow(tau - z_, -1, R) % R)
    check(C, z, y, proof(a, z, y), True)
    check(C, z, (y + 1) % R, proof(a, z, y), False)
    check(C, (z + 1) % R, y, proof(a, z, y), False)
    check(m.g1_mul(m.G1_GEN, y), z, y, None, True)
    check(None, z, 0, None, True)
    check(None, z, 5, None, False)
    check(C, 0, y, proof(a, 0, y), True)
    check(C, z, 0, proof(a, z, 0), True)
    check(C, R - 1, y, proof(a, R - 1, y), True)
    # z = tau: Q is the identity and SCALARS says so (the caller then takes the general path)
    so = schedule2.run_reference2(sc, scalars_inputs(tau, y, tau_g2))
    assert so[411:] == [0, 0]


def test_one_proof_data_files_match_their_generators(proof_programs):
    """kzg_rs_amd/data/slp_scalars.bin, slp_verify3.bin (generated at build time, embedded into the library) are what the
    generator emits; a few entries of fixed_base.bin against the model's scalar multiplications."""
    import struct
    data = os.path.join(ROOT, "kzg_rs_amd", "data")
    for name, blob in (("scalars", proof_programs[0]), ("verify3", proof_programs[2])):
        path = os.path.join(data, "slp_%s.bin" % name)
        if os.path.exists(path):
            assert open(path, "rb").read() == blob, name
    path = os.path.join(data, "fixed_base.bin")
    if not os.path.exists(path):
        pytest.skip("fixed_base.bin not generated yet (python -m kzg_rs_amd.build)")
    b = open(path, "rb").read()
    assert len(b) == 32 * 256 * 9 * 48
    rinv = pow(1 << 384, -1, P)
    fp = lambda off: sum(x << (32 * i) for i, x in enumerate(struct.unpack_from("<12I", b, off))) * rinv % P
    g2e = lambda w, d: [fp(((w * 256 + d) * 6 + i) * 48) for i in range(6)]
    g1e = lambda w, d: [fp(32 * 256 * 6 * 48 + ((w * 256 + d) * 3 + i) * 48) for i in range(3)]
    assert g2e(0, 0) == [0, 0, 1, 0, 0, 0] and g2e(31, 0) == [0, 0, 1, 0, 0, 0] and g1e(5, 0) == [0, 1, 0]
    for w, d in ((0, 1), (0, 255), (1, 1), (7, 200), (31, 255)):
        q = m.g2_mul(m.G2_GEN, (d << (8 * w)) % R)
        assert g2e(w, d) == list(q[0]) + list(q[1]) + [1, 0], (w, d)
        q = m.g1_mul(m.G1_GEN, (d << (8 * w)) % R)
        assert g1e(w, d) == [q[0], q[1], 1], (w, d)
