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
