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
from kzg_rs_amd.slp import gen_pairing, schedule  # noqa: E402

P, R = m.P, m.R


@pytest.fixture(scope="module")
def programs():
    prep, _ = schedule.schedule(gen_pairing.build_prep(), lanes=64, n_instance_inputs=4)
    ver, _ = schedule.schedule(gen_pairing.build_verify(), lanes=64, n_instance_inputs=6)
    return prep, ver


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


def test_committed_programs_match_generator(programs):
    """kzg_rs_amd/data/slp_*.bin (embedded into the library at build time) are what the generator emits."""
    for name, blob in zip(("prep", "verify"), programs):
        path = os.path.join(ROOT, "kzg_rs_amd", "data", "slp_%s.bin" % name)
        if os.path.exists(path):
            assert open(path, "rb").read() == blob
