"""The latency pairing program (kzg_rs_amd/data/slp_verify2.bin, kzg_rs_amd/slp/schedule2.py) through the KERNEL's own step
functions compiled for the host (csrc/slp2.hpp: slp2_lin, slp2_mul, slp2_is_zero over fp29.hpp) - every limb-level
pre-condition checked at every step (operands normalised, limb sums inside 32 bits, no write to the zero slot) - against
the value-level reference interpreter and the independent big-int model.  CPU only; the GPU run of the same program is
tests/test_gpu_parity.py::test_pairing_check / test_pairing_latency_form_matches_one_wave_form."""
import ctypes as C
import os
import random
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as m  # noqa: E402
from kzg_rs_amd.slp import gen_pairing, schedule, schedule2  # noqa: E402

P, R = m.P, m.R
R406 = 1 << 406


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(HERE, "host", "_slp2_host.so")
    src = os.path.join(HERE, "host", "slp2_host.cpp")
    csrc = os.path.join(ROOT, "kzg_rs_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("slp2.hpp", "fp29.hpp", "constants.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", csrc, "-o", out, src])
    L = C.CDLL(out)
    L.h_slp2_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]
    return L


def to_mem(values):
    """plain integers mod p -> the program's representation: x 2^406 mod p, 14 limbs of 29 bits, 16 words each"""
    words = []
    for v in values:
        x = v % P * R406 % P
        words += [(x >> (29 * i)) & 0x1FFFFFFF for i in range(14)] + [0, 0]
    return (C.c_uint32 * len(words))(*words)


def jac(pt, z):
    if pt is None:
        return [0, 1, 0]
    x, y = pt
    return [x * z * z % P, y * z * z * z % P, z]


def test_program_through_kernel_arithmetic(lib):
    blob = open(os.path.join(ROOT, "kzg_rs_amd", "data", "slp_verify2.bin"), "rb").read()
    prep, _ = schedule.schedule(gen_pairing.build_prep(), lanes=64, n_instance_inputs=4)
    rnd = random.Random(99)
    tau = rnd.randrange(1, R)
    tau_g2 = m.g2_mul(m.G2_GEN, tau)
    lines = []
    for Q in (tau_g2, m.G2_GEN):
        (x0, x1), (y0, y1) = Q
        lines += schedule.run_reference(prep, [x0, x1, y0, y1])
    a = rnd.randrange(1, R)
    cases = [(a, a * tau % R, True), (a, (a * tau + 1) % R, False), (0, 0, True), (0, 5, False), (7, 0, False), (3, 3 * tau % R, True)]
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    set_mem = to_mem(lines)
    for i, (ka, kb, expect) in enumerate(cases):
        A = m.g1_mul(m.G1_GEN, ka) if ka else None
        B = m.g1_mul(m.G1_GEN, kb) if kb else None
        za, zb = (1, 1) if i == 5 else (rnd.randrange(1, P), rnd.randrange(1, P))
        inputs = jac(A, za) + jac(B, zb)
        zero = C.create_string_buffer(6)
        limbs = (C.c_uint32 * (6 * 14))()
        rc = lib.h_slp2_run(buf, len(blob) // 4, to_mem(inputs), set_mem, zero, limbs)
        assert rc == 0, rc
        got = all(z == 1 for z in zero.raw)
        assert got == expect == m.pairings_verify(A, tau_g2, B, m.G2_GEN)
        # the six outputs themselves equal the value-level reference interpreter's, mod p
        ref = schedule2.run_reference2(blob, inputs, lines)
        rinv = pow(R406, -1, P)
        for o in range(6):
            v = sum(int(limbs[14 * o + k]) << (29 * k) for k in range(14))
            assert v * rinv % P == ref[o]


def test_one_proof_programs_through_kernel_arithmetic(lib):
    """SCALARS and VERIFY3 (the one-proof path, kzg_rs_amd/slp/gen_pairing.py) through the kernel's own step functions on the
    host with every limb-level pre-condition checked at every step: SCALARS' 413 outputs equal the value-level reference
    interpreter's, mod p; VERIFY3 fed with them says true for a valid tuple, false for a wrong y and true for the zero
    polynomial (identity points), and passes Z of the per-call G2 point through."""
    sc, _ = schedule2.schedule2(gen_pairing.build_scalars(), lanes=gen_pairing.LATENCY_LANES, n_instance_inputs=32 * 9 + 4, out_values=True)
    v3, _ = schedule2.schedule2(gen_pairing.build_verify3(), lanes=gen_pairing.LATENCY_LANES, n_instance_inputs=9 + 68 * 6 + 2)
    prep, _ = schedule.schedule(gen_pairing.build_prep(), lanes=64, n_instance_inputs=4)
    rnd = random.Random(2024)
    tau = rnd.randrange(1, R)
    tau_g2 = m.g2_mul(m.G2_GEN, tau)
    setlines = []
    for Q in (tau_g2, m.G2_GEN):
        (x0, x1), (y0, y1) = Q
        setlines += schedule.run_reference(prep, [x0, x1, y0, y1])
    set_mem = to_mem(setlines)
    rinv = pow(R406, -1, P)

    def entries(k, gen, mul):
        return [(mul(gen, ((k >> (8 * w)) & 255) << (8 * w)) if (k >> (8 * w)) & 255 else None) for w in range(32)]

    def run(blob, inputs, settings_mem, n_out):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        zero = C.create_string_buffer(n_out)
        limbs = (C.c_uint32 * (n_out * 14))()
        rc = lib.h_slp2_run(buf, len(blob) // 4, to_mem(inputs), settings_mem, zero, limbs)
        assert rc == 0, rc
        vals = [sum(int(limbs[14 * o + k]) << (29 * k) for k in range(14)) * rinv % P for o in range(n_out)]
        return [z == 1 for z in zero.raw], vals

    a, z, y = rnd.randrange(1, R), rnd.randrange(R), rnd.randrange(R)
    C_pt = m.g1_mul(m.G1_GEN, a)
    pi = m.g1_mul(m.G1_GEN, (a - y) * pow(tau - z, -1, R) % R)
    proj1 = lambda pt: [0, 1, 0] if pt is None else [pt[0], pt[1], 1]
    for (cc, zz, yy, pp, expect) in ((C_pt, z, y, pi, True), (C_pt, z, (y + 1) % R, pi, False), (None, z, 0, None, True)):
        ti = []
        for e in entries(zz, m.G2_GEN, m.g2_mul):
            ti += [0, 0, 1, 0, 0, 0] if e is None else list(e[0]) + list(e[1]) + [1, 0]
        for e in entries(yy, m.G1_GEN, m.g1_mul):
            ti += proj1(e)
        ti += list(tau_g2[0]) + list(tau_g2[1])
        _, so = run(sc, ti, to_mem([]) if False else (C.c_uint32 * 16)(), 413)
        assert so == schedule2.run_reference2(sc, ti)
        zero, vals = run(v3, proj1(pp) + proj1(cc) + so, set_mem, 8)
        assert all(zero[:6]) == expect == m.verify_kzg_proof_impl(cc, zz, yy, pp, tau_g2)
        assert vals[6:] == so[411:] and not all(zero[6:])
