"""CPU test of kzg_rs_amd/csrc/g1_29_formulas.hpp - the exact point formulas the decode and MSM kernels run (lazy
radix-2^29 arithmetic, static bounds) - compiled for the host with g++ and compared with an integer model of
y^2 = x^3 + 4 over Fp: random points with coordinates lifted to the top of the documented input ranges, every special
case (identity operands with Z = 0, p, 2p; P + P; P - P), chains of operations fed back into each other, and the
output bounds each formula promises."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
MASK = (1 << 29) - 1
RP = 1 << 406
RINV = pow(RP, -1, P)
BETA = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(HERE, "host", "_g1_29_host.so")
    src = os.path.join(HERE, "host", "g1_29_host.cpp")
    csrc = os.path.join(ROOT, "kzg_rs_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("g1_29_formulas.hpp", "fp29.hpp", "constants.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", csrc, "-o", out, src])
    return C.CDLL(out)


def limbs(v):
    assert 0 <= v < (1 << 406)
    return [(v >> (29 * i)) & MASK for i in range(13)] + [v >> 377]


def val(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


# ---- integer model: affine points (x, y) or None for the identity
def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return x, (lam * (a[0] - x) - a[1]) % P


def rand_point(rng):
    while True:
        x = rng.randrange(P)
        y2 = (x * x * x + 4) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            return x, (y if rng.randrange(2) else P - y)


def lift(rng, v, bound):
    """a lazy representative of v (mod p) below bound * p, often close to the top of that range"""
    k = (bound - 1) if rng.randrange(3) == 0 else rng.randrange(bound)
    return v % P + k * P


def to_jac(rng, pt, bx, by, bz):
    """42 words of a Jacobian representative of pt in Montgomery form, coordinates lifted below bx p, by p, bz p"""
    if pt is None:
        z = rng.choice([0, P, 2 * P][: max(1, min(3, bz))])
        return limbs(lift(rng, rng.randrange(P), bx)) + limbs(lift(rng, rng.randrange(1, P), by)) + limbs(z)
    z = rng.randrange(1, P)
    X, Y = pt[0] * z * z % P, pt[1] * z * z * z % P
    return limbs(lift(rng, X * RP, bx)) + limbs(lift(rng, Y * RP, by)) + limbs(lift(rng, z * RP, bz))


def from_jac(w):
    x, y, z = (val(w[0:14]) * RINV % P, val(w[14:28]) * RINV % P, val(w[28:42]) * RINV % P)
    if z == 0:
        return None
    zi = pow(z, -1, P)
    return x * zi * zi % P, y * zi * zi * zi % P


def run(f, *args):
    o = (C.c_uint32 * 42)()
    f(o, *[(C.c_uint32 * len(a))(*a) for a in args])
    return list(o)


def check_bounds(w, bx, by, bz):
    for k, b in enumerate((bx, by, bz)):
        c = w[14 * k: 14 * k + 14]
        assert all(x <= MASK for x in c[:13]), "limb not normalised"
        assert val(c) < b * P, (k, val(c) // P)


def test_doubling(lib):
    rng = random.Random(1)
    for _ in range(400):
        pt = rand_point(rng)
        w = run(lib.h_g1_dbl, to_jac(rng, pt, 1024, 1024, 1024))
        assert from_jac(w) == ec_add(pt, pt)
        check_bounds(w, 130, 34, 4)
    for _ in range(20):  # the identity stays the identity
        assert from_jac(run(lib.h_g1_dbl, to_jac(rng, None, 1024, 1024, 3))) is None
    # extreme limbs: every lower limb 2^29 - 1
    ext = [MASK] * 13 + [13 * 1000]
    w = run(lib.h_g1_dbl, ext + ext + ext)
    X, Y, Z = (val(ext) * RINV % P,) * 3
    # dbl-2009-l on arbitrary (X, Y, Z), not necessarily on the curve: compare the raw formulas
    A, B = X * X % P, Y * Y % P
    Cc = B * B % P
    D = 2 * ((X + B) ** 2 - A - Cc) % P
    E = 3 * A % P
    X3 = (E * E - 2 * D) % P
    assert val(w[0:14]) * RINV % P == X3
    assert val(w[14:28]) * RINV % P == (E * (D - X3) - 8 * Cc) % P
    assert val(w[28:42]) * RINV % P == 2 * Y * Z % P
    check_bounds(w, 130, 34, 4)


def test_general_addition_and_special_cases(lib):
    rng = random.Random(2)
    for i in range(400):
        a, b = rand_point(rng), rand_point(rng)
        kind = i % 8
        if kind == 5:
            b = a                      # P + P
        elif kind == 6:
            b = (a[0], P - a[1])       # P - P
        elif kind == 7:
            a = None                   # identity operand
        wa, wb = to_jac(rng, a, 1024, 1024, 1024), to_jac(rng, b, 1024, 1024, 1024)
        assert from_jac(run(lib.h_g1_add, wa, wb)) == ec_add(a, b)
        assert from_jac(run(lib.h_g1_add, wb, wa)) == ec_add(a, b)
        if a is not None and kind < 5:
            check_bounds(run(lib.h_g1_add, wa, wb), 14, 6, 2)


def test_mixed_addition_and_special_cases(lib):
    rng = random.Random(3)
    for i in range(600):
        a, b = rand_point(rng), rand_point(rng)
        kind = i % 8
        if kind == 5:
            a = b                      # doubling branch
        elif kind == 6:
            a = (b[0], P - b[1])       # identity branch
        elif kind == 7:
            a = None                   # first addition into an empty bucket
        wa = to_jac(rng, a, 256, 256, 1024)
        wq = limbs(lift(rng, b[0] * RP, 8)) + limbs(lift(rng, b[1] * RP, 8))
        w = run(lib.h_g1_add_affine, wa, wq)
        assert from_jac(w) == ec_add(a, b)
        if a is not None and kind < 5:
            check_bounds(w, 14, 6, 2)


def test_two_halves_forms(lib):
    """g1j29_madd_head / _tail (bucket loop) and g1j29_inf_flags / _add_head / _add_tail / _add_same_x_result (reduction
    trees) - what the MSM window kernel's loops run.  The mixed form sends the caller to the complete formula exactly for
    P + P, P - P and an accumulator at infinity; the general form is complete in stages: identity operands (Z = 0, p, 2p)
    pass through, and the same-x ending (P + P as the doubling of (U1, S1, Z1 Z2); P - P) needs nothing but the head."""
    rng = random.Random(6)

    def run_flag(f, *args):
        o = (C.c_uint32 * 42)()
        flag = f(o, *[(C.c_uint32 * len(a))(*a) for a in args])
        return list(o), flag

    for i in range(800):
        a, b = rand_point(rng), rand_point(rng)
        kind = i % 8
        if kind == 5:
            a = b
        elif kind == 6:
            a = (b[0], P - b[1])
        elif kind == 7:
            a = None
        wa = to_jac(rng, a, 256, 256, 1024)
        if a is None:  # the kernel's identity: (0, 1, 0) in any representation of zero
            wa = limbs(rng.choice([0, P])) + wa[14:]
        wq = limbs(lift(rng, b[0] * RP, 8)) + limbs(lift(rng, b[1] * RP, 8))
        w, flag = run_flag(lib.h_g1_madd_split, wa, wq)
        assert flag == (1 if kind in (5, 6, 7) else 0)
        if not flag:
            assert from_jac(w) == ec_add(a, b)
            check_bounds(w, 14, 6, 2)
        wb = to_jac(rng, b, 1024, 1024, 1024)
        wa2 = to_jac(rng, a, 1024, 1024, 1024)
        for x, y in ((wa2, wb), (wb, wa2)):
            w, flag = run_flag(lib.h_g1_add_split, x, y)
            assert flag == (1 if kind in (5, 6) else 2 if kind == 7 else 0)
            assert from_jac(w) == ec_add(a, b)  # the same-x ending is computed from the head alone
            if flag == 0:
                check_bounds(w, 14, 6, 2)
        if kind == 7:  # identity + identity
            w, flag = run_flag(lib.h_g1_add_split, wa2, to_jac(rng, None, 1024, 1024, 3))
            assert flag == 2 and from_jac(w) is None


def test_chains_feed_back(lib):
    """what the kernels do: outputs of one formula are the inputs of the next, never reduced in between"""
    rng = random.Random(4)
    for _ in range(30):
        pt = rand_point(rng)
        acc_w, acc = run(lib.h_g1_identity), None
        base_w = to_jac(rng, pt, 2, 4, 2)
        cur_w, cur = base_w, pt
        for step in range(40):
            op = rng.randrange(4)
            if op == 0:
                cur_w, cur = run(lib.h_g1_dbl, cur_w), ec_add(cur, cur)
            elif op == 1:
                acc_w, acc = run(lib.h_g1_add, acc_w, cur_w), ec_add(acc, cur)
            elif op == 2:
                q = rand_point(rng)
                acc_w, acc = run(lib.h_g1_add_affine, acc_w, limbs(q[0] * RP % P) + limbs(q[1] * RP % P)), ec_add(acc, q)
            else:
                acc_w, acc = run(lib.h_g1_dbl, acc_w), ec_add(acc, acc)
            assert from_jac(cur_w) == cur and from_jac(acc_w) == acc
            check_bounds(cur_w, 256, 256, 1024)
            check_bounds(acc_w, 256, 256, 1024)


def test_neg_phi(lib):
    rng = random.Random(5)
    for _ in range(100):
        pt = rand_point(rng)
        w = run(lib.h_g1_neg_phi, to_jac(rng, pt, 1024, 64, 1024))
        assert from_jac(w) == (BETA * pt[0] % P, P - pt[1])
        check_bounds(w, 2, 128, 1024)
