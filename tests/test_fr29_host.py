"""CPU test of kzg_rs_amd/csrc/fr29.hpp (the radix-2^29 Fr arithmetic of the evaluation kernel), compiled for the
host with g++: results against Python integers, and the accumulator / limb bounds its header claims, on random and on
worst-case inputs.  The GPU kernel built on it is checked bit-exactly against the oracle in test_gpu_parity.py."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
MASK = (1 << 29) - 1
RP = 1 << 261


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(HERE, "host", "_fr29_host.so")
    src = os.path.join(HERE, "host", "fr29_host.cpp")
    hdr = os.path.join(ROOT, "kzg_rs_amd", "csrc", "fr29.hpp")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "kzg_rs_amd", "csrc"),
                               "-o", out, src])
    return C.CDLL(out)


def arr(v, n=9):
    return (C.c_uint32 * n)(*v)


def val(limbs):
    return sum(int(x) << (29 * i) for i, x in enumerate(limbs))


def limbs(v):
    assert v < (1 << 261)
    return [(v >> (29 * i)) & MASK for i in range(9)]


def mul(lib, a, b):
    o = arr([0] * 9)
    lib.h_fr29_mul(o, arr(a), arr(b))
    return list(o)


def column_sums(a, b):
    """The exact 64-bit accumulator values of fr29_mul (Python model of the same loop)."""
    mod = limbs(R)
    acc, m, peak = 0, [], 0
    for k in range(9):
        acc += sum(a[i] * b[k - i] for i in range(k + 1)) + sum(m[i] * mod[k - i] for i in range(k))
        m.append((-acc) & MASK)
        acc += m[k]
        peak = max(peak, acc)
        assert acc & MASK == 0
        acc >>= 29
    out = []
    for k in range(9, 17):
        acc += sum(a[i] * b[k - i] + m[i] * mod[k - i] for i in range(k - 8, 9))
        peak = max(peak, acc)
        out.append(acc & MASK)
        acc >>= 29
    out.append(acc)
    return out, peak


def test_mul_random(lib):
    rng = random.Random(29)
    for _ in range(2000):
        x, y = rng.randrange(8 * R), rng.randrange(2 * R)
        got = mul(lib, limbs(x), limbs(y))
        assert all(g < (1 << 29) for g in got)
        v = val(got)
        assert v % R == x * y * pow(RP, -1, R) % R and v < 2 * R
        assert got == column_sums(limbs(x), limbs(y))[0]


def test_mul_worst_case_bounds(lib):
    """Wide operand with every limb at the documented maximum (2^30 + 2^30 + 2^29 - 1 from a biased difference),
    narrow operand with every limb 2^29 - 1: the accumulator must stay below 2^64 and the C++ result must equal the
    exact model."""
    wide = [(1 << 30) - 2 + (1 << 30) + (1 << 29) - 1] * 8 + [(1 << 27)]
    narrow = [MASK] * 8 + [(2 * R) >> 232]
    out, peak = column_sums(wide, narrow)
    assert peak < (1 << 64), peak.bit_length()
    assert mul(lib, wide, narrow) == out
    assert val(out) % R == val(wide) * val(narrow) * pow(RP, -1, R) % R
    rng = random.Random(1)
    for _ in range(300):
        a = [rng.randrange((1 << 31) + (1 << 29)) for _ in range(8)] + [rng.randrange(1 << 27)]
        b = [rng.randrange(1 << 29) for _ in range(8)] + [rng.randrange(1 << 24)]
        out, peak = column_sums(a, b)
        assert peak < (1 << 64)
        got = mul(lib, a, b)
        assert got == out and val(got) % R == val(a) * val(b) * pow(RP, -1, R) % R


def test_tree_value_discipline(lib):
    """One merge step of the evaluation tree with product outputs as inputs: limb and value bounds as documented."""
    rng = random.Random(7)
    one = limbs(RP % R)
    for _ in range(300):
        prods = [mul(lib, limbs(rng.randrange(8 * R)), limbs(rng.randrange(2 * R))) for _ in range(4)]
        na, nb = [x + y for x, y in zip(prods[0], prods[1])], [x + y for x, y in zip(prods[2], prods[3])]
        o = arr([0] * 9)
        lib.h_fr29_add(o, arr(na), arr(nb))
        s = list(o)
        lib.h_fr29_sub_biased(o, arr(na), arr(nb))
        d = list(o)
        assert val(s) == val(na) + val(nb) and max(s) < (1 << 31)
        assert val(d) == val(na) + 8 * R - val(nb) and max(d) < (1 << 31) + (1 << 29) and val(d) < 12 * R
        z = mul(lib, limbs(rng.randrange(R)), one)
        t = mul(lib, s, z)
        assert val(t) < 2 * R and val(t) % R == val(s) * val(z) * pow(RP, -1, R) % R
        t = mul(lib, d, z)
        assert val(t) < 2 * R and val(t) % R == val(d) * val(z) * pow(RP, -1, R) % R


def column_sums2(a, b, c, d):
    """The exact 64-bit accumulator values of fr29_mul2 (Python model of the same loop)."""
    mod = limbs(R)
    acc, m, peak = 0, [], 0
    for k in range(9):
        acc += sum(a[i] * b[k - i] + c[i] * d[k - i] for i in range(k + 1)) + sum(m[i] * mod[k - i] for i in range(k))
        m.append((-acc) & MASK)
        acc += m[k]
        peak = max(peak, acc)
        assert acc & MASK == 0
        acc >>= 29
    out = []
    for k in range(9, 17):
        acc += sum(a[i] * b[k - i] + c[i] * d[k - i] + m[i] * mod[k - i] for i in range(k - 8, 9))
        peak = max(peak, acc)
        out.append(acc & MASK)
        acc >>= 29
    out.append(acc)
    return out, peak


def mul2(lib, a, b, c, d):
    o = arr([0] * 9)
    lib.h_fr29_mul2(o, arr(a), arr(b), arr(c), arr(d))
    return list(o)


def test_mul2_worst_case_and_merge_discipline(lib):
    """fr29_mul2 = a b + c d under one reduction: worst-case limbs of the merge step (sum of two nodes, 4r-biased
    difference) keep the accumulator below 2^64; iterating the merge keeps node values below 1.2 r."""
    rinv = pow(RP, -1, R)
    a = [(1 << 30) - 2] * 8 + [1 << 26]
    c = [(1 << 29) - 1 + (1 << 30) - 1] * 8 + [1 << 26]
    narrow = [MASK] * 8 + [(2 * R) >> 232]
    out, peak = column_sums2(a, narrow, c, narrow)
    assert peak < (1 << 64), peak.bit_length()
    assert mul2(lib, a, narrow, c, narrow) == out
    assert val(out) % R == (val(a) + val(c)) * val(narrow) * rinv % R
    rng = random.Random(11)
    r2 = limbs(RP * RP % R)
    nodes = [mul(lib, limbs(rng.randrange(R)), r2) for _ in range(64)]
    for _ in range(400):
        na, nb = rng.choice(nodes), rng.choice(nodes)
        z, w = rng.choice(nodes), rng.choice(nodes)
        o = arr([0] * 9)
        lib.h_fr29_add(o, arr(na), arr(nb))
        s = list(o)
        lib.h_fr29_sub_biased4(o, arr(na), arr(nb))
        d = list(o)
        assert val(s) == val(na) + val(nb) and max(s[:8]) < (1 << 30)
        assert val(d) == val(na) + 4 * R - val(nb) and max(d[:8]) < 3 * (1 << 29)
        got = mul2(lib, s, z, d, w)
        exp, peak = column_sums2(s, z, d, w)
        assert got == exp and peak < (1 << 64)
        assert all(x < (1 << 29) for x in got[:8]) and 10 * val(got) < 12 * R
        assert val(got) % R == (val(s) * val(z) + val(d) * val(w)) * rinv % R
        nodes[rng.randrange(64)] = got


def test_words_round_trip_and_normalize(lib):
    rng = random.Random(3)
    for _ in range(500):
        v = rng.randrange(1 << 256)
        w = arr([(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)], 8)
        o = arr([0] * 9)
        lib.h_fr29_from_words(o, w)
        assert list(o) == [(v >> (29 * i)) & MASK for i in range(8)] + [v >> 232]
        w2 = arr([0] * 8, 8)
        lib.h_fr29_to_words(w2, o)
        assert list(w2) == list(w)
        a = [rng.randrange(1 << 32) for _ in range(8)] + [rng.randrange(1 << 28)]
        lib.h_fr29_normalize(o, arr(a))
        assert val(list(o)) == val(a) and all(x < (1 << 29) for x in list(o)[:8])
