"""CPU test of kzg_rs_amd/csrc/fp29.hpp (the radix-2^29 Fp arithmetic of the decode and MSM kernels), compiled for
the host with g++: results against Python integers and the value / limb bounds its header claims, on random and on
worst-case inputs.  The GPU kernels built on it are checked bit-exactly against the oracle in tests/test_gpu_*.py."""
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


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(HERE, "host", "_fp29_host.so")
    src = os.path.join(HERE, "host", "fp29_host.cpp")
    deps = [src, os.path.join(ROOT, "kzg_rs_amd", "csrc", "fp29.hpp"), os.path.join(ROOT, "kzg_rs_amd", "csrc", "constants.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "kzg_rs_amd", "csrc"),
                               "-o", out, src])
    return C.CDLL(out)


def arr(v, n=14):
    return (C.c_uint32 * n)(*v)


def val(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


def limbs(v):
    """normalised: limbs 0..12 < 2^29, the rest in the top limb"""
    return [(v >> (29 * i)) & MASK for i in range(13)] + [v >> 377]


def call2(f, a, b):
    o = arr([0] * 14)
    f(o, arr(a), arr(b))
    return list(o)


def test_mul_sqr_random_and_lazy_bounds(lib):
    rng = random.Random(29)
    for bound in (1, 2, 40, 6000):
        for _ in range(300):
            x, y = rng.randrange(bound * P), rng.randrange(bound * P)
            got = call2(lib.h_fp29_mul, limbs(x), limbs(y))
            assert all(g <= MASK for g in got[:13]) and val(got) < 2 * P
            assert val(got) % P == x * y * pow(RP, -1, P) % P
            o = arr([0] * 14)
            lib.h_fp29_sqr(o, arr(limbs(x)))
            assert val(list(o)) < 2 * P and val(list(o)) % P == x * x * pow(RP, -1, P) % P
    # extreme limbs: every lower limb 2^29 - 1
    x = val([MASK] * 13 + [70000])  # ~5400 p
    got = call2(lib.h_fp29_mul, limbs(x), limbs(x))
    assert val(got) % P == x * x * pow(RP, -1, P) % P and val(got) < 2 * P
    o = arr([0] * 14)
    lib.h_fp29_sqr(o, arr(limbs(x)))
    assert list(o) == got


def test_add_sub_bias(lib):
    rng = random.Random(5)
    for _ in range(500):
        a, b = rng.randrange(600 * P), rng.randrange(600 * P)
        got = call2(lib.h_fp29_add, limbs(a), limbs(b))
        assert val(got) == a + b and all(g <= MASK for g in got[:13])
        for e in range(1, 11):
            bb = rng.randrange((1 << (e - 1)) * P + 1)
            o = arr([0] * 14)
            lib.h_fp29_sub(o, arr(limbs(a)), arr(limbs(bb)), e)
            assert val(list(o)) == a + (1 << e) * P - bb and all(g <= MASK for g in list(o)[:13])
    # worst case for the no-borrow property: b with every lower limb at 2^29 - 1 and the largest admissible top limb
    for e in range(1, 11):
        top = ((1 << (e - 1)) * P) >> 377
        b = [MASK] * 13 + [top - 1]
        assert val(b) <= (1 << (e - 1)) * P
        o = arr([0] * 14)
        lib.h_fp29_sub(o, arr([0] * 14), arr(b), e)
        assert val(list(o)) == (1 << e) * P - val(b)


def test_zero_test_and_words(lib):
    rng = random.Random(3)
    assert lib.h_fp29_is_zero_mod_p(arr(limbs(0))) == 1 and lib.h_fp29_is_zero_mod_p(arr(limbs(P))) == 1
    for v in (1, P - 1, P + 1, 2 * P - 1):
        assert lib.h_fp29_is_zero_mod_p(arr(limbs(v))) == 0
    for _ in range(500):
        v = rng.randrange(1 << 384)
        w = arr([(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)], 12)
        o = arr([0] * 14)
        lib.h_fp29_from_words(o, w)
        assert list(o) == limbs(v)
        w2 = arr([0] * 12, 12)
        lib.h_fp29_to_words(w2, o)
        assert list(w2) == list(w)
