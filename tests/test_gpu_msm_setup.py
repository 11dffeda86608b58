"""kzg_g1_msm_setup: G1Projective::msm_variable_base (call sites src/kzg_proof.rs:419,429,430) over the handle's own Lagrange
points (src/trusted_setup.rs:20-26) - term i = scalars[i] x g1_points[i mod 4096] - against the CPU oracle's MSM.  Both forms
(csrc/capi_prover.hpp): the fixed-base form of csrc/msm_fixed.hpp (16-bit signed windows, partitioned bucket lists; the default at every
size) and the verification path's window kernel over the setup's affine table rows (the fallback, forced through KZG_OPTIONS here).  By linearity the expected value is the oracle's
4 096-term MSM over each point's scalars summed mod r; bit-exact."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from kzg_rs_amd import api
from kzg_rs_amd.api import KzgError, KzgSettings

pytestmark = pytest.mark.gpu
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_INF = bytes([0xC0]) + bytes(47)
N = 4096


@pytest.fixture(scope="module")
def settings():
    return KzgSettings.load_trusted_setup_file()


@pytest.fixture(scope="module")
def base():
    """g1_points as the handle keeps them: the file's Lagrange points, bit-reversal permuted (build.rs:79,89-105)"""
    ts = open(os.path.join(O.ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    brp = lambda i: int(format(i, "012b")[::-1], 2)
    return [bytes.fromhex(ts[2 + brp(i)]) for i in range(N)]


def _expected(base, sc):
    """sc: (n, 32) uint8 big-endian rows"""
    n = len(sc)
    sums = [0] * min(n, N)
    for i in range(n):
        sums[i % N] += int.from_bytes(sc[i].tobytes(), "big")
    if not sums:
        return G1_INF
    return O.g1_msm(b"".join(base[: len(sums)]), b"".join((v % R).to_bytes(32, "big") for v in sums), len(sums))


def _call(settings, sc):
    sc = np.ascontiguousarray(sc, dtype=np.uint8)
    out = C.create_string_buffer(48)
    api._chk(api.lib().kzg_g1_msm_setup(out, sc.ctypes.data_as(C.c_char_p), len(sc), settings._h))
    return out.raw


def _random_scalars(n, seed):
    sc = np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=(n, 32), dtype=np.uint8)
    if n > 3:
        sc[n // 2] = 0                                                      # a zero scalar
        sc[n - 1] = np.frombuffer((R - 1).to_bytes(32, "big"), dtype=np.uint8)
        sc[1] = 0xFF                                                        # 2^256 - 1: reduced mod r twice (Scalar::from_raw)
        sc[2] = np.frombuffer(R.to_bytes(32, "big"), dtype=np.uint8)        # r itself -> 0
    return sc


def test_setup_point_matches_the_handle(settings, base):
    for i in (0, 1, 2, 777, 4095):
        assert settings.g1_point(i) == base[i]
    one = np.zeros((1, 32), dtype=np.uint8)
    one[0, 31] = 1
    assert _call(settings, one) == base[0]            # 1 x g1_points[0]
    assert _call(settings, np.zeros((0, 32), dtype=np.uint8)) == G1_INF  # the empty sum


@pytest.mark.parametrize("n", [1, 2, 5, 767, 768, 769, 1023, 1025, 4095, 4096, 4097, 12289, 32767, 32768, 32769, 100_003, 1 << 18])
def test_g1_msm_setup_vs_oracle(settings, base, n):
    """sizes on either side of the point count, of the fixed-base form's first full slice (768 terms x 16 windows = 12 288 entries), of a
    workgroup's term block (1 024) and around 32 768; random scalars over the whole 256-bit range with 0, r - 1, r and 2^256 - 1 among them"""
    sc = _random_scalars(n, 100 + n)
    assert _call(settings, sc) == _expected(base, sc)


def test_g1_msm_setup_fixed_base_digit_edges(settings, base):
    """The fixed-base form's signed 16-bit digits (csrc/msm_fixed.hpp fb_digits): windows equal to 0x8000 (the one bucket of
    partition 128), 0x8001 (first negative digit), runs of 0xFFFF (a carry through every window), 0x7FFF, 1 and 0; scalars that
    put EVERY entry into one bucket (one workgroup slice after the other of the same partition, and P + P in every bucket: terms
    i and i + 4 096 carry the same row), and scalars with a single non-zero window."""
    n = 40_000
    pats = [bytes.fromhex(h) for h in (
        "0000" * 15 + "8000", "0000" * 15 + "8001", "0000" * 15 + "7fff", "8000" * 16, "0" * 60 + "ffff", "7fff" * 16,
        "0001" + "ffff" * 15, "0" * 63 + "1", "0" * 64, "00ff" * 16, "0100" * 16, "7fff" + "8000" * 15)]
    sc = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        sc[i] = np.frombuffer(pats[(i // 7) % len(pats)], dtype=np.uint8)
    assert _call(settings, sc) == _expected(base, sc)
    same = np.zeros((n, 32), dtype=np.uint8)
    same[:] = np.frombuffer(bytes.fromhex("1234" * 16), dtype=np.uint8)   # every window 0x1234: one bucket, 16 n entries
    same[:, 0] &= 0x0F
    assert _call(settings, same) == _expected(base, same)
    single = np.zeros((n, 32), dtype=np.uint8)
    rng = random.Random(5)
    for i in range(n):
        v = rng.randrange(16)
        single[i, 30 - 2 * v: 32 - 2 * v] = [rng.randrange(256), rng.randrange(256)]
    assert _call(settings, single) == _expected(base, single)


def test_g1_msm_setup_forms_agree():
    """The same sums through both forms forced by KZG_OPTIONS g1_msm_setup_form (read once per process): the window kernel over
    the setup's affine rows and the fixed-base form, at sizes on both sides of the default switch - byte for byte."""
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from kzg_rs_amd import api\n"
        "s = api.KzgSettings.load_trusted_setup_file()\n"
        "for n in (3, 4097, 20000, 70001):\n"
        "    sc = np.random.Generator(np.random.PCG64(n)).integers(0, 256, size=(n, 32), dtype=np.uint8)\n"
        "    out = C.create_string_buffer(48)\n"
        "    api._chk(api.lib().kzg_g1_msm_setup(out, sc.ctypes.data_as(C.c_char_p), n, s._h))\n"
        "    print('SUM', n, out.raw.hex())\n" % O.ROOT)
    outs = []
    for form in ("window", "fixed", ""):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KZG_OPTIONS="g1_msm_setup_form=" + form if form else ""),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (form, r.stdout[-2000:], r.stderr[-2000:])
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("SUM")])
    assert len(outs[0]) == 4 and outs[0] == outs[1] == outs[2]


def test_g1_msm_setup_needs_a_loaded_setup():
    from kzg_rs_amd import synth
    st = KzgSettings.from_tau_g2(synth.synthetic_setup()[1])   # a custom handle carries no G1 section
    with pytest.raises(KzgError):
        api.g1_msm_setup([bytes(31) + b"\x01"], st)


def test_g1_msm_setup_equals_g1_msm_on_tiled_points(settings, base):
    """... and the arbitrary-point entry point over the same points tiled explicitly (decode + tables per call) gives the same sum"""
    n = 50_000
    sc = _random_scalars(n, 9)
    pts = (b"".join(base) * (n // N + 1))[: 48 * n]
    out = C.create_string_buffer(48)
    api._chk(api.lib().kzg_g1_msm(out, pts, sc.ctypes.data_as(C.c_char_p), n, settings._h))
    assert out.raw == _call(settings, sc)


def test_window_form_fold_tree_with_a_ragged_second_level(base):
    """Round-5 advisor finding: above ~4.2 M terms the window form's large-sum tail is off (its save area is capped) and the window
    sums are folded by trees of 64.  With 2 S = 4 160 slice layers (n ~ 12.78 M) the first level leaves 65 partial sums and the
    second reads them as two groups of 64: entries 65 .. 127 must be identities, not whatever the allocation held.  The window form
    is forced (the default at this size is the fixed-base form); scalars come from a table of 64 values so that the expected sum is
    cheap: per point, how often each table entry occurs x the entry."""
    n, K = 12_779_000, 64
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from kzg_rs_amd import api\n"
        "s = api.KzgSettings.load_trusted_setup_file()\n"
        "n, K = %d, %d\n"
        "tab = np.random.Generator(np.random.PCG64(77)).integers(0, 256, size=(K, 32), dtype=np.uint8)\n"
        "tab[:, 0] &= 0x3F\n"
        "idx = (np.arange(n, dtype=np.int64) * 7 + np.arange(n, dtype=np.int64) // 4096) %% K\n"
        "sc = np.ascontiguousarray(tab[idx])\n"
        "out = C.create_string_buffer(48)\n"
        "for _ in range(2):\n"   # (twice: the second call finds the scratch of the first, not fresh memory)
        "    api._chk(api.lib().kzg_g1_msm_setup(out, sc.ctypes.data_as(C.c_char_p), n, s._h))\n"
        "    print('SUM', out.raw.hex())\n" % (O.ROOT, n, K))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KZG_OPTIONS="g1_msm_setup_form=window"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    got = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("SUM")]
    tab = np.random.Generator(np.random.PCG64(77)).integers(0, 256, size=(K, 32), dtype=np.uint8)
    tab[:, 0] &= 0x3F
    vals = [int.from_bytes(tab[k].tobytes(), "big") for k in range(K)]
    i = np.arange(n, dtype=np.int64)
    cnt = np.bincount((i % N) * K + (i * 7 + i // 4096) % K, minlength=N * K).reshape(N, K)
    sums = [sum(int(c) * v for c, v in zip(row, vals) if c) % R for row in cnt]
    want = O.g1_msm(b"".join(base), b"".join(v.to_bytes(32, "big") for v in sums), N)
    assert got == [want.hex(), want.hex()]


def test_g1_msm_setup_two_million_terms(settings, base):
    """beyond BASELINE configs[3]'s size: 2^21 + 5 terms (5 590 workgroups of the fixed-base form, a fold over as many layers)"""
    n = (1 << 21) + 5
    sc = _random_scalars(n, 21)
    words = np.ascontiguousarray(sc).view(">u4")          # (n, 8): the scalars' 32-bit words, most significant first
    idx = np.arange(n) % N
    tot = [0] * N
    for k in range(8):                                       # per point, the sum of word k over its ~513 terms: below 2^42, exact in a double
        part = np.bincount(idx, weights=words[:, k].astype(np.float64), minlength=N)
        for j in range(N):
            tot[j] += int(part[j]) << (32 * (7 - k))
    want = O.g1_msm(b"".join(base), b"".join((v % R).to_bytes(32, "big") for v in tot), N)
    assert _call(settings, sc) == want


def test_g1_msm_setup_with_an_identity_among_the_setup_points():
    """A trusted setup whose G1 section holds the point at infinity (the loader decodes unchecked, build.rs:66-70; the identity is a
    member of G1): its terms add nothing - in the default (fixed-base) form here, and in the window form through a child process."""
    ts = open(os.path.join(O.ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    brp = lambda i: int(format(i, "012b")[::-1], 2)
    lines = list(ts)
    lines[2 + brp(5)] = G1_INF.hex()       # g1_points[5] of the handle
    lines[2 + brp(4095)] = G1_INF.hex()
    txt = "\n".join(lines).encode()
    st = KzgSettings.load_trusted_setup_text(txt)
    pts = [bytes.fromhex(lines[2 + brp(i)]) for i in range(N)]
    assert st.g1_point(5) == G1_INF and st.g1_point(6) == pts[6]
    want = {}
    for n in (4096 + 17, 40_000):
        sc = _random_scalars(n, 300 + n)
        want[n] = _expected(pts, sc)
        assert _call(st, sc) == want[n]
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from kzg_rs_amd import api\n"
        "st = api.KzgSettings.load_trusted_setup_text(open(sys.argv[1], 'rb').read())\n"
        "for n in (4096 + 17, 40000):\n"
        "    sc = np.random.Generator(np.random.PCG64(300 + n)).integers(0, 256, size=(n, 32), dtype=np.uint8)\n"
        "    sc[n // 2] = 0; sc[n - 1] = np.frombuffer((%d - 1).to_bytes(32, 'big'), dtype=np.uint8); sc[1] = 0xFF; sc[2] = np.frombuffer((%d).to_bytes(32, 'big'), dtype=np.uint8)\n"
        "    out = C.create_string_buffer(48)\n"
        "    api._chk(api.lib().kzg_g1_msm_setup(out, sc.ctypes.data_as(C.c_char_p), n, st._h))\n"
        "    print('SUM', n, out.raw.hex())\n" % (O.ROOT, R, R))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".txt") as f:
        f.write(txt)
        f.flush()
        r = subprocess.run([sys.executable, "-c", code, f.name], env=dict(os.environ, KZG_OPTIONS="g1_msm_setup_form=window"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    got = {int(ln.split()[1]): ln.split()[2] for ln in r.stdout.splitlines() if ln.startswith("SUM")}
    assert got == {n: w.hex() for n, w in want.items()}


def test_g1_msm_setup_randomised_check():
    """12 s of tools/fuzz_g1_msm_setup.py: random sizes on both sides of the switch to the fixed-base form, scalar patterns that collide in its
    buckets and partitions (one bucket for the whole call, the recoding's edge digits, one window per term ...), every sum against the
    oracle.  (Longer runs: `python tools/fuzz_g1_msm_setup.py 600 <seed>`.)"""
    out = subprocess.run([sys.executable, os.path.join(O.ROOT, "tools", "fuzz_g1_msm_setup.py"), "12", "20261004"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "no mismatch" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_g1_msm_setup_argument_checks(settings):
    L = api.lib()
    out = C.create_string_buffer(48)
    assert L.kzg_g1_msm_setup(out, None, 5, settings._h) != 0                       # null scalars with n > 0
    assert L.kzg_g1_msm_setup(None, bytes(32), 1, settings._h) != 0                 # null output
    assert L.kzg_g1_msm_setup(out, bytes(32), (1 << 26) + 1, settings._h) != 0      # beyond the documented limit (checked before anything is read)
    assert b"2^26" in L.kzg_last_error()
    assert L.kzg_g1_msm_setup(out, None, 0, settings._h) == 0 and out.raw == G1_INF  # the empty sum needs no scalars
