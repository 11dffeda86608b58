"""Randomised check of kzg_g1_msm_setup (the fixed-base form of csrc/msm_fixed.hpp - the default at every size - or, under
KZG_OPTIONS=g1_msm_setup_form=window, the window kernel over the setup's affine rows) against the CPU oracle by linearity: term i uses g1_points[i mod 4096], so the sum must equal
the oracle's MSM over the first min(n, 4096) setup points with each point's scalars summed mod r.  Scalars are drawn from patterns that
collide in the fixed-base form's buckets and partitions: uniform 256-bit values, ONE 16-bit window pattern repeated in every window and
term (every entry in one bucket), one window set per term, digits at the recoding's edges (0x7fff / 0x8000 / 0x8001 / 0xffff), small
values, mostly-zero with a populated tail, r - 1 and 1 in turn.
    python tools/fuzz_g1_msm_setup.py [seconds] [seed]"""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from kzg_rs_amd import api  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
N = 4096
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
st = api.KzgSettings.load_trusted_setup_file()
L = api.lib()
ts = open(os.path.join(ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
brp = lambda i: int(format(i, "012b")[::-1], 2)
base = [bytes.fromhex(ts[2 + brp(i)]) for i in range(N)]
EDGE = [bytes.fromhex(x) for x in ("7fff", "8000", "8001", "ffff", "0000", "0001", "00ff", "0100", "4000", "c000")]


def expected(sc):
    n = len(sc)
    m = min(n, N)
    words = np.ascontiguousarray(sc).view(">u4")
    idx = np.arange(n) % N
    tot = [0] * m
    for k in range(8):
        part = np.bincount(idx, weights=words[:, k].astype(np.float64), minlength=N)
        for j in range(m):
            tot[j] += int(part[j]) << (32 * (7 - k))
    return O.g1_msm(b"".join(base[:m]), b"".join((v % R).to_bytes(32, "big") for v in tot), m)


t_end = time.time() + seconds
cases, sizes = 0, []
out = C.create_string_buffer(48)
while time.time() < t_end:
    n = rng.choice([rng.randrange(1, 5000), rng.randrange(5000, 32_768), 32_767, 32_768, rng.randrange(32_769, 150_000), rng.randrange(150_000, 600_000)])
    g = np.random.Generator(np.random.PCG64(rng.randrange(1 << 62)))
    kind = rng.randrange(8)
    if kind == 0:
        sc = g.integers(0, 256, size=(n, 32), dtype=np.uint8)
    elif kind == 1:     # one window pattern everywhere: every entry of the call in ONE bucket (and P + P whenever a point repeats)
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:] = np.frombuffer(rng.choice(EDGE[:4] + [bytes([rng.randrange(1, 0x70), rng.randrange(256)])]) * 16, dtype=np.uint8)
        sc[:, 0] &= 0x3F
    elif kind == 2:     # one window set per term
        sc = np.zeros((n, 32), dtype=np.uint8)
        v = g.integers(0, 16, size=n)
        val = g.integers(0, 256, size=(n, 2), dtype=np.uint8)
        sc[np.arange(n), 30 - 2 * v] = val[:, 0]
        sc[np.arange(n), 31 - 2 * v] = val[:, 1]
        sc[:, 0] &= 0x3F
    elif kind == 3:     # every window of every term one of the recoding's edge digits
        choice = g.integers(0, len(EDGE), size=(n, 16))
        tab = np.frombuffer(b"".join(EDGE), dtype=np.uint8).reshape(len(EDGE), 2)
        sc = tab[choice].reshape(n, 32).copy()
    elif kind == 4:     # small scalars
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:, 29:] = g.integers(0, 256, size=(n, 3), dtype=np.uint8)
    elif kind == 5:     # mostly zero, a populated tail
        sc = np.zeros((n, 32), dtype=np.uint8)
        m = rng.randrange(1, min(n, 3000) + 1)
        sc[n - m:] = g.integers(0, 256, size=(m, 32), dtype=np.uint8)
    elif kind == 6:     # r - 1 and 1 in turn (sums cancel point by point when the term count per point is even)
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[0::2] = np.frombuffer((R - 1).to_bytes(32, "big"), dtype=np.uint8)
        sc[1::2, 31] = 1
    else:               # all 0xff (2^256 - 1: reduced twice) with random holes
        sc = np.full((n, 32), 0xFF, dtype=np.uint8)
        sc[g.integers(0, n, size=n // 3)] = 0
    sc = np.ascontiguousarray(sc)
    api._chk(L.kzg_g1_msm_setup(out, sc.ctypes.data_as(C.c_char_p), n, st._h))
    want = expected(sc)
    if out.raw != want:
        np.save("/tmp/fuzz_g1_msm_setup_fail.npy", sc)
        print("MISMATCH", {"n": n, "kind": kind, "seed": seed, "case": cases, "got": out.raw.hex(), "want": want.hex()})
        sys.exit(1)
    cases += 1
    sizes.append(n)
print("fuzz_g1_msm_setup: %d cases in %.0f s, no mismatch (seed %d; sizes %d .. %d; form: %s)"
      % (cases, seconds, seed, min(sizes), max(sizes), os.environ.get("KZG_OPTIONS", "") or "default (fixed-base)"))
