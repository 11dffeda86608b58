"""Randomised check of kzg_g1_msm at the sizes where the large-sum tail runs (csrc/msm.hpp msm_large_tail: n > 24 576), against the
CPU oracle by linearity: D distinct points (multiples of the generator, with repeats, negatives and the identity among them) tiled
over n terms, scalars drawn from patterns that collide in the buckets (all-equal bytes, one byte set, small, zero, r - 1, random);
the sum must equal the oracle's D-term MSM over each point's scalars summed mod r.
    python tools/fuzz_g1_msm.py [seconds] [seed]"""
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
G1_INF = bytes([0xC0] + [0] * 47)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
st = api.KzgSettings.load_trusted_setup_file()
L = api.lib()
pool_k = [rng.randrange(1, R) for _ in range(700)]
pool = api.g1_mul_generator([k.to_bytes(32, "big") for k in pool_k], st)
t_end = time.time() + seconds
cases = 0
sizes = []
while time.time() < t_end:
    n = rng.choice([rng.randrange(24_577, 60_000), rng.randrange(60_000, 400_000), 24_577, 49_152, 98_305, 1 << 17])
    D = rng.choice([1, 2, 3, 17, 64, 257, 600])
    idx = [rng.randrange(700) for _ in range(D)]
    base = [pool[i] for i in idx]
    if D >= 3 and rng.random() < 0.5:
        base[1] = G1_INF
        base[2] = O.g1_mul(base[0], (R - 1).to_bytes(32, "big"))   # -P beside P
    g = np.random.Generator(np.random.PCG64(rng.randrange(1 << 62)))
    kind = rng.randrange(6)
    if kind == 0:
        sc = g.integers(0, 256, size=(n, 32), dtype=np.uint8)
        sc[:, 0] &= 0x7F
    elif kind == 1:   # all terms the same scalar with all-equal bytes
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:] = rng.randrange(1, 0x70)
    elif kind == 2:   # one byte set, the same position everywhere: one window busy, one or two buckets
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:, rng.randrange(1, 32)] = g.integers(1, 3, size=n, dtype=np.uint8)
    elif kind == 3:   # small scalars
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:, 30:] = g.integers(0, 256, size=(n, 2), dtype=np.uint8)
    elif kind == 4:   # mostly zero, a populated tail
        sc = np.zeros((n, 32), dtype=np.uint8)
        m = rng.randrange(1, 3000)
        sc[n - m:] = g.integers(0, 256, size=(m, 32), dtype=np.uint8)
        sc[:, 0] &= 0x3F
    else:             # r - 1 and 1 in turn
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[0::2] = np.frombuffer((R - 1).to_bytes(32, "big"), dtype=np.uint8)
        sc[1::2, 31] = 1
    pts = (b"".join(base) * (n // D + 1))[: 48 * n]
    out = C.create_string_buffer(48)
    api._chk(L.kzg_g1_msm(out, pts, np.ascontiguousarray(sc).ctypes.data_as(C.c_char_p), n, st._h))
    sums = [0] * D
    rows = [int.from_bytes(sc[i].tobytes(), "big") for i in range(n)]
    for i, v in enumerate(rows):
        sums[i % D] += v
    want = O.g1_msm(b"".join(base), b"".join((v % R).to_bytes(32, "big") for v in sums), D)
    if out.raw != want:
        print("MISMATCH n=%d D=%d kind=%d seed=%d case=%d" % (n, D, kind, seed, cases))
        sys.exit(1)
    cases += 1
    sizes.append(n)
print("g1_msm fuzz seed=%d: %d cases (n from %d to %d), no mismatch" % (seed, cases, min(sizes), max(sizes)))
