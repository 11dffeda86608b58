"""Latency of the reference-shaped calls on FEW blobs from host memory: verify_blob_kzg_proof (one blob) and
verify_blob_kzg_proof_batch of n = 2 .. 1 024 host blobs, median of 24 calls each; results checked (valid -> True, a wrong proof
-> False).  KZG_OPTIONS=host_challenge_max_blobs=0 gives the GPU-hashed figures for comparison.
    python tools/prof/small_host_batch_latency.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth  # noqa: E402

blobs, cs, ps, st = synth.make_valid_batch(1024, seed=11, chunk=1024)
L = api.lib()
ok = C.c_bool(False)


def call(n, proofs):
    api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), b"".join(cs[:n]), b"".join(proofs[:n]), n, st._h))
    return bool(ok.value)


for n in (1, 2, 4, 6, 9, 16, 32, 64, 65, 96, 128, 192, 256, 512, 1024):
    bad = list(ps)
    bad[n - 1] = ps[n % 1024] if n < 1024 else ps[0]
    assert call(n, ps) is True and call(n, bad) is False, n
    ts = []
    for _ in range(24):
        t0 = time.perf_counter()
        call(n, ps)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("n = %4d   min %.3f  median %.3f  max %.3f ms" % (n, ts[0], ts[12], ts[-1]))
