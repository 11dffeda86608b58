"""Aggregate rate of SMALL host batches from several host threads, one settings handle per thread (SURVEY 8b threading): T
threads each verify 6-blob batches (a block's worth) in a loop.   python tools/prof/concurrent_small_batches.py"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth  # noqa: E402

N, CALLS = 6, 150
blobs, cs, ps, st0 = synth.make_valid_batch(N, seed=17)
tau_g2 = synth.synthetic_setup()[1]
raw = (blobs.tobytes(), b"".join(cs), b"".join(ps))
L = api.lib()
for T in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]):
    handles = [api.KzgSettings.from_tau_g2(tau_g2) for _ in range(T)]
    bad = []

    def work(h):
        ok = C.c_bool(False)
        for _ in range(CALLS):
            rc = L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), raw[0], raw[1], raw[2], N, h._h)
            if rc != 0 or not ok.value:
                bad.append(rc)

    for h in handles:
        work_one = threading.Thread(target=work, args=(h,))
        work_one.start(); work_one.join()  # warm-up, one handle at a time
    threads = [threading.Thread(target=work, args=(h,)) for h in handles]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    assert not bad, bad[:3]
    print("%2d threads x own handle: %6.0f batches/s = %7.0f blobs/s  (%.2f ms per call per thread)" % (T, T * CALLS / dt, T * CALLS * N / dt, dt / CALLS * 1e3))
    for h in handles:
        h.close()
