"""Rate of the host-fed stream entry point (kzg_verify_blob_kzg_proof_batches) for a few chunk sizes:
    the option host_chunk is read once per process, so each setting runs in a child process.
    python tools/prof/host_stream_rate.py [n_batches]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CODE = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
from kzg_rs_amd import api, synth
n, NB = 1024, %d
blobs, cs, ps, st = synth.make_valid_batch(n, seed=3, chunk=1024)
many = np.ascontiguousarray(np.broadcast_to(blobs, (NB,) + blobs.shape)).reshape(NB * n, -1)
hc, hp = b"".join(cs) * NB, b"".join(ps) * NB
for rep in range(3):
    t = time.perf_counter()
    res = api.verify_blob_kzg_proof_batches(many.ctypes.data, hc, hp, n, NB, st)
    dt = time.perf_counter() - t
    assert all(res)
    print("pass %%d: %%.1f ms, %%.0f blobs/s, %%.1f GB/s" %% (rep, dt * 1e3, n * NB / dt, n * NB * 131168 / dt / 1e9))
""" % (ROOT, NB)
for chunk in (2, 4, 8, 16, 32):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, KZG_OPTIONS="host_chunk=%d" % chunk), capture_output=True, text=True)
    print("host_chunk=%d" % chunk)
    print(out.stdout.strip() if out.returncode == 0 else out.stderr[-800:])
