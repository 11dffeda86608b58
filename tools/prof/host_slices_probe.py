"""One verify_blob_kzg_proof_batch call from host memory (1 024 blobs, pageable) at every slicing: wall time per call and
the library's own intervals (phase 1 = copies + challenge + evaluate + decode; challenge interval ev0 -> ev7 includes the
copies).    python3 tools/prof/host_slices_probe.py [n]      (each slicing in a child process: the switch is read once)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r)
import torch
from kzg_rs_amd import api, synth
n = %d
blobs, cs, ps, st = synth.make_valid_batch(n, seed=5, chunk=1024)
hc, hp = b"".join(cs), b"".join(ps)
ok = C.c_bool(False)
def call():
    api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st._h))
    assert ok.value
for _ in range(3): call()
ts = []
for _ in range(12):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
tm = st.last_timings()
print("wall median %%.3f ms min %%.3f | device %%.3f phase1 %%.3f challenge(+copies) %%.3f evaluate %%.3f decode %%.3f msm %%.3f pairing %%.3f" %% (
    sorted(ts)[6] * 1e3, min(ts) * 1e3, tm[0], tm[1], tm[5], tm[4], tm[6], tm[2], tm[3]))
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for s in ("1", "2", "4", "8", "16"):
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n)], env=dict(os.environ, KZG_OPTIONS="host_slices=" + s), capture_output=True, text=True)
    print("host_slices=%-2s %s" % (s, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))
