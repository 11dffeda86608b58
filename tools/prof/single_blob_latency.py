"""verify_blob_kzg_proof (ONE blob, from host memory) back to back: the call's latency.   python tools/prof/single_blob_latency.py"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth
from kzg_rs_amd.api import Blob, Bytes48, KzgProof
blobs, cs, ps, st = synth.make_valid_batch(4, seed=3, chunk=4)
b = Blob(blobs[0].tobytes()); c = Bytes48(cs[0]); p = Bytes48(ps[0])
for _ in range(3): assert KzgProof.verify_blob_kzg_proof(b, c, p, st)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); assert KzgProof.verify_blob_kzg_proof(b, c, p, st); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort(); print("verify_blob_kzg_proof (host blob): min %.2f median %.2f ms" % (ts[0], ts[10]))
