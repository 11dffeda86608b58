"""Latency of verify_kzg_proof (one tuple) and rate of verify_kzg_proof_batch (n tuples, one RLC + one pairing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth
from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof

cs, zs, ys, ps, st = synth.make_valid_proofs(8192, seed=9)
wrap = lambda n: ([Bytes48(x) for x in cs[:n]], [Bytes32(x) for x in zs[:n]], [Bytes32(x) for x in ys[:n]], [Bytes48(x) for x in ps[:n]])
a = wrap(1)
for _ in range(3):
    assert KzgProof.verify_kzg_proof(a[0][0], a[1][0], a[2][0], a[3][0], st)
t0 = time.perf_counter()
for _ in range(20):
    KzgProof.verify_kzg_proof(a[0][0], a[1][0], a[2][0], a[3][0], st)
print("verify_kzg_proof: %.2f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
for n in (64, 1024, 8192):
    args = wrap(n)
    assert KzgProof.verify_kzg_proof_batch(*args, st)
    t0 = time.perf_counter()
    for _ in range(5):
        KzgProof.verify_kzg_proof_batch(*args, st)
    dt = (time.perf_counter() - t0) / 5
    print("verify_kzg_proof_batch n=%d: %.2f ms per call = %.0f proofs/s (incl. Python marshalling)" % (n, dt * 1e3, n / dt))
