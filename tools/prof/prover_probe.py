"""Prover-side entry points (c-kzg-4844's names): median wall-clock of blob_to_kzg_commitment and
compute_blob_kzg_proof for n host blobs per call; every output checked by the verifier.   python tools/prof/prover_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth  # noqa: E402

st = api.KzgSettings.load_trusted_setup_file()
blobs, _, _, _ = synth.make_valid_batch(64, seed=3)
bl = [blobs[i].tobytes() for i in range(64)]
for n in (1, 6, 64):
    api.blob_to_kzg_commitment(bl[:n], st)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        cs = api.blob_to_kzg_commitment(bl[:n], st)
        ts.append((time.perf_counter() - t0) * 1e3)
    tc = sorted(ts)[3]
    api.compute_blob_kzg_proof(bl[:n], cs, st)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        ps = api.compute_blob_kzg_proof(bl[:n], cs, st)
        ts.append((time.perf_counter() - t0) * 1e3)
    tp = sorted(ts)[3]
    ok = api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in bl[:n]], [api.Bytes48(c) for c in cs], [api.Bytes48(p) for p in ps], st)
    assert ok is True
    print("n = %2d   blob_to_kzg_commitment %.2f ms (%.2f per blob)   compute_blob_kzg_proof %.2f ms (%.2f per blob)   verified: %s"
          % (n, tc, tc / n, tp, tp / n, ok))
