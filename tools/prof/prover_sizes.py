import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth
st = api.KzgSettings.load_trusted_setup_file()
blobs, _, _, _ = synth.make_valid_batch(64, seed=3)
bl = [blobs[i].tobytes() for i in range(64)]
for n in (1, 2, 6, 12, 16, 17, 32, 64):
    api.blob_to_kzg_commitment(bl[:n], st)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); cs = api.blob_to_kzg_commitment(bl[:n], st); ts.append((time.perf_counter() - t0) * 1e3)
    tc = sorted(ts)[3]
    api.compute_blob_kzg_proof(bl[:n], cs, st)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); ps = api.compute_blob_kzg_proof(bl[:n], cs, st); ts.append((time.perf_counter() - t0) * 1e3)
    tp = sorted(ts)[3]
    ok = api.KzgProof.verify_blob_kzg_proof_batch([api.Blob(b) for b in bl[:n]], [api.Bytes48(c) for c in cs], [api.Bytes48(p) for p in ps], st)
    print("n = %2d commit %.2f ms proof %.2f ms verified %s" % (n, tc, tp, ok))
