import os, sys, time
sys.path.insert(0, os.getcwd())
from kzg_rs_amd import api, synth
st = api.KzgSettings.load_trusted_setup_file()
blobs, _, _, _ = synth.make_valid_batch(2, seed=3)
bl = [blobs[i].tobytes() for i in range(2)]
for _ in range(3): cs = api.blob_to_kzg_commitment(bl[:1], st)
t0 = time.perf_counter(); cs = api.blob_to_kzg_commitment(bl[:1], st); print("commit ms", (time.perf_counter() - t0) * 1e3)
for _ in range(2): ps = api.compute_blob_kzg_proof(bl[:1], cs, st)
t0 = time.perf_counter(); ps = api.compute_blob_kzg_proof(bl[:1], cs, st); print("proof ms", (time.perf_counter() - t0) * 1e3)
