import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import golden_data as G
import oracle_lib as O
from kzg_rs_amd import api, synth
from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof
st = api.KzgSettings.load_trusted_setup_file()
bad = 0
V = G.vectors()["verify_kzg_proof"]
for c in V:
    try:
        got = KzgProof.verify_kzg_proof(Bytes48.from_hex(c["commitment"]), Bytes32.from_hex(c["z"]), Bytes32.from_hex(c["y"]), Bytes48.from_hex(c["proof"]), st)
    except api.KzgError:
        got = None
    if got != c["output"]:
        bad += 1
        print("MISMATCH", c["name"], got, c["output"])
print("vectors", len(V), "mismatches", bad)
cs, zs, ys, ps, st2 = synth.make_valid_proofs(4, seed=9)
args = (Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st2)
for _ in range(8): assert KzgProof.verify_kzg_proof(*args)
ts = []
for _ in range(64):
    t0 = time.perf_counter(); assert KzgProof.verify_kzg_proof(*args); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("verify_kzg_proof ms: min %.3f median %.3f max %.3f; device interval %.3f, pairing %.3f" % (ts[0], ts[32], ts[-1], st2.last_timings()[0], st2.last_timings()[3]))
assert KzgProof.verify_kzg_proof(Bytes48(cs[0]), Bytes32(zs[1]), Bytes32(ys[0]), Bytes48(ps[0]), st2) is False
