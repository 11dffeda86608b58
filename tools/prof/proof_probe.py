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
# many independent proofs through one call (kzg_verify_kzg_proofs): rate per launch size
import ctypes as C
for n in (1, 8, 64, 256, 1024, 4096):
    cs, zs, ys, ps, st3 = synth.make_valid_proofs(n, seed=10 + n, settings=st2)
    raw = [b"".join(x) for x in (cs, zs, ys, ps)]
    ok = (C.c_bool * n)(); err = C.create_string_buffer(n)
    call = lambda: api.lib().kzg_verify_kzg_proofs(ok, err, raw[0], raw[1], raw[2], raw[3], n, st2._h)
    for _ in range(3): assert call() == 0 and all(ok)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("verify_kzg_proofs n=%5d: median %.3f ms = %.0f proofs/s (device interval of the last launch %.3f ms)" % (n, ts[6], n / ts[6] * 1e3, st2.last_timings()[0]))
    if n > 1:
        bt = []
        wrap = ([Bytes48(x) for x in cs], [Bytes32(x) for x in zs], [Bytes32(x) for x in ys], [Bytes48(x) for x in ps])
        for _ in range(6):
            t0 = time.perf_counter(); assert KzgProof.verify_kzg_proof_batch(*wrap, st2); bt.append((time.perf_counter() - t0) * 1e3)
        bt.sort()
        print("   (one-boolean verify_kzg_proof_batch of the same tuples: median %.3f ms incl. the mirror's list handling)" % bt[3])
