"""ONE shared settings handle, T host threads (std::threads inside the library: kzg_debug_concurrent_callers), small calls:
verify_kzg_proof, 6-blob verify_blob_kzg_proof_batch, 1-blob verify_blob_kzg_proof.  Prints calls/s, latency and what the
small-call queue did (launches, items per launch) at each T.
    python tools/prof/concurrent_callers.py [--lanes 2,3,4] [--threads 1,8,64,256] [--seconds 2] [--kinds proof,blobs6,blob1]
(GPU_MAX_HW_QUEUES is taken from the environment; api sets 8 when unset.)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", default="3")
ap.add_argument("--threads", default="1,8,64,256")
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--kinds", default="proof,blobs6,blob1")
ap.add_argument("--json", default="")
args = ap.parse_args()

tau, tau_g2 = synth.synthetic_setup()
st0 = api.KzgSettings.from_tau_g2(tau_g2)
NP = 256
cs, zs, ys, ps, _ = synth.make_valid_proofs(NP, seed=3, settings=st0)
ys = list(ys)
exp_p = bytearray([1] * NP)
for i in range(0, NP, 16):  # every 16th tuple carries a wrong y: the answers are checked, true and false
    ys[i] = ys[(i + 1) % NP]
    exp_p[i] = 0
NB = 48
blobs, bc, bp, _ = synth.make_valid_batch(NB, seed=4, settings=st0)
bp = list(bp)
exp6 = bytearray([1] * (NB // 6))
bp[6 * 3 + 2] = bp[6 * 3 + 3]  # call 3 of the 6-blob calls: a swapped proof -> false
exp6[3] = 0
exp1 = bytearray([1] * NB)
exp1[6 * 3 + 2] = 0
raw_blobs = blobs.tobytes()
out = {"hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "runs": []}
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES"))
def cpu_stat():
    out = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            out[k] = int(v)
    except Exception:
        pass
    return out


for lanes in [int(x) for x in args.lanes.split(",")]:
    with api.options(small_lanes=lanes):
        st = api.KzgSettings.from_tau_g2(tau_g2)
    for kind in args.kinds.split(","):
        for T in [int(x) for x in args.threads.split(",")]:
            st.small_queue_stats(reset=True)
            c0 = cpu_stat()
            if kind == "proof":
                r = st.concurrent_callers("proof", T, args.seconds, b"".join(cs), b"".join(ps), bytes(exp_p), z=b"".join(zs), y=b"".join(ys))
                per = 1
            elif kind == "blobs6":
                r = st.concurrent_callers("blobs", T, args.seconds, b"".join(bc), b"".join(bp), bytes(exp6), blobs=raw_blobs, per_call=6)
                per = 6
            else:
                r = st.concurrent_callers("blobs", T, args.seconds, b"".join(bc), b"".join(bp), bytes(exp1), blobs=raw_blobs, per_call=1)
                per = 1
            c1 = cpu_stat()
            cores = (c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e6 / max(r["seconds"], 1e-9)
            sysc = (c1.get("system_usec", 0) - c0.get("system_usec", 0)) / 1e6 / max(r["seconds"], 1e-9)
            thr = c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0)
            q = st.small_queue_stats()
            r.update(host_cores=round(cores, 2), host_cores_system=round(sysc, 2), throttled_periods=thr)
            r.update(kind=kind, threads=T, lanes=lanes, launches=q["launches"], items_per_launch=q["items"] / max(1, q["launches"]), max_items=q["max_items"],
                     items_per_s=r["calls_per_s"] * per)
            out["runs"].append(r)
            print("lanes %d  %-6s T=%3d: %8.0f calls/s (%8.0f items/s)  mean %6.2f ms  max %6.1f ms  wrong %d   launches %5d  items/launch %6.1f (max %d)  host %.1f cores (%.1f system) throttled %d" % (
                lanes, kind, T, r["calls_per_s"], r["items_per_s"], r["mean_ms"], r["max_ms"], r["wrong"], q["launches"], r["items_per_launch"], q["max_items"],
                cores, sysc, thr), flush=True)
    st.close()
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
