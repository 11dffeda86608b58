"""Durations of every dispatch of the kernels of N back-to-back verify_kzg_proof calls, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d DIR -o run --output-format csv -- python3 tools/prof/kernel_duration_modes.py run [N]
    python3 tools/prof/kernel_duration_modes.py show DIR
(shows whether a kernel's time from call to call is one value or several modes)"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    from kzg_rs_amd import synth
    from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof
    cs, zs, ys, ps, st = synth.make_valid_proofs(4, seed=9)
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        assert KzgProof.verify_kzg_proof(Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st)
else:
    path = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(path)):
        d.setdefault(r["Kernel_Name"].split("(")[0][-48:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in d.items():
        if max(v) > 0.2 and len(v) >= 10:
            v = v[-40:]
            print("%-50s n=%d  %s" % (k, len(v), " ".join("%.2f" % x for x in v)))
