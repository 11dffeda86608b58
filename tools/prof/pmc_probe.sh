#!/bin/bash
# One PMC pass over an arbitrary probe script, counters of the kernels matching a pattern (per dispatch averages):
#   tools/prof/pmc_probe.sh <pattern> "<counters>" <script.py> [args...]        (run through gpurun from the repo root)
set -u
pat=$1; ctrs=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/pmc_probe
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace -d $out -o run --output-format csv -- python3 "$@" > $out.log 2>&1
python3 - "$pat" $out <<'PY'
import csv, sys, glob, collections
pat, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(t))}
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if pat in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    n = len(disp[k])
    print(k[:80], "dispatches", n, "avg_ms %.4f" % (sum(dur[i] for i in disp[k]) / n))
    for name, v in sorted(c.items()):
        print("   %-24s %18.0f per dispatch" % (name, v / n))
PY
