#!/bin/bash
# PMC counters of the kernels of 16 back-to-back verify_kzg_proof calls, per dispatch (run through gpurun from the repo root):
#   tools/prof/pmc_proof_calls.sh <kernel pattern> <counter> [<counter> ...]
set -u
pat=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_proof
rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $out -o run --output-format csv -- python3 tools/prof/kernel_duration_modes.py run 16 > $out.log 2>&1
python3 - "$pat" $out <<'PY'
import csv, sys, collections, glob, os
pat, d = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in rows.values() for c in v})
print(" ".join("%18s" % n[-18:] for n in names))
for k, v in list(rows.items())[-16:]:
    print(" ".join("%18.0f" % v.get(n, -1) for n in names))
PY
