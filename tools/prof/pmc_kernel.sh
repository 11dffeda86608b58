#!/bin/bash
# One PMC pass of the bench's launch group (single stream), counters of the kernels matching a pattern:
#   tools/prof/pmc_kernel.sh <pattern> <counter> [<counter> ...]     (run through gpurun from the repo root)
set -u
pat=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_one
rm -rf $out; mkdir -p $out
KZG_OPTIONS=single_stream=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency --no-self-check --group 256 --inflight 1 --steps 1 --warmup 0 > $out.log 2>&1
python3 - "$pat" $out/run_counter_collection.csv <<'PY'
import csv, sys, collections
pat, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if pat in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k[:70])
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print("   %-36s %16.0f  (%d dispatches, last-group share: /%d = %.0f)" % (c, v, n, n, v / n))
PY
