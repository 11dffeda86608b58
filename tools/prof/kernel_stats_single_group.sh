#!/bin/bash
# Per-kernel durations of ONE launch group alone on the chip (single stream, nothing overlapped):
#   tools/prof/kernel_stats_single_group.sh <tag>   -> gpurun_out/<tag>_single_group_stats.txt
set -u
tag=$1
export TMPDIR=/tmp
out=/tmp/sg_$tag
rm -rf $out
KZG_OPTIONS=single_stream=1 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency --no-self-check --group 256 --inflight 1 --steps 3 --warmup 1 > /tmp/sg_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_single_group_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel, calls, avg ms, total ms, share   (one launch group of 256 x 1 024 blobs at a time, single stream; 4 groups)")
for r in rows[:24]:
    print("%-70s %5s %9.3f %9.3f %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat gpurun_out/${tag}_single_group_stats.txt
