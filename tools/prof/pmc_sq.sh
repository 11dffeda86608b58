#!/bin/bash
# SQ-level PMC pass over one launch group (single stream so dispatches do not overlap).
# usage: tools/prof/pmc_sq.sh <tag> <group> "<counters>"
set -u
tag=$1; group=$2; ctrs=$3
export TMPDIR=/tmp KZG_OPTIONS=single_stream=1
out=gpurun_out/pmc_${tag}
rm -rf $out; mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --no-cpu-baseline --group $group --inflight 1 --steps $group --warmup 0 > $out.log 2>&1
python3 tools/prof/pmc_summary.py $out
