#!/bin/bash
# Kernel timelines of ONE 1 024-blob batch and ONE verify_kzg_proof call (run through gpurun from the repo root):
#   tools/prof/collect_timelines.sh <tag>      -> gpurun_out/<tag>_single_batch_timeline.txt
# (rocprofv3 may crash at exit after such a short script; the trace is complete by then)
set -u
tag=$1
export TMPDIR=/tmp
out=gpurun_out/${tag}_single_batch_timeline.txt
: > $out
for what in batch proof; do
    rm -rf /tmp/tl_$what
    timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$what -o run --output-format csv -- python3 tools/prof/call_timeline.py run $what > /dev/null 2>&1
    if [ $what = batch ]; then echo "== one verify_blob_kzg_proof_batch call of 1 024 device-resident blobs" >> $out
    else echo "" >> $out; echo "== one verify_kzg_proof call (round 4: three streams - k_proof_select + SCALARS | k_proof_decompress: the two square roots | the full decode for the subgroup verdict - then VERIFY3; DESIGN.md section 3.7)" >> $out; fi
    python3 tools/prof/call_timeline.py show /tmp/tl_$what | cut -c1-160 >> $out
done
cat $out
