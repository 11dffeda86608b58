#!/bin/bash
# Where k_msm_window's HBM traffic comes from (round-5 verdict: 12x its algorithmic bytes): FETCH_SIZE / WRITE_SIZE / SQ counters of the
# window kernel and the bucket reduction behind it, one launch group of the bench's size on a single stream, with the XCD placement of
# the 8 window workgroups of a (batch, output) ON (default) and OFF (A/B build, KZG_OPTIONS msm_xcd=0).
#   tools/prof/ab_msm_rows.sh > gpurun_out/r6_ab_msm_rows_raw.txt        (through gpurun, from the repo root)
set -u
AB=$PWD/kzg_rs_amd/libkzg_rs_amd_ab.so
for xcd in 1 0; do
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
    echo "=== msm_xcd=$xcd  counters: $ctr"
    KZG_LIB_OVERRIDE=$AB KZG_OPTIONS="single_stream=1;msm_xcd=$xcd" tools/prof/pmc_probe.sh k_msm_ "$ctr" bench.py --no-cpu-baseline --no-latency --no-self-check --group 256 --inflight 1 --steps 1 --warmup 0
  done
done
