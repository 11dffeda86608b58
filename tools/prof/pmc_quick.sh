#!/bin/bash
# The three PMC passes of collect_round.sh alone (no driver-command stats, no final bench): gpurun_out/<tag>_pmc.json and a
# one-line-per-kernel summary (cycles per VALU instruction per SIMD = ms x 2.4 GHz x 1024 SIMDs / SQ_INSTS_VALU).
set -u
tag=$1; group=${2:-256}
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    out=gpurun_out/${tag}_pmc_$i
    rm -rf $out; mkdir -p $out
    KZG_OPTIONS=single_stream=1 KZG_PMC_CALIBRATE=1 rocprofv3 --pmc $ctrs --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency --no-self-check --group $group --inflight 1 --steps 1 --warmup 0 > $out.log 2>&1
done
python3 tools/prof/pmc_to_json.py gpurun_out/${tag}_pmc $group > gpurun_out/${tag}_pmc.json
rm -rf gpurun_out/${tag}_pmc_1 gpurun_out/${tag}_pmc_2 gpurun_out/${tag}_pmc_3
python3 - gpurun_out/${tag}_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("calibration", d["calibration"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -(kv[1].get("ms_single_stream") or 0)):
    ms, iv = v.get("ms_single_stream") or 0, v.get("SQ_INSTS_VALU") or 0
    if ms < 0.05: continue
    print("%-52s %8.3f ms  VALU %6.2f G  cyc/instr %5.2f  wait %4.0f%%  hbm %7.3f GB" % (
        k[:52], ms, iv / 1e9, ms * 1e-3 * 2.4e9 * 1024 / iv if iv else 0,
        100 * v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0, v.get("hbm_bytes_corrected", 0) / 1e9))
PY
