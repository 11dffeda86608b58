#!/bin/bash
# A/B of process-wide switches through bench.py: prints throughput and the stand-alone kernel times for each setting.
#   tools/prof/ab_env.sh "KZG_OPTIONS=msm_xcd=0" "KZG_OPTIONS=msm_xcd=1" ...
# ("-" = the defaults).  Every setting runs twice, alternating, so that clock state of the box shows up as spread.
for rep in 1 2; do
  for setting in "$@"; do
    [ "$setting" = "-" ] && setting=""
    line=$(env $setting python3 bench.py --steps 12 --warmup 4 --no-latency --no-cpu-baseline 2>/dev/null | tail -1)
    python3 - "$setting" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d.get("kernel_ms_standalone") or {}
print("%-44s %9.0f blobs/s  ms/step %.2f | alone: sha %.2f eval %.2f decode %.2f msm %.2f pairing %.2f group %.2f" % (
    sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], k.get("k_blob_challenge", 0), k.get("k_blob_evaluate", 0),
    k.get("k_g1_decode_multiples", 0), k.get("k_msm", 0), k.get("k_slp_run(pairing)", 0), k.get("whole_group", 0)))
PY
  done
done
