# A/B: the large-sum tail of kzg_g1_msm (msm.hpp msm_large_tail) - layers one thread folds in a row (g1_msm_fold_per) - with the kernels' own times
export TMPDIR=/tmp
for o in "g1_msm_fold_per=6" "g1_msm_fold_per=11" "g1_msm_fold_per=16" "g1_msm_fold_per=22" "g1_msm_fold_per=43"; do
  echo "== $o"
  rm -rf /tmp/c4s
  KZG_LIB_OVERRIDE=$PWD/kzg_rs_amd/libkzg_rs_amd_ab.so KZG_OPTIONS="$o" rocprofv3 --kernel-trace --stats -d /tmp/c4s -o run --output-format csv -- python3 tools/prof/config_legs.py --no-cpu 2>/dev/null | grep -o '"ms_msm": [0-9.]*, "ms_decode_and_tables": [0-9.]*, "ms_msm_all_runs": \[[^]]*\]'
  python3 - <<'P'
import csv,glob
f=glob.glob("/tmp/c4s/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("bucket_fold<false","bucket_sum","reduce_quads","combine_quad")): print("   %-50s %s calls  avg %.1f us" % (n.split("(")[0][-50:], r["Calls"], float(r["AverageNs"])/1e3))
P
done
