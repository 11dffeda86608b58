set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "g1_msm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "config4 or msm" 2>&1 | tail -3
echo "== product default"; python tools/prof/config_legs.py --no-cpu 2>&1 | tail -1
for o in "g1_msm_large_tail=0" "g1_msm_fold_per=4" "g1_msm_fold_per=6" "g1_msm_fold_per=8" "g1_msm_fold_per=11" "g1_msm_fold_per=3"; do
  echo "== $o"; KZG_LIB_OVERRIDE=$PWD/kzg_rs_amd/libkzg_rs_amd_ab.so KZG_OPTIONS="$o" python tools/prof/config_legs.py --no-cpu 2>&1 | tail -1
done
