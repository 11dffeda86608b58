#!/bin/bash
# Collect the profile set of a round on the GPU box (run through gpurun from the repo root):
#   tools/prof/collect_round.sh <tag> [group]   (round 6: r6)
# 1. rocprofv3 --kernel-trace --stats of THE DRIVER'S COMMAND (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#                                                                          -> gpurun_out/<tag>_stats/
# 2. PMC passes (each in its own run; single stream so dispatches do not overlap; ONE launch group of the bench's size):
#      FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
#    -> gpurun_out/<tag>_pmc_*/ ; tools/prof/pmc_to_json.py turns them into gpurun_out/<tag>_pmc.json
# 3. the same command once more without the profiler                         -> gpurun_out/<tag>_bench.json
set -u
tag=$1; group=${2:-256}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${tag}_stats; mkdir -p gpurun_out/${tag}_stats
# the JSON line of THIS invocation is the one that is committed beside the stats (profiles/<tag>_bench.json): the kernel's
# duration in the line and its AverageNs in the stats then describe the same run
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o run --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_stats.log 2>&1
grep '^{' gpurun_out/${tag}_stats.log | tail -1 > gpurun_out/${tag}_bench_profiled.json
# the same trace split by launch size: --stats averages a kernel over launches of every size the process makes
python3 tools/prof/trace_by_grid.py gpurun_out/${tag}_stats/run_kernel_trace.csv > gpurun_out/${tag}_kernel_trace_by_grid.json
i=0
# SKIP_PMC=1: only the stats pass and the unprofiled line (the PMC json under profiles/ already describes this code)
[ "${SKIP_PMC:-0}" = "1" ] || for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    out=gpurun_out/${tag}_pmc_$i
    rm -rf $out; mkdir -p $out
    KZG_OPTIONS=single_stream=1 KZG_PMC_CALIBRATE=1 rocprofv3 --pmc $ctrs --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency --no-self-check --group $group --inflight 1 --steps 1 --warmup 0 > $out.log 2>&1
done
[ "${SKIP_PMC:-0}" = "1" ] || python3 tools/prof/pmc_to_json.py gpurun_out/${tag}_pmc $group > gpurun_out/${tag}_pmc.json
# 4. BASELINE configs[2] / configs[3] alone (tools/prof/config_legs.py = bench.config_legs): kernel stats, then the same PMC passes
#    -> gpurun_out/<tag>_config_stats/, gpurun_out/<tag>_config_pmc.json (its "kernels" hold the 2^20-term launch of k_msm_window and
#    the 16 384-blob launch of k_blob_evaluate: their largest dispatches in that process)
rm -rf gpurun_out/${tag}_config_stats; mkdir -p gpurun_out/${tag}_config_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_config_stats -o run --output-format csv -- python3 tools/prof/config_legs.py --no-cpu > gpurun_out/${tag}_config_stats.log 2>&1
grep '^{' gpurun_out/${tag}_config_stats.log | tail -1 > gpurun_out/${tag}_config_legs_profiled.json
i=0
[ "${SKIP_PMC:-0}" = "1" ] || for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    out=gpurun_out/${tag}_config_pmc_$i
    rm -rf $out; mkdir -p $out
    KZG_OPTIONS="single_stream=1" rocprofv3 --pmc $ctrs --kernel-trace -d $out -o run --output-format csv -- python3 tools/prof/config_legs.py --no-cpu > $out.log 2>&1
done
[ "${SKIP_PMC:-0}" = "1" ] || python3 tools/prof/pmc_to_json.py gpurun_out/${tag}_config_pmc 0 > gpurun_out/${tag}_config_pmc.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
# 5. the kernel resource table of the library these profiles describe (code-object metadata: VGPRs, LDS, scratch, waves per SIMD)
python3 tools/prof/kernel_resources.py > gpurun_out/${tag}_kernel_resources.txt 2>/dev/null
