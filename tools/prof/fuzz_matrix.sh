#!/bin/bash
# The differential fuzz (tools/fuzz_campaign.py: 4 entry points against the CPU oracle) through every configuration that
# changes which code answers a small call:   tools/prof/fuzz_matrix.sh <seconds per configuration> <seed base>
secs=${1:-240}; seed=${2:-5000}
AB=kzg_rs_amd/libkzg_rs_amd_ab.so
i=0
run() {  # name, environment assignments...
    i=$((i+1)); name=$1; shift
    printf "%-64s " "$name"
    env "$@" timeout $((secs+180)) python3 tools/fuzz_campaign.py $secs $((seed+i)) ${THREADS:-1} 2>&1 | grep "fuzz campaign\|MISMATCH\|Error\|error" | tail -1
}
run "default (per-tuple pairings, host hashing)"              KZG_OPTIONS=
run "small_batch_pairings_max=0 (combined form, host hashing)" "KZG_OPTIONS=small_batch_pairings_max=0"
run "host_challenge_max_blobs=0 (combined form, GPU hashing)"  "KZG_OPTIONS=host_challenge_max_blobs=0"
run "proof_path=msm (round 3's one-proof path)"               "KZG_OPTIONS=proof_path=msm;small_batch_pairings_max=0"
run "A/B build, fp29=0 (12x32-limb point kernels)"            KZG_LIB_OVERRIDE=$AB "KZG_OPTIONS=fp29=0;small_batch_pairings_max=0"
run "A/B build, evaluate_kernel=32, challenge_occ=3"          KZG_LIB_OVERRIDE=$AB "KZG_OPTIONS=evaluate_kernel=32;challenge_occ=3;small_batch_pairings_max=0"
run "device list [0,0,0] (multi_force)"                       KZG_DEVICES=0,0,0 "KZG_OPTIONS=multi_force=1;multi_min_blobs=2"
# the shared handle under concurrent callers: the same campaign from 16 threads at once (their small calls coalesce into shared launches)
THREADS=16 run "default, 16 threads on ONE handle (coalesced launches)"      KZG_OPTIONS=
THREADS=16 run "coalesce=0, 16 threads on ONE handle (the handle's mutex)"   "KZG_OPTIONS=coalesce=0"
THREADS=8 run "device list [0,0,0], 8 threads on ONE handle"                KZG_DEVICES=0,0,0 "KZG_OPTIONS=multi_force=1;multi_min_blobs=2"
