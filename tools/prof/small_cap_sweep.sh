# A/B: tuples per coalesced launch (small_cap_proofs) at 256 and 512 threads of verify_kzg_proof on one handle
for cap in 1024 160 128 112 96 64; do
  echo "== small_cap_proofs=$cap"
  KZG_LIB_OVERRIDE=$PWD/kzg_rs_amd/libkzg_rs_amd_ab.so KZG_OPTIONS="small_cap_proofs=$cap" python tools/prof/concurrent_callers.py --lanes 2,3 --threads 256,512 --kinds proof --seconds 2 2>&1 | grep "^lanes"
done
