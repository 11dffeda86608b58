# the linger cap of a new leader (small_linger_us; product option), verify_kzg_proof on one handle; two passes against box noise
for pass in 1 2; do
for o in 250 160 120 90 60 30; do
  echo "== small_linger_us=$o (pass $pass)"
  KZG_OPTIONS="small_linger_us=$o" python tools/prof/concurrent_callers.py --lanes 2 --threads 8,64,256 --kinds proof,blobs6 --seconds 1.5 2>&1 | grep "^lanes" | cut -c1-150
done
done
