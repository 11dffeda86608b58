#!/bin/bash
# which HIP runtime / host-memory mode makes the pure-C caller differ from the Python-hosted tests?
set -u
cd "$(dirname "$0")/../.."
python3 - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import test_cabi_c_program as T
print(T._vector_file("/tmp/vectors.bin"))
PY
gcc -std=c11 -O1 -I include -o /tmp/cabi_vectors tests/host/cabi_vectors.c -L kzg_rs_amd -lkzg_rs_amd -Wl,-rpath,$PWD/kzg_rs_amd
TORCHLIB=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
echo "== plain (system libamdhip64)"; /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -4
ldd /tmp/cabi_vectors | grep -i hip
echo "== HIP_HOST_COHERENT=1"; HIP_HOST_COHERENT=1 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== torch's runtime"; LD_LIBRARY_PATH=$TORCHLIB /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== KZG_OPTIONS=cu_mask=0"; KZG_OPTIONS=cu_mask=0 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== KZG_OPTIONS=single_stream=1"; KZG_OPTIONS=single_stream=1 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== KZG_OPTIONS=pairing=1"; KZG_OPTIONS=pairing=1 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== KZG_OPTIONS=decode_quads=0"; KZG_OPTIONS=decode_quads=0 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
echo "== KZG_OPTIONS=msm_latency_layout=0"; KZG_OPTIONS=msm_latency_layout=0 /tmp/cabi_vectors kzg_rs_amd/data/trusted_setup.txt /tmp/vectors.bin 2>&1 | tail -3
