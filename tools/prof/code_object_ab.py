"""Product library (default kernel forms only) against the A/B build (the same + the alternative forms): what stripping the
variants from the shipped code object changes.  Each library in a child process:
  - size of the .so and of its gfx950 code object
  - verify_kzg_proof (one proof at a time): 96 calls, min / median / max / spread (docs/lab_notebook.md 9: a lone wavefront's time
    depends on where the dispatcher put it; both paths: default and KZG_OPTIONS=proof_path=msm, round 3's)
  - one verify_blob_kzg_proof_batch of 1 024 device-resident blobs: 32 calls, min / median / max
    python3 tools/prof/code_object_ab.py            (through gpurun, from the repo root)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, time, json
sys.path.insert(0, %r)
import torch
from kzg_rs_amd import api, synth
from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof
cs, zs, ys, ps, st = synth.make_valid_proofs(4, seed=9)
args = (Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st)
for _ in range(8): assert KzgProof.verify_kzg_proof(*args)
ts = []
for _ in range(96):
    t0 = time.perf_counter(); assert KzgProof.verify_kzg_proof(*args); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
n = 1024
blobs, c, p, st2 = synth.make_valid_batch(n, seed=3, chunk=1024)
d_b = torch.from_numpy(blobs).cuda(); d_c = torch.frombuffer(bytearray(b"".join(c)), dtype=torch.uint8).cuda(); d_p = torch.frombuffer(bytearray(b"".join(p)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(4): assert KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st2)
tb = []
for _ in range(32):
    t0 = time.perf_counter(); assert KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st2); tb.append((time.perf_counter() - t0) * 1e3)
tb.sort()
print(json.dumps({"verify_kzg_proof_ms": {"min": round(ts[0], 3), "median": round(ts[48], 3), "max": round(ts[-1], 3), "spread": round(ts[-1] - ts[0], 3)},
                  "one_batch_1024_ms": {"min": round(tb[0], 3), "median": round(tb[16], 3), "max": round(tb[-1], 3)}}))
""" % ROOT


def code_object_bytes(lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", "-W", lib], capture_output=True, text=True).stdout
    for ln in out.splitlines():
        f = ln.split()
        if ".hip_fatbin" in ln:
            i = f.index(".hip_fatbin")
            return int(f[i + 4], 16)
    return None


res = {}
for name, lib, opts in (("product", os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd.so"), ""), ("ab_build", os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd_ab.so"), ""),
                        ("product_msm_proof_path", os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd.so"), "proof_path=msm"),
                        ("ab_build_msm_proof_path", os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd_ab.so"), "proof_path=msm")):
    env = dict(os.environ, KZG_LIB_OVERRIDE=lib, GPU_MAX_HW_QUEUES="8", KZG_OPTIONS=opts)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    res[name] = {"so_bytes": os.path.getsize(lib), "code_object_bytes": code_object_bytes(lib)}
    res[name].update(json.loads(line[-1]) if line else {"error": r.stderr[-400:]})
print(json.dumps(res, indent=1))
