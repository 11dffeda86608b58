"""verify_kzg_proof back to back, after idle gaps, and beside a kernel that keeps the chip busy: how much of the call's
time is the clock state of an otherwise idle GPU.   python tools/prof/proof_clock_state.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from kzg_rs_amd import synth  # noqa: E402
from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof  # noqa: E402

cs, zs, ys, ps, st = synth.make_valid_proofs(4, seed=9)
args = (Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st)


def run(n, gap=0.0):
    ts = []
    for _ in range(n):
        if gap:
            time.sleep(gap)
        t0 = time.perf_counter()
        assert KzgProof.verify_kzg_proof(*args)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return "min %.2f  median %.2f  max %.2f ms; decode kernel %.2f ms" % (ts[0], ts[len(ts) // 2], ts[-1], st.last_timings()[6])


run(5)
print("back to back      :", run(40))
print("5 ms idle between :", run(20, 0.005))
print("50 ms idle between:", run(10, 0.05))
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(400):
        a @ a
print("beside a GEMM loop:", run(20))
torch.cuda.synchronize()
