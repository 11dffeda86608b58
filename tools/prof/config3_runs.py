"""BASELINE configs[2] (16 384 blobs through kzg_evaluate_polynomials_device) run many times under different regimes, to tell a
launch-quantisation tail from the state of the chip: the round-5 line showed the 6 runs of the leg getting SLOWER one after
the other (0.93, 0.98, 1.11, 1.18, 1.16, 1.14 ms), which a tail of the launch's last wave of workgroups cannot explain.
    python3 tools/prof/config3_runs.py [--blobs 16384]
Regimes: back-to-back calls | a 20 ms host sleep before every call | a large (262 144-blob-sized) torch kernel before every
call | sizes 12 288 (4.0 rounds of 3 072 resident wavefronts), 15 360 (5.0), 16 384 (5.33), 18 432 (6.0)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kzg_rs_amd import api  # noqa: E402

BYTES = 131072
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
st = api.KzgSettings.load_trusted_setup_file()
nmax = 18432
g = torch.Generator(device=dev).manual_seed(3)
d_blobs = torch.randint(0, 256, (nmax, BYTES), dtype=torch.uint8, device=dev, generator=g)
d_blobs[:, 0::32] &= 0x3F
rng = np.random.Generator(np.random.PCG64(5))
z = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)
z[:, 31] &= 0x3F
d_z = torch.from_numpy(z).to(dev)
d_y = torch.zeros(nmax * 32, dtype=torch.uint8, device=dev)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def run(n, reps, before=None):
    out = []
    for _ in range(reps):
        if before:
            before()
        api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, st)
        out.append(round(st.last_timings()[4], 4))
    return out


def busy():
    big.add_(1)
    torch.cuda.synchronize()


res = {}
res["back_to_back_16384"] = run(16384, 24)
res["sleep_20ms_before_16384"] = run(16384, 12, lambda: time.sleep(0.02))
res["busy_kernel_before_16384"] = run(16384, 12, busy)
res["back_to_back_16384_again"] = run(16384, 12)
for n in (12288, 15360, 16384, 18432):
    r = run(n, 10)
    res["size_%d" % n] = {"runs": r, "median_ms": sorted(r[2:])[len(r[2:]) // 2], "us_per_blob_median": round(sorted(r[2:])[len(r[2:]) // 2] * 1e3 / n, 5)}
print(json.dumps(res))
