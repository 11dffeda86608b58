"""kzg_g1_msm_setup at BASELINE configs[3]'s size (2^20 terms over the handle's 4 096 Lagrange points) and around it: the MSM's
own time (HIP events: timings[2]), the call's wall time from pageable host scalars, and the same sum through kzg_g1_msm over the
tiled compressed points (decode + tables per call) - checked against each other.
    python3 tools/prof/msm_setup_probe.py [--sizes 65536,262144,1048576]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kzg_rs_amd import api  # noqa: E402

torch.cuda.set_device(0)
sizes = [65536, 262144, 1048576]
for a in sys.argv[1:]:
    if a.startswith("--sizes="):
        sizes = [int(x) for x in a.split("=")[1].split(",")]
st = api.KzgSettings.load_trusted_setup_file()
L = api.lib()
ts = open(os.path.join(ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
brp = lambda i: int(format(i, "012b")[::-1], 2)
base = b"".join(bytes.fromhex(ts[2 + brp(i)]) for i in range(4096))
out = {"options": os.environ.get("KZG_OPTIONS", ""), "sizes": {}}
o48 = C.create_string_buffer(48)
for n in sizes:
    sc = np.random.Generator(np.random.PCG64(n)).integers(0, 256, size=(n, 32), dtype=np.uint8)
    ms, wall = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        api._chk(L.kzg_g1_msm_setup(o48, sc.ctypes.data_as(C.c_char_p), n, st._h))
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(st.last_timings()[2])
    got = o48.raw
    row = {"ms_msm_runs": [round(x, 4) for x in ms], "ms_wall_runs": [round(x, 3) for x in wall], "ms_msm": round(sorted(ms[1:])[2], 4),
           "ms_call_wall": round(sorted(wall[1:])[2], 3), "ns_per_term": round(sorted(ms[1:])[2] * 1e6 / n, 3)}
    if n <= (1 << 20) and "--no-tiled" not in sys.argv:
        pts = base * (n // 4096)
        t0 = time.perf_counter()
        api._chk(L.kzg_g1_msm(o48, pts, sc.ctypes.data_as(C.c_char_p), n, st._h))
        row["tiled_kzg_g1_msm"] = {"ms_call_wall_first": round((time.perf_counter() - t0) * 1e3, 3), "ms_msm": round(st.last_timings()[2], 4),
                                   "ms_decode_and_tables": round(st.last_timings()[6], 4), "same_sum": o48.raw == got}
    out["sizes"][str(n)] = row
print(json.dumps(out))
