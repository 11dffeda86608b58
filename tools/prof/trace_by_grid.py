"""rocprofv3's --stats averages a kernel over ALL its launches of the process; bench.py launches the big kernels at several sizes
(262 144-blob launch groups, the 16 384-blob / 2^20-term config legs, the 1..256-item launches of the small-call legs).  This
splits the kernel trace of the same run by launch size so that every figure of the benchmark's line can be matched against ITS
population:    python3 tools/prof/trace_by_grid.py <run_kernel_trace.csv> [min_ms] > profiles/<tag>_kernel_trace_by_grid.json
Per kernel and grid size (threads): launches, mean / min / max duration in ms.  Kernels whose longest launch is below min_ms
(default 0.2) are left out."""
import csv
import json
import sys

path = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rows = {}
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    if not (name.startswith("kzg::") or name.startswith("k_")):
        continue
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    rows.setdefault(name, {}).setdefault(grid, []).append(ms)
out = {}
for name, by in sorted(rows.items()):
    if max(max(v) for v in by.values()) < min_ms:
        continue
    out[name] = [{"grid_threads": g, "launches": len(v), "mean_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)}
                 for g, v in sorted(by.items(), key=lambda kv: -kv[0]) if max(v) >= min_ms / 4 or len(by) <= 3][:8]
print(json.dumps({"source": path.split("/")[-2] + "/" + path.split("/")[-1], "what": "kernel durations of one profiled run, by kernel and launch size (threads)",
                  "kernels": out}, indent=1))
