"""The cycle of the small-call queue's launches under load, from a kernel trace:
    rocprofv3 --kernel-trace -d /tmp/cyc -o run --output-format csv -- python3 tools/prof/concurrent_callers.py --lanes 2 --threads 256 --kinds proof --seconds 2
    python3 tools/prof/small_call_cycle.py /tmp/cyc/run_kernel_trace.csv
Kernels are grouped per HIP stream into launches (a gap of more than 150 us starts a new one); prints, for the launches that
have a kernel of at least MIN_WG (second argument, default 32) workgroups: GPU span of a launch (first kernel start .. last kernel end), the gap to the
next launch on the same queue (completion -> results -> the callers' next calls -> linger -> gather -> submit), per-kernel durations."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
MIN_WG = int(sys.argv[2]) if len(sys.argv) > 2 else 32   # a launch counts when one of its kernels has at least this many workgroups
byq = defaultdict(list)
for r in rows:
    wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    byq[r.get("Stream_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "") + " lds=" + r["LDS_Block_Size"], wgs))
spans, gaps, kd = [], [], defaultdict(list)
for q, ks in byq.items():
    ks.sort()
    launches, cur = [], [ks[0]]
    for k in ks[1:]:
        if k[0] - cur[-1][1] > 150_000:
            launches.append(cur)
            cur = [k]
        else:
            cur.append(k)
    launches.append(cur)
    big = [l for l in launches if any(k[3] >= MIN_WG for k in l)]
    for a, b in zip(big, big[1:]):
        gaps.append((b[0][0] - a[-1][1]) / 1e3)
    for l in big:
        spans.append((l[-1][1] - l[0][0]) / 1e3)
        for k in l:
            kd[k[2][:60]].append((k[1] - k[0]) / 1e3)
    print("stream %s: %d launches, %d large" % (q, len(launches), len(big)))
def stat(v):
    v = sorted(v)
    return "n %5d  median %8.1f  p10 %8.1f  p90 %8.1f us" % (len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]) if v else "-"
print("GPU span of a large launch :", stat(spans))
print("gap to the next on its queue:", stat(gaps))
for k, v in sorted(kd.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s %s" % (k, stat(v)))
