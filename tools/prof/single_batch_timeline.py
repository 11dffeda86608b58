"""Print the kernel timeline of the last single-batch step in a rocprofv3 --kernel-trace CSV (start, duration, gap)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the split challenge kernel marks the start of the last single-batch step
idx = max(i for i, r in enumerate(rows) if "k_blob_challenge_split" in r["Kernel_Name"])
# go back to include kernels of the same step launched just before (decode on the other stream)
t0 = int(rows[idx]["Start_Timestamp"])
start = idx
while start > 0 and t0 - int(rows[start - 1]["Start_Timestamp"]) < 200000:
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - t0 > 20e6:
        break
    print("%9.3f ms  +%8.3f ms  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, r["Kernel_Name"].split("(")[0][:50]))
    prev_end = max(prev_end, e)
print("total %.3f ms" % ((prev_end - t0) / 1e6))
