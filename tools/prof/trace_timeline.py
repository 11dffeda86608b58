"""Timeline summary of a rocprofv3 --kernel-trace CSV: per-kernel totals, union busy time, overlap, idle gaps."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
ev.sort()
# steady-state window: middle 60% of the k_blob_evaluate launches
evs = [e for e in ev if "k_blob_evaluate" in e[2]]
lo, hi = evs[len(evs) // 5][0], evs[len(evs) * 4 // 5][0]
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
tot = {}
for s, e, k in win:
    tot.setdefault(k, [0, 0])
    tot[k][0] += e - s
    tot[k][1] += 1
wall = hi - lo
# union busy
busy = 0; cur_s = cur_e = None
for s, e, k in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
ngroups = tot.get([k for k in tot if "k_blob_evaluate" in k][0])[1]
print("window %.2f ms, %d launch groups, %.3f ms/group, GPU busy (union) %.1f%%" % (wall / 1e6, ngroups, wall / 1e6 / ngroups, 100 * busy / wall))
print("%-46s %8s %10s %8s" % ("kernel", "calls", "ms/group", "% wall"))
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-46s %8d %10.3f %8.1f" % (k, c, t / 1e6 / ngroups, 100 * t / wall))
print("sum of kernel time / wall = %.2f" % (sum(t for t, _ in tot.values()) / wall))
