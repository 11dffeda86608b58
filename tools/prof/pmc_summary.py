"""Summarise a rocprofv3 --pmc run: per kernel name, the largest-grid dispatch's counter values."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter_collection.csv under", d); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
best = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    g = int(r["Grid_Size"])
    key = (k, r["Dispatch_Id"])
    best.setdefault(k, {})
    best[k].setdefault(r["Dispatch_Id"], {"grid": g})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, disp in sorted(best.items()):
    did, v = max(disp.items(), key=lambda kv: kv[1]["grid"])
    print(k[:60], " ".join("%s=%.4g" % (a, b) for a, b in v.items()))
