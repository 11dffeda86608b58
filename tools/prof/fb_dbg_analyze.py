import numpy as np, sys, json
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
start = (~d[:, 0]).astype(np.int64); end = d[:, 1].astype(np.int64)
live = d[:, 1] != 0
real = live & (d[:, 3] != 0)
t0 = start[live].min()
s = (start - t0) / 100.0; e = (end - t0) / 100.0   # us (100 MHz)
xcc = (d[:, 2] >> 32) & 0xF
hw = d[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
dur = e - s
print("blocks", len(d), "real", int(real.sum()), "kernel span us", e[live].max())
print("duration us: real blocks mean %.1f min %.1f max %.1f p50 %.1f p90 %.1f" % (dur[real].mean(), dur[real].min(), dur[real].max(), np.percentile(dur[real], 50), np.percentile(dur[real], 90)))
ents = d[:, 3] & 0xFFFFFFFF
print("entries per block (sum over the 4 stamping lanes = their buckets only):", ents[real][:8])
for x in range(8):
    m = real & (xcc == x)
    print("xcc", x, "blocks", int(m.sum()), "first start %.0f last end %.0f  sum dur %.0f" % (s[m].min(), e[m].max(), dur[m].sum()))
# concurrency over time
ts = np.linspace(0, e[live].max(), 25)
print("t_us : running real blocks")
for t in ts:
    print("%8.0f : %d" % (t, int(((s <= t) & (e > t) & real).sum())))
order = np.argsort(s)
print("start times of blocks by id (first 20):", s[:20].round(0))
print("start by id 760..790:", s[760:790].round(0))
print("dur by id 0..20:", dur[:20].round(0))
print("dur by id 760..790:", dur[760:790].round(0))
spec = (d[:, 3] >> 32).astype(np.int64) - 4
print("id   start  dur  xcc se sh cu  ents4  specials4")
for i in list(range(0, len(d), 37)) + [int(np.argmax(np.where(real, dur, 0)))]:
    print(i, round(s[i]), round(dur[i]), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(ents[i]), int(spec[i]))
# per CU: how many real blocks ran there
import collections
c = collections.Counter((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(len(d)) if real[i])
print("distinct CUs used:", len(c), "blocks per CU min/max:", min(c.values()), max(c.values()))
print("blocks started at t<50us:", int((real & (s < 50)).sum()))
