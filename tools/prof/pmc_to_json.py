"""Merge the PMC passes of tools/prof/collect_round.sh into one JSON: per kernel, the counters of its largest dispatch
(one launch group), HBM bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950: 128-byte requests
are tallied at 64 bytes; WRITE_SIZE exact), with the correction checked in the same run on a known-size dispatch:
bench.py under KZG_PMC_CALIBRATE=1 runs ONE elementwise kernel that reads 128 MiB and writes 128 MiB (the only
vectorized_elementwise_kernel of that grid in the process; a hipMemcpy would be ambiguous - device-to-host copies of the
records use the same blit kernel and grid).  Expected: FETCH_SIZE ~ 65 536 KB (x2 = 131 072), WRITE_SIZE = 131 072 KB."""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import build as kbuild  # noqa: E402

# the kernels these counters describe: bench.py (kzg_rs_amd/benchline.py) uses a profile only when its stamp equals the running tree's
KERNEL_KEY = kbuild.kernel_key()
prefix, group = sys.argv[1], int(sys.argv[2])
kern = {}
calib = {}
for d in sorted(glob.glob(prefix + "_*")):
    if not d[-1].isdigit():
        continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        continue
    dur = {}
    for r in csv.DictReader(open(t[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per = {}
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        e = per.setdefault(name, {}).setdefault(r["Dispatch_Id"], {"grid": int(r["Grid_Size"]), "ms": dur.get(r["Dispatch_Id"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for name, disp in per.items():
        did, v = max(disp.items(), key=lambda kv: (kv[1]["grid"], kv[1]["ms"] or 0))
        if "vectorized_elementwise_kernel" in name:
            # calibration: the 128 MiB -> 128 MiB elementwise add is the largest dispatch of this kernel
            for k, x in v.items():
                if k in ("FETCH_SIZE", "WRITE_SIZE"):
                    calib[k + "_KB_for_128MiB_read_128MiB_write"] = x
            continue
        if not name.startswith("kzg::") and not name.startswith("k_"):
            continue
        o = kern.setdefault(name, {"grid_threads": v["grid"]})
        for k, x in v.items():
            if k == "grid":
                continue
            if k == "ms":
                o.setdefault("ms_single_stream", []).append(round(x, 4))
            else:
                o[k] = x
for name, o in kern.items():
    if "ms_single_stream" in o:
        o["ms_single_stream"] = round(sum(o["ms_single_stream"]) / len(o["ms_single_stream"]), 4)
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes_corrected"] = round(2 * o["FETCH_SIZE"] * 1024 + o["WRITE_SIZE"] * 1024)
if group == 0:  # the config legs (tools/prof/config_legs.py): no launch group, the kernels' largest dispatches of that process
    print(json.dumps({
        "method": "rocprofv3 --pmc, one counter set per run, KZG_OPTIONS=single_stream=1, python3 tools/prof/config_legs.py: values are "
                  "per launch of the kernel (its largest dispatch in the process: k_blob_evaluate over 16 384 blobs = BASELINE configs[2]; k_msm_window, "
                  "k_fb_window and the partition, fold and reduction kernels of kzg_g1_msm_setup, k_msm_window, k_g1_decode_multiples29, k_mult_to_affine29 of kzg_g1_msm over 2^20 terms = BASELINE configs[3]). FETCH_SIZE / WRITE_SIZE in KB as "
                  "reported; hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section).",
        "kernel_key": KERNEL_KEY, "config3_blobs": 16384, "config4_pairs": 1 << 20, "kernels": kern}, indent=1))
    sys.exit(0)
print(json.dumps({
    "method": "rocprofv3 --pmc, one counter set per run, KZG_OPTIONS=single_stream=1, bench.py --group %d --inflight 1 --steps 1 --warmup 0: one "
              "launch group of %d batches x 1024 blobs; values are per launch of the kernel (its largest dispatch). FETCH_SIZE / "
              "WRITE_SIZE in KB as reported; hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, "
              "MI355X_MICROARCH.md HBM section); calibration = the same counters on a dispatch that reads 128 MiB and writes 128 MiB." % (group, group),
    "kernel_key": KERNEL_KEY, "blobs_per_launch": 1024 * group, "calibration": calib, "kernels": kern}, indent=1))
