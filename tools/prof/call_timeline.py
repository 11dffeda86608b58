"""Kernel timeline of the LAST call of a short script under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace -d DIR -o run --output-format csv -- python3 tools/prof/call_timeline.py run proof|batch
    python3 tools/prof/call_timeline.py show DIR
`run` makes 4 calls of verify_kzg_proof (proof) or of a 1 024-blob verify_blob_kzg_proof_batch_device (batch) separated by
5 ms sleeps; `show` prints every kernel after the last sleep-sized gap (start, duration, gap to the previous end)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    import torch
    from kzg_rs_amd import api, synth
    from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof
    if sys.argv[2].startswith("proofs"):  # proofsN: verify_kzg_proof_batch over N tuples
        n = int(sys.argv[2][6:])
        cs, zs, ys, ps, st = synth.make_valid_proofs(n, seed=9)
        for _ in range(4):
            time.sleep(0.005)
            assert KzgProof.verify_kzg_proof_batch([Bytes48(c) for c in cs], [Bytes32(z) for z in zs], [Bytes32(y) for y in ys], [Bytes48(p) for p in ps], st)
    elif sys.argv[2] == "proof":
        cs, zs, ys, ps, st = synth.make_valid_proofs(4, seed=9)
        for _ in range(4):
            time.sleep(0.005)
            assert KzgProof.verify_kzg_proof(Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st)
    elif sys.argv[2] == "hostbatch":  # the reference's call shape: a host Vec<Blob> (pageable), copies included
        import ctypes as C
        n = 1024
        blobs, cs, ps, st = synth.make_valid_batch(n, seed=3, chunk=1024)
        hc, hp, ok = b"".join(cs), b"".join(ps), C.c_bool(False)
        for _ in range(4):
            time.sleep(0.005)
            api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(ok), blobs.ctypes.data_as(C.c_char_p), hc, hp, n, st._h))
            assert ok.value
    else:
        n = 1024
        blobs, cs, ps, st = synth.make_valid_batch(n, seed=3, chunk=1024)
        d_b = torch.from_numpy(blobs).cuda()
        d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda()
        d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        for _ in range(4):
            time.sleep(0.005)
            assert KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st)
else:
    import csv
    import glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for m in glob.glob(sys.argv[2] + "/**/*memory_copy_trace.csv", recursive=True):  # (--memory-copy-trace) copies join the timeline
        for r in csv.DictReader(open(m)):
            r["Kernel_Name"] = "[copy %s %s B]" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))
            rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    start = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 3e6:
            start = i
    t0 = int(rows[start]["Start_Timestamp"])
    prev_end = t0
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.3f ms  +%8.3f ms  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
        prev_end = max(prev_end, e)
    print("total %.3f ms" % ((prev_end - t0) / 1e6))
