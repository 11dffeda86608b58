"""PCIe-inclusive rate of the host-buffer entry point (kzg_verify_blob_kzg_proof_batch) next to the device-resident one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, ctypes as C, numpy as np, torch
from kzg_rs_amd import api, synth
n = 1024
blobs, cs, ps, st = synth.make_valid_batch(n, seed=5)
L = api.lib()
cb, pb = b"".join(cs), b"".join(ps)
ok = C.c_bool(False)
def run(ptr, reps=8):
    for _ in range(2):
        api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), C.cast(ptr, C.c_char_p), cb, pb, n, st._h))
    t0 = time.perf_counter()
    for _ in range(reps):
        api._chk(L.kzg_verify_blob_kzg_proof_batch(C.byref(ok), C.cast(ptr, C.c_char_p), cb, pb, n, st._h))
    dt = (time.perf_counter() - t0) / reps
    assert ok.value
    return dt
pageable = np.ascontiguousarray(blobs)
t1 = run(pageable.ctypes.data)
pinned = torch.from_numpy(blobs).pin_memory()
t2 = run(pinned.data_ptr())
d = torch.from_numpy(blobs).cuda(); torch.cuda.synchronize()
dc = torch.frombuffer(bytearray(cb), dtype=torch.uint8).cuda(); dp = torch.frombuffer(bytearray(pb), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
def run_dev(reps=8):
    for _ in range(2): api.KzgProof.verify_blob_kzg_proof_batch_device(d.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, st)
    t0 = time.perf_counter()
    for _ in range(reps): assert api.KzgProof.verify_blob_kzg_proof_batch_device(d.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, st)
    return (time.perf_counter() - t0) / reps
t3 = run_dev()
print("host pageable: %.2f ms/batch = %.0f blobs/s (%.1f GB/s of blob bytes)" % (t1 * 1e3, n / t1, n * 131072 / t1 / 1e9))
print("host pinned  : %.2f ms/batch = %.0f blobs/s (%.1f GB/s)" % (t2 * 1e3, n / t2, n * 131072 / t2 / 1e9))
print("device       : %.2f ms/batch = %.0f blobs/s" % (t3 * 1e3, n / t3))
