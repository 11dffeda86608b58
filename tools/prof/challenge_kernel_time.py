"""Time of the challenge kernel alone (kzg_last_timings[5]) for one batch of n device-resident blobs, verdict NOT checked:
for A/B experiments on a deliberately broken kernel (producer-only / consumer-only builds).
  python tools/prof/challenge_kernel_time.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from kzg_rs_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tau, tau_g2 = synth.synthetic_setup()
st = api.KzgSettings.from_tau_g2(tau_g2)
d_b = torch.from_numpy(synth.random_blobs(n, 5)).cuda()
g = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
d_c = torch.frombuffer(bytearray(g * n), dtype=torch.uint8).cuda()
d_p = d_c.clone()
torch.cuda.synchronize()
ts = []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    try:
        api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st)
    except api.KzgError:
        pass
    ts.append(st.last_timings()[5])
print("n=%d challenge kernel ms: %s" % (n, " ".join("%.3f" % t for t in ts)))
