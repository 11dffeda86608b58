"""Challenge kernel forms by launch size: time of the challenge kernel (kzg_last_timings[5]) for one batch of n device-resident
blobs, each form forced in a child process (KZG_OPTIONS challenge_kernel=...).   python tools/prof/challenge_forms_rate.py [n ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import sys
sys.path.insert(0, %r)
import torch
from kzg_rs_amd import api, synth
n = int(sys.argv[1])
blobs, cs, ps, st = synth.make_valid_batch(min(n, 1024), seed=3, chunk=1024)
reps = (n + 1023) // 1024
d_b = torch.from_numpy(blobs).cuda().repeat(reps, 1)[:n].contiguous()
d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).cuda().view(-1, 48).repeat(reps, 1)[:n].contiguous()
d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).cuda().view(-1, 48).repeat(reps, 1)[:n].contiguous()
torch.cuda.synchronize()
ts = []
for _ in range(4):
    assert api.KzgProof.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st)
    ts.append(st.last_timings())
t = ts[-1]
print("challenge %%.3f ms  whole call %%.3f ms" %% (t[5], t[0]))
""" % ROOT
for n in [int(x) for x in sys.argv[1:]] or [1024, 8192, 16384, 24576, 32768, 49152]:
    for form in ("lane", "split", "split2"):
        out = subprocess.run([sys.executable, "-c", CODE, str(n)], env=dict(os.environ, KZG_OPTIONS="challenge_kernel=" + form), capture_output=True, text=True)
        print("n=%6d %-6s %s" % (n, form, out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-300:]))
