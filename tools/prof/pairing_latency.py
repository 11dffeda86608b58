"""Latency of ONE pairing check (kzg_pairing_check: decode two points + the VERIFY program), per library build:
    python tools/prof/pairing_latency.py [lib.so ...]     (default: the in-tree library; KZG_OPTIONS pairing=1|2 forces a form)
Prints the program's own interval (HIP events around it) as the median of 30 calls."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import sys, statistics
sys.path.insert(0, %r)
from kzg_rs_amd import api
G1 = bytes.fromhex('97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb')
st = api.KzgSettings.load_trusted_setup_file()
ts = []
for i in range(40):
    api.pairing_check(G1, G1, st)
    ts.append(st.last_timings()[3])
print('pairing_ms median %%.4f min %%.4f' %% (statistics.median(ts[10:]), min(ts[10:])))
""" % ROOT
libs = sys.argv[1:] or [None]
for lib in libs:
    for form in ("1", "2"):
        env = dict(os.environ, KZG_OPTIONS="pairing=" + form)
        if lib:
            env["KZG_LIB_OVERRIDE"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(lib or "in-tree", "form", form, out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-500:])
