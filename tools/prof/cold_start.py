import os, sys, time
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
from kzg_rs_amd import api
t1 = time.perf_counter()
st = api.KzgSettings.load_trusted_setup_file()
t2 = time.perf_counter()
st2 = api.KzgSettings.load_trusted_setup_file()
t3 = time.perf_counter()
from kzg_rs_amd import synth
tau, tau_g2 = synth.synthetic_setup()
t4 = time.perf_counter()
st3 = api.KzgSettings.from_tau_g2(tau_g2)
t5 = time.perf_counter()
print("import %.3f s; first load_trusted_setup_file %.3f s (HIP runtime + code object load); second handle %.3f s; from_tau_g2 handle %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t5 - t4))
