"""dev probe: SLP VERIFY program time vs number of concurrent instances + in-kernel shader clock."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kzg_rs_amd import api, synth
st = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1])
L = api.lib()
L.kzg_debug_slp_bench.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]
for inst in (1, 2, 16, 64, 256, 1024, 2048):
    ms, mhz = C.c_float(), C.c_float()
    api._chk(L.kzg_debug_slp_bench(C.byref(ms), C.byref(mhz), inst, 3, st._h))
    print("instances=%5d  %.3f ms per launch  (%.3f us/step)  lone-wave clock %.0f MHz" % (inst, ms.value, ms.value * 1e3 / 5755, mhz.value))
