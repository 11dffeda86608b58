"""Where the lone wavefront of verify_kzg_proof's decode kernel ran (XCC, SE, CU, SIMD) and how long it took, call by call.
    KZG_DECODE_PAIR=0 python tools/prof/decode_placement.py [calls]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kzg_rs_amd import api, synth  # noqa: E402
from kzg_rs_amd.api import Bytes32, Bytes48, KzgProof  # noqa: E402

cs, zs, ys, ps, st = synth.make_valid_proofs(4, seed=9)
L = api.lib()
L.kzg_debug_decode_placement.argtypes = [C.POINTER(C.c_ulonglong), C.c_void_p]
out = (C.c_ulonglong * 4)()
print("call  xcc se sh cu simd wave |  shader cycles   wall us")
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    assert KzgProof.verify_kzg_proof(Bytes48(cs[0]), Bytes32(zs[0]), Bytes32(ys[0]), Bytes48(ps[0]), st)
    api._chk(L.kzg_debug_decode_placement(out, st._h))
    hw = out[0]
    print("%4d  %3d %2d %2d %2d %4d %4d | %14d %9.1f" % (k, out[1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, out[2], out[3] / 100.0))
