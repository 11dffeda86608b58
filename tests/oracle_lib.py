"""ctypes binding of the CPU oracle (oracle/liboracle.so) - test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# KZG_ORACLE_LIB: another build of the same sources (the ASan one, tests/test_host_only_sanitized.py)
LIB = os.environ.get("KZG_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")

OK, BADARGS, ERROR, INVALID_LENGTH = 0, 1, 2, 4


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp, u8p, sz, ip = C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)
        L.oracle_settings_load_txt.restype = vp
        L.oracle_settings_load_txt.argtypes = [u8p, sz, C.c_int]
        L.oracle_settings_from_tau_g2.restype = vp
        L.oracle_settings_from_tau_g2.argtypes = [u8p]
        L.oracle_settings_free.argtypes = [vp]
        L.oracle_settings_root.argtypes = [vp, sz, u8p]
        L.oracle_settings_g1.argtypes = [vp, sz, u8p]
        L.oracle_settings_g2.argtypes = [vp, C.c_int, u8p]
        L.oracle_verify_kzg_proof.argtypes = [ip, u8p, u8p, u8p, u8p, vp]
        L.oracle_verify_blob_kzg_proof.argtypes = [ip, u8p, u8p, u8p, vp]
        L.oracle_verify_blob_kzg_proof_batch.argtypes = [ip, u8p, u8p, u8p, sz, vp, C.c_int, C.c_int]
        L.oracle_verify_blob_kzg_proof_batch_ex.argtypes = [ip, u8p, u8p, u8p, sz, vp, C.c_int, C.c_int,
                                                            u8p, u8p, u8p, u8p, u8p]
        L.oracle_verify_kzg_proof_batch.argtypes = [ip, u8p, u8p, u8p, u8p, sz, vp, C.c_int]
        L.oracle_compute_challenge.argtypes = [u8p, u8p, u8p]
        L.oracle_evaluate_polynomial_in_evaluation_form.argtypes = [u8p, u8p, u8p, vp]
        L.oracle_compute_r.argtypes = [u8p, u8p, u8p, u8p, u8p, sz, C.c_int]
        L.oracle_g1_decompress.argtypes = [u8p, ip, u8p]
        L.oracle_g1_msm.argtypes = [u8p, u8p, u8p, sz]
        L.oracle_g1_mul.argtypes = [u8p, u8p, u8p]
        L.oracle_g2_mul.argtypes = [u8p, u8p, u8p]
        L.oracle_g1_add.argtypes = [u8p, u8p, u8p]
        L.oracle_pairings_verify.argtypes = [ip, u8p, u8p, u8p, u8p]
        L.oracle_sha256.argtypes = [u8p, u8p, sz]
        L.oracle_fr_mul.argtypes = [u8p, u8p, u8p]
        L.oracle_fr_inv.argtypes = [u8p, u8p]
        L.oracle_constants.argtypes = [u8p, u8p, C.POINTER(C.c_uint64), u8p, u8p, C.POINTER(C.c_uint64)]
        L.oracle_bench_threads.argtypes = [C.POINTER(C.c_double), C.c_int, sz, C.c_double, u8p, u8p, u8p, u8p, u8p, sz, sz, vp]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error code %d" % code)
        self.code = code


def _chk(rc):
    if rc != OK:
        raise OracleError(rc)


class Settings:
    def __init__(self, handle):
        if not handle:
            raise OracleError(ERROR)
        self.h = handle

    @classmethod
    def mainnet(cls, load_g1=False):
        txt = open(os.path.join(ROOT, "kzg_rs_amd", "data", "trusted_setup.txt"), "rb").read()
        return cls(lib().oracle_settings_load_txt(txt, len(txt), int(load_g1)))

    @classmethod
    def from_tau_g2(cls, tau_g2):
        return cls(lib().oracle_settings_from_tau_g2(bytes(tau_g2)))

    def root(self, i):
        b = C.create_string_buffer(32)
        lib().oracle_settings_root(self.h, i, b)
        return b.raw

    def g1(self, i):
        b = C.create_string_buffer(48)
        _chk(lib().oracle_settings_g1(self.h, i, b))
        return b.raw

    def g2(self, i):
        b = C.create_string_buffer(96)
        lib().oracle_settings_g2(self.h, i, b)
        return b.raw

    def __del__(self):
        try:
            lib().oracle_settings_free(self.h)
        except Exception:
            pass


def verify_kzg_proof(c, z, y, p, s):
    ok = C.c_int(0)
    _chk(lib().oracle_verify_kzg_proof(C.byref(ok), c, z, y, p, s.h))
    return bool(ok.value)


def verify_blob_kzg_proof(blob, c, p, s):
    ok = C.c_int(0)
    _chk(lib().oracle_verify_blob_kzg_proof(C.byref(ok), blob, c, p, s.h))
    return bool(ok.value)


def verify_blob_kzg_proof_batch(blobs, cs, ps, s, nthreads=1, be=False):
    """blobs/cs/ps: lists of bytes (lengths already validated by the caller)."""
    n = len(blobs)
    ok = C.c_int(0)
    _chk(lib().oracle_verify_blob_kzg_proof_batch(C.byref(ok), b"".join(blobs), b"".join(cs), b"".join(ps), n, s.h,
                                                  nthreads, int(be)))
    return bool(ok.value)


def verify_blob_kzg_proof_batch_ex(blobs, cs, ps, s, nthreads=1, be=False):
    n = len(blobs)
    ok = C.c_int(0)
    zs, ys = C.create_string_buffer(32 * n), C.create_string_buffer(32 * n)
    r, A, B = C.create_string_buffer(32), C.create_string_buffer(48), C.create_string_buffer(48)
    _chk(lib().oracle_verify_blob_kzg_proof_batch_ex(C.byref(ok), b"".join(blobs), b"".join(cs), b"".join(ps), n, s.h,
                                                     nthreads, int(be), zs, ys, r, A, B))
    return bool(ok.value), zs.raw, ys.raw, r.raw, A.raw, B.raw


def verify_kzg_proof_batch(cs, zs, ys, ps, s, be=False):
    """src/kzg_proof.rs:399-444 on lists of bytes."""
    ok = C.c_int(0)
    _chk(lib().oracle_verify_kzg_proof_batch(C.byref(ok), b"".join(cs), b"".join(zs), b"".join(ys), b"".join(ps), len(cs), s.h,
                                             int(be)))
    return bool(ok.value)


def compute_challenge(blob, c):
    z = C.create_string_buffer(32)
    _chk(lib().oracle_compute_challenge(z, blob, c))
    return z.raw


def evaluate_polynomial_in_evaluation_form(blob, z, s):
    y = C.create_string_buffer(32)
    _chk(lib().oracle_evaluate_polynomial_in_evaluation_form(y, blob, z, s.h))
    return y.raw


def compute_r(cs, zs, ys, ps, n, be=False):
    r = C.create_string_buffer(32)
    _chk(lib().oracle_compute_r(r, cs, zs, ys, ps, n, int(be)))
    return r.raw


def g1_decompress(b):
    out = C.create_string_buffer(96)
    inf = C.c_int(0)
    _chk(lib().oracle_g1_decompress(out, C.byref(inf), b))
    return out.raw, bool(inf.value)


def g1_msm(points, scalars, n):
    out = C.create_string_buffer(48)
    _chk(lib().oracle_g1_msm(out, points, scalars, n))
    return out.raw


def g1_mul(p, k):
    out = C.create_string_buffer(48)
    _chk(lib().oracle_g1_mul(out, p, k))
    return out.raw


def g2_mul(p, k):
    out = C.create_string_buffer(96)
    _chk(lib().oracle_g2_mul(out, p, k))
    return out.raw


def g1_add(a, b):
    out = C.create_string_buffer(48)
    _chk(lib().oracle_g1_add(out, a, b))
    return out.raw


def pairings_verify(a1, a2, b1, b2):
    ok = C.c_int(0)
    _chk(lib().oracle_pairings_verify(C.byref(ok), a1, a2, b1, b2))
    return bool(ok.value)


def sha256(data):
    out = C.create_string_buffer(32)
    lib().oracle_sha256(out, data, len(data))
    return out.raw


def fr_mul(a, b):
    out = C.create_string_buffer(32)
    lib().oracle_fr_mul(out, a, b)
    return out.raw


def fr_inv(a):
    out = C.create_string_buffer(32)
    lib().oracle_fr_inv(out, a)
    return out.raw


def constants():
    a, b, c, d = (C.create_string_buffer(32), C.create_string_buffer(32), C.create_string_buffer(48),
                  C.create_string_buffer(48))
    i1, i2 = C.c_uint64(0), C.c_uint64(0)
    lib().oracle_constants(a, b, C.byref(i1), c, d, C.byref(i2))
    return a.raw, b.raw, i1.value, c.raw, d.raw, i2.value


def bench_threads(kind, threads, seconds, s, cs, ps, zs=None, ys=None, blobs=None, per_call=1):
    """oracle/bench_threads.c: T pthreads of independent single-threaded calls for `seconds` - kind "proof": verify_kzg_proof per
    tuple, "blobs": verify_blob_kzg_proof_batch of per_call blobs.  cs / ps / zs / ys / blobs: bytes, back to back (all valid).
    Returns {calls, seconds, bad, calls_per_s, per_thread_min, per_thread_max}."""
    o = (C.c_double * 6)()
    n_items = len(cs) // 48
    _chk(lib().oracle_bench_threads(o, 0 if kind == "proof" else 1, threads, float(seconds), blobs, cs, zs, ys, ps, n_items, per_call, s.h))
    return {"calls": int(o[0]), "seconds": o[1], "bad": int(o[2]), "calls_per_s": o[3], "per_thread_min": o[4], "per_thread_max": o[5]}
