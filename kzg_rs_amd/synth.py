"""Synthetic workloads for benchmarks and full-size tests.

Valid (blob, commitment, proof) triples are produced under a TEST-ONLY trusted setup with a
KNOWN tau (kzg_rs_amd/data/synthetic_setup.json), using the GPU library itself plus Python
integers:      C = p(tau) G1,     pi = ((p(tau) - y) / (tau - z)) G1,   z = challenge(blob, C), y = p(z).
The reference supports such custom settings through EnvKzgSettings::Custom
(src/trusted_setup.rs:52-57).  Blob elements are uniform in [0, 2^254) (always canonical).
"""
import json
import os

import numpy as np

from . import api

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_setup():
    d = json.load(open(os.path.join(_HERE, "data", "synthetic_setup.json")))
    return int(d["tau"], 16), bytes.fromhex(d["tau_g2"])


def random_blobs(n, seed):
    """n blobs as one contiguous uint8 array (n, 131072); every field element < 2^254 < r."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 256, size=(n, api.BYTES_PER_BLOB), dtype=np.uint8)
    a[:, 0::32] &= 0x3F
    return a


def make_valid_batch(n, seed, settings=None, chunk=256):
    """Returns (blobs uint8[n,131072], commitments list[bytes48], proofs list[bytes48], settings).
    All curve and field work is done by the GPU library (through the C ABI, on the numpy buffer itself - no per-blob
    byte strings, so 32 768 blobs = 4 GiB take seconds); only the scalar division uses Python integers."""
    import ctypes as C

    tau, tau_g2 = synthetic_setup()
    if settings is None:
        settings = api.KzgSettings.from_tau_g2(tau_g2)
    blobs = random_blobs(n, seed)
    L, h = api.lib(), settings._h
    tau_be = tau.to_bytes(32, "big")
    commitments, proofs = [], []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        bl = blobs[lo:hi].ctypes.data_as(C.c_char_p)  # rows are contiguous
        p_tau, zs, ys = (C.create_string_buffer(32 * m) for _ in range(3))
        cbuf, qbuf = C.create_string_buffer(48 * m), C.create_string_buffer(48 * m)
        api._chk(L.kzg_evaluate_polynomials(p_tau, bl, tau_be * m, m, h))
        api._chk(L.kzg_g1_mul_generator(cbuf, p_tau, m, h))
        api._chk(L.kzg_compute_challenges(zs, bl, cbuf, m, h))
        api._chk(L.kzg_evaluate_polynomials(ys, bl, zs, m, h))
        qs = bytearray(32 * m)
        pt_b, zs_b, ys_b = p_tau.raw, zs.raw, ys.raw
        for i in range(m):
            sl = slice(32 * i, 32 * i + 32)
            q = (int.from_bytes(pt_b[sl], "big") - int.from_bytes(ys_b[sl], "big")) * pow(tau - int.from_bytes(zs_b[sl], "big"), -1, R) % R
            qs[sl] = q.to_bytes(32, "big")
        api._chk(L.kzg_g1_mul_generator(qbuf, bytes(qs), m, h))
        c_b, q_b = cbuf.raw, qbuf.raw
        commitments += [c_b[48 * i: 48 * i + 48] for i in range(m)]
        proofs += [q_b[48 * i: 48 * i + 48] for i in range(m)]
    return blobs, commitments, proofs, settings


def make_valid_proofs(n, seed, settings=None):
    """n valid (commitment, z, y, proof) tuples for verify_kzg_proof_batch under the synthetic setup: any scalars
    a, z, y with C = [a]G and pi = [(a - y) / (tau - z)]G satisfy the KZG equation.  Returns lists of big-endian
    bytes (48, 32, 32, 48) and the settings."""
    tau, tau_g2 = synthetic_setup()
    if settings is None:
        settings = api.KzgSettings.from_tau_g2(tau_g2)
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.integers(0, 256, size=(3, n, 32), dtype=np.uint8)
    raw[:, :, 0] &= 0x3F
    a, z, y = ([raw[k, i].tobytes() for i in range(n)] for k in range(3))
    qs = []
    for ai, zi, yi in zip(a, z, y):
        q = (int.from_bytes(ai, "big") - int.from_bytes(yi, "big")) * pow(tau - int.from_bytes(zi, "big"), -1, R) % R
        qs.append(q.to_bytes(32, "big"))
    return api.g1_mul_generator(a, settings), z, y, api.g1_mul_generator(qs, settings), settings
