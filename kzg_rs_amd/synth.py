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
    All work is done by the GPU library; only the scalar division uses Python integers."""
    tau, tau_g2 = synthetic_setup()
    if settings is None:
        settings = api.KzgSettings.from_tau_g2(tau_g2)
    blobs = random_blobs(n, seed)
    tau_be = tau.to_bytes(32, "big")
    commitments, proofs = [], []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        bl = [blobs[i].tobytes() for i in range(lo, hi)]
        p_tau = api.evaluate_polynomials(bl, [tau_be] * len(bl), settings)
        cs = api.g1_mul_generator(p_tau, settings)
        zs = api.compute_challenges(bl, cs, settings)
        ys = api.evaluate_polynomials(bl, zs, settings)
        qs = []
        for pt, z, y in zip(p_tau, zs, ys):
            q = (int.from_bytes(pt, "big") - int.from_bytes(y, "big")) * pow(tau - int.from_bytes(z, "big"), -1, R) % R
            qs.append(q.to_bytes(32, "big"))
        commitments += cs
        proofs += api.g1_mul_generator(qs, settings)
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
