"""Loader for the committed golden fixtures (tests/golden/, built by make_fixtures.py)."""
import json
import lzma
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_cache = {}


def vectors():
    if "v" not in _cache:
        _cache["v"] = json.load(open(os.path.join(GOLDEN, "vectors.json")))
    return _cache["v"]


def kat():
    if "k" not in _cache:
        _cache["k"] = json.load(open(os.path.join(GOLDEN, "kat.json")))
    return _cache["k"]


def blob(blob_id):
    if "b" not in _cache:
        raw = lzma.decompress(open(os.path.join(GOLDEN, "blobs.bin.xz"), "rb").read())
        _cache["b"] = {e["id"]: raw[e["offset"]: e["offset"] + e["length"]] for e in vectors()["blob_index"]}
    return _cache["b"][blob_id]


def case(kind, name_suffix):
    for c in vectors()[kind]:
        if c["name"].endswith(name_suffix):
            return c
    raise KeyError(name_suffix)


def valid_blob_tuples():
    """The distinct valid (blob, commitment, proof) mainnet tuples (expected output true)."""
    out, seen = [], set()
    for c in vectors()["verify_blob_kzg_proof"]:
        if c["output"] is True and c["blob"] not in seen:
            seen.add(c["blob"])
            out.append((blob(c["blob"]), bytes.fromhex(c["commitment"]), bytes.fromhex(c["proof"])))
    return out


def off_subgroup_g1():
    """48 compressed bytes of a point ON the curve y^2 = x^3 + 4 but (with overwhelming probability - callers confirm with
    the oracle) OUTSIDE the r-torsion: G1Affine::from_compressed rejects it in its subgroup check (src/kzg_proof.rs:17-25)."""
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    x = 5
    while True:
        y2 = (x * x * x + 4) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            break
        x += 1
    enc = bytearray(x.to_bytes(48, "big"))
    enc[0] |= 0x80 | (0x20 if y > P - y else 0)
    return bytes(enc)
