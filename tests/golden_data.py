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
