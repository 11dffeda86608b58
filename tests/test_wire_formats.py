"""SURVEY 8(f)4 - the serde / rkyv wire formats of Bytes32 / Bytes48 / Blob (src/dtypes.rs:9-17, Cargo.toml:41-43), pinned as far as
they can be without a Rust toolchain: tests/golden/wire_formats.json holds the encodings written out from the formats'
specifications (tests/golden/make_wire_formats.py: rule by rule, without importing the package); the Python mirror's
to_wire_bytes / to_json / from_* (kzg_rs_amd/api.py) must produce and accept exactly those, and the shim's derive attributes must
be the reference's (serde with serde_arrays, rkyv Archive / Serialize / Deserialize, behind the same features)."""
import hashlib
import json
import os
import re

import pytest

from kzg_rs_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = json.load(open(os.path.join(ROOT, "tests", "golden", "wire_formats.json")))
TYPES = {"Bytes32": api.Bytes32, "Bytes48": api.Bytes48, "Blob": api.Blob}


def _input(v):
    if "input_hex" in v:
        return bytes.fromhex(v["input_hex"])
    if v["name"] == "zero":
        return bytes(v["size"])
    b = bytearray((7 * i + 3) % 256 for i in range(v["size"]))   # v["input_rule"]
    for i in range(0, v["size"], 32):
        b[i] = 0
    return bytes(b)


@pytest.mark.parametrize("v", W["vectors"], ids=lambda v: "%s-%s" % (v["type"], v["name"]))
def test_python_mirror_matches_the_written_out_encodings(v):
    cls = TYPES[v["type"]]
    raw = _input(v)
    assert len(raw) == v["size"] == cls.SIZE and hashlib.sha256(raw).hexdigest() == v["input_sha256"]
    x = cls(raw)
    wire = x.to_wire_bytes()
    assert len(wire) == v["bincode_len"] == v["rkyv_len"] == cls.SIZE               # no length prefix, no header, alignment 1
    assert hashlib.sha256(wire).hexdigest() == v["bincode_sha256"] == v["rkyv_sha256"]
    js = x.to_json()
    assert len(js) == v["json_len"] and hashlib.sha256(js.encode()).hexdigest() == v["json_sha256"]
    if "bincode_hex" in v:
        assert wire.hex() == v["bincode_hex"] == v["rkyv_hex"] and js == v["json"]
    else:
        assert js.startswith(v["json_head"]) and js.endswith(v["json_tail"]) and raw[:64].hex() == v["input_head_hex"] and raw[-64:].hex() == v["input_tail_hex"]
    # and back
    assert cls.from_wire_bytes(wire).data == raw and cls.from_json(js).data == raw
    assert cls.from_json(json.dumps(list(raw))).data == raw                          # (a deserialiser accepts whitespace)


def test_wrong_sizes_are_the_deserialisers_error():
    for cls in TYPES.values():
        for bad in (b"", bytes(cls.SIZE - 1), bytes(cls.SIZE + 1)):
            with pytest.raises(api.KzgError) as e:
                cls.from_wire_bytes(bad)
            assert e.value.kind == "InvalidBytesLength"
        for bad in ("[]", "[256]", json.dumps([0] * (cls.SIZE - 1)), json.dumps([0] * cls.SIZE + [1]), json.dumps({"0": 1}), json.dumps([0.5] * cls.SIZE),
                    json.dumps([True] * cls.SIZE), json.dumps([-1] + [0] * (cls.SIZE - 1))):
            with pytest.raises(api.KzgError) as e:
                cls.from_json(bad)
            assert e.value.kind == "InvalidBytesLength"


def test_fixture_says_what_it_is():
    assert W["not_produced_by_the_reference"] is True and set(W["rules"]) == {"R1", "R2", "R3", "R4", "R5"}
    assert {v["type"] for v in W["vectors"]} == {"Bytes32", "Bytes48", "Blob"}


def test_shim_derives_are_the_references():
    """rust/kzg-rs-amd/src/dtypes.rs (uncompiled here) must carry the reference's derive attributes verbatim in meaning: serde
    Serialize / Deserialize behind feature "serde" with the field under serde_arrays, rkyv Archive / Serialize / Deserialize behind
    feature "rkyv"; Cargo.toml must name the same crates and features (kzg-rs Cargo.toml:21-25,41-43)."""
    d = open(os.path.join(ROOT, "rust", "kzg-rs-amd", "src", "dtypes.rs")).read()
    assert re.search(r'cfg_attr\(feature = "serde", derive\(serde::Serialize, serde::Deserialize\)\)', d)
    assert re.search(r'cfg_attr\(feature = "rkyv", derive\(rkyv::Archive, rkyv::Serialize, rkyv::Deserialize\)\)', d)
    assert re.search(r'cfg_attr\(feature = "serde", serde\(with = "serde_arrays"\)\)\] pub \[u8; \$size\]', d)
    c = open(os.path.join(ROOT, "rust", "kzg-rs-amd", "Cargo.toml")).read()
    assert 'serde_arrays = "0.2.0"' in c and re.search(r'rkyv = \{ version = "0\.8\.10", optional = true \}', c)
    assert 'rkyv = ["dep:rkyv"]' in c and 'serde = ["dep:serde"]' in c
