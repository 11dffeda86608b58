"""tests/golden/wire_formats.json: what the reference's optional serde / rkyv derives (src/dtypes.rs:9-17, Cargo.toml:41-43) put on
the wire for Bytes32 / Bytes48 / Blob - WRITTEN OUT FROM THE FORMATS' SPECIFICATIONS, not produced by the reference (no Rust
toolchain in this image; INTEGRATION.md keeps saying so).  The encodings are built here from the rules below alone, without
importing kzg_rs_amd, and tests/test_wire_formats.py holds the Python mirror (kzg_rs_amd/api.py) and the shim's derive attributes
against them.

Rules (each with the specification it comes from):
  R1  `struct Bytes32(#[serde(with = "serde_arrays")] pub [u8; 32])` serialises as serialize_newtype_struct -> the field alone
      (serde data model, "newtype struct": formats treat it as its inner value; serde_json and bincode both do).
  R2  serde_arrays 0.2 (`serde_arrays::serialize`): a `[T; N]` is a TUPLE of N elements - serialize_tuple(N) then N
      serialize_element calls (the crate's README: "serialize ... arrays of any size ... as tuples", the form serde itself uses for
      arrays up to 32).
  R3  bincode 1.x (`bincode::serialize`, fixint little-endian default): a tuple is its elements in order with NO length prefix; a u8
      is one byte.  -> [u8; N] = the N raw bytes.  (bincode spec: "tuples and fixed-size arrays: encoded as their elements")
  R4  serde_json: a tuple is a JSON array; a u8 is a JSON number; `to_string` writes no whitespace.  -> "[1,2,...]"
  R5  rkyv 0.8 (`rkyv::to_bytes::<Error>`): `[T; N]` archives as `[Archived<T>; N]`, `u8` archives as `u8`, a one-field tuple
      struct archives as a one-field struct of the archived field (no padding at alignment 1); the root object is written last and
      nothing precedes it when the value owns no out-of-line data.  -> the buffer is the N raw bytes, alignment 1.
"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def bincode_u8_array(data):   # R1 + R2 + R3
    out = bytearray()
    for b in data:            # tuple: elements in order, no length prefix; u8: one byte
        out.append(b)
    return bytes(out)


def json_u8_array(data):      # R1 + R2 + R4
    return "[" + ",".join(str(b) for b in data) + "]"


def rkyv_u8_array(data):      # R5
    return bytes(data)


def vec(kind, name, data):
    big = len(data) > 64
    bc, js, rk = bincode_u8_array(data), json_u8_array(data), rkyv_u8_array(data)
    v = {"type": kind, "name": name, "size": len(data), "bincode_len": len(bc), "rkyv_len": len(rk), "rkyv_align": 1, "json_len": len(js),
         "input_sha256": hashlib.sha256(bytes(data)).hexdigest(), "bincode_sha256": hashlib.sha256(bc).hexdigest(),
         "rkyv_sha256": hashlib.sha256(rk).hexdigest(), "json_sha256": hashlib.sha256(js.encode()).hexdigest()}
    if big:
        v.update({"input_head_hex": bytes(data[:64]).hex(), "input_tail_hex": bytes(data[-64:]).hex(), "json_head": js[:80], "json_tail": js[-40:],
                  "input_rule": "byte i = (7 * i + 3) % 256 with byte 0 of every 32-byte element forced to 0 (canonical field elements)"})
    else:
        v.update({"input_hex": bytes(data).hex(), "bincode_hex": bc.hex(), "rkyv_hex": rk.hex(), "json": js})
    return v


def blob_pattern():
    b = bytearray((7 * i + 3) % 256 for i in range(131072))
    for i in range(0, 131072, 32):
        b[i] = 0
    return bytes(b)


G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
R_MINUS_1 = (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001 - 1).to_bytes(32, "big")
out = {
    "what": __doc__.strip().split("\n\n")[0],
    "not_produced_by_the_reference": True,
    "reference": {"derives": "src/dtypes.rs:9-17", "features": "Cargo.toml:41-43", "serde_arrays": "Cargo.toml:24 (0.2.0)", "rkyv": "Cargo.toml:25 (0.8.10)"},
    "rules": {"R1": "newtype struct -> its field (serde data model)", "R2": "serde_arrays: [T; N] as a tuple of N elements",
              "R3": "bincode 1.x fixint: tuple = elements in order, no length prefix; u8 = 1 byte", "R4": "serde_json: tuple = JSON array, no whitespace",
              "R5": "rkyv 0.8: [u8; N] newtype archives as the N bytes, alignment 1, root object at the end of the buffer (= the whole buffer)"},
    "vectors": [
        vec("Bytes32", "zero", bytes(32)), vec("Bytes32", "ascending", bytes(range(32))), vec("Bytes32", "all_ff", bytes([255] * 32)),
        vec("Bytes32", "r_minus_1", R_MINUS_1),
        vec("Bytes48", "g1_generator", G1_GEN), vec("Bytes48", "g1_identity", bytes([0xC0]) + bytes(47)), vec("Bytes48", "descending", bytes(range(47, -1, -1))),
        vec("Blob", "zero", bytes(131072)), vec("Blob", "pattern", blob_pattern()),
    ],
}
json.dump(out, open(os.path.join(HERE, "wire_formats.json"), "w"), indent=1)
print("wrote", os.path.join(HERE, "wire_formats.json"))
