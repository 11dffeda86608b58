#!/usr/bin/env python3
"""Build the committed golden fixtures from the reference's own test DATA files.

Run once in the build container (needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_fixtures.py

Inputs (data only, no reference source code is read or copied):
  /root/reference/tests/verify_kzg_proof/*/data.yaml            (122 c-kzg-4844 vectors)
  /root/reference/tests/verify_blob_kzg_proof/*/data.yaml       (29)
  /root/reference/tests/verify_blob_kzg_proof_batch/*/data.yaml (24; unused by the
        reference's own tests but valid c-kzg-4844 mainnet vectors)
  /root/reference/src/trusted_setup.txt  (public Ethereum KZG ceremony output; the
        product ships its own copy under kzg_rs_amd/data/, this script only checks
        that the two are byte-identical)

Outputs (all under tests/golden/):
  vectors.json        manifest: every case with hex inputs; blobs replaced by ids
  blobs.bin.xz        the distinct blob byte strings, concatenated, LZMA-compressed
  kat.json            the two scalar known-answer tests the reference asserts
                      (src/kzg_proof.rs:739-753 and :755-778) + the derived (z, y)
                      table and batch intermediates of SURVEY.md 10 / 10.2 / 10.3
"""
import hashlib
import json
import lzma
import os
import sys

import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def hx(s):
    assert s.startswith("0x")
    return s[2:]


def main():
    blobs = {}  # sha256 -> bytes
    order = []

    def blob_id(hexstr):
        raw = bytes.fromhex(hx(hexstr))
        h = hashlib.sha256(raw).hexdigest()[:16]
        if h not in blobs:
            blobs[h] = raw
            order.append(h)
        return h

    out = {"verify_kzg_proof": [], "verify_blob_kzg_proof": [], "verify_blob_kzg_proof_batch": []}

    d = os.path.join(REF, "tests/verify_kzg_proof")
    for name in sorted(os.listdir(d)):
        t = yaml.safe_load(open(os.path.join(d, name, "data.yaml")))
        i = t["input"]
        out["verify_kzg_proof"].append(
            {
                "name": name,
                "commitment": hx(i["commitment"]),
                "z": hx(i["z"]),
                "y": hx(i["y"]),
                "proof": hx(i["proof"]),
                "output": t["output"],
            }
        )

    d = os.path.join(REF, "tests/verify_blob_kzg_proof")
    for name in sorted(os.listdir(d)):
        t = yaml.safe_load(open(os.path.join(d, name, "data.yaml")))
        i = t["input"]
        out["verify_blob_kzg_proof"].append(
            {
                "name": name,
                "blob": blob_id(i["blob"]),
                "commitment": hx(i["commitment"]),
                "proof": hx(i["proof"]),
                "output": t["output"],
            }
        )

    d = os.path.join(REF, "tests/verify_blob_kzg_proof_batch")
    for name in sorted(os.listdir(d)):
        t = yaml.safe_load(open(os.path.join(d, name, "data.yaml")))
        i = t["input"]
        out["verify_blob_kzg_proof_batch"].append(
            {
                "name": name,
                "blobs": [blob_id(b) for b in i["blobs"]],
                "commitments": [hx(c) for c in i["commitments"]],
                "proofs": [hx(p) for p in i["proofs"]],
                "output": t["output"],
            }
        )

    index = []
    cat = bytearray()
    for h in order:
        index.append({"id": h, "offset": len(cat), "length": len(blobs[h])})
        cat += blobs[h]
    out["blob_index"] = index

    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, "blobs.bin.xz"), "wb") as f:
        f.write(lzma.compress(bytes(cat), preset=9 | lzma.PRESET_EXTREME))

    # trusted setup: the product's copy must be byte-identical to the reference's data file
    ours = os.path.join(HERE, "../../kzg_rs_amd/data/trusted_setup.txt")
    ref_ts = open(os.path.join(REF, "src/trusted_setup.txt"), "rb").read()
    if os.path.exists(ours):
        assert open(ours, "rb").read() == ref_ts, "kzg_rs_amd/data/trusted_setup.txt differs from the reference"
    ts_sha = hashlib.sha256(ref_ts).hexdigest()

    kat = {
        "trusted_setup_sha256": ts_sha,
        # reference src/kzg_proof.rs:739-753
        "compute_challenge": {
            "case": "verify_blob_kzg_proof_case_correct_proof_fb324bc819407148",
            "z": "4f00eef944a21cb9f3ac3390702621e4bbf1198767c43c0fb9c8e9923bfbb31a",
        },
        # reference src/kzg_proof.rs:755-778
        "evaluate_polynomial_in_evaluation_form": {
            "case": "verify_blob_kzg_proof_case_correct_proof_19b3f3f8c98ea31e",
            "z": "637c904d316955b7282f980433d5cd9f40d0533c45d0a233c009bc7fe28b92e3",
            "y": "1bdfc5da40334b9c51220e8cbea1679c20a7f32dd3d7f3c463149bb4b41a7d18",
        },
        # SURVEY.md 10 (derived with an independent Python big-int restatement; the two
        # rows above are members of this table and are asserted by the reference itself)
        "zy_table": {
            "correct_proof_0951cfd9ab47a8d3": [
                "04b7b22af63d2b2f1ced8d550560e5d1e4b01e355903dee22781e87826856096",
                "0000000000000000000000000000000000000000000000000000000000000000"],
            "correct_proof_19b3f3f8c98ea31e": [
                "637c904d316955b7282f980433d5cd9f40d0533c45d0a233c009bc7fe28b92e3",
                "1bdfc5da40334b9c51220e8cbea1679c20a7f32dd3d7f3c463149bb4b41a7d18"],
            "correct_proof_84d8089232bc23a8": [
                "5935f3d4dc5393d54160cdb591503bb3875ecb08cb27a8d1d05269bb8b0305d4",
                "0339395aabbec4e6653d783d8cd077f85c19b715cfeffec691d6e52b6e0812fd"],
            "correct_proof_a87a4e636e0f58fb": [
                "42f49b423e71eb01edad0c68a59717e35d404de582fbf6fa9a2ec6096ef9261e",
                "0000000000000000000000000000000000000000000000000000000000000002"],
            "correct_proof_c40b9b515df8721b": [
                "0ea8a7dd57973d93d9a70414c7396d72a101671d86b2f3b10143f6046dfd879d",
                "6b277e8bdd0677e91ee54a5e2777ad1bc363a43a33e46313221584bf255389f8"],
            "correct_proof_cdb3e6d49eb12307": [
                "087327c18b4ae771a880eea6dc5db3a1d208a37cc42b3cc4e179c13dde3b56ce",
                "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000"],
            "correct_proof_fb324bc819407148": [
                "4f00eef944a21cb9f3ac3390702621e4bbf1198767c43c0fb9c8e9923bfbb31a",
                "3921e40e41bc755dafbcf0d0985a1647dff2ae053b014bdeefe490a1c22f9f27"],
        },
        # r, A = sum r^i pi_i, B = sum r^i (C_i - y_i G) + sum r^i z_i pi_i
        # "le" = reference convention (Scalar::to_bytes, src/kzg_proof.rs:321,326),
        # "be" = c-kzg-4844 convention.
        "batch_intermediates": [
            {"case": "verify_blob_kzg_proof_batch_case_e61aafba051ddf79", "n": 2,
             "le": {"r": "29e8a68544f48a2ebcfcc18d9cbfc09ea2986dd50bbfdaacd32e17e9432ae188",
                    "A": "c0" + "00" * 47, "B": "c0" + "00" * 47},
             "be": {"r": "4535ea8cd1e1dc9a939f9367f78372df1c21a391e9949528593a9c59b2e8f213",
                    "A": "c0" + "00" * 47, "B": "c0" + "00" * 47}},
            {"case": "verify_blob_kzg_proof_batch_case_12c097d7ca0261e3", "n": 6,
             "le": {"r": "10b029414f41f3cc85805f00897afdde288d7cdba43527061aff763c1db17a6c",
                    "A": "a2c2b6105879c9d236d37d791622cf5683580ef2954e7c586301dd2bb763de0a0e9cc9c597b6d5666fe1934cdf3a42b7",
                    "B": "95631c28fd1700caa18e972d21ec3f0b85de85cac2212eab5c8f676634f067f12fc2f68bb39a5d14fbb6dd964e1eff9c"},
             "be": {"r": "37b47652f5824edc0894a4f01e2aef5286e7743785c135f1871aeb0968e4dee4",
                    "A": "aaf3a64980464a207f6183d4dc429bee0fdfc16c90011ba8a852f380c5f22f176868dbb5b1f9c2427ccd4b04476c7cd3",
                    "B": "98f1229c1ef789a63ed7c95f641bf0d929a233d61b38eaa35a5e62b7966f55314128a30e382fc8dfd41cc6eb156f13e4"}},
            {"case": "verify_blob_kzg_proof_batch_case_incorrect_proof_add_one", "n": 7,
             "le": {"r": "4b956bb24d486a7dc58ee07e897e63a992cc779c86e2b08a8b2eb309e875f094",
                    "A": "912fc525432420e018e14338b08a29c3cd72a757815c372a72ededd0a270b1c2b29b6f94e37f8d03f7395041f3d0168b",
                    "B": "82c51ffe511f25eb18d73f34cc51ee9602a3587b2bf6f67a8532cc57457eb46e9ed667f252a6946af8f0b5d57e251ea9"},
             "be": {"r": "10c3ca14e10b086481ea2c366f700dfd4e58150ac3d5f1cef062940eeaa58861",
                    "A": "b03ee0a6d5cf3ce2c11d76a8d224bb819fb45b617048cce6323d66f2e0743aea157cb9d4ae2c1a3376117f9608155328",
                    "B": "b6bda1af628634e0501e04a323a610d0b8c817bde11267a0304a3b7e3e54543772de0dd2c76a6c9b7e5a82d9147ddd1e"}},
        ],
        # SURVEY.md 10.3
        "anchors": {
            "tau_g2": {
                "x_c0": "185cbfee53492714734429b7b38608e23926c911cceceac9a36851477ba4c60b087041de621000edc98edada20c1def2",
                "x_c1": "15bfd7dd8cdeb128843bc287230af38926187075cbfbefa81009a2ce615ac53d2914e5870cb452d2afaaab24f3499f72",
                "y_c0": "014353bdb96b626dd7d5ee8599d1fca2131569490e28de18e82451a496a9c9794ce26d105941f383ee689bfbbb832a99",
                "y_c1": "1666c54b0a32529503432fcae0181b4bef79de09fc63671fda5ed1ba9bfa07899495346f3d7ac9cd23048ef30d0a154f"},
            "commitment_fb324": {
                "compressed": "a421e229565952cfff4ef3517100a97da1d4fe57956fa50a442f92af03b1bf37adacc8ad4ed209b31287ea5bb94d9d06",
                "x": "0421e229565952cfff4ef3517100a97da1d4fe57956fa50a442f92af03b1bf37adacc8ad4ed209b31287ea5bb94d9d06",
                "y": "0ee3c0592ab28e14268541106e48ae30618a7aa78851024e647b234f16da727a91162b4c361eafce2fd3c48c006e77c5"},
            "fr_R": "1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe",
            "fr_R2": "0748d9d99f59ff1105d314967254398f2b6cedcb87925c23c999e990f3f29c6d",
            "fr_inv64": "fffffffeffffffff",
            "fp_R": "15f65ec3fa80e4935c071a97a256ec6d77ce5853705257455f48985753c758baebf4000bc40c0002760900000002fffd",
            "fp_R2": "11988fe592cae3aa9a793e85b519952d67eb88a9939d83c08de5476c4c95b6d50a76e6a609d104f1f4df1f341c341746",
            "fp_inv64": "89f3fffcfffcfffd",
            "roots_of_unity_first4": [
                "0000000000000000000000000000000000000000000000000000000000000001",
                "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000",
                "00000000000000008d51ccce760304d0ec030002760300000001000000000000",
                "73eda753299d7d47a5e80b39939ed33467baa40089fb5bfefffeffff00000001"],
        },
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)

    n = {k: len(v) for k, v in out.items()}
    print("cases:", n, "distinct blobs:", len(order), "raw blob bytes:", len(cat))
    print("blobs.bin.xz:", os.path.getsize(os.path.join(HERE, "blobs.bin.xz")))
    print("trusted setup sha256:", ts_sha)


if __name__ == "__main__":
    sys.exit(main())
