"""include/kzg_rs_amd.h as a C header, bound by a C caller (tests/host/cabi_vectors.c): compiled with
gcc -std=c11 -Wall -Wextra -Werror and linked against libkzg_rs_amd.so - a prototype that drifted from its definition, a
C++-ism in the header or a missing symbol fails here, not in a downstream build.  Without a GPU the program checks the
no-fallback contract (clean KZG_ERROR); on the GPU box it runs all 175 vectors the reference ships through
kzg_verify_kzg_proof / kzg_verify_blob_kzg_proof / kzg_verify_blob_kzg_proof_batch (strict null <=> error)."""
import os
import struct
import subprocess

import pytest

import golden_data as G

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "kzg_rs_amd")
SETUP = os.path.join(LIBDIR, "data", "trusted_setup.txt")


@pytest.fixture(scope="module")
def program(tmp_path_factory):
    from kzg_rs_amd import build
    build.build()
    exe = str(tmp_path_factory.mktemp("cabi") / "cabi_vectors")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(HERE, "host", "cabi_vectors.c"), "-L", LIBDIR, "-lkzg_rs_amd", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _unhex(s):
    s = s[2:] if s.startswith("0x") else s
    return bytes.fromhex(s)


def _field(b):
    return struct.pack("<I", len(b)) + b


def _vector_file(path):
    V = G.vectors()
    out, count = [], 0
    exp = lambda o: {True: 1, False: 0, None: -1}[o]
    for c in V["verify_kzg_proof"]:
        out.append(struct.pack("<Bb", 1, exp(c["output"])) + b"".join(_field(_unhex(c[k])) for k in ("commitment", "z", "y", "proof")))
        count += 1
    for c in V["verify_blob_kzg_proof"]:
        out.append(struct.pack("<Bb", 2, exp(c["output"])) + _field(G.blob(c["blob"])) + _field(_unhex(c["commitment"])) + _field(_unhex(c["proof"])))
        count += 1
    for c in V["verify_blob_kzg_proof_batch"]:
        rec = struct.pack("<Bb", 3, exp(c["output"]))
        rec += struct.pack("<I", len(c["blobs"])) + b"".join(_field(G.blob(b)) for b in c["blobs"])
        rec += struct.pack("<I", len(c["commitments"])) + b"".join(_field(_unhex(x)) for x in c["commitments"])
        rec += struct.pack("<I", len(c["proofs"])) + b"".join(_field(_unhex(x)) for x in c["proofs"])
        out.append(rec)
        count += 1
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x56475A4B, count) + b"".join(out))
    return count


def test_header_compiles_as_c_and_links(program):
    """No GPU here: the C caller must get KZG_ERROR with a message from the constructor, KZG_BADARGS for null handles,
    and the handle-free host entry point (kzg_batch_challenges) must work."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-device contract is checked on the CPU box")
    r = subprocess.run([program, SETUP, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "no-gpu ok" in r.stdout, r.stdout + r.stderr


def test_header_declares_nothing_but_c(program):
    """every prototype of the header is visible to a C compiler with the strictest flags as well (declarations only)"""
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wstrict-prototypes", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "kzg_rs_amd.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_caller_runs_all_175_reference_vectors(program, tmp_path):
    vec = str(tmp_path / "vectors.bin")
    assert _vector_file(vec) == 175
    r = subprocess.run([program, SETUP, vec], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "175 vectors: 175 ok (verify_kzg_proof 122, verify_blob_kzg_proof 29, verify_blob_kzg_proof_batch 24)" in r.stdout
