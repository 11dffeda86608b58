"""CPU-only: the C-ABI shared library loads and exports every entry point include/kzg_rs_amd.h
declares (no compute calls without a GPU), and fails loudly - not silently - without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kzg_rs_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_reference_entry_points():
    syms = declared_symbols()
    for s in ("kzg_verify_kzg_proof", "kzg_verify_kzg_proof_batch", "kzg_verify_blob_kzg_proof", "kzg_verify_blob_kzg_proof_batch",
              "kzg_verify_blob_kzg_proof_batch_device", "kzg_settings_load_trusted_setup", "kzg_settings_free"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from kzg_rs_amd import api, build
    build.build()
    L = ctypes.CDLL(api.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), "libkzg_rs_amd.so does not export " + s


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the settings constructor must fail with an error, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kzg_rs_amd import api
    with pytest.raises(api.KzgError) as e:
        api.KzgSettings.load_trusted_setup_file()
    assert e.value.kind == "InternalError"


def test_host_sha256_both_paths():
    """The host-side SHA-256 that hashes the batch transcripts (capi_host_util.hpp: SHA-NI when the CPU has it, a
    portable compression otherwise) against hashlib - no GPU involved."""
    import ctypes as C
    import hashlib
    import random
    from kzg_rs_amd import api, build
    build.build()
    L = C.CDLL(api.LIB_PATH)
    L.kzg_debug_host_sha256.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int]
    rng = random.Random(256)
    out = C.create_string_buffer(32)
    for n in [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 1000, 32 + 160 * 1024, 32 + 160 * 7]:
        data = rng.randbytes(n)
        rc = L.kzg_debug_host_sha256(out, data, n, 0)
        assert rc in (0, 1) and out.raw == hashlib.sha256(data).digest(), n
    # the portable compression on whole blocks: one more block of padding makes it a full SHA-256
    for n in [0, 64, 128, 64 * 37]:
        data = rng.randbytes(n)
        padded = data + b"\x80" + bytes(55) + (8 * n).to_bytes(8, "big")
        assert L.kzg_debug_host_sha256(out, padded, len(padded), 1) == 0
        assert out.raw == hashlib.sha256(data).digest(), n
