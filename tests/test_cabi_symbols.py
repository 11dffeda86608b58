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


def test_batch_challenges_is_host_code_and_matches_the_oracle():
    """kzg_batch_challenges (the hash half of compute_r_powers, src/kzg_proof.rs:291-334) needs neither a handle nor a GPU:
    both record layouts against the oracle's compute_r on canonical random records, n = 1 included."""
    import ctypes as C
    import random

    import oracle_lib as O
    from kzg_rs_amd import api

    L = C.CDLL(api.LIB_PATH)
    L.kzg_batch_challenges.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
    rng = random.Random(11)
    for world, B, n in ((0, 3, 5), (0, 1, 1), (2, 4, 3), (3, 2, 7)):
        n_total = (world or 1) * n
        # records of batch b in global order: C(48) || z(32 LE) || y(32 LE) || pi(48); z, y canonical (top byte 0)
        flat = []
        for b in range(B):
            recs = []
            for _ in range(n_total):
                z, y = bytearray(rng.randbytes(32)), bytearray(rng.randbytes(32))
                z[31] = y[31] = 0
                recs.append(rng.randbytes(48) + bytes(z) + bytes(y) + rng.randbytes(48))
            flat.append(recs)
        if world == 0:
            buf = b"".join(b"".join(r) for r in flat)
        else:  # [world][B][n]
            buf = b"".join(b"".join(flat[b][k * n:(k + 1) * n]) for k in range(world) for b in range(B))
        out = C.create_string_buffer(32 * B)
        assert L.kzg_batch_challenges(out, buf, world, B, n) == 0
        for b in range(B):
            rec = flat[b]
            want = O.compute_r(b"".join(x[:48] for x in rec), b"".join(x[48:80][::-1] for x in rec), b"".join(x[80:112][::-1] for x in rec),
                               b"".join(x[112:] for x in rec), n_total)
            assert out.raw[32 * b: 32 * b + 32][::-1] == want, (world, B, n, b)


def test_loading_the_library_leaves_the_environment_alone():
    """Rounds 1-3 set GPU_MAX_HW_QUEUES from a load-time constructor; the library now reads its environment (KZG_DEVICES,
    KZG_OPTIONS) and writes nothing: the host sets GPU_MAX_HW_QUEUES=8 itself, and a constructor that sees
    fewer says so in the handle's note, kzg_settings_note() (GPU test: test_gpu_parity.py::test_constructor_notes_too_few_hardware_queues)."""
    import subprocess
    import sys
    from kzg_rs_amd import api, build
    build.build()
    code = ("import ctypes, os\n"
            "before = dict(os.environ)\n"
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p\n"
            "for lib in %r:\n"
            "    ctypes.CDLL(lib)\n"
            "print(g(b'GPU_MAX_HW_QUEUES'), dict(os.environ) == before)\n" % ([api.LIB_PATH, api.LIB_AB_PATH],))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "None True"


def test_kzg_options_string_is_parsed_as_documented():
    """KZG_OPTIONS = "key=value;key=value" (';' or blanks between entries; a bare key means 1), read through the library's own
    parser (csrc/capi_host_util.hpp) - re-read when the string changes, unknown keys harmless; and which of the two libraries
    is the A/B build."""
    import ctypes as C
    from kzg_rs_amd import api, build
    build.build()

    def get(L, key):
        buf = C.create_string_buffer(64)
        n = L.kzg_debug_option(key.encode(), buf, 64, None)
        return None if n < 0 else buf.value.decode()

    L = C.CDLL(api.LIB_PATH)
    L.kzg_debug_option.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    keep = os.environ.get("KZG_OPTIONS")
    try:
        os.environ["KZG_OPTIONS"] = "challenge_kernel=lane; multi_min_blobs=2\tsingle_stream host_slices=16;;x=a=b"
        assert get(L, "challenge_kernel") == "lane" and get(L, "multi_min_blobs") == "2" and get(L, "single_stream") == "1"
        assert get(L, "host_slices") == "16" and get(L, "x") == "a=b" and get(L, "pairing") is None
        os.environ["KZG_OPTIONS"] = "pairing=2"  # the string has changed: parsed again
        assert get(L, "pairing") == "2" and get(L, "challenge_kernel") is None
        with api.options(msm_cpb=4):
            assert get(L, "msm_cpb") == "4" and get(L, "pairing") == "2"
        assert get(L, "msm_cpb") is None
        os.environ.pop("KZG_OPTIONS")
        assert get(L, "pairing") is None
    finally:
        if keep is None:
            os.environ.pop("KZG_OPTIONS", None)
        else:
            os.environ["KZG_OPTIONS"] = keep
    ab = C.c_int(-1)
    L.kzg_debug_option(b"x", None, 0, C.byref(ab))
    assert ab.value == 0
    LA = C.CDLL(api.LIB_AB_PATH)
    LA.kzg_debug_option.argtypes = L.kzg_debug_option.argtypes
    LA.kzg_debug_option(b"x", None, 0, C.byref(ab))
    assert ab.value == 1


def test_build_checks_the_kernels_the_host_object_launches(tmp_path):
    """kzg_rs_amd/build.py caches the gfx950 code object across host-only edits; the key is a heuristic, so the link step
    compares the kernels the host object launches (its __device_stub__ symbols) with the kernels the cached code object
    defines.  Here: a device pass that instantiates k_t<3> and a host pass that launches k_t<4> - the difference is seen."""
    import subprocess
    from kzg_rs_amd import build
    src = tmp_path / "t.hip"
    src.write_text("#include <hip/hip_runtime.h>\n"
                   "template <int N> __global__ void k_t(int* p) { p[0] = N; }\n"
                   "__global__ void k_u(float* q, int n) { q[0] = n; }\n"
                   "void f(int* p, float* q) { hipLaunchKernelGGL(k_t<VARIANT>, dim3(1), dim3(1), 0, 0, p); hipLaunchKernelGGL(k_u, dim3(1), dim3(1), 0, 0, q, 1); }\n")
    dev, host_same, host_other = (str(tmp_path / n) for n in ("t.out", "same.o", "other.o"))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-DVARIANT=3", "--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", dev, str(src)])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-DVARIANT=3", "--cuda-host-only", "-c", "-o", host_same, str(src)])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-DVARIANT=4", "--cuda-host-only", "-c", "-o", host_other, str(src)])
    defined = build._kernel_names(dev, stubs=False)
    assert defined == {"void k_t<3>(int*)", "k_u(float*, int)"}
    assert build._kernel_names(host_same, stubs=True) - defined == set()
    assert build._kernel_names(host_other, stubs=True) - defined == {"void k_t<4>(int*)"}


def test_host_blob_challenge_matches_the_reference_kat_and_the_oracle():
    """Small host batches take their Fiat-Shamir challenges from the host's SHA-NI cores (csrc/host_only.hpp
    host_blob_challenge: the same serial-chain argument as for the batch transcript).  Against the reference's own known
    answer (test_compute_challenge, src/kzg_proof.rs:739-752), the derived (z) table of the valid mainnet blobs, and the oracle
    on random blobs and commitments (digests above r included)."""
    import ctypes as C
    import random
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd import api, build
    build.build()
    L = C.CDLL(api.LIB_PATH)
    L.kzg_debug_host_blob_challenge.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.kzg_debug_host_blob_challenge.restype = None

    def z(blob, c):
        out = C.create_string_buffer(32)
        L.kzg_debug_host_blob_challenge(out, blob, c)
        return out.raw

    for blob, c, _ in G.valid_blob_tuples():
        assert z(blob, c) == O.compute_challenge(blob, c)
    k = G.kat()["compute_challenge"]  # the reference's own known answer
    c = G.case("verify_blob_kzg_proof", k["case"])
    assert z(G.blob(c["blob"]), bytes.fromhex(c["commitment"])).hex() == k["z"]
    for suffix, (zz, _) in G.kat()["zy_table"].items():
        c = G.case("verify_blob_kzg_proof", suffix)
        assert z(G.blob(c["blob"]), bytes.fromhex(c["commitment"])).hex() == zz
    rng = random.Random(2718)
    above_r = 0
    for _ in range(24):
        blob, c = rng.randbytes(131072), rng.randbytes(48)
        want = O.compute_challenge(blob, c)
        assert z(blob, c) == want
        above_r += int.from_bytes(O.sha256(b"FSBLOBVERIFY_V1_" + bytes(14) + b"\x10\x00" + blob + c), "big") >= 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    assert above_r >= 3  # (a digest is above r with probability 0.55: the reduction was exercised)
