"""The register and LDS budgets DESIGN.md quotes for the hot kernels, read from the gfx950 code object inside the built library
(tools/prof/kernel_resources.py: the ELF's own metadata, not a compile log) - a compiler or source change that moves a kernel to
another occupancy class shows up here before it shows up as a benchmark regression.  profiles/r6_kernel_resources.txt is the same
table for every kernel of the product library."""
import importlib.util
import os
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")


def _table():
    from kzg_rs_amd import build
    lib = build.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "prof", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    with tempfile.TemporaryDirectory() as tmp:
        ks = kr.kernels(kr.code_object(lib, tmp))
    return dict(zip(kr.demangle([k["name"] for k in ks]), ks))


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no ROCm LLVM tools")
def test_hot_kernels_keep_their_occupancy_class():
    t = _table()
    waves = lambda k: min(8, 512 // ((k["vgpr"] + k["agpr"] + 7) // 8 * 8))
    # kernel -> (waves per SIMD by registers, may it use scratch?)   [DESIGN.md 4, docs/lab_notebook.md 9]
    want = {
        "kzg::k_blob_challenge_t<4>": (4, False),                  # 114 VGPRs: the whole launch resident at once
        "kzg::k_blob_evaluate_t<true>": (3, False),                # 166 VGPRs, no scratch (round 4)
        "kzg::k_g1_decode_multiples29<4, true>": (2, True),        # 256 VGPRs, a few spills accepted
        "kzg::k_msm_window<kzg::Curve29Aff, true>": (3, True),     # 146 VGPRs; 116 B of scratch for the rare complete addition
        "kzg::k_fb_window": (3, True),                             # the fixed-base MSM's bucket kernel: the same budget
        "kzg::k_msm_reduce<false>": (2, False),
    }
    for name, (w, scratch_ok) in want.items():
        k = t[name]
        assert waves(k) == w, (name, k)
        assert k["vspill"] <= (16 if scratch_ok else 0), (name, k)
        if not scratch_ok:
            assert k["scratch"] == 0, (name, k)
    # static LDS: three window workgroups (3 KB static + <= 48 KB of sorted list) and three evaluation workgroups (38 KB) per CU fit 160 KB
    assert t["kzg::k_fb_window"]["lds"] + 4 * 12288 + 16 <= 160 * 1024 // 3
    assert t["kzg::k_msm_window<kzg::Curve29Aff, true>"]["lds"] <= 3200 and t["kzg::k_blob_evaluate_t<true>"]["lds"] * 3 <= 160 * 1024


def test_committed_resource_table_matches_the_built_library():
    path = os.path.join(ROOT, "profiles", "r6_kernel_resources.txt")
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")) or not os.path.exists(path):
        pytest.skip("no ROCm LLVM tools / table not collected")
    from kzg_rs_amd import build
    head = open(path).readline()
    if build.kernel_key() not in head:
        pytest.skip("the committed table describes another kernel key (re-collect with tools/prof/collect_round.sh)")
    t = _table()
    rows = {}
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel "):
            continue
        f = ln.split()
        rows[" ".join(f[:-8])] = [int(x) for x in f[-8:]]
    for name in ("kzg::k_blob_evaluate_t<true>", "kzg::k_fb_window", "kzg::k_blob_challenge_t<4>"):
        k = t[name]
        assert rows[name][:5] == [k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"]], (name, rows[name], k)
