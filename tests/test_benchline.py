"""The benchmark line's roofline / path / valu blocks (kzg_rs_amd/benchline.py) are a pure function of what a run measured and
the PMC profile under profiles/: here it is fed the measurement of the round-5 line (tests/golden/bench_measurement_r5.json,
built from profiles/r5_bench_unprofiled.json) and every figure is checked against the ones it must follow from.  Round 5
printed the pairing's 3.2 ms as the dominant kernel's standalone_ms next to an achieved_standalone that belonged to 20.7 ms
(a rebound loop variable): the first assertions are exactly that."""
import copy
import json
import os

import pytest

from kzg_rs_amd import benchline as BL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_measurement_r5.json")))
PMC = json.load(open(os.path.join(ROOT, "profiles", "r5_pmc.json")))


def _rel(a, b):
    return abs(a - b) / abs(b)


def _assemble(m=None, pmc=None, key="K1", pmc_key="K1"):
    pmc = copy.deepcopy(PMC if pmc is None else pmc)
    if pmc_key is not None:
        pmc["kernel_key"] = pmc_key
    return BL.assemble(copy.deepcopy(m or FIX["measurement"]), "r5_pmc.json", pmc, key)


def test_dominant_kernel_columns_follow_from_each_other():
    out = _assemble()
    rf = out["roofline"]
    rows = {r["kernel"]: r for r in rf["kernels"]}
    dom = rf["kernel"]
    assert dom == "k_blob_challenge"   # largest stand-alone duration of the round-5 run
    assert dom == max(BL.STAMPED, key=lambda k: rows[k]["standalone_ms"])
    # the round-5 defect: standalone_ms must be the DOMINANT kernel's, and achieved_standalone must follow from it
    assert rf["standalone_ms"] == rows[dom]["standalone_ms"] == FIX["measurement"]["solo_stamps_ms"][dom]
    assert rf["standalone_ms"] != FIX["round5_line"]["standalone_ms_WRONG_in_round5"]
    alg = BL.ALG_BYTES[dom] * 1024 * 256
    assert rf["algorithmic_bytes_per_launch"] == alg == rows[dom]["algorithmic_bytes_per_launch"]
    assert _rel(rf["achieved_standalone"], alg / rf["standalone_ms"] / 1e6) < 1e-4
    assert _rel(rf["achieved_standalone"], FIX["round5_line"]["achieved_standalone"]) < 1e-3
    assert _rel(rf["frac_standalone"], rf["achieved_standalone"] / 8000.0) < 1e-3
    # in-flight column: the same kernel's interval over the timed region
    assert rf["launch_ms"] == FIX["measurement"]["in_flight_ms"][dom]
    assert _rel(rf["achieved"], alg / rf["launch_ms"] / 1e6) < 1e-4
    assert _rel(rf["frac"], rf["achieved"] / rf["peak"]) < 1e-3
    assert _rel(rf["frac"], FIX["round5_line"]["roofline_frac"]) < 1e-3
    # the other defensible choice is named, with its own fraction
    dfl = rf["dominant_in_flight"]
    assert dfl["kernel"] == "k_blob_evaluate" and dfl["in_flight_ms"] == rows["k_blob_evaluate"]["in_flight_ms"]
    assert dfl["in_flight_ms"] > FIX["round5_line"]["ms_per_step"]   # residency of overlapping groups: longer than a step, and said so
    assert "may exceed ms_per_step" in rf["kernels_note"] and "may exceed ms_per_step" in dfl["note"]


def test_every_kernel_row_is_consistent():
    out = _assemble()
    for r in out["roofline"]["kernels"]:
        if r["kernel"] not in BL.STAMPED:
            continue
        alg = BL.ALG_BYTES[r["kernel"]] * 262144
        assert r["algorithmic_bytes_per_launch"] == alg
        assert _rel(r["achieved_standalone_GBps"], alg / r["standalone_ms"] / 1e6) < 1e-3
        assert _rel(r["frac_standalone"], r["achieved_standalone_GBps"] / 8000.0) < 1e-3
        assert _rel(r["achieved_in_flight_GBps"], alg / r["in_flight_ms"] / 1e6) < 1e-3
        assert _rel(r["hbm_traffic_ratio"], r["hbm_traffic_bytes"] / alg) < 1e-3
        pk = PMC["kernels"][BL.PMC_NAME[r["kernel"]]]
        assert r["hbm_traffic_bytes"] == round(pk["hbm_bytes_corrected"])
        assert _rel(r["cycles_per_inst_standalone"], r["standalone_ms"] * 1e-3 * 2319.6e6 * 1024 / pk["SQ_INSTS_VALU"]) < 1e-3


def test_path_and_binding_bound():
    out = _assemble()
    rf, path, valu = out["roofline"], out["path"], out["valu"]
    m = FIX["measurement"]
    blobs_per_s = m["n"] * m["G"] * m["K"] / m["elapsed_s"]
    assert _rel(blobs_per_s, FIX["round5_line"]["value"]) < 1e-4
    assert _rel(path["algorithmic_GBps"], BL.PATH_ALG_BYTES * blobs_per_s / 1e9) < 1e-4
    assert rf["frac_path"] == path["frac"] and _rel(path["frac"], FIX["round5_line"]["path_frac"]) < 1e-3
    assert rf["frac_of_binding_bound"] == path["valu_frac_of_mix_ceiling"]
    assert _rel(rf["frac_of_binding_bound"], path["valu_mix_ceiling_cycles_per_inst"] / valu["cycles_per_inst_at_measured_clock"]) < 1e-3
    assert _rel(rf["frac_of_binding_bound"], FIX["round5_line"]["valu_frac_of_mix_ceiling"]) < 2e-3
    assert 0.5 < rf["frac_of_binding_bound"] <= 1.0
    # consistency the driver checks: bytes/step / ms_per_step below the peak; the dominant kernel's stand-alone cost below a step
    assert path["algorithmic_GBps"] < 8000.0 and rf["standalone_ms"] < m["elapsed_s"] / m["K"] * 1e3
    assert rf["inputs"]["elapsed_s"] == m["elapsed_s"]   # the line carries what it was assembled from


def test_stale_or_unstamped_pmc_is_flagged_and_not_used():
    fresh = _assemble()
    assert fresh["roofline"]["traffic_stale"] is None and fresh["roofline"]["traffic"] > 0
    assert "kernel key K1" in fresh["roofline"]["traffic_source"]
    for pmc_key, word in ((None, "no kernel_key stamp"), ("OTHER", "NOT used")):
        out = _assemble(pmc_key=pmc_key)
        rf = out["roofline"]
        assert rf["traffic"] is None and rf["traffic_source"] is None and word in rf["traffic_stale"]
        assert out["valu"] is None and out["path"]["hbm_traffic_ratio"] is None and rf["frac_of_binding_bound"] is None
        assert all(r.get("hbm_traffic_bytes") is None for r in rf["kernels"])
        # what the run itself measured is still there
        assert rf["standalone_ms"] == fresh["roofline"]["standalone_ms"] and rf["frac"] == fresh["roofline"]["frac"]
    none = BL.assemble(copy.deepcopy(FIX["measurement"]), None, None, "K1")
    assert none["roofline"]["traffic_stale"] == "no PMC profile under profiles/"


def test_without_a_standalone_group_the_rule_falls_back_and_says_so():
    m = copy.deepcopy(FIX["measurement"])
    m["solo_stamps_ms"] = None
    m["standalone_event_ms"] = None
    rf = _assemble(m)["roofline"]
    assert rf["kernel"] == "k_blob_evaluate" and "no stand-alone group" in rf["kernel_chosen_by"]
    assert rf["standalone_ms"] is None and rf["achieved_standalone"] is None


def test_extrapolated_launch_size_is_said():
    m = copy.deepcopy(FIX["measurement"])
    m["G"] = 128
    rf = _assemble(m)["roofline"]
    assert "EXTRAPOLATED" in rf["traffic_source"]
    rows = {r["kernel"]: r for r in rf["kernels"]}
    assert rows["k_blob_challenge"]["hbm_traffic_bytes"] == round(PMC["kernels"][BL.PMC_NAME["k_blob_challenge"]]["hbm_bytes_corrected"] / 2)


def test_pmc_file_order_prefers_the_newest_round():
    assert BL.PMC_FILES[0].startswith("r6_") and list(BL.PMC_FILES) == sorted(BL.PMC_FILES, reverse=True)
    name, pmc = BL.load_pmc(ROOT)
    assert name in BL.PMC_FILES and "kernels" in pmc


@pytest.mark.parametrize("name", ["r6_pmc.json"])
def test_committed_profile_carries_the_tree_key(name):
    """The PMC profile of this round is stamped with the kernel key of the tree it was collected on; when the kernels changed
    after it was collected the benchmark line says so (traffic_stale) - this test only checks that the stamp exists."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip("not collected yet")
    assert json.load(open(path)).get("kernel_key")
