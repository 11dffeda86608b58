"""Assembly of the benchmark line's `roofline`, `path` and `valu` blocks from what a run measured (bench.py) and the PMC profile
committed under profiles/ - a PURE function of its inputs, so that (i) tests/test_benchline.py can feed it a recorded measurement
and check that every figure follows from the others, and (ii) anyone can recompute the line of a BENCH_rNN.json from the
`roofline.inputs` it carries and the profiles/ files it names.

Rules, stated once:
  * units: one launch = one launch group = n x G blobs; algorithmic bytes per blob per kernel = ALG_BYTES.
  * THE DOMINANT KERNEL is the one with the largest STAND-ALONE duration: its own execution interval (in-kernel s_memrealtime stamps,
    first wavefront in to last wavefront out) in one launch group run alone on a single-stream handle.  Cost, not residency.
    Every top-level figure of `roofline` (launch_ms, achieved, frac, standalone_ms, achieved_standalone, frac_standalone, traffic)
    is about THAT kernel; `dominant_in_flight` names the kernel with the largest in-flight interval, the other defensible choice.
  * launch_ms / achieved / frac: the dominant kernel's interval averaged over the launch groups of the timed region, F groups in
    flight - what rocprofv3 --kernel-trace --stats averages for the same command.  With groups overlapping this is RESIDENCY: an
    in-flight interval may exceed ms_per_step (four groups share the chip, each kernel's interval spans the time it shared it).
  * standalone_ms / achieved_standalone / frac_standalone: the same kernel alone on the chip; equal to kernels[dominant].standalone_ms.
  * frac_path: whole-path algorithmic bytes per blob x blobs/s / HBM peak.  frac_of_binding_bound: the VALU-issue fraction - the
    mix ceiling (cycles per wave-instruction the instruction mix can issue at) / the cycles per instruction the path achieved at the
    measured shader clock.  The path is VALU-issue bound; the HBM fractions are what BASELINE.json's metric asks to be reported.
  * traffic: HBM bytes of the dominant kernel per launch from the PMC profile (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction),
    only when the profile's `kernel_key` equals the running tree's (kzg_rs_amd.build.kernel_key()); otherwise None and
    `traffic_stale` says why."""
import json
import os

BYTES_PER_BLOB = 131072
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable
# algorithmic bytes per blob, per kernel
ALG_BYTES = {
    "k_blob_challenge": BYTES_PER_BLOB + 48 + 32,   # blob + commitment read, z written
    "k_blob_evaluate": BYTES_PER_BLOB + 32 + 32,    # blob + z read, y written
    "k_g1_decode_multiples": 2 * (48 + 96 + 4 + 4 * 128 + 192),  # two points per blob: compressed in; affine, flag, 4 affine table rows, 2^64 P out
    "k_msm_window": 3 * 128,                        # three (point, scalar) terms per blob, 96 + 32 B each
}
PATH_ALG_BYTES = BYTES_PER_BLOB + 48 + 48 + 64      # full verify, per blob: blob + commitment + proof read, z and y written
PMC_NAME = {"k_blob_challenge": "kzg::k_blob_challenge_t<4>", "k_blob_evaluate": "kzg::k_blob_evaluate_t<true>",
            "k_g1_decode_multiples": "kzg::k_g1_decode_multiples29<4, true>", "k_msm_window": "kzg::k_msm_window<kzg::Curve29Aff, true>",
            "k_slp_run(pairing)": "kzg::k_slp_run<false>"}
# cycles per wave-instruction a saturated SIMD issues the kernel's instruction mix at (builder's microbenchmarks:
# profiles/r1_issuebench_valu_issue_cost.txt - v_mad_u64_u32 / carry-chain code 4.2; profiles/r1_shabench_sha256_compress.txt - 3.9)
CEIL = {"k_blob_challenge": 3.9, "k_blob_evaluate": 4.2, "k_g1_decode_multiples": 4.2, "k_msm_window": 4.2}
STAMPED = ("k_blob_challenge", "k_blob_evaluate", "k_g1_decode_multiples", "k_msm_window")
PATH_KERNELS = list(PMC_NAME.values()) + ["kzg::k_msm_combine", "kzg::k_msm_combine_lanes", "kzg::k_batch_scalars", "kzg::k_glv_split", "kzg::k_mult_to_affine29",
                                          "kzg::k_eval_powers", "kzg::k_eval_finish", "kzg::k_msm_reduce<false>", "kzg::k_msm_reduce<true>"]
PMC_FILES = ("r6_pmc.json", "r5_pmc.json", "r4_pmc.json", "r3_pmc.json")  # newest first; the first that exists is used


def load_pmc(root, names=PMC_FILES):
    for name in names:
        try:
            return name, json.load(open(os.path.join(root, "profiles", name)))
        except Exception:
            continue
    return None, None


def _r(x, nd=4):
    return round(x, nd) if x is not None else None


def assemble(m, pmc_file, pmc, kernel_key):
    """m: the run's measurement (see bench.py `measurement`): n, G, K, F, elapsed_s, shader_mhz, in_flight_ms{}, stamp_cnt, stamp_sum_ms{},
    other_stamp_sum_ms{}, other_stamp_cnt, solo_stamps_ms{} | None, standalone_event_ms{} | None, challenge_event_population{}.
    Returns {"roofline", "path", "valu", "kernel_ms_standalone", "kernel_ms_in_flight"}."""
    n, G, K, F = m["n"], m["G"], m["K"], m["F"]
    units = n * G
    elapsed = m["elapsed_s"]
    blobs_per_s = units * K / elapsed            # per GPU
    ms_per_step = elapsed / K * 1e3
    shader_mhz = m.get("shader_mhz")
    clock_hz = (shader_mhz or 2400.0) * 1e6
    prof = (pmc or {}).get("kernels", {})
    pmc_units = (pmc or {}).get("blobs_per_launch")
    scale = units / pmc_units if pmc_units else 1.0
    pmc_key = (pmc or {}).get("kernel_key")
    if not pmc:
        stale = "no PMC profile under profiles/"
    elif not pmc_key:
        stale = "profiles/%s carries no kernel_key stamp (collected before round 6): cannot tell which build it describes" % pmc_file
    elif kernel_key and pmc_key != kernel_key:
        stale = "profiles/%s was collected on kernel key %s, this tree is %s: counters NOT used" % (pmc_file, pmc_key, kernel_key)
    else:
        stale = None
    use_pmc = stale is None
    solo = m.get("solo_stamps_ms") or {}
    ev_sa = m.get("standalone_event_ms") or {}
    in_flight = m.get("in_flight_ms") or {}

    def standalone_of(k):  # the stand-alone figure of a kernel and where it comes from
        if solo.get(k):
            return solo[k], "in-kernel stamps, one launch group alone on a single-stream handle"
        if ev_sa.get(k):
            return ev_sa[k], "HIP events around the kernel on a single-stream handle, one launch group alone on the chip"
        return None, None

    # ---- rows: every big kernel of the path
    rows, mix_num, mix_den = [], 0.0, 0.0
    n_all = m.get("stamp_cnt", 0) + m.get("other_stamp_cnt", 0)
    for k in STAMPED:
        sa, sa_src = standalone_of(k)
        fl = in_flight.get(k) or None
        pk = prof.get(PMC_NAME[k], {}) if use_pmc else {}
        insts = pk.get("SQ_INSTS_VALU")
        all_ms = ((m.get("stamp_sum_ms") or {}).get(k, 0.0) + (m.get("other_stamp_sum_ms") or {}).get(k, 0.0)) / n_all if n_all else None
        alg = ALG_BYTES[k] * units
        rows.append({
            "kernel": k, "units_per_launch": units, "algorithmic_bytes_per_launch": alg,
            "standalone_ms": _r(sa), "standalone_ms_source": sa_src, "in_flight_ms": _r(fl),
            # every launch of this size in the process (warm-up, timed, stand-alone and self-check groups): the population behind the
            # kernel's launches of this grid in a rocprofv3 kernel trace of this command (profiles/<round>_kernel_trace_by_grid.json)
            "all_launches_ms": _r(all_ms) if all_ms else None, "launches": n_all,
            "achieved_standalone_GBps": _r(alg / sa / 1e6, 2) if sa else None, "frac_standalone": _r(alg / sa / 1e6 / HBM_PEAK_GBS, 6) if sa else None,
            "achieved_in_flight_GBps": _r(alg / fl / 1e6, 2) if fl else None, "frac_in_flight": _r(alg / fl / 1e6 / HBM_PEAK_GBS, 6) if fl else None,
            "hbm_traffic_bytes": round(pk["hbm_bytes_corrected"] * scale) if "hbm_bytes_corrected" in pk else None,
            "hbm_traffic_ratio": _r(pk["hbm_bytes_corrected"] * scale / alg, 3) if "hbm_bytes_corrected" in pk else None,
            "valu_wave_insts_per_launch": round(insts * scale) if insts else None,
            "cycles_per_inst_standalone": _r(sa * 1e-3 * clock_hz * 1024 / (insts * scale), 3) if sa and insts else None,
            "issue_ceiling_cycles_per_inst": CEIL[k]})
        if insts:
            mix_num += insts * CEIL[k]
            mix_den += insts
    for k, pname, ev_ms in (("k_msm_reduce", "kzg::k_msm_reduce<false>", None), ("k_slp_run(pairing)", PMC_NAME["k_slp_run(pairing)"], ev_sa.get("k_slp_run(pairing)"))):
        pk = prof.get(pname, {}) if use_pmc else {}
        ms = ev_ms or pk.get("ms_single_stream")
        insts = pk.get("SQ_INSTS_VALU")
        rows.append({"kernel": k, "units_per_launch": units, "algorithmic_bytes_per_launch": 0, "standalone_ms": _r(ms),
                     "standalone_ms_source": "HIP events on a single-stream handle, this run" if ev_ms else ("kernel-trace duration in profiles/%s (not stamped)" % pmc_file if ms else None),
                     "in_flight_ms": None, "valu_wave_insts_per_launch": round(insts * scale) if insts else None,
                     "cycles_per_inst_standalone": _r(ms * 1e-3 * clock_hz * 1024 / (insts * scale), 3) if ms and insts else None,
                     "issue_ceiling_cycles_per_inst": 4.2,
                     "note": "reads the bucket sums k_msm_window wrote (no input bytes of its own): latency chain of ~28 point additions per window slot"
                     if k == "k_msm_reduce" else "one wavefront per pairing instance, LDS-resident straight-line program: dependency-depth bound"})
        if insts:
            mix_num += insts * 4.2
            mix_den += insts
    mix_ceiling = mix_num / mix_den if mix_den else None
    by_name = {r["kernel"]: r for r in rows}

    # ---- the dominant kernel: largest stand-alone duration; without a stand-alone group, largest in-flight interval
    have_sa = [k for k in STAMPED if by_name[k]["standalone_ms"]]
    if have_sa:
        dom = max(have_sa, key=lambda k: by_name[k]["standalone_ms"])
        dom_rule = "largest stand-alone duration in this run (one launch group alone on a single-stream handle)"
    else:
        cands = [k for k in STAMPED if in_flight.get(k)]
        dom = max(cands, key=lambda k: in_flight[k]) if cands else "k_blob_challenge"
        dom_rule = "largest in-flight interval (this run measured no stand-alone group: --no-self-check)"
    drow = by_name[dom]
    alg_dom = ALG_BYTES[dom] * units
    launch_ms = in_flight.get(dom) or drow["standalone_ms"] or 0.0
    achieved = alg_dom / (launch_ms * 1e-3) / 1e9 if launch_ms else 0.0
    sa_dom = drow["standalone_ms"]
    sa_achieved = alg_dom / (sa_dom * 1e-3) / 1e9 if sa_dom else None
    fl_cands = [k for k in STAMPED if in_flight.get(k)]
    dom_fl = max(fl_cands, key=lambda k: in_flight[k]) if fl_cands else None

    # ---- whole path
    path_gbps = PATH_ALG_BYTES * blobs_per_s / 1e9
    valu, path_ratio = None, None
    if use_pmc and pmc_units:
        insts = sum(prof[k].get("SQ_INSTS_VALU", 0) for k in PATH_KERNELS if k in prof)
        per_blob = insts / pmc_units  # wave-instructions per blob, all kernels of the path
        if per_blob:
            simds, nominal = 1024, 2.4e9
            valu = {"wave_insts_per_blob": round(per_blob), "insts_per_cycle_per_simd": _r(per_blob * blobs_per_s / (simds * nominal)),
                    "cycles_per_inst": _r((simds * nominal) / (per_blob * blobs_per_s), 3),
                    "shader_clock_mhz_measured": _r(shader_mhz, 1) if shader_mhz else None,
                    "cycles_per_inst_at_measured_clock": _r((simds * shader_mhz * 1e6) / (per_blob * blobs_per_s), 3) if shader_mhz else None,
                    "ceiling_insts_per_cycle_per_simd": {"multiply_add_kernels (evaluate, decode, MSM window, MSM reduce)": 0.238, "sha256 (k_blob_challenge)": 0.256},
                    "note": "insts_per_cycle_per_simd / cycles_per_inst assume the nominal 2.4 GHz; shader_clock_mhz_measured is what the SIMDs ran at during "
                            "the timed region (s_memtime / s_memrealtime stamped by every wave of k_blob_challenge) and cycles_per_inst_at_measured_clock "
                            "the figure to hold against the ceilings.  SQ_INSTS_VALU of every kernel of the path from profiles/%s.  Ceilings per kernel class, "
                            "from the builder's microbenchmarks: v_mad_u64_u32 / carry-chain code 4.2 cycles per wave-instruction on a saturated SIMD = 0.238 "
                            "(profiles/r1_issuebench_valu_issue_cost.txt; with the 2-4 wavefronts per SIMD their register budgets allow: 5.5 / 5.15 / 4.9 cycles, "
                            "profiles/r3_depbench_mad_issue_vs_occupancy.txt); SHA-256 mixes 4.2-cycle rotates with 2.4-cycle logic at 3.9 cycles = 0.256 "
                            "(profiles/r1_shabench_sha256_compress.txt)" % pmc_file}
        path_bytes = sum(prof[k].get("hbm_bytes_corrected", 0) for k in PATH_KERNELS if k in prof)
        path_ratio = _r(path_bytes / (PATH_ALG_BYTES * pmc_units), 3)
    cpi = valu.get("cycles_per_inst_at_measured_clock") if valu else None
    frac_binding = _r(mix_ceiling / cpi) if mix_ceiling and cpi else None

    traffic = drow["hbm_traffic_bytes"]
    if use_pmc and traffic is not None:
        same = pmc_units == units
        traffic_source = ("profiles/%s (kernel key %s = this tree's): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 gfx950 correction, "
                          "collected on a launch of %d blobs - %s" % (pmc_file, pmc_key, pmc_units, "the same launch size as this run, not re-measured by it" if same else
                                                                      "EXTRAPOLATED linearly to this run's %d blobs per launch" % units))
    else:
        traffic_source = None
    pop = m.get("challenge_event_population") or {}
    roofline = {
        "bound": "valu-issue",
        "bound_note": "the bound that binds is VALU issue (wide-integer modular arithmetic and SHA-256: ~106 k wave-instructions per blob against 131 KB of input): "
                      "frac_of_binding_bound; achieved / peak / frac are the HBM figures BASELINE.json's metric asks for, for the dominant kernel; frac_path the same for "
                      "the whole path; kernels[] carries every big kernel with its cycles per instruction against the issue ceiling of its instruction mix",
        "kernel": dom, "kernel_chosen_by": dom_rule, "units_per_launch": units, "algorithmic_bytes_per_launch": alg_dom,
        "launch_ms": _r(launch_ms), "achieved": _r(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _r(achieved / HBM_PEAK_GBS, 6),
        "standalone_ms": sa_dom, "achieved_standalone": _r(sa_achieved, 3) if sa_achieved else None,
        "frac_standalone": _r(sa_achieved / HBM_PEAK_GBS, 6) if sa_achieved else None,
        "frac_path": _r(path_gbps / HBM_PEAK_GBS, 6), "frac_of_binding_bound": frac_binding,
        "dominant_in_flight": {"kernel": dom_fl, "in_flight_ms": _r(in_flight[dom_fl]), "frac_in_flight": by_name[dom_fl]["frac_in_flight"],
                               "note": "the kernel with the largest in-flight interval - the other defensible 'dominant': its interval is residency beside %d "
                                       "groups in flight and may exceed ms_per_step (%.2f ms)" % (F, ms_per_step)} if dom_fl else None,
        "traffic": traffic if use_pmc else None, "traffic_source": traffic_source, "traffic_stale": stale,
        "pmc_file": pmc_file, "pmc_kernel_key": pmc_key, "kernel_key": kernel_key,
        "launch_ms_all_launches": _r(pop.get("all_launches_ms")), "launches": pop.get("launches"),
        "launch_ms_incl_warmup": _r(pop.get("incl_warmup_ms")),
        "launch_ms_source": "in-kernel stamps (s_memrealtime: first wavefront in, last wavefront out) averaged over the %d launch groups of the timed region, %d in flight "
                            "(residency, not cost); launch_ms_all_launches: k_blob_challenge's interval over every launch of this size in the process - the population "
                            "rocprofv3 --kernel-trace --stats of this command averages into AverageNs; standalone_ms: the same stamp, one launch group alone on the chip"
                            % (m.get("stamp_cnt", 0), F),
        "kernels": rows,
        "kernels_note": "standalone_ms: the kernel's own interval in one launch group alone on a single-stream handle (cost) | in_flight_ms: averaged over the %d launch "
                        "groups of the timed region with %d in flight (RESIDENCY: the kernel shares the chip, so an in-flight interval may exceed ms_per_step = %.2f ms and the "
                        "in-flight intervals of one step add up to more than the step).  cycles_per_inst_standalone = standalone_ms x the measured shader clock x 1 024 SIMDs / "
                        "SQ_INSTS_VALU (profiles/%s), to hold against issue_ceiling_cycles_per_inst" % (m.get("stamp_cnt", 0), F, ms_per_step, pmc_file),
        "inputs": m,
        "note": "path is integer-ALU / latency bound, not HBM bound (DESIGN.md 4)"}
    path = {"algorithmic_bytes_per_blob": PATH_ALG_BYTES, "algorithmic_GBps": _r(path_gbps, 2), "frac": _r(path_gbps / HBM_PEAK_GBS, 6),
            "hbm_traffic_ratio": path_ratio, "valu_mix_ceiling_cycles_per_inst": _r(mix_ceiling, 3) if mix_ceiling else None,
            "valu_frac_of_mix_ceiling": frac_binding,
            "note": "whole path per GPU: 131 232 algorithmic bytes per blob x blobs/s; hbm_traffic_ratio = PMC HBM bytes of every kernel of the path / algorithmic bytes "
                    "(profiles/%s; the blob is streamed twice: hash, then evaluate - 2.2x is what that costs, accepted while the path is issue-bound)" % pmc_file}
    return {"roofline": roofline, "path": path, "valu": valu,
            "kernel_ms_standalone": {k: _r(v) for k, v in ev_sa.items()} if ev_sa else None,
            "kernel_ms_in_flight": dict({k: _r(v) for k, v in in_flight.items()},
                                        note="in-kernel stamps averaged over the timed region's launch groups, %d in flight: residency, not cost" % F)}
