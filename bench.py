#!/usr/bin/env python3
"""Benchmark of the north-star path: verify_blob_kzg_proof_batch on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|configs1|config5] [--blobs n] [--group G] [--inflight 4]
                    [--force-collectives] [--no-shard-leg] [--no-configs] [--no-concurrent] [--no-latency] [--no-self-check] [--no-cpu-baseline]

One "step" = one LAUNCH GROUP: `--group` (256) independent verify_blob_kzg_proof_batch calls
(src/kzg_proof.rs:472-525) of `--blobs` (1 024, BASELINE.json configs[1]) synthetic blobs each, all already resident
in HBM - 262 144 blobs, 32 GiB of blob bytes per step and GPU.  Every batch of the group is a complete, independent
verification (its own transcript, challenge r, MSMs and pairing; its own boolean).  At n = 1 024 each phase of ONE
batch is a latency-bound serial chain that occupies a sliver of the chip, so the batch dimension lives inside every
kernel, and `--inflight` groups are kept in flight by a fixed-order software pipeline on one host thread.
The latency of ONE batch at a time (the reference's call shape) is measured too and reported as `single_batch`
(device-resident) and `end_to_end` (host `Vec<Blob>` layout through kzg_verify_blob_kzg_proof_batch, PCIe included).

With N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) the default workload is BASELINE.json
configs[4]: every batch is 32 768 x N blobs sharded by blob (at N = 8: 262 144 blobs over 8 GPUs), 8 batches per step -
the same 262 144 blobs = 32 GiB per GPU and step as at N = 1, so scaling is "weak".  Each batch's 160 n-byte transcript
is hashed once (the batches of a step are dealt to the ranks), r travels as 32 bytes, and the "G1 all-reduce" is an
all-gather of 288-byte partial sums folded on every rank (kzg_rs_amd/distributed.py).  `--workload configs1` keeps
1 024 blobs per GPU and batch at any N; `--workload config5` (= `--config5`) runs the 32 768-blob shard shape on one GPU.
Rank 0 prints ONE JSON line.

`python3 bench.py --gpus N` from a bare command (no WORLD_SIZE in the environment) starts the N ranks itself, as child
processes, BEFORE anything in this process has touched the GPU, and exits with their worst return code.

The default one-GPU run also measures BASELINE configs[4]'s SHARD SHAPE on that one GPU (`configs.config5_shard`: 8 batches x 32 768 blobs
per step through the very code path `--gpus N` runs - `--workload config5 --force-collectives` in a child process, a world of one rank over
RCCL - with its stage split and a poisoned batch), so that a later `--gpus N` curve has a like-for-like N = 1 point
(`multi_gpu.efficiency_vs`); a pre-flight sizes the variants against free HBM and shrinks the launch group instead of dying.  `configs.config3`
/ `configs.config4`: BASELINE configs[2] (evaluation only) and configs[3] (2^20-term MSM over the setup's points, kzg_g1_msm_setup);
`end_to_end.large_calls`: ONE call with a Vec<Blob> of 8 192 / 32 768 blobs.  The `roofline` / `path` / `valu` blocks are assembled by
kzg_rs_amd/benchline.py (its docstring states the rules; `roofline.inputs` carries the measurement they follow from).

Also in the line: `self_check` (the negative control THROUGH THE BENCHMARKED ENTRY POINT: five launch groups of the full
shape through kzg_verify_blob_kzg_proof_batch_groups_device with the benchmark's groups in flight, a wrong proof, a field
element equal to r and a commitment outside G1 poisoned into batches of DIFFERENT groups and lanes - the call must say
false / Err / Err for exactly those), `roofline.standalone_ms` (the dominant kernel alone on the chip: one launch group on a
single-stream handle, nothing else in flight), `path` (whole-path algorithmic bandwidth and the HBM traffic ratio),
`verify_kzg_proof_ms`, and at N > 1 `multi_gpu.single_process` (ONE process driving all N GPUs through a multi-device
settings handle: launch groups routed to the devices that own them, one sharded batch at a time, and a stream of sharded
batches in flight).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm gives a process 4 hardware queues by default; the pipeline uses 2 streams per in-flight group
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from kzg_rs_amd.benchline import BYTES_PER_BLOB, HBM_PEAK_GBS  # noqa: E402  (per-kernel algorithmic bytes, PMC names and the line's assembly live there)


def cpu_baseline(blobs, cs, ps, tau_g2, max_blobs):
    """The CPU oracle (a C restatement of the reference's operation sequence, oracle/kzg.c) timed on
    the host cores of this box on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    ost = O.Settings.from_tau_g2(tau_g2)
    m = min(len(cs), max_blobs)
    bl = [blobs[i].tobytes() for i in range(m)]
    t = time.perf_counter()
    ok, zs, ys = O.verify_blob_kzg_proof_batch_ex(bl, cs[:m], ps[:m], ost, nthreads=1)[:3]
    dt1 = time.perf_counter() - t
    ncores = os.cpu_count() or 1
    t = time.perf_counter()
    ok2 = O.verify_blob_kzg_proof_batch(bl, cs[:m], ps[:m], ost, nthreads=ncores)
    dtn = time.perf_counter() - t
    # the part no thread count shortens: decode of 2 m points, r, the m scalar multiplications G y_i, three MSMs, the pairing
    # (src/kzg_proof.rs:399-444, single-threaded in the reference and in the port)
    t = time.perf_counter()
    ok3 = O.verify_kzg_proof_batch(cs[:m], [zs[32 * i: 32 * i + 32] for i in range(m)], [ys[32 * i: 32 * i + 32] for i in range(m)], ps[:m], ost)
    dtail = time.perf_counter() - t
    assert ok and ok2 and ok3, "oracle rejects the synthetic batch"
    # The honest whole-host figure: the reference has no threads (src/kzg_proof.rs:251-277, :399-444 are plain loops), so an
    # operator fills a host with INDEPENDENT verifications.  oracle/bench_threads.c: T pthreads (no interpreter, no ctypes in the
    # loop), each making single-threaded verify_blob_kzg_proof_batch calls of its own 64-blob batch for ~2.5 s, at T = 1 / 8 /
    # 64 / all cores, with the per-thread rate at each T - a baseline that can be explained: per-thread rates that fall with T
    # measure the host (shared caches, SMT siblings, clocks), not the port (its per-blob temporaries are per-thread scratch
    # since round 5; rounds 1-4 malloc'ed 0.5 MB per blob, i.e. mmap + munmap + page faults under the process's address-space lock).
    mb = min(64, m)
    m64 = m // mb * mb
    raw = (b"".join(bl[:m64]), b"".join(cs[:m64]), b"".join(ps[:m64]))
    ladder = []
    quota, cg0 = cgroup_cpu()
    qt = int(quota) if quota and quota >= 1 else None
    for T in sorted({1, 8, 64, ncores} | ({qt} if qt else set())):
        if T > ncores:
            continue
        r = O.bench_threads("blobs", T, 2.5, ost, raw[1], raw[2], blobs=raw[0], per_call=mb)
        assert r["bad"] == 0, "oracle rejects a synthetic sub-batch"
        ladder.append({"threads": T, "blobs_per_s": round(r["calls_per_s"] * mb, 1), "blobs_per_s_per_thread": round(r["calls_per_s"] * mb / T, 1),
                       "slowest_thread": round(r["per_thread_min"] * mb, 1), "fastest_thread": round(r["per_thread_max"] * mb, 1)})
    top = max(ladder, key=lambda x: x["blobs_per_s"])
    nthr = top["threads"]
    return {
        "all_cores_independent": {"value": top["blobs_per_s"], "unit": "blobs/s", "cores": nthr, "host_cores": ncores, "host_cpu_quota_cores": quota,
                                  "throttled_periods_during_ladder": (cgroup_cpu()[1].get("nr_throttled", 0) - cg0.get("nr_throttled", 0)) if cg0 else None,
                                  "explanation": "per-thread rate is flat up to the cgroup's CPU quota (cpu.max: %s cores on this %d-thread host) and falls beyond it: "
                                                 "T threads then share the quota's core-seconds (and are throttled in 100 ms periods).  Rounds 1-4 reported 104 "
                                                 "blobs/s per thread at T = 64 against 489 alone and could not say why: it is this quota (64 threads on 16 cores' "
                                                 "worth of time = a quarter each), not the port" % (quota, ncores),
                                  "threads_ladder": ladder,
                                  "sample": "oracle/bench_threads.c: T pthreads, each making single-threaded verify_blob_kzg_proof_batch calls of its own "
                                            "%d-blob batch for 2.5 s, T = %s; value = the best aggregate (T = %d) - the process-level parallelism an "
                                            "operator of the single-threaded reference would use" % (mb, " / ".join(str(x["threads"]) for x in ladder), nthr)},
        "value": round(m / dt1, 2), "unit": "blobs/s", "cores": 1, "kind": "port",
        "sample": "%d of the same synthetic blobs, one verify_blob_kzg_proof_batch call, 1 thread (the reference is single-threaded): "
                  "%.2f s, of which per-blob phase (challenge + evaluation, src/kzg_proof.rs:251-277) %.2f s and random linear combination + "
                  "pairing (:399-444) %.2f s.  (ONE call with its per-blob loop threaded over all %d host cores: %.1f blobs/s - the %.2f s of the "
                  "second phase stay serial, Amdahl limit %.0f blobs/s; the whole-host figure is all_cores_independent)"
                  % (m, dt1, max(dt1 - dtail, 0.0), dtail, ncores, m / dtn, dtail, m / dtail),
        "phases_s": {"per_blob": round(max(dt1 - dtail, 0.0), 3), "rlc_and_pairing": round(dtail, 3), "all_cores_call": round(dtn, 3)},
    }


def cgroup_cpu():
    """(cpu quota of this process's cgroup in cores or None, {usage_usec, nr_throttled, throttled_usec}) - cgroup v2 files; the GPU boxes
    of this project give a job 16 cores' worth of quota on a 256-thread host, which is what bounds any T > 16 host threads."""
    quota, stat = None, {}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            if k in ("usage_usec", "nr_throttled", "throttled_usec"):
                stat[k] = int(v)
    except Exception:
        pass
    return quota, stat


def concurrent_callers(settings, blobs, cs, ps, synth, qc, qz, qy, qp, no_cpu=False, seconds=1.2):
    """ONE shared settings handle, T host threads, one small call each - the reference's &KzgSettings semantics
    (src/trusted_setup.rs:44-50,80-92) under its named workload (the revm precompile's verify_kzg_proof; a beacon node's 6-blob
    verify_blob_kzg_proof_batch).  The threads are std::threads INSIDE the library (kzg_debug_concurrent_callers: the public
    entry points, no interpreter lock) and every answer is checked against an expected table that holds true and false cases.
    Beside each figure: the CPU oracle under the same thread count (oracle/bench_threads.c, T pthreads of single-threaded calls)."""
    n = 256
    c_, z_, p_ = b"".join(qc[:n]), b"".join(qz[:n]), b"".join(qp[:n])
    ys = list(qy[:n])
    exp = bytearray([1] * n)
    for i in range(0, n, 16):   # every 16th claim is wrong: exactly those verdicts must be False
        ys[i] = ys[(i + 1) % n] if ys[(i + 1) % n] != ys[i] else ys[(i + 2) % n]
        exp[i] = 0
    exp[7] = 0 if qy[7] == qy[8] and 7 % 16 else exp[7]  # (the caller's list already carries one wrong claim at index 7)
    nb = 48
    bl_raw = blobs[:nb].tobytes()
    bc, bp = b"".join(cs[:nb]), list(ps[:nb])
    bp[6 * 3 + 2], bp[6 * 3 + 3] = bp[6 * 3 + 3], bp[6 * 3 + 2]   # call 3: two proofs swapped -> false
    exp6 = bytearray([1] * (nb // 6))
    exp6[3] = 0
    ost = None
    if not no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ost = O.Settings.from_tau_g2(synth.synthetic_setup()[1])
        ok_c, ok_z, ok_y, ok_p = (b"".join(x[8:n]) for x in (qc, qz, qy, qp))  # valid tuples only (index 7 is the caller's wrong claim)
    ncores = os.cpu_count() or 1
    quota = cgroup_cpu()[0]

    def host_use(before, after, wall):  # what the host side cost, and whether the cgroup's CPU quota throttled the process meanwhile
        if not before or not after:
            return None
        return {"cores_used": round((after.get("usage_usec", 0) - before.get("usage_usec", 0)) / 1e6 / wall, 2),
                "throttled_periods": after.get("nr_throttled", 0) - before.get("nr_throttled", 0),
                "throttled_ms": round((after.get("throttled_usec", 0) - before.get("throttled_usec", 0)) / 1e3, 1)}

    out = {"handle": "one shared KzgSettings handle; small-call queue of csrc/capi_coalesce.hpp", "seconds_per_point": seconds, "host_cores": ncores,
           "callers": "T std::threads inside the library (kzg_debug_concurrent_callers), closed loop: each calls the public entry point again as soon as its answer is "
                      "back and checked; the threads wait for the start asleep and make their first call at their own moment within 2.5 ms (independent callers do not "
                      "arrive in lock-step; released together they would travel as one cohort with the second lane idle: 86-90 k instead of 94-108 k calls/s at T = 256)",
           "host_cpu_quota_cores": quota,
           "verify_kzg_proof": [], "verify_blob_kzg_proof_batch_6_host_blobs": []}
    settings.concurrent_callers("proof", 8, 0.3, c_, p_, bytes(exp), z=z_, y=b"".join(ys))  # lanes and their workspaces
    for T in (1, 8, 64, 256):
        settings.small_queue_stats(reset=True)
        c0 = cgroup_cpu()[1]
        r = settings.concurrent_callers("proof", T, seconds, c_, p_, bytes(exp), z=z_, y=b"".join(ys))
        c1 = cgroup_cpu()[1]
        q = settings.small_queue_stats()
        row = {"threads": T, "calls_per_s": round(r["calls_per_s"], 1), "mean_ms": round(r["mean_ms"], 3), "max_ms": round(r["max_ms"], 2), "wrong_answers": r["wrong"],
               "launches": q["launches"], "calls_per_launch": round(q["items"] / max(1, q["launches"]), 1), "host": host_use(c0, c1, r["seconds"])}
        if ost is not None and T <= ncores:
            o = O.bench_threads("proof", T, seconds, ost, ok_c, ok_p, zs=ok_z, ys=ok_y)
            row["cpu_oracle_calls_per_s"] = round(o["calls_per_s"], 1)
            row["cpu_oracle_per_thread"] = round(o["calls_per_s"] / T, 1)
        out["verify_kzg_proof"].append(row)
    settings.concurrent_callers("blobs", 8, 0.3, bc, b"".join(bp), bytes(exp6), blobs=bl_raw, per_call=6)
    for T in (1, 8, 64):
        settings.small_queue_stats(reset=True)
        c0 = cgroup_cpu()[1]
        r = settings.concurrent_callers("blobs", T, seconds, bc, b"".join(bp), bytes(exp6), blobs=bl_raw, per_call=6)
        c1 = cgroup_cpu()[1]
        q = settings.small_queue_stats()
        row = {"threads": T, "batches_per_s": round(r["calls_per_s"], 1), "blobs_per_s": round(6 * r["calls_per_s"], 1), "mean_ms": round(r["mean_ms"], 3),
               "max_ms": round(r["max_ms"], 2), "wrong_answers": r["wrong"], "launches": q["launches"], "blobs_per_launch": round(q["items"] / max(1, q["launches"]), 1),
               "host": host_use(c0, c1, r["seconds"])}
        if ost is not None and T <= ncores:
            o = O.bench_threads("blobs", T, seconds, ost, bc, b"".join(ps[:nb]), blobs=bl_raw, per_call=6)
            row["cpu_oracle_batches_per_s"] = round(o["calls_per_s"], 1)
        out["verify_blob_kzg_proof_batch_6_host_blobs"].append(row)
    out["all_answers_correct"] = all(r["wrong_answers"] == 0 for k in ("verify_kzg_proof", "verify_blob_kzg_proof_batch_6_host_blobs") for r in out[k])
    return out


def config_legs(settings, torch, dev, no_cpu=False):
    """BASELINE.json configs[2] and configs[3] as measured legs of the default run (they are parity-tested at full size in
    tests/test_gpu_baseline_sizes.py; here they get a number the driver's line carries), each with a result check.
      config3: evaluate_polynomial_in_evaluation_form only (src/kzg_proof.rs:94-133), 16 384 device-resident blobs (2 GiB) through
               kzg_evaluate_polynomials_device.  Check: z = a root of unity for 32 of the blobs -> y must be the blob's own element
               (the :104-108 early return), and blobs [8192, 16384) repeat [0, 8192) -> the halves agree bit for bit.
      config4: G1 msm_variable_base (call sites :419,429,430) over 2^20 (point, scalar) pairs - the 4 096 Lagrange points of the
               mainnet setup tiled 256 times x uniform random scalars - through kzg_g1_msm.  Check: by linearity the result equals
               the 4 096-term MSM over the distinct points with each point's 256 scalars summed mod r, computed by the SAME entry
               point in its one-slice shape (and by the CPU oracle when the CPU baseline is on)."""
    import ctypes as C
    import random

    import numpy as np

    from kzg_rs_amd import api
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    out = {}
    # ---- config 3
    n, half = 16384, 8192
    g = torch.Generator(device=dev).manual_seed(3)
    d_blobs = torch.empty((n, BYTES_PER_BLOB), dtype=torch.uint8, device=dev)
    d_blobs[:half] = torch.randint(0, 256, (half, BYTES_PER_BLOB), dtype=torch.uint8, device=dev, generator=g)
    d_blobs[:half, 0::32] &= 0x3F
    d_blobs[half:] = d_blobs[:half]
    rng = random.Random(33)
    zs = [rng.randrange(R) for _ in range(half)]
    roots = {}
    for i in range(0, half, 256):
        k = rng.randrange(4096)
        roots[i] = k
        zs[i] = int.from_bytes(settings.root_of_unity(k), "big")
    z_le = np.frombuffer(b"".join(z.to_bytes(32, "little") for z in zs + zs), dtype=np.uint8).copy()
    d_z = torch.from_numpy(z_le).to(dev)
    d_y = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # What this 1 ms VALU-bound launch costs depends on what the chip did in the milliseconds before it (profiles/r6_config3_regimes.json):
    #   alone - 20 ms of idleness before the call:           0.911-0.924 ms, 12 calls, spread 1.4 %   -> `ms` (what one such call costs a caller)
    #   back to back: 0.93, 0.98, then 1.1-1.25 ms for ~5 calls, decaying to 0.92 over ~20 calls       -> `ms_back_to_back_*` (a power / clock transient:
    #   the same happens behind a memory-bound torch kernel, and not at all behind a pause)
    # It is the state of the chip, not the launch: the time per blob is the same at 4.0, 5.0 and 5.33 rounds of resident wavefronts
    # (12 288 / 15 360 / 16 384 blobs: 54.7 / 54.6 / 54.0 ns per blob), so there is no quantisation tail for a persistent grid to remove.
    api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)   # warm-up (scratch allocation)
    ms3 = []
    for _ in range(11):
        time.sleep(0.02)
        api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
        ms3.append(settings.last_timings()[4])
    b2b = []
    for _ in range(32):
        api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
        b2b.append(settings.last_timings()[4])
    y = d_y.cpu().numpy().reshape(n, 32)
    ok3 = bool((y[:half] == y[half:]).all())
    for i, k in roots.items():
        ok3 = ok3 and y[i].tobytes()[::-1] == d_blobs[i, 32 * k: 32 * k + 32].cpu().numpy().tobytes()
    t3 = sorted(ms3[1:])[len(ms3[1:]) // 2]
    alg3 = BYTES_PER_BLOB + 64
    out["config3"] = {"workload": "evaluate_polynomial_in_evaluation_form only, %d device-resident blobs (BASELINE.json configs[2])" % n,
                      "entry_point": "kzg_evaluate_polynomials_device", "blobs": n, "ms": round(t3, 4), "ms_all_runs": [round(x, 4) for x in ms3],
                      "spread": round((max(ms3[1:]) - min(ms3[1:])) / t3, 4),
                      "ms_back_to_back_all_runs": [round(x, 4) for x in b2b], "ms_back_to_back_steady": round(sorted(b2b[-8:])[4], 4),
                      "ms_back_to_back_worst": round(max(b2b), 4),
                      "blobs_per_s": round(n / t3 * 1e3, 1),
                      "roofline": {"bound": "valu-issue", "algorithmic_bytes_per_blob": alg3, "achieved": round(alg3 * n / t3 / 1e6, 2), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(alg3 * n / t3 / 1e6 / HBM_PEAK_GBS, 6)},
                      "checked": {"passed": ok3, "what": "%d blobs evaluated at a root of unity return their own element; the two identical halves agree bit for bit"
                                                         % len(roots)},
                      "timing": "HIP events on the library's stream around k_eval_powers + k_blob_evaluate + k_eval_finish; ms = median of 10 calls, each after 20 ms of "
                                "idleness (one warm-up before); ms_back_to_back_*: 32 calls one behind the other - the first calls of such a burst run into a power / clock "
                                "transient (worst), the last 8 are the steady state (see profiles/r6_config3_regimes.json: regimes and a size sweep)"}
    del d_blobs, d_z, d_y
    torch.cuda.empty_cache()
    # ---- config 4
    ts = open(os.path.join(ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    brp = lambda i: int(format(i, "012b")[::-1], 2)
    base = b"".join(bytes.fromhex(ts[2 + brp(i)]) for i in range(4096))   # g1_points as the handle keeps them (bit-reversal permuted, build.rs:79)
    reps = 256
    n4 = 4096 * reps
    sc = np.random.Generator(np.random.PCG64(4)).integers(0, 256, size=(n4, 32), dtype=np.uint8)
    sc[:, 0] &= 0x7F   # some scalars land in [r, 2^255): the entry points reduce them mod r like Scalar::from_raw
    L = api.lib()
    o48 = C.create_string_buffer(48)
    bench_settings = settings
    settings = api.KzgSettings.load_trusted_setup_file()   # the mainnet setup (the benchmark's own handle is the known-tau test setup: no G1 section)
    # (a) the setup form: the handle's own points, no per-call decode (kzg_g1_msm_setup)
    ms4, wall4 = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        api._chk(L.kzg_g1_msm_setup(o48, sc.ctypes.data_as(C.c_char_p), n4, settings._h))
        wall4.append((time.perf_counter() - t0) * 1e3)
        ms4.append(settings.last_timings()[2])
    dec_setup = settings.last_timings()[6]
    got = o48.raw
    # (b) arbitrary points: the same 2^20 pairs as compressed bytes (kzg_g1_msm decodes, subgroup-tests and tabulates them per call)
    pts = base * reps
    msA, decA, wallA = [], [], []
    for _ in range(3):
        t0 = time.perf_counter()
        api._chk(L.kzg_g1_msm(o48, pts, sc.ctypes.data_as(C.c_char_p), n4, settings._h))
        wallA.append((time.perf_counter() - t0) * 1e3)
        tm = settings.last_timings()
        msA.append(tm[2])
        decA.append(tm[6])
    same_forms = o48.raw == got
    # sum of each point's 256 scalars mod r, in Python integers (256 x 4096 additions of 32-byte numbers)
    sums = [0] * 4096
    as_int = [int.from_bytes(sc[i].tobytes(), "big") for i in range(n4)]
    for i, v in enumerate(as_int):
        sums[i & 4095] += v
    small = b"".join((v % R).to_bytes(32, "big") for v in sums)
    api._chk(L.kzg_g1_msm(o48, base, small, 4096, settings._h))
    ok4 = o48.raw == got and same_forms
    how = ("kzg_g1_msm_setup = kzg_g1_msm over the same points as compressed bytes = the 4 096-term MSM over the distinct points with each point's 256 scalars summed mod r")
    if not no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        ok4 = ok4 and O.g1_msm(base, small, 4096) == got
        how += " = the CPU oracle's 4 096-term MSM"
    settings.close()
    settings = bench_settings
    t4 = sorted(ms4[1:])[len(ms4[1:]) // 2]
    w4 = sorted(wall4[1:])[len(wall4[1:]) // 2]
    tA = sorted(msA[1:])[0]
    out["config4"] = {"workload": "G1 msm_variable_base, 2^20 trusted-setup points (the handle's 4 096 Lagrange points, term i -> point i mod 4 096) x random Fr scalars "
                                  "(BASELINE.json configs[3])",
                      "entry_point": "kzg_g1_msm_setup", "pairs": n4, "ms_msm": round(t4, 4), "ms_decode_and_tables": round(dec_setup, 4), "ms_msm_all_runs": [round(x, 4) for x in ms4],
                      "ms_call_wall": round(w4, 3), "ms_call_wall_all_runs": [round(x, 3) for x in wall4], "pairs_per_s": round(n4 / t4 * 1e3, 1), "ns_per_term": round(t4 * 1e6 / n4, 3),
                      "roofline": {"bound": "valu-issue", "algorithmic_bytes_per_pair": 32, "achieved": round(32 * n4 / t4 / 1e6, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(32 * n4 / t4 / 1e6 / HBM_PEAK_GBS, 6),
                                   "note": "algorithmic bytes = the scalars (32 B per term): the points are the handle's, resident as table rows (16 MB).  What the bucket "
                                           "kernel moves beyond them (hbm_traffic_bytes below: ~1.4 GB at 2^20 terms) is 16 row reads of 128 B per term = 2.1 GB of requests against "
                                           "a 16 MB table, of which the ~60 % that miss the 4 MB L2 of their XCD are counted (they are served by the 256 MB MALL, not by HBM: the "
                                           "counter is L2-to-fabric traffic), plus the partitioned entry list written and read once (2 x 67 MB) and the workgroups' bucket sums "
                                           "(70 MB).  The same launch over a 128 KB table (rows of 64 points) takes the same time (profiles/r6_fb_window_timeline.txt): the kernel is "
                                           "issue-bound, the row traffic is not what limits it"},
                      "form": "fixed base (csrc/msm_fixed.hpp): 16 signed 16-bit windows, 16 bucket additions per term into ONE set of 2^15 buckets; entries partitioned in HBM by "
                              "the bucket's high 7 bits (counting sort, 4 B per entry), each workgroup sorts <= 12 288 entries of one partition by the low 8 bits in LDS (48 KB) and "
                              "accumulates one bucket per lane in registers (mixed additions of 128-B affine rows 2^(16 v) P_j)",
                      "arbitrary_points": {"entry_point": "kzg_g1_msm", "ms_msm": round(tA, 4), "ms_decode_and_tables": round(sorted(decA[1:])[0], 4), "ms_call_wall": round(sorted(wallA[1:])[0], 3),
                                           "what": "the same pairs with the points as 2^20 x 48 compressed bytes: decompression + subgroup test + table rows per call, then GLV + 8-bit "
                                                   "windows (32 bucket additions per term) on the verification path's window kernel"},
                      "checked": {"passed": ok4, "what": how},
                      "timing": "HIP events on the library's stream: ms_msm = digits + partition (count, plan, scatter) + bucket accumulation + fold of the workgroups' bucket sums + "
                                "the two 256-bucket reductions + join; ms_call_wall = the whole call from pageable host scalars (32 MB of PCIe, reduction mod r on the device); "
                                "median of 5 runs after one warm-up"}
    return out


def spawn_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: N fresh child processes, one rank each, started before this
    process has imported torch or made any GPU call (never re-exec a process that touched the GPU); rank 0's stdout is
    ours.  Returns the worst child return code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KZG_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def single_process_leg(spec, n, steps, warmup):
    """`--single-process-devices 0,1,...`: ONE process, ONE settings handle over the listed devices (include/kzg_rs_amd.h:
    kzg_settings_from_tau_g2_devices) - what a caller of the reference's one-process signature has on a multi-GPU node.
    n blobs resident PER DEVICE.  Three legs, each checked with a negative control:
      sharded_batch   one verify_blob_kzg_proof_batch of n x D blobs at a time (BASELINE configs[4] behind the reference's call
                      shape: one call, one bool) through kzg_verify_blob_kzg_proof_batch_sharded, with its host stage split
      sharded_stream  the same batches through kzg_verify_blob_kzg_proof_batch_sharded_stream, 4 in flight: the transcript
                      hash of one batch beside the device phases of the others
      groups          independent 1 024-blob batches, launch groups of 32 routed to the device that owns them
                      (kzg_verify_blob_kzg_proof_batch_groups_device, 4 groups in flight per device, no exchange)
    Prints one JSON object."""
    import ctypes as C

    import torch

    from kzg_rs_amd import api, synth

    devs = [int(x) for x in spec.split(",")]
    D = len(devs)
    torch.cuda.set_device(devs[0])
    blobs, cs, ps, st0 = synth.make_valid_batch(n, seed=77, chunk=1024)
    st = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1], devices=devs)
    hc, hp = b"".join(cs), b"".join(ps)
    bad = bytearray(hp)
    bad[48 * (n - 1): 48 * n] = ps[0] if n > 1 else cs[0]   # a valid G1 point, but not this blob's proof
    shards, shards_bad, keep = [], [], []
    src = torch.from_numpy(blobs)
    for k, d in enumerate(devs):
        dev = torch.device("cuda", d)
        t = (src.to(dev), torch.frombuffer(bytearray(hc), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(hp), dtype=torch.uint8).to(dev),
             torch.frombuffer(bad, dtype=torch.uint8).to(dev))
        keep.append(t)
        shards.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n))
        shards_bad.append((t[0].data_ptr(), t[1].data_ptr(), (t[3] if k == D - 1 else t[2]).data_ptr(), n))
    for d in devs:
        torch.cuda.synchronize(d)
    stage_names = ("call", "copy_and_phase1", "transcript_hash_after_last_piece", "phase2_launch", "exchange", "fold_and_pairing", "transcript_hash_busy", "pieces")
    # ---- one sharded batch at a time
    for _ in range(max(warmup, 1)):
        assert api.verify_blob_kzg_proof_batch_sharded(shards, st) is True
    t0 = time.perf_counter()
    for _ in range(steps):
        ok = api.verify_blob_kzg_proof_batch_sharded(shards, st)
    dt = (time.perf_counter() - t0) / steps
    stages = st.multi_last_timings()
    neg = api.verify_blob_kzg_proof_batch_sharded(shards_bad, st)
    # ---- a stream of sharded batches, several in flight
    NB, F = max(8, 2 * steps), 4
    stream_in = [shards] * (NB - 1) + [shards_bad]
    want = [True] * (NB - 1) + [False]
    assert api.verify_blob_kzg_proof_batch_sharded_stream(stream_in[-F - 1:], st, in_flight=F) == want[-F - 1:]   # lanes and workspaces
    t0 = time.perf_counter()
    got = api.verify_blob_kzg_proof_batch_sharded_stream(stream_in, st, in_flight=F)
    dts = time.perf_counter() - t0
    sstages = st.multi_last_timings()
    stream = {"batches": NB, "in_flight": F, "ms_per_batch": round(dts / NB * 1e3, 4), "value": round(n * D * NB / dts, 2), "unit": "blobs/s",
              "results_as_expected": got == want,
              "stage_ms_avg_per_batch": {k: round(v, 4) for k, v in zip(stage_names[1:7], sstages[1:7])}}
    # ---- independent batches: launch groups routed to the devices that own them
    groups_leg = None
    nb = 1024
    if n >= nb:
        B = min(n // nb, 32)
        KG = max(6, steps)  # groups per device (the pointers repeat)
        glist, gwant = [], []
        for g in range(KG):
            for k in range(D):
                last = g == KG - 1 and k == D - 1
                off = (n // nb - B) * nb  # the group that ends with the shard's last batch (where the bad proof sits)
                t = keep[k]
                glist.append((t[0].data_ptr() + off * BYTES_PER_BLOB, t[1].data_ptr() + 48 * off, (t[3] if last else t[2]).data_ptr() + 48 * off))
                gwant.append([True] * (B - 1) + [not last])
        assert api.verify_blob_kzg_proof_batch_groups_device(glist[-3 * D:], nb, B, st, in_flight=4) == gwant[-3 * D:]
        t0 = time.perf_counter()
        gres = api.verify_blob_kzg_proof_batch_groups_device(glist, nb, B, st, in_flight=4)
        dtg = time.perf_counter() - t0
        groups_leg = {"groups": len(glist), "groups_per_device": KG, "batches_per_group": B, "blobs_per_batch": nb, "in_flight_per_device": 4,
                      "value": round(len(glist) * B * nb / dtg, 2), "unit": "blobs/s", "ms": round(dtg * 1e3, 3), "results_as_expected": gres == gwant,
                      "what": "independent %d-blob batches in launch groups of %d, every group on the device that holds it, one pipeline and host "
                              "thread per device, no exchange" % (nb, B)}
    # the same batch from HOST memory through the unchanged kzg_verify_blob_kzg_proof_batch (interleaved chunks, each over its device's PCIe link)
    host = None
    if n * D <= 8192:
        hb = np_tile(blobs, D)
        okh = C.c_bool(False)
        for _ in range(2):
            api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(okh), hb.ctypes.data_as(C.c_char_p), hc * D, hp * D, n * D, st._h))
        t0 = time.perf_counter()
        for _ in range(steps):
            api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(okh), hb.ctypes.data_as(C.c_char_p), hc * D, hp * D, n * D, st._h))
        hst = st.multi_last_timings()
        host = {"ms_per_call": round((time.perf_counter() - t0) / steps * 1e3, 4), "ok": bool(okh.value),
                "stage_ms": {k: round(v, 4) for k, v in zip(stage_names, hst)}}
    devices, exchange = st.devices()
    print(json.dumps({"devices": devices, "exchange": exchange, "exchange_note": st.note(), "batch": n * D, "blobs_per_device": n, "steps": steps,
                      "sharded_batch": {"ms_per_call": round(dt * 1e3, 4), "value": round(n * D / dt, 2), "unit": "blobs/s", "ok": bool(ok),
                                        "corrupted_proof_on_last_device": neg,
                                        "stage_ms": {k: round(v, 4) for k, v in zip(stage_names, stages)}},
                      "sharded_stream": stream, "groups": groups_leg, "host_vec_blob": host,
                      "what": "one process, one multi-device KzgSettings handle over %d device(s); sharded batches of %d blobs (%d resident per device), "
                              "partial sums exchanged by %s" % (D, n * D, n, exchange)}))


def off_subgroup_g1():
    """48 compressed bytes of a point ON y^2 = x^3 + 4 but outside the r-torsion (the first x >= 5 with a square right-hand
    side; a random curve point lies in G1 with probability ~2^-126): from_compressed's subgroup check rejects it."""
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    x = 5
    while True:
        y2 = (x * x * x + 4) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            break
        x += 1
    enc = bytearray(x.to_bytes(48, "big"))
    enc[0] |= 0x80 | (0x20 if y > P - y else 0)
    return bytes(enc)


def np_tile(blobs, k):
    import numpy as np
    return np.ascontiguousarray(np.broadcast_to(blobs, (k,) + blobs.shape)).reshape(k * blobs.shape[0], -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="launch groups timed")
    ap.add_argument("--warmup", type=int, default=5, help="untimed launch groups (at least one per pipeline handle is always run)")
    ap.add_argument("--workload", choices=["auto", "configs1", "config5"], default="auto",
                    help="configs1: 1 024 blobs per GPU per batch, 256 batches per step; config5: 32 768 blobs per GPU per batch "
                         "(BASELINE configs[4] at 8 GPUs), 8 batches per step; auto: configs1 on one GPU, config5 on several")
    ap.add_argument("--config5", action="store_true", help="same as --workload config5")
    ap.add_argument("--blobs", type=int, default=None, help="blobs per GPU per batch (overrides the workload's)")
    ap.add_argument("--group", type=int, default=None, help="independent batches per launch group = per step (overrides the workload's)")
    ap.add_argument("--inflight", type=int, default=4, help="launch groups kept in flight by the fixed-order software pipeline")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single_batch / end_to_end legs (profiling runs)")
    ap.add_argument("--no-self-check", action="store_true", help="skip the poisoned control group and the stand-alone group (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[2] / configs[3] legs (evaluation only; 2^20-term MSM)")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent_callers block (T threads on one shared handle)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the one-process-per-GPU path (PipelinedVerifier: kzg_shard_* phases + torch.distributed exchanges) even in a world of ONE rank - "
                         "what the default run's configs.config5_shard leg does in a child process, so that the 1-GPU figure of the sharded shape comes from the "
                         "same code path --gpus N runs")
    ap.add_argument("--no-shard-leg", action="store_true", help="skip configs.config5_shard (BASELINE configs[4]'s shard shape on one GPU, in a child process)")
    ap.add_argument("--precall-single", action="store_true",
                    help="measurement: one 1 024-blob call on the handle BEFORE the warm-up (it creates the handle's CU-masked stream pair, "
                         "two more users of the 8 hardware queues the pipeline's lanes share)")
    ap.add_argument("--single-process-devices", default=None, metavar="0,1,...",
                    help="only the single-process multi-device leg: one handle over these devices, --blobs per device (default 32768)")
    args = ap.parse_args()

    if args.single_process_devices:
        return single_process_leg(args.single_process_devices, args.blobs or 32768, max(args.steps, 1) if args.steps != 20 else 5, args.warmup if args.warmup != 5 else 2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if os.environ.get("KZG_BENCH_SPAWNED"):
            raise SystemExit("bench.py: spawned rank without WORLD_SIZE")
        raise SystemExit(spawn_ranks(args.gpus))
    if os.environ.get("KZG_BENCH_DRY_SPAWN") == "1":  # CPU test of the launcher: say who we are, touch nothing
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    # KZG_BENCH_SHARE_GPU=1 (test rig only): all ranks on cuda:0 with gloo for the exchanges - lets the N > 1 code
    # path of this script run on a one-GPU box; the numbers it prints then mean nothing
    share = os.environ.get("KZG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = "cpu" if share else dev  # where the collectives' tensors live
    dist = None
    use_pipe = world > 1 or args.force_collectives   # the one-process-per-GPU path (also in a world of one when asked for)
    if use_pipe:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))

    from kzg_rs_amd import api, synth
    from kzg_rs_amd.distributed import HipBackend, PipelinedVerifier, verify_blob_kzg_proof_batch_sharded

    workload = "config5" if args.config5 else args.workload
    if workload == "auto":
        workload = "configs1" if world == 1 else "config5"
    n = args.blobs or (1024 if workload == "configs1" else 32768)
    G = max(1, args.group or (256 if workload == "configs1" else 8))
    F = max(1, args.inflight)
    # pre-flight: the variants (one per group in flight: G x n blobs each) against this GPU's free memory - a run that does not fit
    # shrinks its launch group (and says so) instead of dying in an allocation
    n_var = 1 if F <= 1 else (max(1, F - 2) + 3 if use_pipe else F + 1)   # = sum(depth) + 1 below: one variant per pipeline handle
    free_b = torch.cuda.mem_get_info(dev)[0]
    G_asked = G
    def _need(g):
        return n_var * g * n * (BYTES_PER_BLOB + 96) + n * (BYTES_PER_BLOB + 96) + (14 << 30)   # + the synthetic batch + workspaces, tables, pinned mirrors
    while G > 1 and _need(G) > free_b:
        G = max(1, G // 2)
    preflight = {"free_hbm_GiB": round(free_b / 2**30, 1), "variants": n_var, "needed_GiB": round(_need(G) / 2**30, 1), "batches_per_step": G,
                 "batches_per_step_asked": G_asked, "shrunk": G != G_asked}
    n_src = min(n, 1024) if args.force_collectives and world == 1 else n   # (the shard leg of the default run: 1 024 valid blobs, tiled)
    blobs, cs, ps, settings = synth.make_valid_batch(n_src, seed=1000 + rank, chunk=1024)
    if n_src != n:
        reps = (n + n_src - 1) // n_src
        blobs = np.ascontiguousarray(np.tile(blobs, (reps, 1))[:n])
        cs, ps = (cs * reps)[:n], (ps * reps)[:n]
    d_blobs = torch.from_numpy(blobs).to(dev)
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    backend0 = HipBackend(settings)

    # ---- pipeline: depth (d1, d2, d3) groups between the phases; one handle (2 HIP streams + workspace) per group in flight
    if F <= 1:
        depth = (0, 0, 0)
    elif use_pipe:
        depth = (max(1, F - 2), 1, 1)
    else:
        depth = (F - 1, 0, 1)
    n_handles = sum(depth) + 1
    if not use_pipe:
        # ONE GPU: the groups are kept in flight INSIDE the library (kzg_verify_blob_kzg_proof_batch_groups_device,
        # csrc/capi_pipeline.hpp: the same fixed-order pipeline behind one C call - the entry point a C / Rust caller with many
        # batches has); the handle grows its own per-group lanes
        handles, backends, pipe = [settings], [backend0], None
    else:
        handles = [settings] + [api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1]) for _ in range(n_handles - 1)]
        backends = [backend0] + [HipBackend(h) for h in handles[1:]]
        pipe = PipelinedVerifier(backends, dist, coll_dev, depth, equal_shards=True, force_collectives=args.force_collectives)
    # every batch is a different permutation of the rank's shard (different transcript and r), at its own HBM address:
    # one variant (G x n blobs = G x 128 MiB) per handle, so the groups in flight stream disjoint memory
    gen = torch.Generator(device="cpu").manual_seed(7 + rank)
    c_t, p_t = d_c.view(n, 48), d_p.view(n, 48)
    variants = []
    for v in range(n_handles):
        perms = [torch.randperm(n, generator=gen).to(dev) if (v or g) else torch.arange(n, device=dev) for g in range(G)]
        variants.append((torch.cat([d_blobs[p] for p in perms]).contiguous(), torch.cat([c_t[p] for p in perms]).contiguous(),
                         torch.cat([p_t[p] for p in perms]).contiguous()))
    torch.cuda.synchronize()
    if os.environ.get("KZG_PMC_CALIBRATE") == "1":
        # a known-size dispatch for the FETCH_SIZE / WRITE_SIZE scaling check of a PMC run: an elementwise kernel (the
        # only one of this name and grid in the process) that reads 128 MiB and writes 128 MiB
        cal = d_blobs.view(torch.int32).add(1)
        torch.cuda.synchronize()
        del cal

    def run_groups(k):
        """k launch groups of G independent batches each through the pipeline."""
        groups = []
        for i in range(k):
            v = variants[i % n_handles]
            groups.append(((v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), n), G))
        if pipe is None:
            res = api.verify_blob_kzg_proof_batch_groups_device([g[0][:3] for g in groups], n, G, settings, in_flight=F)
        else:
            res = pipe.run(groups)
        if not all(all(r) for r in res):
            raise SystemExit("verification of a valid synthetic batch returned false")
        return groups

    def timed(fn):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    if args.precall_single:
        v0 = variants[0]
        assert api.KzgProof.verify_blob_kzg_proof_batch_device(v0[0].data_ptr(), v0[1].data_ptr(), v0[2].data_ptr(), n, settings) is True
    # warm-up: every handle allocates its workspace and runs every kernel once before the timed region
    W = max(args.warmup, n_handles) if args.warmup else 0
    if W:
        run_groups(W)
    K = args.steps
    warm_sum5, warm_cnt = 0.0, 0
    other_stamps, other_stamp_cnt = {}, 0   # in-kernel stamps of the launch groups OUTSIDE the timed region: warm-up, stand-alone, self-check
    for h in handles:
        t, c = h.timing_totals(reset=True)  # the warm-up groups do not count (kept apart for the profiler cross-check below)
        warm_sum5 += t[5]
        warm_cnt += c
        h.shader_clock(reset=True)
        ws, wc, _ = h.kernel_stamp_totals(reset=True)
        other_stamps = {k: other_stamps.get(k, 0.0) + v for k, v in ws.items()}
        other_stamp_cnt += wc
    elapsed, _ = timed(lambda: run_groups(K))
    clk_cycles = clk_ticks = 0.0
    for h in handles:  # the clock the SIMDs ran at during the timed region (every wave of the challenge kernel stamps it)
        a, b = h.shader_clock(reset=True)
        clk_cycles += a
        clk_ticks += b
    shader_mhz = 100.0 * clk_cycles / clk_ticks if clk_ticks else None
    # kernel times: on the library's own streams, AVERAGED over every launch group of the timed region (all handles)
    # - the quantity rocprofv3 --stats reports as the kernel's average duration for the same command
    sums, cnt = [0.0] * 8, 0
    for h in handles:
        t, c = h.timing_totals(reset=True)
        sums = [a + b for a, b in zip(sums, t)]
        cnt += c
    tm = [x / max(cnt, 1) for x in sums]
    # the kernels' OWN intervals in the timed region (in-kernel stamps: first wavefront in, last out - residency beside the other
    # groups in flight, without the queueing a HIP-event pair would add)
    stamp_sum, stamp_cnt = {}, 0
    for h in handles:
        ssum, c, _ = h.kernel_stamp_totals(reset=True)
        stamp_sum = {k: stamp_sum.get(k, 0.0) + v for k, v in ssum.items()}
        stamp_cnt += c
    in_flight = {k: v / max(stamp_cnt, 1) for k, v in stamp_sum.items()}
    kernels = {"k_blob_challenge": tm[5], "k_blob_evaluate": tm[4], "k_g1_decode_multiples": tm[6], "k_msm": tm[2],
               "k_slp_run(pairing)": tm[3]}

    # ---- ONE batch at a time (the reference's call shape): device-resident, then from host memory
    single = end2end = None
    if not args.no_latency:
        stage_s = {}

        def seq_steps(k, tm=None):
            for _ in range(k):
                if not use_pipe:
                    ok = api.KzgProof.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, settings)
                else:
                    ok = verify_blob_kzg_proof_batch_sharded((d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n), n, backend0, dist, coll_dev, timings=tm)
                if not ok:
                    raise SystemExit("verification of a valid synthetic batch returned false")

        seq_steps(2)
        KS = 16 if n <= 4096 else 4
        seq_elapsed, _ = timed(lambda: seq_steps(KS, stage_s))
        seq_tm = settings.last_timings()
        single = {"value": round(n * world * KS / seq_elapsed, 2), "unit": "blobs/s", "ms_per_step": round(seq_elapsed / KS * 1e3, 4),
                  "steps": KS, "what": "one verify_blob_kzg_proof_batch call of %d device-resident blobs at a time%s"
                                       % (n * world, "" if world == 1 else ", sharded by blob over %d ranks (transcript hashed once on rank 0)" % world),
                  "stage_ms_rank0": {k[:-2] + "_ms": round(v / KS * 1e3, 4) for k, v in stage_s.items()} if use_pipe else None,
                  "kernel_ms": {"k_blob_challenge": round(seq_tm[5], 4), "k_blob_evaluate": round(seq_tm[4], 4),
                                "k_g1_decode_multiples": round(seq_tm[6], 4), "k_msm": round(seq_tm[2], 4),
                                "k_slp_run(pairing)": round(seq_tm[3], 4)}}
        if not use_pipe and n <= 4096:
            import ctypes as C

            h_c, h_p = b"".join(cs), b"".join(ps)
            h_blobs = np.ascontiguousarray(blobs)  # pageable host memory, exactly a Rust Vec<Blob>
            ok = C.c_bool(False)

            def host_steps(k):
                for _ in range(k):
                    api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(ok), h_blobs.ctypes.data_as(C.c_char_p), h_c, h_p, n, settings._h))
                    if not ok.value:
                        raise SystemExit("verification of a valid synthetic batch returned false")

            host_steps(2)
            KH = 16
            h_elapsed, _ = timed(lambda: host_steps(KH))
            # a STREAM of host batches (kzg_verify_blob_kzg_proof_batches): chunked copies overlapped with verification
            NB = 64
            h_many = np.ascontiguousarray(np.broadcast_to(h_blobs, (NB,) + h_blobs.shape)).reshape(NB * n, -1)  # 8 GiB, pageable
            hc_many, hp_many = h_c * NB, h_p * NB

            def host_stream():
                res = api.verify_blob_kzg_proof_batches(h_many.ctypes.data, hc_many, hp_many, n, NB, settings)
                if not all(res):
                    raise SystemExit("verification of a valid synthetic batch returned false")

            cold, _ = timed(host_stream)   # first touch of the pages by the driver (it pins pageable memory on the fly)
            warm, _ = timed(host_stream)
            end2end = {"value": round(n * KH / h_elapsed, 2), "unit": "blobs/s", "ms_per_step": round(h_elapsed / KH * 1e3, 4), "steps": KH,
                       "source": "host Vec<Blob> layout (pageable memory) through kzg_verify_blob_kzg_proof_batch, one %d-blob batch at a "
                                 "time, PCIe transfer included" % n,
                       "stream": {"value": round(n * NB / warm, 2), "unit": "blobs/s", "batches": NB, "ms_per_batch": round(warm / NB * 1e3, 4),
                                  "first_pass_blobs_per_s": round(n * NB / cold, 2), "pcie_GBps": round(n * NB * (BYTES_PER_BLOB + 96) / warm / 1e9, 2),
                                  "source": "%d host batches of %d blobs back to back in pageable memory through kzg_verify_blob_kzg_proof_batches "
                                            "(chunked copies on a copy stream overlapped with the previous chunk's verification); first_pass = the same "
                                            "call on never-touched pages" % (NB, n)}}
            del h_many
            # ONE reference-shaped call with a LARGE Vec<Blob> (src/kzg_proof.rs:472-525; an unchanged caller does not restructure its blobs
            # into launch groups): 8 192 and 32 768 blobs through kzg_verify_blob_kzg_proof_batch from pageable host memory, and
            # device-resident through kzg_verify_blob_kzg_proof_batch_device - next to the 1 024-blob figures above
            large = []
            for nl in (8192, 32768):
                rl = nl // n
                hb = np.ascontiguousarray(np.broadcast_to(h_blobs, (rl,) + h_blobs.shape)).reshape(nl, -1)   # pageable, 1 / 4 GiB
                hc_l, hp_l = h_c * rl, h_p * rl
                tsl, kms = [], None
                for _ in range(4):
                    t0 = time.perf_counter()
                    api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(ok), hb.ctypes.data_as(C.c_char_p), hc_l, hp_l, nl, settings._h))
                    tsl.append(time.perf_counter() - t0)
                    if not ok.value:
                        raise SystemExit("verification of a valid synthetic batch returned false (large host call)")
                host_ms = sorted(tsl[1:])[1] * 1e3
                db = torch.from_numpy(hb).to(dev)
                dc_l = torch.frombuffer(bytearray(hc_l), dtype=torch.uint8).to(dev)
                dp_l = torch.frombuffer(bytearray(hp_l), dtype=torch.uint8).to(dev)
                torch.cuda.synchronize()
                tsd = []
                for _ in range(4):
                    t0 = time.perf_counter()
                    okd = api.KzgProof.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc_l.data_ptr(), dp_l.data_ptr(), nl, settings)
                    tsd.append(time.perf_counter() - t0)
                    if not okd:
                        raise SystemExit("verification of a valid synthetic batch returned false (large device call)")
                ltm = settings.last_timings()
                dev_ms = sorted(tsd[1:])[1] * 1e3
                large.append({"blobs": nl, "host_pageable": {"ms": round(host_ms, 3), "blobs_per_s": round(nl / host_ms * 1e3, 1), "pcie_GBps": round(nl * (BYTES_PER_BLOB + 96) / host_ms / 1e6, 2)},
                              "device_resident": {"ms": round(dev_ms, 3), "blobs_per_s": round(nl / dev_ms * 1e3, 1),
                                                  "kernel_ms": {"k_blob_challenge": round(ltm[5], 4), "k_blob_evaluate": round(ltm[4], 4), "k_g1_decode_multiples": round(ltm[6], 4),
                                                                "k_msm": round(ltm[2], 4), "k_slp_run(pairing)": round(ltm[3], 4)}}})
                del db, dc_l, dp_l, hb
                torch.cuda.empty_cache()
            end2end["large_calls"] = {"what": "ONE kzg_verify_blob_kzg_proof_batch call (one transcript, one r, one pairing) of a large Vec<Blob>: median of 3 calls after one "
                                              "warm-up; the blobs are the 1 024 valid synthetic blobs repeated (a batch may hold a blob twice)", "calls": large}
    # ---- the kernels ALONE on the chip: one launch group on a single-stream handle, nothing else in flight.  With several
    # groups in flight a kernel's interval is its residency (how long it shared the chip), not its cost.
    standalone = None
    solo_stamps = None
    self_check = None
    proof_ms = None
    blob_ms = None
    small_ms = None
    proofs_per_s = None
    concurrent = None
    configs = None
    solo_sums, solo_cnt = [0.0] * 8, 0
    sc_sums, sc_cnt = [0.0] * 8, 0
    if not args.no_self_check:
        with api.options(single_stream=1):   # read when a handle is made
            solo = api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1])
        v = variants[0]
        solo_res = None
        for _ in range(2):  # first pass: workspace allocation
            torch.cuda.synchronize()
            solo_res = api.verify_blob_kzg_proof_batches_device(v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), n, G, solo)
        st_tm = solo.last_timings()
        ss, sc_, solo_stamps = solo.kernel_stamp_totals()   # the in-kernel stamps (ms) of the handle's groups, and of its last one
        other_stamps = {k: other_stamps.get(k, 0.0) + v for k, v in ss.items()}
        other_stamp_cnt += sc_
        standalone = {"k_blob_challenge": st_tm[5], "k_blob_evaluate": st_tm[4], "k_g1_decode_multiples": st_tm[6], "k_msm": st_tm[2],
                      "k_slp_run(pairing)": st_tm[3], "whole_group": st_tm[0]}
        if not all(r is True for r in solo_res):
            raise SystemExit("verification of a valid synthetic batch returned false (stand-alone group)")
        solo_sums, solo_cnt = solo.timing_totals()
        solo.close()
        # ---- self check THROUGH THE BENCHMARKED ENTRY POINT at the benchmarked shape (src/kzg_proof.rs:436-444 -> Ok(false);
        # src/dtypes.rs:48-57, src/kzg_proof.rs:17-25 -> Err): n_handles + 1 full launch groups through ONE
        # kzg_verify_blob_kzg_proof_batch_groups_device call with the benchmark's groups in flight, so every lane of the
        # pipeline carries a group and lane 0 carries two.  Poisoned in DIFFERENT groups (= different lanes): group 0 a valid
        # G1 point that is not its blob's proof (-> false), group 1 a field element equal to r (-> Err), group 2 a
        # commitment on the curve but outside G1 (-> Err); the last group repeats group 0's memory (-> false again).
        NG = n_handles + 1
        order = [i % n_handles for i in range(NG)]
        bf, be, bc = G // 3, (2 * G) // 3, G - 1
        poison = {}
        if n_handles >= 3 and G >= 3:
            poison = {0: ("wrong_proof", bf, None), 1: ("element_equals_r", be, None), 2: ("commitment_outside_g1", bc, None)}
        else:  # (a reduced --inflight / --group: everything in the one variant there is)
            poison = {0: ("wrong_proof", 0, None)}
        saved = []
        r_be = torch.tensor(list((0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001).to_bytes(32, "big")), dtype=torch.uint8, device=dev)
        off_g1 = torch.tensor(list(off_subgroup_g1()), dtype=torch.uint8, device=dev)
        for vi, (kind, b, _) in poison.items():
            vv = variants[vi]
            if kind == "wrong_proof":
                i = b * n + n // 2
                saved.append((vv[2], i, vv[2][i].clone()))
                vv[2][i] = vv[2][(i + 1) % (n * G)]
            elif kind == "element_equals_r":
                i = b * n + n - 1
                saved.append((vv[0], (i, slice(64, 96)), vv[0][i, 64:96].clone()))
                vv[0][i, 64:96] = r_be
            else:
                i = b * n + 1
                saved.append((vv[1], i, vv[1][i].clone()))
                vv[1][i] = off_g1
        torch.cuda.synchronize()
        for h in handles:  # (the single-batch legs above ran on this handle too: only the control's launches are counted below)
            h.timing_totals(reset=True)
            h.kernel_stamp_totals(reset=True)
        res = api.verify_blob_kzg_proof_batch_groups_device([tuple(t.data_ptr() for t in variants[vi]) for vi in order], n, G, settings, in_flight=F)
        for t, i, old in saved:
            t[i] = old
        torch.cuda.synchronize()
        want = [[True] * G for _ in range(NG)]
        for g, vi in enumerate(order):
            if vi in poison:
                want[g][poison[vi][1]] = False if poison[vi][0] == "wrong_proof" else None
        false_b = [[g, b] for g in range(NG) for b in range(G) if res[g][b] is False]
        err_b = [[g, b] for g in range(NG) for b in range(G) if res[g][b] is None]
        self_check = {"entry_point": "kzg_verify_blob_kzg_proof_batch_groups_device", "groups": NG, "groups_in_flight": F, "batches_per_group": G,
                      "blobs": n * G * NG,
                      "poisoned": [{"group": g, "lane": g % (n_handles), "batch": poison[vi][1], "how": poison[vi][0]} for g, vi in enumerate(order) if vi in poison],
                      "false_batches": false_b, "err_batches": err_b, "true_batches": sum(1 for row in res for r in row if r is True),
                      "passed": res == want}
        # the clean groups again: nothing of the poisoned call may linger on the lanes
        run_groups(n_handles)
        for h in handles:  # (the control and the clean groups after it are launches of this process too: see launch_ms_all_launches)
            t, c = h.timing_totals(reset=True)
            sc_sums = [a + b for a, b in zip(sc_sums, t)]
            sc_cnt += c
            ks, kc, _ = h.kernel_stamp_totals(reset=True)
            other_stamps = {k: other_stamps.get(k, 0.0) + v for k, v in ks.items()}
            other_stamp_cnt += kc
        if world == 1:
            # one proof at a time (src/kzg_proof.rs:353-397, the revm precompile's call): median of 32 calls
            pc, pz, py, pp, _ = synth.make_valid_proofs(1, seed=5, settings=settings)
            args4 = (api.Bytes48(pc[0]), api.Bytes32(pz[0]), api.Bytes32(py[0]), api.Bytes48(pp[0]))
            ts = []
            for _ in range(36):
                t0 = time.perf_counter()
                okp = api.KzgProof.verify_kzg_proof(*args4, settings)
                ts.append(time.perf_counter() - t0)
            assert okp is True
            proof_ms = round(sorted(ts[4:])[16] * 1e3, 4)
            # one blob from host memory (src/kzg_proof.rs:446-470): median of 16 calls
            b0 = api.Blob(blobs[0].tobytes())
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                okb = api.KzgProof.verify_blob_kzg_proof(b0, api.Bytes48(cs[0]), api.Bytes48(ps[0]), settings)
                ts.append(time.perf_counter() - t0)
            assert okb is True
            blob_ms = round(sorted(ts[4:])[8] * 1e3, 4)
            # a block's worth of blobs from host memory (the size a beacon node calls verify_blob_kzg_proof_batch with): median of 16
            import ctypes as C
            okc = C.c_bool(False)
            raw6 = (blobs[:6].tobytes(), b"".join(cs[:6]), b"".join(ps[:6]))
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                api._chk(api.lib().kzg_verify_blob_kzg_proof_batch(C.byref(okc), raw6[0], raw6[1], raw6[2], 6, settings._h))
                ts.append(time.perf_counter() - t0)
            assert okc.value is True
            small_ms = round(sorted(ts[4:])[8] * 1e3, 4)
            # 1 024 INDEPENDENT proofs, a verdict each, through one call (kzg_verify_kzg_proofs): median of 5 calls
            qc, qz, qy, qp, _ = synth.make_valid_proofs(1024, seed=6, settings=settings)
            qy[7] = qy[8]  # one wrong claim: exactly that verdict must be False
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                verdicts = api.verify_kzg_proofs(qc, qz, qy, qp, settings)
                ts.append(time.perf_counter() - t0)
            assert verdicts == [i != 7 for i in range(1024)]
            proofs_per_s = round(1024 / sorted(ts[1:])[2])
            if not args.no_concurrent:
                concurrent = concurrent_callers(settings, blobs, cs, ps, synth, qc, qz, qy, qp, no_cpu=args.no_cpu_baseline)
            if not args.no_configs:
                configs = config_legs(settings, torch, dev, no_cpu=args.no_cpu_baseline)
    backend_name = dist.get_backend() if dist else None
    pipe_check = rank_stats = None
    if pipe is not None:
        # the negative control THROUGH THE PIPELINED PATH: one group of the timed shape with one proof swapped for another blob's in
        # batch G // 2 - that batch false on every rank, every other batch true (src/kzg_proof.rs:436-444)
        v0 = variants[0]
        bad_b = G // 2
        i = bad_b * n + n // 2
        keep_p = v0[2][i].clone()
        v0[2][i] = v0[2][(i + 1) % (n * G)] if not torch.equal(v0[2][i], v0[2][(i + 1) % (n * G)]) else v0[2][(i + 2) % (n * G)]
        torch.cuda.synchronize()
        res_bad = pipe.run([((v0[0].data_ptr(), v0[1].data_ptr(), v0[2].data_ptr(), n), G)])[0]
        v0[2][i] = keep_p
        torch.cuda.synchronize()
        res_ok = pipe.run([((v0[0].data_ptr(), v0[1].data_ptr(), v0[2].data_ptr(), n), G)])[0]
        pipe_check = {"poisoned_batch": bad_b, "how": "one proof of this rank's shard replaced by another blob's (a valid G1 point)",
                      "results": [r for r in res_bad], "passed": list(res_bad) == [b != bad_b for b in range(G)] and all(r is True for r in res_ok)}
        # every rank's host-side stage split (per step, warm-up groups included), gathered on rank 0
        mine = dict(pipe.stats, rank=rank)
        rank_stats = [None] * world
        if world > 1:
            dist.all_gather_object(rank_stats, mine)
        else:
            rank_stats = [mine]
    if dist:
        # every rank leaves its GPU before rank 0 reports (and, at N > 1, drives all of them from one process)
        del variants, d_blobs
        pipe = type("Stats", (), {"stats": dict(pipe.stats)})
        del backends, handles, backend0, settings
        torch.cuda.empty_cache()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # ---- roofline / path / valu: assembled by kzg_rs_amd/benchline.py (a pure function of this measurement and the PMC profile under
    # profiles/; its docstring states the dominance rule and what every column means; tests/test_benchline.py checks its arithmetic)
    from kzg_rs_amd import benchline
    from kzg_rs_amd import build as kbuild
    pmc_file, pmc = benchline.load_pmc(ROOT)
    try:
        kkey = kbuild.kernel_key()
    except BaseException:
        kkey = None
    n_chal = cnt + warm_cnt + solo_cnt + sc_cnt
    measurement = {
        "n": n, "G": G, "K": K, "F": F, "world": world, "elapsed_s": elapsed, "shader_mhz": shader_mhz,
        "in_flight_ms": {k: v for k, v in in_flight.items() if v}, "stamp_cnt": stamp_cnt, "stamp_sum_ms": stamp_sum,
        "other_stamp_sum_ms": other_stamps, "other_stamp_cnt": other_stamp_cnt,
        "solo_stamps_ms": {k: v for k, v in (solo_stamps or {}).items() if v} or None,
        "standalone_event_ms": standalone,
        "challenge_event_population": {"all_launches_ms": (sums[5] + warm_sum5 + solo_sums[5] + sc_sums[5]) / n_chal if n_chal and kernels["k_blob_challenge"] > 0 else None,
                                       "launches": n_chal if kernels["k_blob_challenge"] > 0 else None,
                                       "incl_warmup_ms": (sums[5] + warm_sum5) / max(cnt + warm_cnt, 1) if kernels["k_blob_challenge"] > 0 else None},
    }
    blocks = benchline.assemble(measurement, pmc_file, pmc, kkey)
    clock_hz = (shader_mhz or 2400.0) * 1e6
    out = {
        "metric": "blobs/sec verify_blob_kzg_proof_batch",
        "value": round(n * world * G * K / elapsed, 2),
        "unit": "blobs/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (255-bit Fr / 381-bit Fp modular integers)",
        "data": "synthetic",
        "config": {"workload": "verify_blob_kzg_proof_batch, batches of %d synthetic blobs (4096 Fr each)%s, device-resident, known-tau test setup (%s); "
                               "one step = one launch group of %d independent batches = %d blobs per GPU"
                               % (n * world, "" if world == 1 else " sharded by blob over %d GPUs (%d per GPU)" % (world, n),
                                  "BASELINE.json configs[1]" if n == 1024 and world == 1 else
                                  "BASELINE.json configs[4]" + ("" if world == 8 else " shard shape: 32 768 blobs per GPU") if n == 32768 else "custom size", G, n * G),
                   "blobs_per_gpu_per_batch": n, "batch": n * world, "batches_per_step": G, "blobs_per_step": n * world * G,
                   "parallelism": "shard-by-blob x%d" % world, "groups_in_flight": F,
                   "entry_point": "kzg_verify_blob_kzg_proof_batch_groups_device: ONE C call for all %d launch groups of the timed region, %d in flight "
                                  "inside the library" % (K, F) if not use_pipe else
                                  "kzg_shard_*_launch/_wait phases driven by kzg_rs_amd.distributed.PipelinedVerifier, one process per GPU%s"
                                  % (" (a world of ONE rank: --force-collectives)" if world == 1 else "")},
        "roofline": blocks["roofline"],
        "path": blocks["path"],
        "valu": blocks["valu"],
        "kernel_ms_standalone": blocks["kernel_ms_standalone"],
        "kernel_ms_in_flight": blocks["kernel_ms_in_flight"],
        "self_check": self_check,
        "single_batch": single,
        "end_to_end": end2end,
        "verify_kzg_proof_ms": proof_ms,
        "verify_blob_kzg_proof_ms": blob_ms,
        "verify_blob_kzg_proof_batch_6_host_blobs_ms": small_ms,
        "verify_kzg_proofs_independent_per_s": proofs_per_s,
        "concurrent_callers": concurrent,
        "configs": configs,
    }
    if configs:   # cycles per VALU instruction of the two config kernels: this run's time x the measured clock / the instruction counts of the committed PMC profile
        try:
            cfile = "r6_config_pmc.json"
            cj = json.load(open(os.path.join(ROOT, "profiles", cfile)))
            cp = cj["kernels"]
            fresh = kkey is not None and cj.get("kernel_key") == kkey
            for leg, kn, ms_key in (("config3", "kzg::k_blob_evaluate_t<true>", "ms"), ("config4", "kzg::k_fb_window", None)):
                if leg not in configs or kn not in cp:
                    continue
                if not fresh:
                    configs[leg]["roofline"]["pmc_stale"] = "profiles/%s was collected on kernel key %s, this tree is %s: counters NOT used" % (cfile, cj.get("kernel_key"), kkey)
                    continue
                insts = cp[kn]["SQ_INSTS_VALU"]
                ms = configs[leg]["ms"] if ms_key else cp[kn]["ms_single_stream"]
                configs[leg]["roofline"].update({
                    "valu_wave_insts": round(insts), "valu_cycles_per_inst": round(ms * 1e-3 * clock_hz * 1024 / insts, 3), "issue_ceiling_cycles_per_inst": 4.2,
                    "valu_source": "profiles/%s (kernel key %s = this tree's) SQ_INSTS_VALU of the same launch size; %s; shader clock %s" % (
                        cfile, kkey, "this run's ms" if ms_key else "the bucket kernel's own duration in that profile (%.2f ms: ms_msm above also holds the partition passes, folds and reductions)" % ms,
                        "%.0f MHz measured in this run's timed region" % shader_mhz if shader_mhz else "2 400 MHz nominal"),
                    "hbm_traffic_bytes": cp[kn].get("hbm_bytes_corrected"), "sq_wait_any_frac": round(cp[kn]["SQ_WAIT_ANY"] / cp[kn]["SQ_WAVE_CYCLES"], 3)})
        except Exception:
            pass
    out["preflight"] = preflight
    if use_pipe:
        g = max(pipe.stats["groups"], 1)  # warm-up groups included; per-step averages of THIS rank's host time
        bulk_once = G % world == 0
        shard_1gpu = None
        try:
            shard_1gpu = json.load(open(os.path.join(ROOT, "profiles", "r6_config5_shard_1gpu.json")))
        except Exception:
            pass
        out["multi_gpu"] = {"ranks": world, "backend": backend_name, "rccl_ranks": world if backend_name == "nccl" else 0,
                            "exchange_ran": {"exchange1": ("all_to_all_single of [src rank][%d batches][%d blobs] x 160 B records to the batch's owner, then all_gather of r "
                                                           "(32 B per batch) + error flags" % (G // world, n)) if bulk_once else
                                                          "all_gather of every rank's records (batches per step not divisible by the rank count)",
                                             "exchange2": "all_gather of %d x 288 B partial sums (A_k, B_k), folded on every rank" % G,
                                             "transport": "RCCL (torch.distributed backend nccl) on device buffers" if backend_name == "nccl" else
                                                          "gloo on host buffers (test rig: ranks share one GPU)",
                                             "world_of_one": world == 1},
                            "per_rank_stage_ms_per_step": [{"rank": r_["rank"], "groups": r_["groups"], "r_hash_ms": round(r_["r_hash_s"] / max(r_["groups"], 1) * 1e3, 4),
                                                            "exchange1_ms": round(r_["exchange1_s"] / max(r_["groups"], 1) * 1e3, 4),
                                                            "exchange2_ms": round(r_["exchange2_s"] / max(r_["groups"], 1) * 1e3, 4)} for r_ in (rank_stats or []) if r_],
                            "pipe_check": pipe_check,
                            "efficiency_vs": {"value": round(world * shard_1gpu["value"], 2), "unit": "blobs/s", "one_gpu_value": shard_1gpu["value"],
                                              "what": "N x the ONE-GPU figure of this same shard shape through this same code path (profiles/r6_config5_shard_1gpu.json: "
                                                      "bench.py --workload config5 --force-collectives on one MI355X, %s) - the like-for-like base of a scaling curve; "
                                                      "configs1's 1 024-blob batches at N = 1 are a different workload" % shard_1gpu.get("measured", "")} if shard_1gpu else None,
                            "transcript_hash": "once per batch: rank j hashes batches [jB/N, (j+1)B/N) of every step" if G % world == 0 else
                                               "every rank hashes every batch (batches per step not divisible by the rank count)",
                            "r_hash_ms_per_step": round(pipe.stats["r_hash_s"] / g * 1e3, 4),
                            "exchange1_ms_per_step": round(pipe.stats["exchange1_s"] / g * 1e3, 4),
                            "exchange2_ms_per_step": round(pipe.stats["exchange2_s"] / g * 1e3, 4),
                            "note": "exchange 1 = all-to-all of 160 B transcript records + all-gather of r (32 B per batch) and error flags; exchange 2 = "
                                    "all-gather of the 288 B partial sums (A_k, B_k) - the G1 all-reduce of the north star, folded on every rank; host "
                                    "times of rank 0, overlapped with the GPU work of the other groups in flight"}
        # ONE process over all the GPUs through a multi-device handle and the reference's call shape (the ranks have left the
        # GPUs by now).  A child process: a failure or a hang there costs this leg, not the line.
        sp = os.environ.get("KZG_BENCH_SINGLE_PROCESS", "1")
        if world > 1 and sp != "0" and (not share or sp == "force"):  # (ranks sharing one GPU: only when a test asks for it - the list then names device 0 `world` times)
            out["multi_gpu"]["single_process"] = run_single_process_child(",".join("0" if share else str(i) for i in range(world)), n)
    if world == 1 and not use_pipe and not args.no_configs and not args.no_self_check and not args.no_shard_leg:
        # BASELINE configs[4]'s SHARD SHAPE on this one GPU, through the code path --gpus N runs (a child process: this one gives
        # its memory back first; a failure or a hang there costs the leg, not the line)
        variants.clear()   # (and every other name that still points into them)
        v = vv = v0 = saved = res = t = solo_res = d_blobs = d_c = d_p = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out.setdefault("configs", {})
        if out["configs"] is None:
            out["configs"] = {}
        out["configs"]["config5_shard"] = run_shard_leg_child(args.steps, args.warmup)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blobs, cs, ps, synth.synthetic_setup()[1], args.cpu_sample)
    print(json.dumps(out))


def run_shard_leg_child(steps, warmup, timeout=420):
    """configs.config5_shard: `bench.py --gpus 1 --workload config5 --force-collectives` in a child process - 8 batches x 32 768 blobs
    per step on ONE GPU through PipelinedVerifier (kzg_shard_* phases, the records all-to-all, r as 32 bytes, the all-gather of
    288-byte partial sums: a world of one rank over RCCL), i.e. the N = 1 point of the curve `--gpus N` draws."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "config5", "--force-collectives", "--steps", str(steps), "--warmup", str(warmup),
           "--no-latency", "--no-self-check", "--no-cpu-baseline", "--no-configs", "--no-concurrent"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    for k in list(env):   # a profiler around this process (rocprofv3 preloads its tool library) does not follow the child: it runs plain
        if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") or k.startswith(("ROCPROF", "ROCPROFILER", "ROCTX")):
            env.pop(k)
    try:
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        wall = time.perf_counter() - t0
        line = None
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                line = json.loads(ln)
                break
        if line is None:
            return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-800:])}
        mg = line.get("multi_gpu") or {}
        rk = (mg.get("per_rank_stage_ms_per_step") or [{}])[0]
        return {"workload": line["config"]["workload"], "entry_point": line["config"]["entry_point"], "value": line["value"], "unit": line["unit"],
                "ms_per_step": line["ms_per_step"], "steps": line["steps"], "warmup": line["warmup"], "blobs_per_step": line["config"]["blobs_per_step"],
                "batches_per_step": line["config"]["batches_per_step"], "blobs_per_gpu_per_batch": line["config"]["blobs_per_gpu_per_batch"],
                "r_hash_s": round(rk.get("r_hash_ms", 0.0) / 1e3, 6), "exchange1_s": round(rk.get("exchange1_ms", 0.0) / 1e3, 6), "exchange2_s": round(rk.get("exchange2_ms", 0.0) / 1e3, 6),
                "stage_note": "host seconds per step of the one rank: r_hash = SHA-256 of the step's 8 transcripts of 32 768 x 160 B (42 MB) on the host pool, overlapped with the "
                              "GPU phases of the other groups in flight; exchange 1 / 2 = the collectives (a world of one: the transport is exercised, nothing crosses a link)",
                "poisoned_batch_false": (mg.get("pipe_check") or {}).get("passed"), "pipe_check": mg.get("pipe_check"),
                "exchange_ran": mg.get("exchange_ran"), "backend": mg.get("backend"), "preflight": line.get("preflight"),
                "roofline": {k: line["roofline"].get(k) for k in ("kernel", "launch_ms", "frac", "frac_path", "frac_of_binding_bound")},
                "child_wall_s": round(wall, 1),
                "what": "BASELINE.json configs[4]'s shard shape on ONE GPU: 8 batches x 32 768 blobs per step = 262 144 blobs, the per-GPU work of every step at any N, "
                        "through the one-process-per-GPU path (PipelinedVerifier with force_collectives, world of one rank)"}
    except Exception as e:  # timeout included
        return {"error": repr(e)[:800]}


def run_single_process_child(devices, n, timeout=300):
    """The one-process leg in child processes (a failure or a hang there costs that leg, not the line): first with the exchange
    the handle's own self-test selects (csrc/capi_multi.hpp multi_exchange_selftest: host memory vs ncclAllGather compared bit for
    bit at construction; `exchange` and `exchange_note` of the leg say which one won and why), then with the OTHER exchange
    forced (KZG_OPTIONS multi_exchange=host | rccl) - on a list of distinct devices the RCCL one is the north star's collective
    over xGMI."""
    import subprocess
    out = {}
    distinct = len(set(devices.split(","))) == len(devices.split(","))

    def child(opts):
        env = dict(os.environ)
        if opts:
            env["KZG_OPTIONS"] = ";".join(x for x in (env.get("KZG_OPTIONS"), opts) if x)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--single-process-devices", devices, "--blobs", str(n)],
                               capture_output=True, text=True, timeout=timeout, env=env)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{"):
                    return json.loads(line)
            return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-600:])}
        except Exception as e:  # timeout included
            return {"error": repr(e)[:600]}

    out["selected_exchange"] = child(None)
    picked = out["selected_exchange"].get("exchange")
    if picked == "rccl":
        out["host_exchange"] = child("multi_exchange=host")
        out["rccl_exchange"] = {"same_as": "selected_exchange"}
    else:
        out["host_exchange"] = {"same_as": "selected_exchange"} if picked == "host" else child("multi_exchange=host")
        out["rccl_exchange"] = child("multi_exchange=rccl") if distinct else {"skipped": "the device list names a device twice: ncclCommInitAll needs distinct devices"}
    return out


if __name__ == "__main__":
    main()
