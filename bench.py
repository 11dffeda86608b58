#!/usr/bin/env python3
"""Benchmark of the north-star path: verify_blob_kzg_proof_batch on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blobs 1024] [--group G] [--inflight F]

One "step" = one verify_blob_kzg_proof_batch over a batch of `--blobs` synthetic blobs per GPU that are already
resident in HBM (BASELINE.json configs[1]: 1 024 blobs of 4096 Fr per GPU).  Every step is a complete, independent
verification (its own transcript, challenge r, MSMs and pairing; its own boolean).  At this batch size each phase
is a latency-bound serial chain that occupies a sliver of the chip, so steps are issued in LAUNCH GROUPS of G
independent batches (a batch dimension inside every kernel) and F groups are kept in flight by a fixed-order
software pipeline on one host thread.  `--group 1 --inflight 1` gives strictly sequential single-batch steps; that
latency is also measured and reported as `single_batch` in the same JSON line.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every batch is N * blobs, sharded by blob, with the
two small all-gathers of kzg_rs_amd/distributed.py (RCCL); per-GPU work is fixed, so scaling is "weak".
Rank 0 prints ONE JSON line.
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

BYTES_PER_BLOB = 131072
# algorithmic bytes per unit, per kernel (DESIGN.md 5 / SURVEY.md 8d)
ALG_BYTES = {
    "k_blob_challenge": BYTES_PER_BLOB + 48 + 32,   # blob + commitment read, z written       (per blob)
    "k_blob_evaluate": BYTES_PER_BLOB + 32 + 32,    # blob + z read, y written                (per blob)
    "k_g1_decode_multiples": 2 * (48 + 96 + 4 + 4 * 128 + 192),  # two points per blob: compressed in; affine, flag, 4 affine table rows, 2^64 P out
    "k_msm": 3 * 128,                               # three (point, scalar) terms per blob, 96 + 32 B each
    "k_slp_run(pairing)": 0,
}
PMC_FILE = "r1h_pmc.json"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable


def cpu_baseline(blobs, cs, ps, tau_g2, max_blobs):
    """The CPU oracle (a C restatement of the reference's operation sequence, oracle/kzg.c) timed on
    the host cores of this box on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    ost = O.Settings.from_tau_g2(tau_g2)
    m = min(len(cs), max_blobs)
    bl = [blobs[i].tobytes() for i in range(m)]
    t = time.perf_counter()
    ok = O.verify_blob_kzg_proof_batch(bl, cs[:m], ps[:m], ost, nthreads=1)
    dt1 = time.perf_counter() - t
    ncores = os.cpu_count() or 1
    t = time.perf_counter()
    ok2 = O.verify_blob_kzg_proof_batch(bl, cs[:m], ps[:m], ost, nthreads=ncores)
    dtn = time.perf_counter() - t
    assert ok and ok2, "oracle rejects the synthetic batch"
    return {
        "value": round(m / dt1, 2), "unit": "blobs/s", "cores": 1, "kind": "port",
        "sample": "%d of the same synthetic blobs, one verify_blob_kzg_proof_batch call, 1 thread "
                  "(the reference is single-threaded); all %d host cores (per-blob loop threaded): %.1f blobs/s"
                  % (m, ncores, m / dtn),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6144)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blobs", type=int, default=1024, help="blobs per GPU per step (batch)")
    ap.add_argument("--group", type=int, default=256, help="independent batches per launch group (batch dimension inside the kernels)")
    ap.add_argument("--inflight", type=int, default=3, help="launch groups kept in flight by the fixed-order software pipeline")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from kzg_rs_amd import api, synth
    from kzg_rs_amd.distributed import HipBackend, PipelinedVerifier, verify_blob_kzg_proof_batch_sharded

    n, G, F = args.blobs, max(1, args.group), max(1, args.inflight)
    blobs, cs, ps, settings = synth.make_valid_batch(n, seed=1000 + rank)
    d_blobs = torch.from_numpy(blobs).to(dev)
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    if os.environ.get("KZG_PMC_CALIBRATE") == "1":  # a 128 MiB device-to-device copy: the known-size dispatch that
        cal = torch.empty_like(d_blobs)              # checks the FETCH_SIZE / WRITE_SIZE scaling in a PMC run
        cal.copy_(d_blobs)
        torch.cuda.synchronize()
        del cal
    backend0 = HipBackend(settings)

    # ---- pipeline: depth (d1, d2, d3) groups between the phases; one handle (2 HIP streams + workspace) per group in flight
    if F <= 1:
        depth = (0, 0, 0)
    elif world > 1:
        depth = (max(1, F - 2), 1, 1)
    else:
        depth = (F - 1, 0, 1)
    n_handles = sum(depth) + 1
    handles = [settings] + [api.KzgSettings.from_tau_g2(synth.synthetic_setup()[1]) for _ in range(n_handles - 1)]
    backends = [backend0] + [HipBackend(h) for h in handles[1:]]
    pipe = PipelinedVerifier(backends, dist, coll_dev, depth, equal_shards=True)
    # every batch is a different permutation of the rank's shard (different transcript and r), at its own HBM address
    gen = torch.Generator(device="cpu").manual_seed(7 + rank)
    c_t, p_t = d_c.view(n, 48), d_p.view(n, 48)
    variants = []
    for v in range(n_handles):
        perms = [torch.randperm(n, generator=gen).to(dev) if (v or g) else torch.arange(n, device=dev) for g in range(G)]
        variants.append((torch.cat([d_blobs[p] for p in perms]).contiguous(), torch.cat([c_t[p] for p in perms]).contiguous(),
                         torch.cat([p_t[p] for p in perms]).contiguous()))
    torch.cuda.synchronize()

    def run_batches(k):
        """k independent batches = ceil(k / G) launch groups (the last one possibly smaller)."""
        groups, left, i = [], k, 0
        while left > 0:
            g = min(G, left)
            v = variants[i % n_handles]
            groups.append(((v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), n), g))
            left -= g
            i += 1
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

    if args.warmup:
        run_batches(args.warmup * G)
    K = args.steps
    for h in handles:
        h.timing_totals(reset=True)  # the warm-up groups do not count
    elapsed, groups = timed(lambda: run_batches(K))
    # kernel times: HIP events on the library's own streams, AVERAGED over every full launch group of the timed region
    # (all handles) - the quantity rocprofv3 --stats reports as the kernel's average duration for the same command
    sums, cnt = [0.0] * 8, 0
    for h in handles:
        t, c = h.timing_totals(reset=True)
        sums = [a + b for a, b in zip(sums, t)]
        cnt += c
    tm = [x / max(cnt, 1) for x in sums]
    g0 = min(G, K)  # batches per launch group the averages refer to (a shorter last group is averaged in; K % G == 0 by default)
    kernels = {"k_blob_challenge": tm[5], "k_blob_evaluate": tm[4], "k_g1_decode_multiples": tm[6], "k_msm": tm[2],
               "k_slp_run(pairing)": tm[3]}

    # ---- strictly sequential single-batch steps (latency), same inputs
    def seq_steps(k):
        for _ in range(k):
            if world == 1:
                ok = api.KzgProof.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, settings)
            else:
                ok = verify_blob_kzg_proof_batch_sharded((d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n), n, backend0, dist, coll_dev)
            if not ok:
                raise SystemExit("verification of a valid synthetic batch returned false")

    seq_steps(2)
    KS = 8
    seq_elapsed, _ = timed(lambda: seq_steps(KS))
    seq_tm = settings.last_timings()
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    # The dominant kernel.  With several launch groups in flight the event-to-event time of a kernel measures how long
    # it SHARED the chip, not what it costs, so dominance is decided by the kernels' stand-alone cost recorded in the
    # committed PMC profile (the largest VALU instruction count: the challenge kernel); its duration is still the live
    # one, measured with HIP events on the library's own stream over the timed region.
    PMC_NAME = {"k_blob_challenge": "kzg::k_blob_challenge", "k_blob_evaluate": "kzg::k_blob_evaluate",
                "k_g1_decode_multiples": "kzg::k_g1_decode_multiples29<4, true>", "k_msm": "kzg::k_msm_window<kzg::Curve29Aff>",
                "k_slp_run(pairing)": "kzg::k_slp_run<false>"}
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))["kernels"]
        dom = max(kernels, key=lambda k: prof.get(PMC_NAME[k], {}).get("SQ_INSTS_VALU", 0))
    except Exception:
        dom = max(kernels, key=kernels.get)
    units = n * g0
    achieved = ALG_BYTES[dom] * units / (kernels[dom] * 1e-3) / 1e9 if kernels[dom] > 0 else 0.0
    # HBM traffic of that kernel and the VALU instruction counts of all kernels, from the PMC passes committed under
    # profiles/ (tools/prof/collect_round.sh; rocprofv3 --pmc cannot run inside this process): per launch of
    # `blobs_per_launch` blobs, scaled to this launch's unit count
    traffic, valu = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        pk = pmc["kernels"].get(PMC_NAME[dom])
        if pk and "hbm_bytes_corrected" in pk:
            traffic = round(pk["hbm_bytes_corrected"] * units / pmc["blobs_per_launch"])
        path = list(PMC_NAME.values()) + ["kzg::k_msm_combine", "kzg::k_msm_combine_lanes", "kzg::k_batch_scalars", "kzg::k_glv_split", "kzg::k_mult_to_affine29",
                                           "kzg::k_eval_powers", "kzg::k_eval_finish"]
        insts = sum(pmc["kernels"][k].get("SQ_INSTS_VALU", 0) for k in path if k in pmc["kernels"])
        per_blob = insts / pmc["blobs_per_launch"]  # wave-instructions per blob, all kernels of the path
        simds, clock = 1024, 2.4e9
        valu = {"wave_insts_per_blob": round(per_blob), "insts_per_cycle_per_simd": round(per_blob * (n * K / elapsed) / (simds * clock), 4),
                "note": "VALU wave-instructions issued per SIMD cycle at the measured throughput (SQ_INSTS_VALU of every kernel of the path, "
                        + PMC_FILE + "); gfx950 issues the path's instruction mix at 2.4-4.3 cycles per wave-instruction "
                        "(profiles/r1_issuebench_valu_issue_cost.txt), i.e. 0.23-0.42 is the ceiling"}
    except Exception:
        pass
    out = {
        "metric": "blobs/sec verify_blob_kzg_proof_batch",
        "value": round(n * world * K / elapsed, 2),
        "unit": "blobs/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (255-bit Fr / 381-bit Fp modular integers)",
        "data": "synthetic",
        "config": {"workload": "verify_blob_kzg_proof_batch, %d synthetic blobs (4096 Fr each) per GPU per step, device-resident, "
                               "known-tau test setup (BASELINE.json configs[1]); every step is an independent batch verification"
                               % n,
                   "blobs_per_gpu": n, "batch": n * world, "parallelism": "shard-by-blob x%d" % world,
                   "batches_per_launch_group": G, "groups_in_flight": F},
        "roofline": {"bound": "hbm", "kernel": dom, "units_per_launch": units, "launch_ms": round(kernels[dom], 4),
                     "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                     "traffic": traffic, "traffic_source": "profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, FETCH x2 gfx950 correction)",
                     "algorithmic_bytes_per_launch": ALG_BYTES[dom] * units,
                     "launch_ms_source": ("the kernel's own execution interval, stamped inside the kernel with s_memrealtime and averaged over "
                                          "the launch groups of the timed region (agrees with rocprofv3 --stats of this command, profiles/"
                                          + PMC_FILE.replace("pmc.json", "kernel_stats.csv") + ")") if dom == "k_blob_challenge" else
                                         "HIP events on the kernel's stream (includes waiting behind the other launch groups' kernels)",
                     "note": "path is integer-ALU / latency bound, not HBM bound (DESIGN.md 5)"},
        "valu": valu,
        "kernel_ms_per_launch_group": {k: round(v, 4) for k, v in kernels.items()},
        "single_batch": {"value": round(n * world * KS / seq_elapsed, 2), "unit": "blobs/s", "ms_per_step": round(seq_elapsed / KS * 1e3, 4),
                         "steps": KS, "kernel_ms": {"k_blob_challenge": round(seq_tm[5], 4), "k_blob_evaluate": round(seq_tm[4], 4),
                                                     "k_g1_decode_multiples": round(seq_tm[6], 4), "k_msm": round(seq_tm[2], 4),
                                                     "k_slp_run(pairing)": round(seq_tm[3], 4)}},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blobs, cs, ps, synth.synthetic_setup()[1], args.cpu_sample)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
