#!/usr/bin/env python3
"""Benchmark of the north-star path: verify_blob_kzg_proof_batch on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blobs 1024]

One "step" = one verify_blob_kzg_proof_batch call over a batch of synthetic blobs already resident
in HBM (BASELINE.json configs[1]: 1 024 blobs of 4096 Fr per GPU).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the batch is N * blobs, sharded by blob, with the two tiny
all-gathers of kzg_rs_amd/distributed.py; per-GPU work is fixed, so scaling is "weak".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_BLOB = 131072
# algorithmic bytes per unit, per kernel (DESIGN.md 5 / SURVEY.md 8d)
ALG_BYTES = {
    "k_blob_challenge": BYTES_PER_BLOB + 48 + 32,   # blob + commitment read, z written       (per blob)
    "k_blob_evaluate": BYTES_PER_BLOB + 32 + 32,    # blob + z read, y written                (per blob)
    "k_g1_decode": 2 * (48 + 96 + 4),               # two points per blob: bytes in, affine + flag out
    "k_msm": 3 * 128 ,                              # three (point, scalar) terms per blob, 96 + 32 B each
    "k_slp_run(pairing)": 0,
}
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blobs", type=int, default=1024, help="blobs per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from kzg_rs_amd import api, synth
    from kzg_rs_amd.distributed import HipBackend, verify_blob_kzg_proof_batch_sharded

    n = args.blobs
    blobs, cs, ps, settings = synth.make_valid_batch(n, seed=1000 + rank)
    d_blobs = torch.from_numpy(blobs).to(dev)
    d_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    d_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    backend = HipBackend(settings)
    shard = (d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)

    def step():
        if world == 1:
            ok = api.KzgProof.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, settings)
        else:
            ok = verify_blob_kzg_proof_batch_sharded(shard, n, backend, dist, dev)
        if not ok:
            raise SystemExit("verification of a valid synthetic batch returned false")

    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kt = [0.0] * 8
    for _ in range(args.steps):
        step()
        tm = settings.last_timings()
        kt = [a + b for a, b in zip(kt, tm)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    K = args.steps
    kt = [x / K for x in kt]  # average ms per launch over the timed region (HIP events on the library's streams)
    kernels = {"k_blob_challenge": kt[5], "k_blob_evaluate": kt[4], "k_g1_decode": kt[6], "k_msm": kt[2], "k_slp_run(pairing)": kt[3]}
    dom = max(kernels, key=kernels.get)
    achieved = ALG_BYTES[dom] * n / (kernels[dom] * 1e-3) / 1e9 if kernels[dom] > 0 else 0.0
    total_blobs = n * world * K
    out = {
        "metric": "blobs/sec verify_blob_kzg_proof_batch",
        "value": round(total_blobs / elapsed, 2),
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
        "config": {"workload": "verify_blob_kzg_proof_batch, %d synthetic blobs (4096 Fr each) per GPU, device-resident, "
                               "known-tau test setup (BASELINE.json configs[1])" % n,
                   "blobs_per_gpu": n, "batch": n * world, "parallelism": "shard-by-blob x%d" % world},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                     "note": "path is integer-ALU / latency bound, not HBM bound (DESIGN.md 5)"},
        "kernel_ms": {k: round(v, 4) for k, v in kernels.items()},
        "device_ms_per_step": round(kt[0], 4),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blobs, cs, ps, synth.synthetic_setup()[1], args.cpu_sample)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
