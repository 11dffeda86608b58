"""The N > 1 product path on real hardware: 2 and 3 ranks (one process each, all on cuda:0 of the test box, gloo for
the two small exchanges) run the HIP shard phases of include/kzg_rs_amd.h through kzg_rs_amd/distributed.py and must
agree with the unsharded oracle - the same scenarios as the CPU choreography tests of test_distributed_cpu.py (which
use a backend made of oracle primitives), here with HipBackend: power offsets r^offset per rank, the 288-byte Jacobian
partials, the fold kernel and the pairing.  On the 8-GPU node the only difference is the transport (RCCL)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _device_shard(torch, blobs, cs, ps):
    """-> (keepalive tensors, (d_blobs, d_commitments, d_proofs, n)); an empty shard still gets valid pointers"""
    n = len(blobs)
    t = [torch.frombuffer(bytearray(b"".join(x) or b"\0"), dtype=torch.uint8).cuda() for x in (blobs, cs, ps)]
    torch.cuda.synchronize()
    return t, (t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n)


def _worker(rank, world, port, scenario, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd import api
    from kzg_rs_amd.distributed import HipBackend, PipelinedVerifier, verify_blob_kzg_proof_batch_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ost = O.Settings.mainnet()
    st = api.KzgSettings.load_trusted_setup_file()
    tuples = G.valid_blob_tuples()  # 7 valid mainnet (blob, C, pi)
    blobs, cs, ps = [list(x) for x in zip(*tuples)]
    if scenario == "bad_proof":
        ps[5] = O.g1_add(ps[5], G1_GEN)
    if scenario == "bad_blob":
        b = bytearray(blobs[6])
        b[64:96] = R.to_bytes(32, "big")
        blobs[6] = bytes(b)
    n = len(blobs)
    if scenario.startswith("pipelined"):
        # three launch groups (2, 1, 2 batches) of 6-blob batches, every rank holding 6 / world blobs of each batch
        def batch(rot, corrupt=False):
            t = (tuples[rot:] + tuples[:rot])[:6]
            bl, c, p = [list(x) for x in zip(*t)]
            if corrupt:
                p[4] = O.g1_add(p[4], G1_GEN)
            return bl, c, p
        def invalid(rot, where):  # a non-canonical field element in one blob (in the LAST rank's slice): Err in the reference
            bl, c, p = batch(rot)
            b = bytearray(bl[where])
            b[64:96] = R.to_bytes(32, "big")
            bl[where] = bytes(b)
            return bl, c, p
        # groups of 2, 1 and 3 batches: with 2 ranks the first, with 3 ranks the last is divisible by the world size (the
        # hash-once all-to-all exchange of the bulk path); the others take the all-gather form
        plan = [[batch(0), batch(1, True)], [batch(2)], [batch(3), invalid(4, 5), batch(5)]]
        per = 6 // world
        keep, groups, want = [], [], []
        for grp in plan:
            gb, gc, gp = [], [], []
            for bl, c, p in grp:
                try:
                    want.append(O.verify_blob_kzg_proof_batch(bl, c, p, ost))
                except O.OracleError:
                    want.append(None)
                sl = slice(rank * per, (rank + 1) * per)
                gb += bl[sl]; gc += c[sl]; gp += p[sl]
            t, (db, dc, dp, _) = _device_shard(torch, gb, gc, gp)
            keep.append(t)
            groups.append(((db, dc, dp, per), len(grp)))
        handles = [st] + [api.KzgSettings.load_trusted_setup_file() for _ in range(2)]
        pipe = PipelinedVerifier([HipBackend(h) for h in handles], dist, "cpu", (1, 0, 1), equal_shards=scenario == "pipelined_bulk")
        got = [x for res in pipe.run(groups) for x in res]
        q.put((rank, got, want))
    else:
        if scenario == "uneven":
            bounds = [0, 1, n] if world == 2 else [0, 1, 1, n]  # an empty shard in the 3-rank case
        else:
            bounds = [n * k // world for k in range(world + 1)]
        lo, hi = bounds[rank], bounds[rank + 1]
        keep, shard = _device_shard(torch, blobs[lo:hi], cs[lo:hi], ps[lo:hi])
        try:
            got = verify_blob_kzg_proof_batch_sharded(shard, hi - lo, HipBackend(st), dist, "cpu")
        except api.KzgError:
            got = "error"
        try:
            want = O.verify_blob_kzg_proof_batch(blobs, cs, ps, ost)
        except O.OracleError:
            want = "error"
        q.put((rank, got, want))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,scenario", [(2, "valid"), (2, "bad_proof"), (2, "bad_blob"), (2, "uneven"), (3, "uneven")])
def test_sharded_hip_path_matches_oracle(world, scenario):
    expected = {"valid": True, "uneven": True, "bad_proof": False, "bad_blob": "error"}[scenario]
    for rank, got, want in _run(world, scenario):
        assert got == want == expected, (rank, got, want)


@pytest.mark.parametrize("world,scenario", [(2, "pipelined"), (3, "pipelined"), (2, "pipelined_bulk"), (3, "pipelined_bulk")])
def test_pipelined_groups_hip_path(world, scenario):
    """Launch groups through the fixed-order pipeline: the byte-string exchange (any shard sizes) and the bulk exchange
    (equal shards: records leave the library in device memory, come back as one gathered buffer)."""
    for rank, got, want in _run(world, scenario):
        assert got == want == [True, False, True, True, None, True], (rank, got, want)


def test_bench_script_two_ranks_shared_gpu():
    """bench.py's N > 1 code path (fixed-order pipeline, bulk exchange, sharded single-batch leg, max-over-ranks timing)
    with two ranks sharing the test box's one GPU over gloo (KZG_BENCH_SHARE_GPU=1): must print one JSON line for
    n_gpus = 2.  The driver's 8-GPU run differs only in the transport (RCCL) and in one GPU per rank."""
    import json
    import subprocess
    env = dict(os.environ, KZG_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--group", "16", "--no-cpu-baseline", "--workload", "configs1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["config"]["batch"] == 2048
    assert d["config"]["batches_per_step"] == 16 and d["multi_gpu"]["ranks"] == 2
    mg = d["multi_gpu"]
    # the run describes itself: which exchange ran over which transport, every rank's stage split, the negative control through the pipeline
    assert "all_to_all_single" in mg["exchange_ran"]["exchange1"] and "288 B" in mg["exchange_ran"]["exchange2"] and "gloo" in mg["exchange_ran"]["transport"]
    assert [r["rank"] for r in mg["per_rank_stage_ms_per_step"]] == [0, 1] and all(r["groups"] >= 4 and r["r_hash_ms"] > 0 for r in mg["per_rank_stage_ms_per_step"])
    assert mg["pipe_check"]["passed"] is True and mg["pipe_check"]["results"].count(False) == 1 and mg["pipe_check"]["results"][8] is False
    assert d["preflight"]["shrunk"] is False and d["preflight"]["batches_per_step"] == 16
    assert d["roofline"]["frac_path"] > 0 and "inputs" in d["roofline"]


def test_bench_shard_leg_world_of_one():
    """`bench.py --workload config5 --force-collectives` on ONE GPU (what the default run's configs.config5_shard leg starts as a child):
    BASELINE configs[4]'s shard shape through PipelinedVerifier with the exchanges of a world of one rank over RCCL - here at a reduced
    size (2 048-blob shards, 4 batches per step) - prints the standard line with the stage split, the exchange that ran and the
    poisoned batch's false."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "config5", "--force-collectives", "--steps", "3", "--warmup", "1",
           "--blobs", "2048", "--group", "4", "--no-latency", "--no-self-check", "--no-cpu-baseline", "--no-configs", "--no-concurrent"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["blobs_per_step"] == 4 * 2048
    assert "PipelinedVerifier" in d["config"]["entry_point"] and "world of ONE rank" in d["config"]["entry_point"]
    mg = d["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["exchange_ran"]["world_of_one"] is True and "RCCL" in mg["exchange_ran"]["transport"]
    assert mg["pipe_check"]["passed"] is True and mg["pipe_check"]["results"] == [True, True, False, True]
    assert mg["per_rank_stage_ms_per_step"][0]["r_hash_ms"] > 0


def test_bench_script_bare_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` WITHOUT a launcher (no WORLD_SIZE in the environment): the script must start the two
    ranks itself - fresh child processes, before anything touched the GPU - run the same N > 1 path as under
    torch.distributed.run and print rank 0's one JSON line (KZG_BENCH_SHARE_GPU=1: both ranks on the box's one GPU, gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["KZG_BENCH_SHARE_GPU"] = "1"
    env["KZG_BENCH_SINGLE_PROCESS"] = "force"  # also the one-process leg (a child process over the device list "0,0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--group", "4", "--blobs", "256",
           "--no-cpu-baseline", "--no-latency", "--workload", "configs1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["multi_gpu"]["ranks"] == 2
    assert d["self_check"]["passed"] is True
    sp = d["multi_gpu"]["single_process"]["selected_exchange"]  # (the shared-GPU rig names device 0 twice: host memory is what the self-test can select)
    assert d["multi_gpu"]["single_process"]["host_exchange"] == {"same_as": "selected_exchange"}
    assert sp["batch"] == 512 and sp["devices"] == [0, 0] and sp["exchange"] == "host", sp
    assert sp["sharded_batch"]["ok"] is True and sp["sharded_batch"]["corrupted_proof_on_last_device"] is False, sp
    assert sp["sharded_stream"]["results_as_expected"] is True and sp["sharded_stream"]["in_flight"] == 4, sp
    assert sp["host_vec_blob"]["ok"] is True
    assert "skipped" in d["multi_gpu"]["single_process"]["rccl_exchange"]  # (the shared-GPU rig names device 0 twice)
    sel = d["multi_gpu"]["single_process"]["selected_exchange"]  # what the handle's own self-test picked, and why
    assert sel["exchange"] == "host" and "names a device twice" in sel["exchange_note"] and sel["sharded_batch"]["ok"] is True, sel


def _nccl_worker(port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd import api
    from kzg_rs_amd.distributed import HipBackend, PipelinedVerifier, verify_blob_kzg_proof_batch_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ost = O.Settings.mainnet()
    st = api.KzgSettings.load_trusted_setup_file()
    tuples = G.valid_blob_tuples()
    blobs, cs, ps = [list(x) for x in zip(*tuples)]
    keep, shard = _device_shard(torch, blobs, cs, ps)
    got = [verify_blob_kzg_proof_batch_sharded(shard, len(blobs), HipBackend(st), dist, dev, force_collectives=True)]
    want = [O.verify_blob_kzg_proof_batch(blobs, cs, ps, ost)]
    bad = list(ps)
    bad[2] = O.g1_add(ps[2], G1_GEN)
    keep2, shard2 = _device_shard(torch, blobs + blobs, cs + cs, ps + bad)  # a group of two batches: valid, corrupted
    for bulk in (False, True):
        handles = [st] + [api.KzgSettings.load_trusted_setup_file() for _ in range(3)]
        pipe = PipelinedVerifier([HipBackend(h) for h in handles], dist, dev, (1, 1, 1), equal_shards=bulk, force_collectives=True)
        res = pipe.run([((shard2[0], shard2[1], shard2[2], len(blobs)), 2), ((shard[0], shard[1], shard[2], len(blobs)), 1)])
        got += [x for r in res for x in r]
        want += [True, False, True]
    q.put((got, want))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_transport_world_of_one():
    """The exchanges over RCCL itself (backend "nccl"), forced in a world of one rank: all_reduce of the error flag,
    all_gather of byte strings and of the bulk record buffer on device tensors, barrier.  What a one-GPU box can check of
    the 8-GPU transport: the calls, dtypes and buffer handling - not the wire."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    got, want = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got == want == [True, True, False, True, True, False, True]
