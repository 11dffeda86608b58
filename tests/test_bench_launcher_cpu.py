"""`python3 bench.py --gpus N` from a bare command: the launcher half (no GPU needed).  The script must start the N ranks
itself as fresh child processes - before importing torch or touching a GPU - hand each its RANK / LOCAL_RANK / WORLD_SIZE and
a 127.0.0.1 rendezvous, show rank 0's output only, and return the worst child return code.  (KZG_BENCH_DRY_SPAWN=1 makes the
ranks report their environment instead of benchmarking; the real two-rank run is tests/test_gpu_multirank.py.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True, timeout=120)


def test_bare_command_spawns_the_ranks():
    r = _run({"KZG_BENCH_DRY_SPAWN": "1"}, "--gpus", "3", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["LOCAL_RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and int(d["MASTER_PORT"]) > 0


def test_single_gpu_form_does_not_spawn():
    r = _run({"KZG_BENCH_DRY_SPAWN": "1"}, "--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["WORLD_SIZE"] is None and d["RANK"] is None


def test_child_failure_is_the_launcher_s_return_code():
    # a rank that dies (here: an argument error in every child) must not look like success
    r = _run({"KZG_BENCH_DRY_SPAWN": "1"}, "--gpus", "2", "--workload", "nonsense")
    assert r.returncode != 0


def test_launched_rank_keeps_the_torchrun_contract():
    # under a launcher (WORLD_SIZE set) the script must NOT spawn again
    r = _run({"KZG_BENCH_DRY_SPAWN": "1", "WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"},
             "--gpus", "2")
    assert r.returncode == 0
    assert json.loads(r.stdout.strip())["RANK"] == "1"
