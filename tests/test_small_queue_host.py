"""The small-call queue of a shared settings handle (csrc/small_queue.hpp: leader / follower coalescing, futex words, requeue,
linger) on the CPU: tests/host/small_queue_main.cpp drives the very submit loop the library runs - with a stand-in for the GPU
launch - from many threads, under ThreadSanitizer and under AddressSanitizer + UBSan.  What a GPU test cannot show as sharply:
no data race, no lost wake-up (a watchdog), every caller its own results, launches within capacity, one launch per lane."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
CSRC = os.path.join(ROOT, "kzg_rs_amd", "csrc")


def _build(tag, flags):
    exe = os.path.join(HOST, "_small_queue_%s" % tag)
    src = os.path.join(HOST, "small_queue_main.cpp")
    deps = [src, os.path.join(CSRC, "small_queue.hpp"), os.path.join(CSRC, "host_only.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-I", CSRC] + flags + ["-o", exe, src])
    return exe


@pytest.mark.parametrize("tag,flags,runs", [
    ("tsan", ["-fsanitize=thread"], [(48, 60, 2), (96, 25, 3), (3, 150, 2)]),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], [(64, 80, 2), (200, 20, 1)]),
])
def test_small_queue_under_sanitizers(tag, flags, runs):
    exe = _build(tag, flags)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    for threads, calls, lanes in runs:
        out = subprocess.run([exe, str(threads), str(calls), str(lanes)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, (threads, calls, lanes, out.stdout[-1500:], out.stderr[-3000:])
        assert "failures 0" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, (out.stdout[-500:], out.stderr[-3000:])
