"""The library's HIP-free host code (kzg_rs_amd/csrc/host_only.hpp: SHA-256 in both forms, the trusted-setup text parser,
the batch-transcript hash) built with g++ -fsanitize=address,undefined and exercised by tests/host/host_only_main.cpp: known
answers, 3 000 mutated trusted-setup files (truncated, bad hex, wrong counts, CRLF - the behaviour of build.rs:23-56: a
clean rejection, never a crash), both record layouts of the transcript hash - and its r against the CPU oracle's compute_r.
The same vectors also run through the ASan build of the oracle itself (liboracle_asan.so).  CPU only."""
import os
import subprocess
import sys

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_only_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_only_main")
    csrc = os.path.join(ROOT, "kzg_rs_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                           "-I", csrc, "-o", exe, os.path.join(HERE, "host", "host_only_main.cpp")])
    out = subprocess.run([exe, os.path.join(ROOT, "kzg_rs_amd", "data", "trusted_setup.txt"), "12345", "3000"], capture_output=True, text=True,
                         timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    lines = dict(ln.split(" ", 1) for ln in out.stdout.strip().splitlines() if " " in ln)
    assert lines["failures"] == "0"
    mut = lines["parser"].split()
    assert int(mut[1]) == 3000 and int(mut[3]) >= 1500 and int(mut[5]) >= 600  # most mutations rejected, CRLF / trailing-line forms accepted
    # the transcript hash against the oracle: batch 0 = 21 records
    recs = bytes.fromhex(lines["records"])
    n = len(recs) // 160
    rec = [recs[160 * i: 160 * i + 160] for i in range(n)]
    want = O.compute_r(b"".join(x[:48] for x in rec), b"".join(x[48:80][::-1] for x in rec), b"".join(x[80:112][::-1] for x in rec),
                       b"".join(x[112:] for x in rec), n)
    assert bytes.fromhex(lines["r0"])[::-1] == want


def test_oracle_vectors_under_asan():
    """tests/test_oracle_vectors.py once more in a child process whose oracle is the -fsanitize=address build
    (oracle/Makefile: liboracle_asan.so; libasan preloaded because the interpreter itself is not instrumented)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", KZG_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_oracle_vectors.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "passed" in out.stdout and "AddressSanitizer" not in out.stderr
