"""Build libkzg_rs_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m kzg_rs_amd.build          # regenerates constants / SLP programs, then compiles

hipcc cross-compiles without a GPU; the resulting .so travels with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libkzg_rs_amd.so")
SRC = os.path.join(HERE, "csrc", "kzg_capi.hip")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    csrc = os.path.join(HERE, "csrc")
    data = os.path.join(HERE, "data")
    # 1. generated sources
    gens = [os.path.join(ROOT, "tools", "gen_constants.py"), os.path.join(ROOT, "tools", "gen_mac_chains.py")]
    if force or _newer(os.path.join(csrc, "constants.inc"), [gens[0], os.path.join(ROOT, "tools", "bls_params.py")]):
        subprocess.check_call([sys.executable, gens[0]])
    if force or _newer(os.path.join(csrc, "mac_chains.inc"), [gens[1]]):
        subprocess.check_call([sys.executable, gens[1]])
    slp_dir = os.path.join(HERE, "slp")
    slp_src = [os.path.join(slp_dir, f) for f in ("trace.py", "schedule.py", "gen_pairing.py")]
    slp_src.append(os.path.join(slp_dir, "schedule2.py"))
    if force or any(_newer(os.path.join(data, "slp_%s.bin" % nm), slp_src) for nm in ("verify", "prep", "verify2")):
        subprocess.check_call([sys.executable, "-m", "kzg_rs_amd.slp.gen_pairing"], cwd=ROOT)
    # 2. the library
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(data, "slp_prep.bin"),
                                                               os.path.join(data, "slp_verify.bin"), os.path.join(data, "slp_verify2.bin"),
                                                               os.path.join(ROOT, "include", "kzg_rs_amd.h")]
    if force or _newer(LIB, deps):
        _compile(csrc, data, force, verbose)
    return LIB


# Files of the single translation unit that hold no device code (no __global__ / __device__): a change in one of them
# leaves the gfx950 code object as it is unless it instantiates another kernel template, so the device pass (3.5 of the
# 3.7 minutes of a build) is cached under build/devcache, keyed by the device-side sources and by the kernel
# instantiations the host files launch.  `--force` (and any cold build) runs every step.
HOST_ONLY = ("capi_host_util.hpp", "capi_pieces.hpp", "capi_prover.hpp", "capi_settings.hpp", "capi_verify.hpp",
             "capi_multi.hpp", "capi_pipeline.hpp", "host_only.hpp", "kzg_capi.hip")


def _device_key(csrc, flags):
    import hashlib
    import re
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(os.listdir(csrc)):
        body = open(os.path.join(csrc, f), "rb").read()
        if f in HOST_ONLY:
            if re.search(rb"__global__|__device__|__constant__", body):
                raise SystemExit("build.py: %s is listed as host-only but holds device code" % f)
            # what the device pass can see of a host-only file: the kernels it launches / takes the address of, and constants
            body = b"\n".join(sorted(set(re.findall(rb"\bk_[A-Za-z0-9_]+\s*(?:<[^;(]*?>)?", body) + re.findall(rb"constexpr[^;]*;", body))))
        h.update(f.encode() + b"\0" + body + b"\0")
    return h.hexdigest()[:24]


def _compile(csrc, data, force, verbose):
    cache = os.path.join(ROOT, "build", "devcache")
    os.makedirs(cache, exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-result",
             '-DKZG_DATA_DIR="%s"' % data, "-I", csrc]
    key = _device_key(csrc, flags)
    fb = os.path.join(cache, key + ".hipfb")
    if force or not os.path.exists(fb):
        dev = os.path.join(cache, key + ".out")
        # (--no-gpu-bundle-output: the linked gfx950 code object itself - by default the device-only output is already a bundle)
        cmd = ["hipcc"] + flags + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", dev, SRC]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        bundler = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang-offload-bundler")
        subprocess.check_call([bundler, "-type=o", "-bundle-align=4096",
                               "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                               "-input=/dev/null", "-input=" + dev, "-output=" + fb + ".tmp"])
        os.replace(fb + ".tmp", fb)
        os.remove(dev)
        for old in os.listdir(cache):  # one entry is enough
            if old != os.path.basename(fb):
                os.remove(os.path.join(cache, old))
    host = os.path.join(cache, "host.o")
    subprocess.check_call(["hipcc"] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                                               "-c", "-o", host, SRC])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB + ".tmp", host, "-ldl"])
    os.replace(LIB + ".tmp", LIB)
    os.remove(host)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
