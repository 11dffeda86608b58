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
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wno-unused-result",
               '-DKZG_DATA_DIR="%s"' % data, "-I", csrc, "-o", LIB, SRC]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
