"""Build the HIP library (kernels + C ABI) for gfx950, in-tree.

    python -m kzg_rs_amd.build          # regenerates constants / SLP programs, then compiles

Two libraries come out of the one source tree:
    libkzg_rs_amd.so      the PRODUCT: -DKZG_AB_VARIANTS=0, only the kernel forms the dispatch by launch size selects
    libkzg_rs_amd_ab.so   the A/B build: -DKZG_AB_VARIANTS=1 adds the alternative forms kept for measurement and for the
                          differential fuzz (12x32-limb point kernels, the 8x32 evaluation, the 16-chunk proofs layout),
                          selected through KZG_OPTIONS (csrc/capi_host_util.hpp); tests load it through KZG_LIB_OVERRIDE
hipcc cross-compiles without a GPU; the resulting .so files travel with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libkzg_rs_amd.so")
LIB_AB = os.path.join(HERE, "libkzg_rs_amd_ab.so")
SRC = os.path.join(HERE, "csrc", "kzg_capi.hip")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variants=(0, 1)):
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
    if force or any(_newer(os.path.join(data, "slp_%s.bin" % nm), slp_src) for nm in ("verify", "prep", "verify2", "scalars", "verify3")):
        subprocess.check_call([sys.executable, "-m", "kzg_rs_amd.slp.gen_pairing"], cwd=ROOT)
    fb_gen = os.path.join(ROOT, "tools", "gen_fixed_base.py")
    if force or _newer(os.path.join(data, "fixed_base.bin"), [fb_gen, os.path.join(ROOT, "tools", "bls_params.py")]):
        subprocess.check_call([sys.executable, fb_gen])
    # 2. the libraries (the two device passes side by side: each is one hipcc process)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(data, "slp_prep.bin"),
                                                               os.path.join(data, "slp_verify.bin"), os.path.join(data, "slp_verify2.bin"),
                                                               os.path.join(data, "slp_scalars.bin"), os.path.join(data, "slp_verify3.bin"), os.path.join(data, "fixed_base.bin"),
                                                               os.path.join(ROOT, "include", "kzg_rs_amd.h")]
    todo = [v for v in variants if force or _newer(LIB_AB if v else LIB, deps)]
    if len(todo) > 1:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(len(todo)) as ex:
            for f in [ex.submit(_compile, csrc, data, force, verbose, v) for v in todo]:
                f.result()
    elif todo:
        _compile(csrc, data, force, verbose, todo[0])
    return LIB


# Files of the single translation unit that hold no device code (no __global__ / __device__): a change in one of them
# leaves the gfx950 code object as it is unless it instantiates another kernel template, so the device pass (3.5 of the
# 3.7 minutes of a build) is cached under build/devcache, keyed by the device-side sources and by what the device pass can
# see of the host files: the kernels and launch helpers they name (with their template arguments), their preprocessor
# lines and constants.  The key is a heuristic, so the link step CHECKS it: every kernel the host object launches (its
# __device_stub__ symbols) must be defined in the cached code object, else the device pass runs again.
# `--force` (and any cold build) runs every step.
HOST_ONLY = ("capi_host_util.hpp", "capi_pieces.hpp", "capi_prover.hpp", "capi_settings.hpp", "capi_verify.hpp",
             "capi_multi.hpp", "capi_coalesce.hpp", "small_queue.hpp", "dyn_lds.hpp", "capi_pipeline.hpp", "host_only.hpp", "kzg_capi.hip")
LLVM_BIN = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")


def _device_key(csrc, flags):
    import hashlib
    import re
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(os.listdir(csrc)):
        body = open(os.path.join(csrc, f), "rb").read()
        if f in HOST_ONLY:
            if re.search(rb"__global__|__device__|__constant__", body):
                raise SystemExit("build.py: %s is listed as host-only but holds device code" % f)
            seen = re.findall(rb"\b(?:k_[A-Za-z0-9_]+|msm_window_launch|launch_program2|run_program2?)\s*(?:<[^;(]*?>)?", body)
            seen += re.findall(rb"constexpr[^;]*;", body) + re.findall(rb"(?m)^\s*#\s*(?:if|ifdef|ifndef|elif|else|endif|define|undef)\b[^\n]*", body)
            body = b"\n".join(sorted(set(seen)))
        h.update(f.encode() + b"\0" + body + b"\0")
    return h.hexdigest()[:24]


def kernel_key(variant=0):
    """What identifies the KERNELS of a build, wherever the tree lies: the device-side sources in full plus what the device pass sees
    of the host-only files (_device_key), without the absolute paths of the compile flags.  PMC profiles under profiles/ are stamped
    with it (tools/prof/pmc_to_json.py) and bench.py flags a profile whose stamp differs from the tree it runs in."""
    return _device_key(os.path.join(HERE, "csrc"), ["kernel-key", "-DKZG_AB_VARIANTS=%d" % variant])


def _kernel_names(obj, stubs):
    """demangled kernel names an object defines (device code object) or launches (host object: its __device_stub__ symbols)"""
    out = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "-sW", "--demangle", obj], capture_output=True, text=True, check=True).stdout
    names = set()
    for ln in out.splitlines():
        f = ln.split(None, 7)
        if len(f) < 8 or not f[0].rstrip(":").isdigit() or f[6] == "UND" or f[3] != "FUNC":
            continue
        name = f[7].strip()
        if stubs:
            if "__device_stub__" in name:
                names.add(name.replace("__device_stub__", ""))
        else:
            names.add(name)
    return names


def _device_pass(flags, dev, fb, verbose):
    # (--no-gpu-bundle-output: the linked gfx950 code object itself - by default the device-only output is already a bundle)
    cmd = ["hipcc"] + flags + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", dev, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
                           "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                           "-input=/dev/null", "-input=" + dev, "-output=" + fb + ".tmp"])
    with open(fb + ".syms", "w") as f:
        f.write("\n".join(sorted(_kernel_names(dev, stubs=False))))
    os.replace(fb + ".tmp", fb)
    os.remove(dev)


def _compile(csrc, data, force, verbose, variant=0):
    cache = os.path.join(ROOT, "build", "devcache")
    os.makedirs(cache, exist_ok=True)
    lib = LIB_AB if variant else LIB
    tag = "ab%d-" % variant
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-result",
             "-DKZG_AB_VARIANTS=%d" % variant, '-DKZG_DATA_DIR="%s"' % data, "-I", csrc]
    key = _device_key(csrc, flags)
    fb = os.path.join(cache, tag + key + ".hipfb")
    dev = os.path.join(cache, tag + key + ".out")
    fresh = force or not (os.path.exists(fb) and os.path.exists(fb + ".syms"))
    if fresh:
        _device_pass(flags, dev, fb, verbose)
    host = os.path.join(cache, tag + "host.o")

    def host_pass():
        subprocess.check_call(["hipcc"] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                                                   "-c", "-o", host, SRC])
        return _kernel_names(host, stubs=True) - set(open(fb + ".syms").read().splitlines())

    missing = host_pass()
    if missing and not fresh:  # the key missed a new instantiation: the cached code object lacks kernels the host launches
        print("build.py: cached device code lacks %d kernel(s) (%s ...): running the device pass" % (len(missing), sorted(missing)[0]))
        _device_pass(flags, dev, fb, verbose)
        missing = host_pass()
    if missing:
        raise SystemExit("build.py: the device code object does not define: %s" % ", ".join(sorted(missing)))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib + ".tmp", host, "-ldl"])
    os.replace(lib + ".tmp", lib)
    os.remove(host)
    for old in os.listdir(cache):  # one entry per variant is enough
        if old.startswith(tag) and not old.startswith(tag + key):
            os.remove(os.path.join(cache, old))
        elif not old.startswith("ab"):
            os.remove(os.path.join(cache, old))


if __name__ == "__main__":
    only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--variant=")]
    build(force="--force" in sys.argv, verbose="-v" in sys.argv, variants=tuple(only) or (0, 1))
    print(LIB)
