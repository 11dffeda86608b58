"""Kernel resource table of the built library: VGPRs (arch + accumulation), SGPRs, static LDS, scratch, spills and the wavefronts
per SIMD the register budget allows, for every kernel of the PRODUCT code object - read from the gfx950 code object's own metadata
(the .note section of the ELF inside libkzg_rs_amd.so's fat binary), so the table describes the library that ships, not a compile log.
    python3 tools/prof/kernel_resources.py [path/to/lib.so] > profiles/r6_kernel_resources.txt
gfx950: 512 unified registers (VGPR + AGPR) per lane and SIMD, allocated in blocks of 8; at most 8 wavefronts per SIMD; 160 KB of
LDS per CU (MI355X_MICROARCH.md).  Occupancy by LDS depends on the launch's dynamic LDS and is noted in DESIGN.md per kernel."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")


def code_object(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    elf = os.path.join(tmp, "dev.elf")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-targets=hipv4-amdgcn-amd-amdhsa--gfx950", "-input=" + fat,
                           "-output=" + elf, "-unbundle"])
    return elf


def kernels(elf):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        f = lambda k, b=blk: re.search(r"\.%s:\s+(\S+)" % k, b)
        name = f("name").group(1)
        g = lambda k, b=blk: int(f(k, b).group(1)) if f(k, b) else 0
        out.append({"name": name, "vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"), "lds": g("group_segment_fixed_size"),
                    "scratch": g("private_segment_fixed_size"), "vspill": g("vgpr_spill_count"), "sspill": g("sgpr_spill_count"),
                    "wg": g("max_flat_workgroup_size")})
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    short = []
    for d in r:
        d = re.sub(r"\(.*$", "", d)            # drop the argument list
        d = re.sub(r"^void ", "", d)
        short.append(d)
    return short


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        ks = kernels(code_object(lib, tmp))
    names = demangle([k["name"] for k in ks])
    sys.path.insert(0, ROOT)
    try:
        from kzg_rs_amd import build
        key = build.kernel_key()
    except Exception:
        key = "?"
    print("# kernel resources of %s (gfx950 code object metadata); kernel key %s" % (os.path.relpath(lib, ROOT), key))
    print("# waves/SIMD = min(8, 512 // roundup(vgpr + agpr, 8)) - the register bound; LDS = static bytes only (dynamic LDS is per launch)")
    print("%-64s %5s %5s %5s %7s %8s %6s %6s %10s" % ("kernel", "vgpr", "agpr", "sgpr", "lds_B", "scratchB", "vspill", "sspill", "waves/SIMD"))
    for k, nm in sorted(zip(ks, names), key=lambda x: x[1]):
        regs = (k["vgpr"] + k["agpr"] + 7) // 8 * 8
        waves = min(8, 512 // regs) if regs else 8
        print("%-64s %5d %5d %5d %7d %8d %6d %6d %10d" % (nm[:64], k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"], k["vspill"], k["sspill"], waves))


if __name__ == "__main__":
    main()
