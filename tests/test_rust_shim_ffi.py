"""rust/kzg-rs-amd cannot be compiled here (no Rust toolchain): what CAN be checked mechanically is that its `extern "C"`
block binds the header - every function it declares exists in include/kzg_rs_amd.h with the same number of arguments and
in the built library - that its return-code constants equal the header's enum, and that its numeric constants equal the
ones the tests already pin (r, the 4096th root of unity, the sizes)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "kzg-rs-amd")
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
OMEGA = 0x564C0A11A0F704F4FC3E8ACFE0F8245F0AD1347B378FBF96E206DA11A5D36306  # SURVEY.md 9 (SCALE2_ROOT_OF_UNITY[12])


def _header_functions():
    h = open(os.path.join(ROOT, "include", "kzg_rs_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:KzgRet|void|const char \*)\s*\*?(kzg_\w+)\(([^;]*?)\);", h, re.S):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = len(args)
    return out


def _c_arg_to_rust(arg):
    """`const uint8_t commitment[48]` -> `*const u8`, `KzgSettings **out` -> `*mut *mut RawSettings`, `size_t n` -> `usize` ...: the Rust
    FFI type a C parameter of include/kzg_rs_amd.h must be bound with (arrays decay to pointers; `const` binds to the pointee)."""
    a = " ".join(arg.replace("*", " * ").split())
    depth = a.count("*") + (1 if "[" in a else 0)
    a = re.sub(r"\[.*?\]", "", a)
    toks = [t for t in a.split() if t != "*"]
    const = "const" in toks
    toks = [t for t in toks if t != "const"]
    base = toks[0] if toks[0] != "unsigned" else " ".join(toks[:2])
    rust = {"uint8_t": "u8", "bool": "bool", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void", "float": "f32", "double": "f64",
            "KzgSettings": "RawSettings", "uint64_t": "u64", "uint32_t": "u32", "unsigned long long": "u64"}[base]
    if depth == 0:
        return rust
    inner = ("*const " if const else "*mut ") + rust
    return "*mut " * (depth - 1) + inner


def _header_signatures():
    h = open(os.path.join(ROOT, "include", "kzg_rs_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    out = {}
    for m in re.finditer(r"(KzgRet|void|const char \*)\s*\*?(kzg_\w+)\(([^;]*?)\);", h, re.S):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        ret = {"KzgRet": "c_int", "void": "", "const char *": "*const c_char"}[m.group(1)]
        out[m.group(2)] = ([_c_arg_to_rust(a) for a in args], ret)
    return out


def test_c_to_rust_type_mapping():
    assert _c_arg_to_rust("const uint8_t commitment[48]") == "*const u8"
    assert _c_arg_to_rust("uint8_t out[48]") == "*mut u8"
    assert _c_arg_to_rust("KzgSettings **out") == "*mut *mut RawSettings"
    assert _c_arg_to_rust("const KzgSettings *s") == "*const RawSettings"
    assert _c_arg_to_rust("const char *txt") == "*const c_char"
    assert _c_arg_to_rust("size_t n") == "usize" and _c_arg_to_rust("bool *ok") == "*mut bool"
    assert _c_arg_to_rust("const int *devices") == "*const c_int" and _c_arg_to_rust("size_t *n_devices") == "*mut usize"


def test_extern_block_argument_types_match_the_header():
    """Round 5 compared argument COUNTS only: a `*const u8` bound as `*const c_void`, a `usize` as `c_int` or a `*mut bool` as
    `*mut u8` would have passed.  Every parameter and the return type of every function the shim declares must be the Rust
    spelling of the header's C type."""
    sig = _header_signatures()
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', ffi, re.S).group(1)
    decl = re.findall(r"pub fn (kzg_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, re.S)
    assert len(decl) >= 18
    for name, args, ret in decl:
        want_args, want_ret = sig[name]
        got = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        assert got == want_args, (name, got, want_args)
        assert " ".join((ret or "").split()) == want_ret, (name, ret, want_ret)


def test_extern_block_matches_the_header_and_the_library():
    from kzg_rs_amd import build
    build.build()
    hd = _header_functions()
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    decl = re.findall(r"pub fn (kzg_\w+)\((.*?)\)", ffi, re.S)
    assert len(decl) >= 14
    L = ctypes.CDLL(os.path.join(ROOT, "kzg_rs_amd", "libkzg_rs_amd.so"))
    for name, args in decl:
        n = len([a for a in args.split(",") if a.strip()])
        assert name in hd, name
        assert hd[name] == n, (name, hd[name], n)
        assert hasattr(L, name), name


def test_return_codes_and_constants():
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    h = open(os.path.join(ROOT, "include", "kzg_rs_amd.h")).read()
    for name in ("KZG_OK", "KZG_BADARGS", "KZG_ERROR", "KZG_MALLOC", "KZG_INVALID_LENGTH", "KZG_BAD_SETUP"):
        hv = int(re.search(name + r"\s*=\s*(\d+)", h).group(1))
        rv = int(re.search(r"pub const " + name + r": c_int = (\d+);", ffi).group(1))
        assert hv == rv, name
    c = open(os.path.join(CRATE, "src", "consts.rs")).read()

    def limbs(name):
        body = re.search(name + r": \[u64; 4\] = \[(.*?)\];", c, re.S).group(1)
        v = [int(x.replace("_", ""), 16) for x in re.findall(r"0x[0-9a-f_]+", body)]
        return sum(x << (64 * i) for i, x in enumerate(v))

    assert limbs("MODULUS") == R
    assert limbs("PRIMITIVE_ROOT_OF_UNITY_4096") == OMEGA
    assert pow(OMEGA, 4096, R) == 1 and pow(OMEGA, 2048, R) != 1
    for name, val in (("BYTES_PER_FIELD_ELEMENT", 32), ("NUM_G1_POINTS", 4096), ("NUM_G2_POINTS", 65), ("BYTES_PER_COMMITMENT", 48),
                      ("BYTES_PER_PROOF", 48), ("BYTES_PER_G2_POINT", 96), ("NUM_FIELD_ELEMENTS_PER_BLOB", 4096)):
        assert re.search(r"pub const %s: usize = %d;" % (name, val), c), name
    assert os.path.exists(os.path.join(CRATE, "src", "../../../kzg_rs_amd/data/trusted_setup.txt"))  # the include_str! target
