"""Host-side mirror of the reference's public API (succinctlabs/kzg-rs v0.2.8, src/lib.rs:12-18)
over the C ABI of libkzg_rs_amd.so (include/kzg_rs_amd.h).

Same names, argument meaning and error behaviour as the Rust crate so parity tests read like the
reference's own tests (src/kzg_proof.rs:604-737):

    Bytes32 / Bytes48 / Blob          src/dtypes.rs:7-57      (from_slice length check)
    KzgError                          src/enums.rs:6-18
    KzgSettings.load_trusted_setup_file()   src/trusted_setup.rs:94-98
    KzgProof.verify_kzg_proof / verify_blob_kzg_proof / verify_blob_kzg_proof_batch
                                      src/kzg_proof.rs:353-525

This module is a ctypes binding only: there is no Python or CPU implementation behind it.  If the
HIP library is missing, or no gfx950 device is usable, every call raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZG_LIB_OVERRIDE") or os.path.join(HERE, "libkzg_rs_amd.so")
# the A/B build (python -m kzg_rs_amd.build): the product plus the alternative kernel forms kept for measurement and for the
# differential fuzz; a process selects it with KZG_LIB_OVERRIDE=LIB_AB_PATH and a form with KZG_OPTIONS (options_string)
LIB_AB_PATH = os.path.join(HERE, "libkzg_rs_amd_ab.so")
TRUSTED_SETUP_PATH = os.path.join(HERE, "data", "trusted_setup.txt")

BYTES_PER_FIELD_ELEMENT = 32
FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_BLOB = 131072
BYTES_PER_COMMITMENT = 48
BYTES_PER_PROOF = 48

KZG_OK, KZG_BADARGS, KZG_ERROR, KZG_MALLOC, KZG_INVALID_LENGTH, KZG_BAD_SETUP = range(6)


class KzgError(Exception):
    """src/enums.rs:6-18.  `.kind` is the variant name."""

    def __init__(self, kind, msg=""):
        super().__init__("%s: %s" % (kind, msg))
        self.kind = kind
        self.msg = msg


def BadArgs(msg):
    return KzgError("BadArgs", msg)


def InvalidBytesLength(msg):
    return KzgError("InvalidBytesLength", msg)


def options_string(**kw):
    """The value of KZG_OPTIONS for the given switches (csrc/capi_host_util.hpp): options_string(single_stream=1,
    challenge_kernel="lane") -> "single_stream=1;challenge_kernel=lane"."""
    return ";".join("%s=%s" % (k, v) for k, v in kw.items())


class options:
    """with api.options(multi_min_blobs=2): ... - KZG_OPTIONS of THIS process for the duration of the block (added to what
    is already set).  The library re-reads the string whenever it has changed; switches that are read when a handle is made
    (single_stream, multi_*) apply to handles made inside the block, switches latched on first use only if this is it."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.prev = os.environ.get("KZG_OPTIONS")
        os.environ["KZG_OPTIONS"] = ";".join(x for x in (self.prev, options_string(**self.kw)) if x)
        return self

    def __exit__(self, *exc):
        if self.prev is None:
            os.environ.pop("KZG_OPTIONS", None)
        else:
            os.environ["KZG_OPTIONS"] = self.prev
        return False


_KIND = {KZG_BADARGS: "BadArgs", KZG_ERROR: "InternalError", KZG_MALLOC: "InternalError",
         KZG_INVALID_LENGTH: "InvalidBytesLength", KZG_BAD_SETUP: "InvalidTrustedSetup"}

_lib = None


def lib():
    """Load libkzg_rs_amd.so.  No fallback: a missing library is a hard error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KzgError("InternalError", "libkzg_rs_amd.so is not built (python -m kzg_rs_amd.build); "
                                            "there is no CPU fallback")
        # The HOST side of the boundary asks for 8 HIP hardware queues (ROCm's default is 4) before the HIP runtime starts, as
        # INTEGRATION.md tells a Rust host to: the launch-group pipeline and the small-call lanes (3 streams each) want them.
        # A value the process already has is kept; a runtime that is already initialised ignores it (KzgSettings.note() says so).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes cannot share a process.
        # When torch is installed, let it load its runtime first so this library binds to the same one.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        vp, pp, u8, sz, bp = C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t, C.POINTER(C.c_bool)
        L.kzg_settings_load_trusted_setup.argtypes = [pp, u8, sz]
        L.kzg_settings_from_tau_g2.argtypes = [pp, u8]
        L.kzg_settings_load_trusted_setup_devices.argtypes = [pp, u8, sz, C.POINTER(C.c_int), sz]
        L.kzg_settings_from_tau_g2_devices.argtypes = [pp, u8, C.POINTER(C.c_int), sz]
        L.kzg_settings_devices.argtypes = [vp, C.POINTER(sz), C.POINTER(C.c_int), sz, C.POINTER(C.c_int)]
        L.kzg_verify_blob_kzg_proof_batch_sharded.argtypes = [bp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), sz, vp]
        L.kzg_verify_blob_kzg_proof_batch_sharded_stream.argtypes = [bp, u8, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), sz, sz, sz, vp]
        L.kzg_multi_last_timings.argtypes = [vp, C.POINTER(C.c_float)]
        L.kzg_settings_free.argtypes = [vp]
        L.kzg_settings_free.restype = None
        L.kzg_settings_root_of_unity.argtypes = [vp, sz, u8]
        L.kzg_settings_tau_g2.argtypes = [vp, u8]
        L.kzg_settings_g1_point.argtypes = [vp, sz, u8]
        L.kzg_settings_g2_point.argtypes = [vp, sz, u8]
        L.kzg_settings_is_monomial_form.argtypes = [bp, vp]
        L.kzg_blob_to_kzg_commitment.argtypes = [u8, u8, sz, vp]
        L.kzg_compute_kzg_proof.argtypes = [u8, u8, u8, u8, sz, vp]
        L.kzg_compute_blob_kzg_proof.argtypes = [u8, u8, u8, sz, vp]
        L.kzg_verify_kzg_proof.argtypes = [bp, u8, u8, u8, u8, vp]
        L.kzg_verify_kzg_proof_batch.argtypes = [bp, u8, u8, u8, u8, sz, vp]
        L.kzg_verify_kzg_proofs.argtypes = [bp, u8, u8, u8, u8, u8, sz, vp]
        L.kzg_verify_blob_kzg_proof.argtypes = [bp, u8, u8, u8, vp]
        L.kzg_verify_blob_kzg_proof_batch.argtypes = [bp, u8, u8, u8, sz, vp]
        L.kzg_verify_blob_kzg_proof_batch_device.argtypes = [bp, vp, vp, vp, sz, vp]
        L.kzg_verify_blob_kzg_proof_batches_device.argtypes = [bp, u8, vp, vp, vp, sz, sz, vp]
        L.kzg_verify_blob_kzg_proof_batches.argtypes = [bp, u8, vp, vp, vp, sz, sz, vp]
        L.kzg_verify_blob_kzg_proof_batch_groups_device.argtypes = [bp, u8, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), sz, sz, sz, sz, vp]
        L.kzg_compute_challenges.argtypes = [u8, u8, u8, sz, vp]
        L.kzg_evaluate_polynomials.argtypes = [u8, u8, u8, sz, vp]
        L.kzg_evaluate_polynomials_device.argtypes = [vp, vp, vp, sz, vp]
        L.kzg_g1_decompress.argtypes = [u8, u8, u8, sz, vp]
        L.kzg_g1_msm.argtypes = [u8, u8, u8, sz, vp]
        L.kzg_g1_msm_setup.argtypes = [u8, u8, sz, vp]
        L.kzg_pairing_check.argtypes = [bp, u8, u8, vp]
        L.kzg_pairings_verify.argtypes = [bp, u8, u8, u8, u8, vp]
        L.kzg_g1_mul_generator.argtypes = [u8, u8, sz, vp]
        L.kzg_last_timings.argtypes = [vp, C.POINTER(C.c_float)]
        L.kzg_timing_totals.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.kzg_debug_shader_clock.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
        L.kzg_kernel_stamp_totals.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_int]
        L.kzg_last_error.restype = C.c_char_p
        L.kzg_settings_note.argtypes = [vp]
        L.kzg_settings_note.restype = C.c_char_p
        L.kzg_debug_small_queue_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.kzg_debug_concurrent_callers.argtypes = [C.POINTER(C.c_double), C.c_int, sz, C.c_double, u8, u8, u8, u8, u8, u8, sz, sz, vp]
        _lib = L
    return _lib


def _chk(rc):
    if rc != KZG_OK:
        raise KzgError(_KIND.get(rc, "InternalError"), lib().kzg_last_error().decode(errors="replace"))


class _BytesN:
    SIZE = 0

    def __init__(self, data):
        self.data = bytes(data)

    @classmethod
    def from_slice(cls, data):
        """src/dtypes.rs:20-29."""
        if len(data) != cls.SIZE:
            raise InvalidBytesLength("Invalid slice length")
        return cls(data)

    @classmethod
    def from_hex(cls, s):
        return cls.from_slice(bytes.fromhex(s[2:] if s.startswith("0x") else s))

    def as_slice(self):
        return self.data

    # ---- wire formats of the reference's optional derives (src/dtypes.rs:9-17, Cargo.toml:41-43) ----
    # `serde`: the newtype serialises as its inner [u8; N] through serde_arrays, i.e. a fixed-size sequence of N u8 -
    # N raw bytes in a binary format such as bincode, a list of N numbers in a self-describing one such as JSON.
    # `rkyv`: the archived form of a [u8; N] newtype is the N bytes themselves (alignment 1, no header).
    def to_wire_bytes(self):
        """bincode(serde) / rkyv archived bytes: exactly SIZE raw bytes."""
        return self.data

    @classmethod
    def from_wire_bytes(cls, data):
        """Inverse of to_wire_bytes; a wrong length is the deserialiser's error (InvalidBytesLength here)."""
        return cls.from_slice(bytes(data))

    def to_json(self):
        """serde_json form: a JSON array of SIZE integers 0..255."""
        import json
        return json.dumps(list(self.data), separators=(",", ":"))

    @classmethod
    def from_json(cls, text):
        import json
        v = json.loads(text)
        if not isinstance(v, list) or len(v) != cls.SIZE or any((not isinstance(x, int)) or isinstance(x, bool) or not 0 <= x <= 255 for x in v):
            raise InvalidBytesLength("expected an array of %d u8" % cls.SIZE)
        return cls(bytes(v))


class Bytes32(_BytesN):
    SIZE = 32


class Bytes48(_BytesN):
    SIZE = 48


class Blob(_BytesN):
    SIZE = BYTES_PER_BLOB


class KzgSettings:
    """Opaque device-side settings (replaces src/trusted_setup.rs:44-50)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def _devs(devices):
        """devices: "all" or a list of HIP ordinals -> (int array or None, count)"""
        if devices == "all":
            return None, 0
        arr = (C.c_int * len(devices))(*devices)
        return arr, len(devices)

    @classmethod
    def load_trusted_setup_file(cls, path=None, devices=None):
        """src/trusted_setup.rs:94-98 (the reference embeds the file at build time; here the same
        public ceremony file ships as package data).  devices: None = the current device (or KZG_DEVICES from the
        environment), "all" or a list of ordinals = one handle over several GPUs (include/kzg_rs_amd.h)."""
        txt = open(path or TRUSTED_SETUP_PATH, "rb").read()
        h = C.c_void_p()
        if devices is None:
            _chk(lib().kzg_settings_load_trusted_setup(C.byref(h), txt, len(txt)))
        else:
            arr, n = cls._devs(devices)
            _chk(lib().kzg_settings_load_trusted_setup_devices(C.byref(h), txt, len(txt), arr, n))
        return cls(h)

    @classmethod
    def load_trusted_setup_text(cls, txt):
        """The same parser on text held in memory (bytes)."""
        h = C.c_void_p()
        _chk(lib().kzg_settings_load_trusted_setup(C.byref(h), bytes(txt), len(txt)))
        return cls(h)

    @classmethod
    def from_tau_g2(cls, tau_g2, devices=None):
        """EnvKzgSettings::Custom (src/trusted_setup.rs:52-57) from g2_points[1] alone; devices as in load_trusted_setup_file."""
        h = C.c_void_p()
        if devices is None:
            _chk(lib().kzg_settings_from_tau_g2(C.byref(h), bytes(tau_g2)))
        else:
            arr, n = cls._devs(devices)
            _chk(lib().kzg_settings_from_tau_g2_devices(C.byref(h), bytes(tau_g2), arr, n))
        return cls(h)

    def devices(self):
        """(device ordinals of the handle's shards, exchange) - exchange: "none" (one device), "host" or "rccl"."""
        n, ex = C.c_size_t(0), C.c_int(0)
        arr = (C.c_int * 64)()
        _chk(lib().kzg_settings_devices(self._h, C.byref(n), arr, 64, C.byref(ex)))
        return list(arr[: n.value]), ("none", "host", "rccl")[ex.value]

    def note(self):
        """What the constructor wants its caller to know about a handle it made successfully ("" = nothing): fewer than 8 HIP
        hardware queues, how a multi-device handle exchanges its partial sums (include/kzg_rs_amd.h kzg_settings_note)."""
        return lib().kzg_settings_note(self._h).decode(errors="replace")

    def small_queue_stats(self, reset=False):
        """The small-call queue of this handle (csrc/capi_coalesce.hpp) since the last reset:
        {launches, requests, items, max_items (the largest launch), lanes}."""
        o = (C.c_uint64 * 5)()
        _chk(lib().kzg_debug_small_queue_stats(self._h, o, int(reset)))
        return dict(zip(("launches", "requests", "items", "max_items", "lanes"), (int(x) for x in o)))

    def concurrent_callers(self, kind, threads, seconds, c, p, expect, z=None, y=None, blobs=None, per_call=1):
        """T host threads INSIDE the library (no interpreter lock) calling the public small entry points on this one handle for
        `seconds`, every answer checked against `expect` (0 false / 1 true / 2 Err): kind "proof" = verify_kzg_proof per tuple,
        "blobs" = verify_blob_kzg_proof_batch of per_call blobs (expect per call), "proofs" = kzg_verify_kzg_proofs of per_call
        tuples (expect per tuple).  Returns {calls, seconds, calls_per_s, wrong, mean_ms, max_ms}."""
        k = {"proof": 0, "blobs": 1, "proofs": 2}[kind]
        n_items = len(c) // 48
        o = (C.c_double * 5)()
        _chk(lib().kzg_debug_concurrent_callers(o, k, threads, float(seconds), blobs, bytes(c), z, y, bytes(p), bytes(expect), n_items, per_call, self._h))
        return {"calls": int(o[0]), "seconds": o[1], "calls_per_s": o[0] / o[1] if o[1] else 0.0, "wrong": int(o[2]), "mean_ms": o[3], "max_ms": o[4]}

    def multi_last_timings(self):
        t = (C.c_float * 8)()
        _chk(lib().kzg_multi_last_timings(self._h, t))
        return list(t)

    def root_of_unity(self, i):
        out = C.create_string_buffer(32)
        _chk(lib().kzg_settings_root_of_unity(self._h, i, out))
        return out.raw

    def tau_g2(self):
        out = C.create_string_buffer(96)
        _chk(lib().kzg_settings_tau_g2(self._h, out))
        return out.raw

    def g1_point(self, i):
        """g1_points[i] (bit-reversal permuted, build.rs:79) as 48 compressed bytes."""
        out = C.create_string_buffer(48)
        _chk(lib().kzg_settings_g1_point(self._h, i, out))
        return out.raw

    def g2_point(self, i):
        out = C.create_string_buffer(96)
        _chk(lib().kzg_settings_g2_point(self._h, i, out))
        return out.raw

    def is_monomial_form(self):
        """build.rs:107-129 (the reference computes this at load time and discards it)."""
        ok = C.c_bool(False)
        _chk(lib().kzg_settings_is_monomial_form(C.byref(ok), self._h))
        return bool(ok.value)

    def last_timings(self):
        t = (C.c_float * 8)()
        _chk(lib().kzg_last_timings(self._h, t))
        return list(t)

    def timing_totals(self, reset=False):
        """(sums of the last_timings intervals over the groups finished since the last reset, number of groups)"""
        t, c = (C.c_double * 8)(), C.c_uint64(0)
        _chk(lib().kzg_timing_totals(self._h, t, C.byref(c), int(reset)))
        return list(t), int(c.value)

    def kernel_stamp_totals(self, reset=False):
        """The kernels' OWN execution intervals (in-kernel stamps, no queueing in them), ms: ({challenge, evaluate, decode, msm_window}
        summed over the launch groups finished on this handle and its lanes since the last reset, groups, the same four of the
        handle's last group)."""
        t, c, last = (C.c_double * 4)(), C.c_uint64(0), (C.c_float * 4)()
        _chk(lib().kzg_kernel_stamp_totals(self._h, t, C.byref(c), last, int(reset)))
        names = ("k_blob_challenge", "k_blob_evaluate", "k_g1_decode_multiples", "k_msm_window")
        return dict(zip(names, t)), int(c.value), dict(zip(names, last))

    def shader_clock(self, reset=False):
        """(shader cycles, 100 MHz reference ticks) summed over the waves of the throughput-form challenge kernel since the last
        reset: MHz = 100 * cycles / ticks"""
        o = (C.c_double * 2)()
        _chk(lib().kzg_debug_shader_clock(self._h, o, int(reset)))
        return float(o[0]), float(o[1])

    def close(self):
        if self._h:
            lib().kzg_settings_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_settings = None


class EnvKzgSettings:
    """src/trusted_setup.rs:52-92: Default (cached mainnet setup) or Custom(settings)."""

    def __init__(self, custom=None):
        self.custom = custom

    def get(self):
        global _default_settings
        if self.custom is not None:
            return self.custom
        if _default_settings is None:
            _default_settings = KzgSettings.load_trusted_setup_file()
        return _default_settings


class KzgProof:
    @staticmethod
    def verify_kzg_proof(commitment_bytes, z_bytes, y_bytes, proof_bytes, kzg_settings):
        """src/kzg_proof.rs:353-397."""
        ok = C.c_bool(False)
        _chk(lib().kzg_verify_kzg_proof(C.byref(ok), commitment_bytes.data, z_bytes.data, y_bytes.data, proof_bytes.data,
                                        kzg_settings._h))
        return bool(ok.value)

    @staticmethod
    def verify_kzg_proof_batch(commitments, zs, ys, proofs, kzg_settings):
        """src/kzg_proof.rs:399-444: n (commitment, z, y, proof) tuples, one random linear combination, one pairing.
        The reference takes decoded &[G1Affine] / &[Scalar]; here they are Bytes48 / Bytes32 (big-endian, canonical)
        and decoding (with the subgroup check) is part of the call.  Slices of unequal length index out of bounds in
        the reference (a panic): IndexError here."""
        n = len(commitments)
        if len(zs) < n or len(ys) < n or len(proofs) < n:
            raise IndexError("index out of bounds")
        ok = C.c_bool(False)
        _chk(lib().kzg_verify_kzg_proof_batch(
            C.byref(ok), b"".join(c.data for c in commitments), b"".join(z.data for z in zs[:n]),
            b"".join(y.data for y in ys[:n]), b"".join(p.data for p in proofs[:n]), n, kzg_settings._h))
        return bool(ok.value)

    @staticmethod
    def verify_blob_kzg_proof(blob, commitment_bytes, proof_bytes, kzg_settings):
        """src/kzg_proof.rs:446-470."""
        ok = C.c_bool(False)
        _chk(lib().kzg_verify_blob_kzg_proof(C.byref(ok), blob.data, commitment_bytes.data, proof_bytes.data,
                                             kzg_settings._h))
        return bool(ok.value)

    @staticmethod
    def verify_blob_kzg_proof_batch(blobs, commitments_bytes, proofs_bytes, kzg_settings):
        """src/kzg_proof.rs:472-525, including the order of its early returns (quirk Q2:
        empty -> Ok(true) and the single-blob shortcut come before the length checks)."""
        if len(blobs) == 0:
            return True
        if len(blobs) == 1:
            if not commitments_bytes or not proofs_bytes:
                raise IndexError("index out of bounds")  # the reference panics on [0] here
            return KzgProof.verify_blob_kzg_proof(blobs[0], commitments_bytes[0], proofs_bytes[0], kzg_settings)
        if len(blobs) != len(commitments_bytes):
            raise InvalidBytesLength("Invalid commitments length")
        if len(blobs) != len(proofs_bytes):
            raise InvalidBytesLength("Invalid proofs length")
        ok = C.c_bool(False)
        _chk(lib().kzg_verify_blob_kzg_proof_batch(
            C.byref(ok), b"".join(b.data for b in blobs), b"".join(c.data for c in commitments_bytes),
            b"".join(p.data for p in proofs_bytes), len(blobs), kzg_settings._h))
        return bool(ok.value)

    @staticmethod
    def verify_blob_kzg_proof_batch_device(d_blobs, d_commitments, d_proofs, n, kzg_settings):
        """Device-resident form: arguments are device pointers (ints), e.g. torch tensor .data_ptr()."""
        ok = C.c_bool(False)
        _chk(lib().kzg_verify_blob_kzg_proof_batch_device(C.byref(ok), d_blobs, d_commitments, d_proofs, n,
                                                          kzg_settings._h))
        return bool(ok.value)


def verify_kzg_proofs(commitments, zs, ys, proofs, kzg_settings):
    """n INDEPENDENT verify_kzg_proof calls (src/kzg_proof.rs:353-397) through one C call, each with its own pairing: lists of
    Bytes48 / Bytes32 (or bytes) of equal length -> a list with True / False per proof, or None where the reference would
    return Err."""
    n = len(commitments)
    if not (len(zs) == len(ys) == len(proofs) == n):
        raise InvalidBytesLength("verify_kzg_proofs: lists of unequal length")
    if n == 0:
        return []
    raw = lambda xs: b"".join(x.data if hasattr(x, "data") else bytes(x) for x in xs)
    ok = (C.c_bool * n)()
    err = C.create_string_buffer(n)
    _chk(lib().kzg_verify_kzg_proofs(ok, err, raw(commitments), raw(zs), raw(ys), raw(proofs), n, kzg_settings._h))
    return [None if err.raw[i] else bool(ok[i]) for i in range(n)]


def verify_blob_kzg_proof_batch_sharded(shards, kzg_settings):
    """ONE batch whose shards are resident on the devices of a multi-device handle: shards = [(d_blobs, d_commitments,
    d_proofs, n_local), ...] in the order of the handle's device list (device pointers as ints)."""
    k = len(shards)
    vp = C.c_void_p
    b, c, p = (vp * k)(*[s[0] for s in shards]), (vp * k)(*[s[1] for s in shards]), (vp * k)(*[s[2] for s in shards])
    nl = (C.c_size_t * k)(*[s[3] for s in shards])
    ok = C.c_bool(False)
    _chk(lib().kzg_verify_blob_kzg_proof_batch_sharded(C.byref(ok), b, c, p, nl, k, kzg_settings._h))
    return bool(ok.value)


def verify_blob_kzg_proof_batch_sharded_stream(batches, kzg_settings, in_flight=0):
    """A STREAM of sharded batches, `in_flight` of them at once inside the library (0: its default): batches = [shards, ...],
    shards as in verify_blob_kzg_proof_batch_sharded.  Returns True / False per batch, or None where the reference would
    return Err."""
    nb = len(batches)
    if nb == 0:
        return []
    k = len(batches[0])
    if any(len(b) != k for b in batches):
        raise BadArgs("every batch needs one shard per device of the handle")
    vp = C.c_void_p
    flat = [s for b in batches for s in b]
    b, c, p = (vp * (nb * k))(*[s[0] for s in flat]), (vp * (nb * k))(*[s[1] for s in flat]), (vp * (nb * k))(*[s[2] for s in flat])
    nl = (C.c_size_t * (nb * k))(*[s[3] for s in flat])
    ok = (C.c_bool * nb)()
    err = C.create_string_buffer(nb)
    _chk(lib().kzg_verify_blob_kzg_proof_batch_sharded_stream(ok, err, b, c, p, nl, k, nb, in_flight, kzg_settings._h))
    return [None if err.raw[j] else bool(ok[j]) for j in range(nb)]


def verify_blob_kzg_proof_batches_device(d_blobs, d_commitments, d_proofs, n, n_batches, kzg_settings):
    """n_batches independent verify_blob_kzg_proof_batch calls of n blobs each in one launch group (device pointers).
    Returns a list with True / False per batch, or None where the reference would return Err."""
    ok = (C.c_bool * n_batches)()
    err = C.create_string_buffer(n_batches)
    _chk(lib().kzg_verify_blob_kzg_proof_batches_device(ok, err, d_blobs, d_commitments, d_proofs, n, n_batches,
                                                        kzg_settings._h))
    return [None if err.raw[b] else bool(ok[b]) for b in range(n_batches)]


def verify_blob_kzg_proof_batch_groups_device(groups, n, batches_per_group, kzg_settings, in_flight=0):
    """Many launch groups through ONE C call, `in_flight` of them (0: the library's default, 4) overlapping inside the library: groups = [(d_blobs,
    d_commitments, d_proofs), ...] device pointers of groups of `batches_per_group` batches of n blobs.  Returns one list per
    group with True / False per batch, or None where the reference would return Err."""
    k, B, vp = len(groups), batches_per_group, C.c_void_p
    if k == 0:
        return []
    b, c, p = (vp * k)(*[g[0] for g in groups]), (vp * k)(*[g[1] for g in groups]), (vp * k)(*[g[2] for g in groups])
    ok = (C.c_bool * (k * B))()
    err = C.create_string_buffer(k * B)
    _chk(lib().kzg_verify_blob_kzg_proof_batch_groups_device(ok, err, b, c, p, n, B, k, in_flight, kzg_settings._h))
    return [[None if err.raw[g * B + i] else bool(ok[g * B + i]) for i in range(B)] for g in range(k)]


def verify_blob_kzg_proof_batches(blobs, commitments, proofs, n, n_batches, kzg_settings):
    """The host-memory form: n_batches batches of n blobs each, back to back in host memory (bytes, or a host address as
    an int); copies overlap verification.  Returns True / False / None (Err) per batch."""
    ok = (C.c_bool * n_batches)()
    err = C.create_string_buffer(n_batches)
    keep = []
    ptrs = [_host_ptr(x, want, what, keep) for x, want, what in
            ((blobs, n * n_batches * BYTES_PER_BLOB, "blobs"), (commitments, n * n_batches * BYTES_PER_COMMITMENT, "commitments"),
             (proofs, n * n_batches * BYTES_PER_PROOF, "proofs"))]
    _chk(lib().kzg_verify_blob_kzg_proof_batches(ok, err, ptrs[0], ptrs[1], ptrs[2], n, n_batches, kzg_settings._h))
    del keep
    return [None if err.raw[b] else bool(ok[b]) for b in range(n_batches)]


def _host_ptr(x, want_len, what, keep):
    """A host address for the C ABI.  bytes-like objects are length-checked (the reference's Err(InvalidBytesLength) for
    mismatched lengths, src/kzg_proof.rs:491-501); an int is taken as a raw address the caller vouches for - the
    explicitly unsafe form."""
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, (bytes, bytearray, memoryview)):
        if len(x) != want_len:
            raise InvalidBytesLength("Invalid %s length: %d bytes, expected %d" % (what, len(x), want_len))
        if isinstance(x, bytes):
            return C.cast(C.c_char_p(x), C.c_void_p)
        buf = (C.c_char * len(x)).from_buffer(x)  # bytearray / writable memoryview: no copy
        keep.append(buf)
        return C.cast(buf, C.c_void_p)
    raise TypeError("%s: bytes, bytearray or an int address expected" % what)


# ---- pieces of the path (parity tests / per-kernel benchmarks) ----

def compute_challenges(blobs, commitments, kzg_settings):
    """compute_challenge (src/kzg_proof.rs:46-72) for a list of blobs; returns list of 32-byte BE scalars."""
    n = len(blobs)
    out = C.create_string_buffer(32 * max(n, 1))
    _chk(lib().kzg_compute_challenges(out, b"".join(blobs), b"".join(commitments), n, kzg_settings._h))
    return [out.raw[32 * i: 32 * i + 32] for i in range(n)]


def evaluate_polynomials(blobs, zs, kzg_settings):
    """evaluate_polynomial_in_evaluation_form (src/kzg_proof.rs:94-133) for lists of blobs / 32-byte BE points."""
    n = len(blobs)
    out = C.create_string_buffer(32 * max(n, 1))
    _chk(lib().kzg_evaluate_polynomials(out, b"".join(blobs), b"".join(zs), n, kzg_settings._h))
    return [out.raw[32 * i: 32 * i + 32] for i in range(n)]


def evaluate_polynomials_device(d_y, d_blobs, d_z, n, kzg_settings):
    _chk(lib().kzg_evaluate_polynomials_device(d_y, d_blobs, d_z, n, kzg_settings._h))


def g1_decompress(points, kzg_settings, want_xy=True):
    n = len(points)
    st = C.create_string_buffer(max(n, 1))
    xy = C.create_string_buffer(96 * max(n, 1)) if want_xy else None
    _chk(lib().kzg_g1_decompress(st, xy, b"".join(points), n, kzg_settings._h))
    return list(st.raw[:n]), ([xy.raw[96 * i: 96 * i + 96] for i in range(n)] if want_xy else None)


def g1_msm(points, scalars, kzg_settings):
    n = len(points)
    out = C.create_string_buffer(48)
    _chk(lib().kzg_g1_msm(out, b"".join(points), b"".join(scalars), n, kzg_settings._h))
    return out.raw


def g1_msm_setup(scalars, kzg_settings):
    """sum_i scalars[i] * g1_points[i mod 4096] over the handle's own Lagrange points (kzg_g1_msm_setup): scalars = a list of
    32-byte big-endian values, or one bytes object of n x 32."""
    raw = scalars if isinstance(scalars, (bytes, bytearray)) else b"".join(scalars)
    out = C.create_string_buffer(48)
    _chk(lib().kzg_g1_msm_setup(out, bytes(raw), len(raw) // 32, kzg_settings._h))
    return out.raw


def pairing_check(a, b, kzg_settings):
    ok = C.c_bool(False)
    _chk(lib().kzg_pairing_check(C.byref(ok), a, b, kzg_settings._h))
    return bool(ok.value)


def pairings_verify(a1, a2, b1, b2, kzg_settings):
    """pairings_verify (src/pairings.rs:5-9, src/lib.rs:15): e(a1, a2) == e(b1, b2) for compressed G1 (48 B) / G2 (96 B)
    points; the settings supply the device and the pairing programs only."""
    if len(a1) != 48 or len(b1) != 48 or len(a2) != 96 or len(b2) != 96:
        raise InvalidBytesLength("pairings_verify: 48-byte G1 and 96-byte G2 encodings expected")
    ok = C.c_bool(False)
    _chk(lib().kzg_pairings_verify(C.byref(ok), bytes(a1), bytes(a2), bytes(b1), bytes(b2), kzg_settings._h))
    return bool(ok.value)


def blob_to_kzg_commitment(blobs, kzg_settings):
    """Prover side (not in the reference; c-kzg-4844's name): commitments of a list of blobs (bytes) under the
    settings' Lagrange G1 points, as 48-byte strings."""
    n = len(blobs)
    out = C.create_string_buffer(48 * max(n, 1))
    _chk(lib().kzg_blob_to_kzg_commitment(out, b"".join(blobs), n, kzg_settings._h))
    return [out.raw[48 * i: 48 * i + 48] for i in range(n)]


def compute_kzg_proof(blobs, zs, kzg_settings):
    """c-kzg-4844's compute_kzg_proof for lists of blobs and 32-byte big-endian z: -> (proofs, ys)."""
    n = len(blobs)
    pr, ys = C.create_string_buffer(48 * max(n, 1)), C.create_string_buffer(32 * max(n, 1))
    _chk(lib().kzg_compute_kzg_proof(pr, ys, b"".join(blobs), b"".join(zs), n, kzg_settings._h))
    return [pr.raw[48 * i: 48 * i + 48] for i in range(n)], [ys.raw[32 * i: 32 * i + 32] for i in range(n)]


def compute_blob_kzg_proof(blobs, commitments, kzg_settings):
    """c-kzg-4844's compute_blob_kzg_proof: the proofs verify_blob_kzg_proof accepts."""
    n = len(blobs)
    pr = C.create_string_buffer(48 * max(n, 1))
    _chk(lib().kzg_compute_blob_kzg_proof(pr, b"".join(blobs), b"".join(commitments), n, kzg_settings._h))
    return [pr.raw[48 * i: 48 * i + 48] for i in range(n)]


def g1_mul_generator(scalars, kzg_settings):
    """[k]G1 for a list of 32-byte big-endian scalars; returns list of 48-byte compressed points."""
    n = len(scalars)
    out = C.create_string_buffer(48 * max(n, 1))
    _chk(lib().kzg_g1_mul_generator(out, b"".join(scalars), n, kzg_settings._h))
    return [out.raw[48 * i: 48 * i + 48] for i in range(n)]
