"""CPU tests of the host-side mirror of the reference API (kzg_rs_amd/api.py): the parts that run before any device work
- byte-string types (src/dtypes.rs:7-46) and the early returns of verify_blob_kzg_proof_batch in the reference's own
order (src/kzg_proof.rs:478-501, SURVEY quirk Q2: empty -> Ok(true) and the single-blob shortcut come BEFORE the
length checks)."""
import pytest

from kzg_rs_amd import api
from kzg_rs_amd.api import Blob, Bytes32, Bytes48, KzgError, KzgProof


def test_bytes_types_from_slice():
    for cls, n in ((Bytes32, 32), (Bytes48, 48), (Blob, 131072)):
        assert cls.from_slice(bytes(n)).as_slice() == bytes(n)
        for bad in (n - 1, n + 1, 0):
            with pytest.raises(KzgError) as e:
                cls.from_slice(bytes(bad))
            assert e.value.kind == "InvalidBytesLength"
    assert Bytes48.from_hex("0x" + "ab" * 48).data == bytes([0xAB]) * 48
    assert Bytes32.from_hex("cd" * 32).data == bytes([0xCD]) * 32
    with pytest.raises(KzgError):
        Bytes32.from_hex("cd" * 31)


def test_batch_early_returns_need_no_device():
    # src/kzg_proof.rs:478-480: an empty batch is true whatever the other vectors hold (no settings needed either)
    assert KzgProof.verify_blob_kzg_proof_batch([], [Bytes48(bytes(48))], [], None) is True
    # :491-501 length mismatches are InvalidBytesLength, checked after the n == 0 / n == 1 shortcuts
    two = [Blob(bytes(131072))] * 2
    with pytest.raises(KzgError) as e:
        KzgProof.verify_blob_kzg_proof_batch(two, [Bytes48(bytes(48))], [Bytes48(bytes(48))] * 2, None)
    assert e.value.kind == "InvalidBytesLength" and "commitments" in e.value.msg
    with pytest.raises(KzgError) as e:
        KzgProof.verify_blob_kzg_proof_batch(two, [Bytes48(bytes(48))] * 2, [Bytes48(bytes(48))] * 3, None)
    assert e.value.kind == "InvalidBytesLength" and "proofs" in e.value.msg
    # :482-489 the single-blob shortcut indexes commitments[0] / proofs[0] before any length check: a panic in Rust
    with pytest.raises(IndexError):
        KzgProof.verify_blob_kzg_proof_batch(two[:1], [], [Bytes48(bytes(48))], None)


def test_proof_batch_slices_of_unequal_length():
    with pytest.raises(IndexError):  # zs[i] out of bounds in src/kzg_proof.rs:422-426
        KzgProof.verify_kzg_proof_batch([Bytes48(bytes(48))] * 2, [Bytes32(bytes(32))], [Bytes32(bytes(32))] * 2,
                                        [Bytes48(bytes(48))] * 2, None)


def test_constants_match_reference():
    # src/consts.rs:1-15
    assert (api.BYTES_PER_FIELD_ELEMENT, api.FIELD_ELEMENTS_PER_BLOB, api.BYTES_PER_BLOB) == (32, 4096, 131072)
    assert (api.BYTES_PER_COMMITMENT, api.BYTES_PER_PROOF) == (48, 48)


def test_serde_rkyv_wire_formats_round_trip():
    """the reference's optional `serde` / `rkyv` derives on Bytes32 / Bytes48 / Blob (src/dtypes.rs:9-17).
    Binary form (bincode over serde_arrays, rkyv archive of a [u8; N] newtype): exactly N raw bytes; serde_json form:
    an array of N numbers.  Round trips are byte-exact; wrong sizes and out-of-range elements are rejected."""
    import json
    import random
    from kzg_rs_amd.api import Blob, Bytes32, Bytes48, KzgError
    rng = random.Random(5)
    for cls in (Bytes32, Bytes48, Blob):
        raw = bytes(rng.getrandbits(8) for _ in range(cls.SIZE))
        v = cls.from_slice(raw)
        w = v.to_wire_bytes()
        assert w == raw and len(w) == cls.SIZE
        assert cls.from_wire_bytes(w).data == raw
        j = v.to_json()
        assert json.loads(j) == list(raw)
        assert cls.from_json(j).data == raw
        for bad in (raw[:-1], raw + b"\0"):
            with pytest.raises(KzgError) as e:
                cls.from_wire_bytes(bad)
            assert e.value.kind == "InvalidBytesLength"
        for bad in (json.dumps(list(raw[:-1])), json.dumps(list(raw[:-1]) + [256]), json.dumps(list(raw[:-1]) + [-1]), json.dumps({"a": 1})):
            with pytest.raises(KzgError):
                cls.from_json(bad)


def test_host_stream_wrapper_checks_buffer_lengths_before_the_call():
    """verify_blob_kzg_proof_batches (the host-memory stream form): bytes-like arguments whose length is not exactly
    n * n_batches entries raise InvalidBytesLength (the reference's error for mismatched lengths, src/kzg_proof.rs:491-501)
    BEFORE anything is handed to the C entry point - it would read n * n_batches * 131072 bytes whatever the object holds.
    bytearray is accepted (no copy); other types are a TypeError.  No GPU is touched: the checks come first."""
    from kzg_rs_amd import api
    n, B = 2, 3
    blobs, cs, ps = bytes(131072 * n * B), bytes(48 * n * B), bytes(48 * n * B)
    for bad in ((blobs[:-1], cs, ps), (blobs, cs[:-48], ps), (blobs, cs, ps + b"\0"), (bytearray(blobs[:131072]), cs, ps)):
        with pytest.raises(api.KzgError) as e:
            api.verify_blob_kzg_proof_batches(*bad, n, B, None)
        assert e.value.kind == "InvalidBytesLength"
    with pytest.raises(TypeError):
        api.verify_blob_kzg_proof_batches([1, 2, 3], cs, ps, n, B, None)
    keep = []
    p = api._host_ptr(bytearray(b"abc"), 3, "x", keep)
    assert p.value and len(keep) == 1
    from kzg_rs_amd.distributed import HipBackend
    with pytest.raises(api.KzgError):
        HipBackend.batch_challenges(bytes(159), 0, 1, 1)
