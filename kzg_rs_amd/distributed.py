"""Multi-GPU verify_blob_kzg_proof_batch: one process per GPU, the batch sharded by blob.

The reference is single-threaded; the only data-parallel axis is the per-blob
loop of src/kzg_proof.rs:261-273.  Rank k owns the contiguous global index range
[offset_k, offset_k + n_k).  Two tiny exchanges are real data dependencies of the algorithm:

  1. all-gather of the per-blob transcript records (160 B per blob: C || z || y || pi) - the batch
     challenge r of src/kzg_proof.rs:291-348 hashes ALL of them in index order;
  2. all-gather of each rank's partial sums (A_k, B_k) = 288 B per rank - elliptic-curve addition is
     not an RCCL reduction op (rccl.h offers sum/prod/min/max/avg only), so the "G1 all-reduce" is
     an all-gather followed by a local fold of `world` points on every rank, then the single
     pairing.  Message sizes are latency-bound; xGMI bandwidth is irrelevant here.

Transport is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The compute backend is injectable so the collective logic can be tested on CPU against
the oracle; the default backend is the HIP library.
"""
import ctypes as C

from . import api

RECORD_BYTES = 160
PARTIAL_BYTES = 288


class HipBackend:
    """The three shard phases of include/kzg_rs_amd.h (device pointers in, host bytes out)."""

    def __init__(self, settings):
        self.settings = settings
        L = api.lib()
        vp, u8, sz = C.c_void_p, C.c_char_p, C.c_size_t
        L.kzg_shard_phase1.argtypes = [u8, vp, vp, vp, sz, vp]
        L.kzg_shard_phase2.argtypes = [u8, u8, sz, sz, sz, vp]
        L.kzg_shard_finish.argtypes = [C.POINTER(C.c_bool), u8, sz, vp]

    def phase1(self, shard):
        d_blobs, d_commitments, d_proofs, n_local = shard
        out = C.create_string_buffer(RECORD_BYTES * n_local)
        api._chk(api.lib().kzg_shard_phase1(out, d_blobs, d_commitments, d_proofs, n_local, self.settings._h))
        return out.raw

    def phase2(self, all_records, n_total, offset, n_local):
        out = C.create_string_buffer(PARTIAL_BYTES)
        api._chk(api.lib().kzg_shard_phase2(out, all_records, n_total, offset, n_local, self.settings._h))
        return out.raw

    def finish(self, partials, world):
        ok = C.c_bool(False)
        api._chk(api.lib().kzg_shard_finish(C.byref(ok), partials, world, self.settings._h))
        return bool(ok.value)


def _all_gather_bytes(dist, payload, device):
    """All-gather variable-length byte strings in rank order."""
    import torch

    world = dist.get_world_size()
    ln = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, ln)
    lens = [int(x.item()) for x in lens]
    mx = max(lens + [1])
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [bytes(o[:l].cpu().numpy().tobytes()) for o, l in zip(outs, lens)]


def verify_blob_kzg_proof_batch_sharded(shard, n_local, backend, dist=None, device="cpu"):
    """Every rank calls this with its own shard (rank order = global blob order).  Returns the batch
    result on every rank.  Raises KzgError on every rank if any shard holds an invalid input
    (first-error identity is not observable in the reference beyond "is Err")."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if n_local == 0:
            return True
        recs = backend.phase1(shard)
        part = backend.phase2(recs, n_local, 0, n_local)
        return backend.finish(part, 1)
    world = dist.get_world_size()
    err, recs = None, b""
    if n_local:
        try:
            recs = backend.phase1(shard)
        except api.KzgError as e:  # keep taking part in the collectives, then raise everywhere
            err = e
    flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        raise err if err else api.KzgError("BadArgs", "invalid input on another rank")
    gathered = _all_gather_bytes(dist, recs, device)  # exchange 1: 160 B per blob
    counts = [len(g) // RECORD_BYTES for g in gathered]
    n_total = sum(counts)
    if n_total == 0:
        return True  # src/kzg_proof.rs:478-480
    offset = sum(counts[: dist.get_rank()])
    part = backend.phase2(b"".join(gathered), n_total, offset, n_local) if n_local else b""
    parts = [p for p in _all_gather_bytes(dist, part, device) if p]  # exchange 2: 288 B per rank
    return backend.finish(b"".join(parts), len(parts))
