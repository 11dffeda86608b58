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
        L.kzg_shard_phase1_launch.argtypes = [vp, vp, vp, sz, sz, vp]
        L.kzg_shard_phase1_wait.argtypes = [u8, u8, vp]
        L.kzg_shard_phase2_launch.argtypes = [u8, sz, sz, vp]
        L.kzg_shard_phase2_launch_gathered.argtypes = [vp, sz, sz, vp]
        L.kzg_shard_records_device.argtypes = [vp, vp]
        L.kzg_shard_phase2_wait.argtypes = [u8, vp]
        L.kzg_shard_finish_launch.argtypes = [u8, sz, sz, vp]
        L.kzg_shard_finish_wait.argtypes = [C.POINTER(C.c_bool), vp]

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

    # launch / wait halves over a launch group of n_batches batches (see PipelinedVerifier)
    def phase1_launch(self, shard, n_batches=1):
        d_blobs, d_commitments, d_proofs, n_local = shard
        self._n, self._b = n_local, n_batches
        api._chk(api.lib().kzg_shard_phase1_launch(d_blobs, d_commitments, d_proofs, n_local, n_batches, self.settings._h))

    def phase1_wait(self, want_records=True):
        """want_records=False leaves the records inside the handle (phase2_launch(None, ...) or the device exchange)."""
        if not want_records:
            api._chk(api.lib().kzg_shard_phase1_wait(None, None, self.settings._h))
            return None
        out = C.create_string_buffer(RECORD_BYTES * self._n * self._b)
        api._chk(api.lib().kzg_shard_phase1_wait(out, None, self.settings._h))
        return out.raw

    def phase2_launch(self, all_records, n_total, offset):
        """all_records None: the handle's own records (single rank)."""
        api._chk(api.lib().kzg_shard_phase2_launch(all_records, n_total, offset, self.settings._h))

    # bulk exchange without host copies (equal shards): records leave through device memory, come back gathered
    def records_to_device(self, d_ptr):
        api._chk(api.lib().kzg_shard_records_device(d_ptr, self.settings._h))

    def phase2_launch_gathered(self, h_ptr, world, rank):
        api._chk(api.lib().kzg_shard_phase2_launch_gathered(h_ptr, world, rank, self.settings._h))

    def phase2_wait(self):
        out = C.create_string_buffer(PARTIAL_BYTES * self._b)
        api._chk(api.lib().kzg_shard_phase2_wait(out, self.settings._h))
        return out.raw

    def finish_launch(self, partials, world):
        api._chk(api.lib().kzg_shard_finish_launch(partials, world, self._b, self.settings._h))

    def finish_wait(self):
        ok = (C.c_bool * self._b)()
        api._chk(api.lib().kzg_shard_finish_wait(ok, self.settings._h))
        return [bool(x) for x in ok]


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


def verify_blob_kzg_proof_batch_sharded(shard, n_local, backend, dist=None, device="cpu", force_collectives=False):
    """Every rank calls this with its own shard (rank order = global blob order).  Returns the batch
    result on every rank.  Raises KzgError on every rank if any shard holds an invalid input
    (first-error identity is not observable in the reference beyond "is Err")."""
    import torch

    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives):
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


class PipelinedVerifier:
    """Throughput mode: launch GROUPS of independent batches, and keep several groups in flight from ONE host
    thread with a fixed-order software pipeline.

    At n = 1024 every phase of a batch is a latency-bound serial chain that occupies a sliver of the chip (the
    SHA-256 challenge chain runs on 16 of 1024 SIMDs for ~7 ms), so `group` batches share each kernel launch
    (batch dimension inside the kernels) and `depth` groups overlap through separate handles (= HIP streams).
    Iteration t runs, in this order:   phase1_launch(t);   phase1_wait(t-d1) + exchange 1 + phase2_launch;
    phase2_wait(t-d1-d2) + exchange 2 + finish_launch;   finish_wait(t-d1-d2-d3).
    The order is the same on every rank, so the collectives of different groups never interleave differently
    on different ranks.  With one rank the exchanges and the phase-2 host round trip disappear."""

    def __init__(self, backends, dist=None, device="cpu", depth=(1, 1, 1), equal_shards=False, force_collectives=False):
        """equal_shards: every rank holds the same number of blobs of every batch - enables the bulk exchange (records
        go from the library's device buffer straight into the all-gather and come back through one pinned buffer)."""
        self.backends = backends
        self.equal_shards = equal_shards
        self._bufs = {}
        # force_collectives: run the exchanges even in a world of one (exercises the transport in tests)
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives)) else None
        self.device = device
        self.depth = depth if self.dist else (depth[0], 0, depth[2])
        assert len(backends) >= sum(self.depth) + 1, "need depth+1 handles"

    def _regroup(self, gathered, n_batches):
        """[rank][batch][n_local] records -> [batch][rank][n_local] (each batch's global transcript order)."""
        per_rank = [[g[i * (len(g) // n_batches): (i + 1) * (len(g) // n_batches)] for i in range(n_batches)] for g in gathered]
        return b"".join(per_rank[r][b] for b in range(n_batches) for r in range(len(gathered)))

    def _exchange_buffers(self, slot, nbytes):
        """Per handle: device send buffer, gathered receive buffer and its pinned host mirror (cached by size)."""
        import torch
        world = self.dist.get_world_size()
        key = (slot, nbytes)
        if key not in self._bufs:
            on_gpu = str(self.device) != "cpu"
            send = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            recv = torch.empty(world * nbytes, dtype=torch.uint8, device=self.device)
            host = torch.empty(world * nbytes, dtype=torch.uint8, pin_memory=True) if on_gpu else recv
            self._bufs[key] = (send, recv, host)
        return self._bufs[key]

    def run(self, groups):
        """groups: list of ((d_blobs, d_commitments, d_proofs, n_local), n_batches): this rank's shard of every batch
        of the group, batches contiguous.  Returns the per-batch results of every group, in order."""
        import torch
        K, S = len(groups), len(self.backends)
        d1, d2, d3 = self.depth
        results = [None] * K
        # bulk path: equal shards on every rank and a backend that can hand its records over in device memory
        bulk = bool(self.dist) and self.equal_shards and hasattr(self.backends[0], "records_to_device")
        for t in range(K + d1 + d2 + d3):
            if t < K:
                b = self.backends[t % S]
                b.phase1_launch(groups[t][0], groups[t][1])
                if bulk:
                    send, _, _ = self._exchange_buffers(t % S, RECORD_BYTES * groups[t][0][3] * groups[t][1])
                    b.records_to_device(send.data_ptr())
            i = t - d1
            if 0 <= i < K:
                b, nb = self.backends[i % S], groups[i][1]
                if bulk:
                    world, rank = self.dist.get_world_size(), self.dist.get_rank()
                    send, recv, host = self._exchange_buffers(i % S, RECORD_BYTES * groups[i][0][3] * nb)
                    b.phase1_wait(want_records=False)  # the handle's stream is drained: `send` is complete
                    src = send if recv.is_cuda else send.cpu()
                    self.dist.all_gather(list(recv.chunk(world)), src)  # exchange 1, [world][batch][n_local] records
                    if recv.is_cuda:
                        host.copy_(recv, non_blocking=True)
                        torch.cuda.current_stream().synchronize()
                    b.phase2_launch_gathered(host.data_ptr(), world, rank)
                elif self.dist:
                    recs = b.phase1_wait()
                    gathered = _all_gather_bytes(self.dist, recs, self.device)
                    counts = [len(g) // RECORD_BYTES // nb for g in gathered]
                    b.phase2_launch(self._regroup(gathered, nb), sum(counts), sum(counts[: self.dist.get_rank()]))
                else:
                    if hasattr(b, "records_to_device"):  # the records never leave the handle
                        b.phase1_wait(want_records=False)
                        b.phase2_launch(None, groups[i][0][3], 0)
                    else:
                        recs = b.phase1_wait()
                        b.phase2_launch(recs, len(recs) // RECORD_BYTES // nb, 0)
                    b.finish_launch(None, 1)
            j = t - d1 - d2
            if self.dist and 0 <= j < K:
                b = self.backends[j % S]
                parts = _all_gather_bytes(self.dist, b.phase2_wait(), self.device)
                b.finish_launch(b"".join(parts), len(parts))
            k = t - d1 - d2 - d3
            if 0 <= k < K:
                results[k] = self.backends[k % S].finish_wait()
        return results
