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
        self.bad = []
        L = api.lib()
        vp, u8, sz = C.c_void_p, C.c_char_p, C.c_size_t
        L.kzg_shard_phase1.argtypes = [u8, vp, vp, vp, sz, vp]
        L.kzg_shard_phase2.argtypes = [u8, u8, sz, sz, sz, vp]
        L.kzg_shard_finish.argtypes = [C.POINTER(C.c_bool), u8, sz, vp]
        L.kzg_shard_phase1_launch.argtypes = [vp, vp, vp, sz, sz, vp]
        L.kzg_shard_phase1_wait.argtypes = [u8, u8, vp]
        L.kzg_shard_phase2_launch.argtypes = [u8, sz, sz, vp]
        L.kzg_shard_phase2_launch_gathered.argtypes = [vp, sz, sz, vp]
        L.kzg_shard_phase2_launch_r.argtypes = [u8, sz, sz, vp]
        L.kzg_batch_challenges.argtypes = [u8, vp, sz, sz, sz]
        L.kzg_shard_records_device.argtypes = [vp, vp]
        L.kzg_shard_phase2_wait.argtypes = [u8, vp]
        L.kzg_shard_finish_launch.argtypes = [u8, sz, sz, vp]
        L.kzg_shard_finish_wait.argtypes = [C.POINTER(C.c_bool), vp]

    def phase1(self, shard):
        d_blobs, d_commitments, d_proofs, n_local = shard
        out = C.create_string_buffer(RECORD_BYTES * n_local)
        self._n, self._b = n_local, 1
        api._chk(api.lib().kzg_shard_phase1(out, d_blobs, d_commitments, d_proofs, n_local, self.settings._h))
        return out.raw

    def phase2(self, all_records, n_total, offset, n_local):
        out = C.create_string_buffer(PARTIAL_BYTES)
        api._chk(api.lib().kzg_shard_phase2(out, all_records, n_total, offset, n_local, self.settings._h))
        return out.raw

    # hash once: the transcript hash is host code that needs no handle; phase 2 then takes r (32 B per batch) instead
    @staticmethod
    def batch_challenges(records, world, n_batches, n_local):
        """r of n_batches batches (32 little-endian bytes each) from records laid out [n_batches][n_local] (world = 0)
        or [world][n_batches][n_local] (world > 0); `records` is bytes or a host address."""
        out = C.create_string_buffer(32 * n_batches)
        keep = []
        src = api._host_ptr(records, 160 * max(world, 1) * n_batches * n_local, "records", keep)
        api._chk(api.lib().kzg_batch_challenges(out, src, world, n_batches, n_local))
        del keep
        return out.raw

    def phase2_r(self, r_le, n_total, offset, n_local):
        self.phase2_launch_r(r_le, n_total, offset)
        return self.phase2_wait()

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
        """Leaves self.bad = one flag per batch (True: the batch holds an undecodable point or a non-canonical blob
        element, i.e. the reference returns Err for it); the group keeps going, its result for that batch is forced
        to None by the caller.  want_records=False leaves the records inside the handle (phase2_launch(None, ...) or
        the device exchange)."""
        bad = C.create_string_buffer(self._b)
        out = C.create_string_buffer(RECORD_BYTES * self._n * self._b) if want_records else None
        api._chk(api.lib().kzg_shard_phase1_wait(out, bad, self.settings._h))
        self.bad = [x != 0 for x in bad.raw]
        return out.raw if want_records else None

    def phase2_launch(self, all_records, n_total, offset):
        """all_records None: the handle's own records (single rank)."""
        api._chk(api.lib().kzg_shard_phase2_launch(all_records, n_total, offset, self.settings._h))

    def phase2_launch_r(self, r_le, n_total, offset):
        api._chk(api.lib().kzg_shard_phase2_launch_r(r_le, n_total, offset, self.settings._h))

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


def _gather_bytes_to_root(dist, payload, lens, device):
    """Variable-length byte strings of every rank, in rank order, on rank 0 only (others get None)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    mx = max(lens + [1])
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, outs, dst=0)
    if rank != 0:
        return None
    return [bytes(o[:l].cpu().numpy().tobytes()) for o, l in zip(outs, lens)]


def verify_blob_kzg_proof_batch_sharded(shard, n_local, backend, dist=None, device="cpu", force_collectives=False, timings=None):
    """ONE batch sharded by blob over the ranks (BASELINE.json configs[4]: 262 144 blobs over 8 GPUs).  Every rank calls
    this with its own shard (rank order = global blob order).  Returns the batch result on every rank.  Raises KzgError
    on every rank if any shard holds an invalid input (first-error identity is not observable in the reference beyond
    "is Err").  The 160 n_total-byte transcript is hashed ONCE, on rank 0 (a serial SHA-256 chain, ~2 GB/s on a host
    core), and the 32-byte challenge is broadcast.  timings (optional dict) receives the seconds spent in each stage."""
    import time

    import torch

    def lap(key, t0):
        if timings is not None:
            timings[key] = timings.get(key, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t = time.perf_counter()
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives):
        if n_local == 0:
            return True
        recs = backend.phase1(shard)
        t = lap("phase1_s", t)
        part = backend.phase2(recs, n_local, 0, n_local)
        t = lap("phase2_s", t)
        ok = backend.finish(part, 1)
        lap("finish_s", t)
        return ok
    world, rank = dist.get_world_size(), dist.get_rank()
    err, recs = None, b""
    if n_local:
        try:
            recs = backend.phase1(shard)
        except api.KzgError as e:  # keep taking part in the collectives, then raise everywhere
            err = e
    t = lap("phase1_s", t)
    # one small all-gather carries the error flag and the shard sizes
    mine = torch.tensor([1 if err else 0, len(recs) // RECORD_BYTES], dtype=torch.int64, device=device)
    meta = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(meta, mine)
    meta = [[int(v) for v in m.tolist()] for m in meta]
    if any(m[0] for m in meta):
        raise err if err else api.KzgError("BadArgs", "invalid input on another rank")
    counts = [m[1] for m in meta]
    n_total = sum(counts)
    if n_total == 0:
        return True  # src/kzg_proof.rs:478-480
    offset = sum(counts[:rank])
    if hasattr(backend, "batch_challenges") and hasattr(backend, "phase2_r"):
        # exchange 1: records to rank 0 only (160 B per blob), which hashes the transcript; 32 bytes come back
        gathered = _gather_bytes_to_root(dist, recs, [RECORD_BYTES * c for c in counts], device)
        t = lap("exchange1_s", t)
        r = torch.zeros(32, dtype=torch.uint8, device=device)
        if rank == 0:
            r_le = backend.batch_challenges(b"".join(gathered), 0, 1, n_total)
            r = torch.frombuffer(bytearray(r_le), dtype=torch.uint8).to(device)
        t = lap("r_hash_s", t)
        dist.broadcast(r, src=0)
        r_le = bytes(r.cpu().numpy().tobytes())
        t = lap("r_broadcast_s", t)
        part = backend.phase2_r(r_le, n_total, offset, n_local) if n_local else b""
    else:
        gathered = _all_gather_bytes(dist, recs, device)  # exchange 1: 160 B per blob, every rank hashes
        t = lap("exchange1_s", t)
        part = backend.phase2(b"".join(gathered), n_total, offset, n_local) if n_local else b""
    t = lap("phase2_s", t)
    parts = [p for p in _all_gather_bytes(dist, part, device) if p]  # exchange 2: 288 B per rank
    t = lap("exchange2_s", t)
    ok = backend.finish(b"".join(parts), len(parts))
    lap("finish_s", t)
    return ok


class PipelinedVerifier:
    """Throughput mode: launch GROUPS of independent batches, and keep several groups in flight from ONE host
    thread with a fixed-order software pipeline.

    At n = 1024 every phase of a batch is a latency-bound serial chain that occupies a sliver of the chip (the
    SHA-256 challenge chain runs on 16 of 1024 SIMDs for ~7 ms), so `group` batches share each kernel launch
    (batch dimension inside the kernels) and `depth` groups overlap through separate handles (= HIP streams).
    Iteration t runs, in this order:   phase1_launch(t);   phase1_wait(t-d1) + exchange 1 + phase2_launch;
    phase2_wait(t-d1-d2) + exchange 2 + finish_launch;   finish_wait(t-d1-d2-d3).
    The order is the same on every rank, so the collectives of different groups never interleave differently
    on different ranks.  With one rank the exchanges and the phase-2 host round trip disappear.

    Exchange 1 with equal shards and a batch count divisible by the world size hashes every transcript ONCE: rank j
    owns the batches [j B / world, (j + 1) B / world) of the group, an all-to-all brings it their records from every
    rank ([src][B / world][n] x 160 B - 1/world of what an all-gather would deliver), it hashes them on its host
    cores, and one all-gather returns every r (32 B per batch) together with the per-batch error flags.

    A batch with an invalid input (the reference's Err) does not stop anything: it runs through the group like any
    other batch, and its result is None - on every rank."""

    def __init__(self, backends, dist=None, device="cpu", depth=(1, 1, 1), equal_shards=False, force_collectives=False):
        """equal_shards: every rank holds the same number of blobs of every batch - enables the bulk exchange (records
        go from the library's device buffer straight into the collective and come back through one pinned buffer)."""
        self.backends = backends
        self.equal_shards = equal_shards
        self._bufs = {}
        # force_collectives: run the exchanges even in a world of one (exercises the transport in tests)
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives)) else None
        self.device = device
        self.depth = depth if self.dist else (depth[0], 0, depth[2])
        self.stats = {"r_hash_s": 0.0, "exchange1_s": 0.0, "exchange2_s": 0.0, "groups": 0}
        assert len(backends) >= sum(self.depth) + 1, "need depth+1 handles"

    def _regroup(self, gathered, n_batches):
        """[rank][batch][n_local] records -> [batch][rank][n_local] (each batch's global transcript order)."""
        per_rank = [[g[i * (len(g) // n_batches): (i + 1) * (len(g) // n_batches)] for i in range(n_batches)] for g in gathered]
        return b"".join(per_rank[r][b] for b in range(n_batches) for r in range(len(gathered)))

    def _exchange_buffers(self, slot, nbytes, recv_bytes):
        """Per handle: device send buffer, receive buffer and its pinned host mirror (cached by size)."""
        import torch
        key = (slot, nbytes, recv_bytes)
        if key not in self._bufs:
            on_gpu = str(self.device) != "cpu"
            send = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            recv = torch.empty(recv_bytes, dtype=torch.uint8, device=self.device)
            host = torch.empty(recv_bytes, dtype=torch.uint8, pin_memory=True) if on_gpu else recv
            self._bufs[key] = (send, recv, host)
        return self._bufs[key]

    def _exchange1_bulk(self, b, slot, n, nb):
        """Equal shards, records in device memory.  Returns after phase 2 of the group has been launched."""
        import time

        import torch
        world, rank = self.dist.get_world_size(), self.dist.get_rank()
        nbytes = RECORD_BYTES * n * nb
        once = nb % world == 0  # every transcript hashed once, by the rank that owns the batch
        send, recv, host = self._exchange_buffers(slot, nbytes, nbytes if once else world * nbytes)
        b.phase1_wait(want_records=False)  # the handle's stream is drained: `send` is complete
        t0 = time.perf_counter()
        src = send if recv.is_cuda else send.cpu()
        if once:
            self.dist.all_to_all_single(recv, src)  # recv = [src rank][nb / world][n] records of MY batches
        else:
            self.dist.all_gather(list(recv.chunk(world)), src)  # [world][batch][n_local] records
        if recv.is_cuda:
            host.copy_(recv, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        flags = torch.tensor([1 if x else 0 for x in b.bad], dtype=torch.uint8)
        if once:
            r_mine = b.batch_challenges(host.data_ptr(), world, nb // world, n)
            t2 = time.perf_counter()
            mine = torch.cat([torch.frombuffer(bytearray(r_mine), dtype=torch.uint8), flags]).to(self.device)
            outs = [torch.empty_like(mine) for _ in range(world)]
            self.dist.all_gather(outs, mine)
            outs = [o.cpu() for o in outs]
            r_all = b"".join(bytes(o[: 32 * (nb // world)].numpy().tobytes()) for o in outs)
            fl = torch.stack([o[32 * (nb // world):] for o in outs]).amax(dim=0)
            b.bad = [bool(x) for x in fl.tolist()]
            b.phase2_launch_r(r_all, world * n, rank * n)
            t3 = time.perf_counter()
            self.stats["r_hash_s"] += t2 - t1
            self.stats["exchange1_s"] += (t1 - t0) + (t3 - t2)
        else:
            fl = flags.to(self.device)
            self.dist.all_reduce(fl, op=self.dist.ReduceOp.MAX)
            b.bad = [bool(x) for x in fl.cpu().tolist()]
            t2 = time.perf_counter()
            b.phase2_launch_gathered(host.data_ptr(), world, rank)  # hashes every batch's transcript on this rank
            self.stats["r_hash_s"] += time.perf_counter() - t2
            self.stats["exchange1_s"] += t2 - t0

    def run(self, groups):
        """groups: list of ((d_blobs, d_commitments, d_proofs, n_local), n_batches): this rank's shard of every batch
        of the group, batches contiguous.  Returns the per-batch results of every group, in order: True / False, or
        None where the reference would return Err."""
        import time
        K, S = len(groups), len(self.backends)
        d1, d2, d3 = self.depth
        results = [None] * K
        bad = [None] * K
        # bulk path: equal shards on every rank and a backend that can hand its records over in device memory
        bulk = bool(self.dist) and self.equal_shards and hasattr(self.backends[0], "records_to_device")
        for t in range(K + d1 + d2 + d3):
            if t < K:
                b = self.backends[t % S]
                b.phase1_launch(groups[t][0], groups[t][1])
                if bulk:
                    nbytes = RECORD_BYTES * groups[t][0][3] * groups[t][1]
                    world = self.dist.get_world_size()
                    send, _, _ = self._exchange_buffers(t % S, nbytes, nbytes if groups[t][1] % world == 0 else world * nbytes)
                    b.records_to_device(send.data_ptr())
            i = t - d1
            if 0 <= i < K:
                b, nb = self.backends[i % S], groups[i][1]
                if bulk:
                    self._exchange1_bulk(b, i % S, groups[i][0][3], nb)
                elif self.dist:
                    recs = b.phase1_wait()
                    t0 = time.perf_counter()
                    flags = bytes(1 if x else 0 for x in getattr(b, "bad", None) or [False] * nb)
                    gathered = _all_gather_bytes(self.dist, flags + recs, self.device)
                    b.bad = [any(g[k] for g in gathered) for k in range(nb)]
                    gathered = [g[nb:] for g in gathered]
                    counts = [len(g) // RECORD_BYTES // nb for g in gathered]
                    self.stats["exchange1_s"] += time.perf_counter() - t0
                    b.phase2_launch(self._regroup(gathered, nb), sum(counts), sum(counts[: self.dist.get_rank()]))
                else:
                    if hasattr(b, "records_to_device"):  # the records never leave the handle
                        b.phase1_wait(want_records=False)
                        b.phase2_launch(None, groups[i][0][3], 0)
                    else:
                        recs = b.phase1_wait()
                        b.phase2_launch(recs, len(recs) // RECORD_BYTES // nb, 0)
                    b.finish_launch(None, 1)
                bad[i] = list(getattr(b, "bad", None) or [False] * nb)
            j = t - d1 - d2
            if self.dist and 0 <= j < K:
                b = self.backends[j % S]
                part = b.phase2_wait()
                t0 = time.perf_counter()
                parts = _all_gather_bytes(self.dist, part, self.device)
                self.stats["exchange2_s"] += time.perf_counter() - t0
                b.finish_launch(b"".join(parts), len(parts))
            k = t - d1 - d2 - d3
            if 0 <= k < K:
                res = self.backends[k % S].finish_wait()
                results[k] = [None if e else r for r, e in zip(res, bad[k])]
                self.stats["groups"] += 1
        return results
