"""CPU tests (gloo, world_size 2 and 3) of the multi-GPU sharding logic in kzg_rs_amd/distributed.py.

The collective choreography (shard by contiguous blob range -> all-gather transcript records ->
per-shard partial sums with the global power offset -> all-gather partials -> fold + one pairing)
is exercised with a compute backend built from the CPU ORACLE's primitives, and the result must
equal the unsharded oracle result on the same inputs - true batches, a corrupted proof on one
rank, an invalid blob on one rank, empty shards and uneven shards.  The HIP backend implements the
same three phases on the GPU (tests/test_gpu_parity.py covers it against the oracle)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G1_INF = bytes([0xC0]) + bytes(47)


class OracleBackend:
    """The three shard phases restated with oracle primitives (src/kzg_proof.rs:251-277, :291-348, :399-444).
    Partials travel as two 48-byte compressed points padded to 288 bytes (opaque to the wrapper)."""

    def __init__(self, O, osettings):
        self.O, self.st = O, osettings

    def phase1(self, shard):
        from kzg_rs_amd import api
        blobs, cs, ps = shard
        out = b""
        for b, c, p in zip(blobs, cs, ps):
            try:
                self.O.g1_decompress(c)
                self.O.g1_decompress(p)
                z = self.O.compute_challenge(b, c)
                y = self.O.evaluate_polynomial_in_evaluation_form(b, z, self.st)
            except self.O.OracleError:
                raise api.KzgError("BadArgs", "invalid input")
            out += c + z[::-1] + y[::-1] + p  # z, y little-endian inside the transcript (quirk Q1)
        self._mine = (cs, ps, out)
        return out

    def batch_challenges(self, records, world, n_batches, n_local):
        """Same contract as HipBackend.batch_challenges for the [n_batches][n_total] layout (world = 0): r per batch,
        32 little-endian bytes each."""
        assert world == 0
        out = b""
        for b in range(n_batches):
            recs = [records[160 * (b * n_local + i): 160 * (b * n_local + i) + 160] for i in range(n_local)]
            out += self.O.compute_r(b"".join(x[:48] for x in recs), b"".join(x[48:80][::-1] for x in recs),
                                    b"".join(x[80:112][::-1] for x in recs), b"".join(x[112:] for x in recs), n_local)[::-1]
        return out

    def phase2_r(self, r_le, n_total, offset, n_local):
        return self.phase2(None, n_total, offset, n_local, r=int.from_bytes(r_le, "little"))

    def phase2(self, all_records, n_total, offset, n_local, r=None):
        O = self.O
        cs, ps, mine = self._mine
        if n_total == 1:
            r = 1
        elif r is None:
            r = int.from_bytes(self.batch_challenges(all_records, 0, 1, n_total), "little")
        A, B, g = G1_INF, G1_INF, 0
        for i in range(n_local):
            rec = mine[160 * i: 160 * i + 160]
            z, y = int.from_bytes(rec[48:80], "little"), int.from_bytes(rec[80:112], "little")
            rp = pow(r, offset + i, R)
            A = O.g1_add(A, O.g1_mul(ps[i], rp.to_bytes(32, "big")))
            B = O.g1_add(B, O.g1_mul(cs[i], rp.to_bytes(32, "big")))
            B = O.g1_add(B, O.g1_mul(ps[i], (rp * z % R).to_bytes(32, "big")))
            g = (g + rp * y) % R
        B = O.g1_add(B, O.g1_mul(G1_GEN, ((R - g) % R).to_bytes(32, "big")))
        return (A + B).ljust(288, b"\0")

    def finish(self, partials, world):
        O = self.O
        A, B = G1_INF, G1_INF
        for k in range(world):
            A = O.g1_add(A, partials[288 * k: 288 * k + 48])
            B = O.g1_add(B, partials[288 * k + 48: 288 * k + 96])
        return O.pairings_verify(A, self.st.g2(1), B, self.st.g2(0))

    # launch / wait halves over a launch group (same contract as HipBackend; "launch" just records the arguments)
    def phase1_launch(self, shard, n_batches=1):
        self._grp = (shard, n_batches)

    def phase1_wait(self):
        """Like HipBackend.phase1_wait: a batch with an invalid input does not raise, it sets self.bad[b] (and leaves
        zero records); the pipeline carries the flag and reports None for that batch."""
        from kzg_rs_amd import api
        (blobs, cs, ps), nb = self._grp
        n = len(blobs) // nb
        self._per_batch, out, self.bad = [], b"", []
        for b in range(nb):
            sl = slice(b * n, (b + 1) * n)
            try:
                out += self.phase1((blobs[sl], cs[sl], ps[sl]))
                self._per_batch.append(self._mine)
                self.bad.append(False)
            except api.KzgError:
                out += bytes(160 * n)
                self._per_batch.append(None)
                self.bad.append(True)
        self._nloc = n
        return out

    def phase2_launch(self, all_records, n_total, offset):
        self._parts = b""
        for b, mine in enumerate(self._per_batch):
            if mine is None or self.bad[b]:  # flagged on this or another rank: the result is forced to None anyway
                self._parts += (G1_INF + G1_INF).ljust(288, b"\0")
                continue
            self._mine = mine
            self._parts += self.phase2(all_records[160 * n_total * b: 160 * n_total * (b + 1)], n_total, offset, self._nloc)

    def phase2_wait(self):
        return self._parts

    def finish_launch(self, partials, world):
        nb = len(self._per_batch)
        if partials is None:
            partials, world = self._parts, 1
        self._res = []
        for b in range(nb):
            mine = b"".join(partials[288 * (k * nb + b): 288 * (k * nb + b + 1)] for k in range(world))
            self._res.append(self.finish(mine, world))

    def finish_wait(self):
        return self._res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scenario, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd import api
    from kzg_rs_amd.distributed import verify_blob_kzg_proof_batch_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = O.Settings.mainnet()
    tuples = G.valid_blob_tuples()  # 7 valid mainnet (blob, C, pi)
    blobs, cs, ps = [list(x) for x in zip(*tuples)]
    if scenario == "bad_proof":
        ps[5] = O.g1_add(ps[5], G1_GEN)
    if scenario == "bad_blob":
        b = bytearray(blobs[6])
        b[64:96] = R.to_bytes(32, "big")
        blobs[6] = bytes(b)
    n = len(blobs)
    if scenario == "uneven":
        bounds = [0, 1, n] if world == 2 else [0, 1, 1, n]  # an empty shard in the 3-rank case
    else:
        bounds = [n * k // world for k in range(world + 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    shard = (blobs[lo:hi], cs[lo:hi], ps[lo:hi])
    backend = OracleBackend(O, st)
    try:
        got = verify_blob_kzg_proof_batch_sharded(shard, hi - lo, backend, dist, "cpu")
    except api.KzgError:
        got = "error"
    try:
        want = O.verify_blob_kzg_proof_batch(blobs, cs, ps, st)
    except O.OracleError:
        want = "error"
    q.put((rank, got, want))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scenario", [(2, "valid"), (2, "bad_proof"), (2, "bad_blob"), (2, "uneven"), (3, "uneven"), (3, "valid")])
def test_sharded_matches_unsharded(world, scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = {"valid": True, "uneven": True, "bad_proof": False, "bad_blob": "error"}[scenario]
    for rank, got, want in res:
        assert got == want == expected, (rank, got, want)


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import golden_data as G
    import oracle_lib as O
    from kzg_rs_amd.distributed import PipelinedVerifier

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = O.Settings.mainnet()
    tuples = G.valid_blob_tuples()[:4]  # 4 valid mainnet tuples; per rank shard = 2 blobs of each batch
    # three launch groups: 2 batches, 1 batch, 2 batches; batch contents are rotations; one batch is corrupted
    def batch(rot, corrupt=False):
        t = tuples[rot:] + tuples[:rot]
        blobs, cs, ps = [list(x) for x in zip(*t)]
        if corrupt:
            ps[3] = O.g1_add(ps[3], G1_GEN)
        return blobs, cs, ps
    def invalid(rot, where):  # a non-canonical field element in one blob: the reference returns Err (src/dtypes.rs:48-57)
        blobs, cs, ps = batch(rot)
        b = bytearray(blobs[where])
        b[64:96] = R.to_bytes(32, "big")
        blobs[where] = bytes(b)
        return blobs, cs, ps
    # the invalid blob sits in the shard of rank 1 only (index 3): every rank must still report None for that batch
    plan = [[batch(0), batch(1, True)], [batch(2)], [invalid(3, 3), batch(1)], [batch(3), batch(1)]]
    want = [[True, False], [True], [None, True], [True, True]]
    groups = []
    for grp in plan:
        gb, gc, gp = [], [], []
        for blobs, cs, ps in grp:  # this rank's contiguous slice of every batch, batches concatenated
            lo, hi = 4 * rank // world, 4 * (rank + 1) // world
            gb += blobs[lo:hi]; gc += cs[lo:hi]; gp += ps[lo:hi]
        groups.append(((gb, gc, gp), len(grp)))
    def unsharded(b, c, p):
        try:
            return O.verify_blob_kzg_proof_batch(b, c, p, st)
        except O.OracleError:
            return None
    for grp, w in zip(plan, want):  # sanity: the unsharded oracle agrees with the plan
        assert [unsharded(b, c, p) for b, c, p in grp] == w
    pv = PipelinedVerifier([OracleBackend(O, st) for _ in range(4)], dist, "cpu", depth=(1, 1, 1))
    got = pv.run(groups)
    q.put((rank, got, want))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_groups_two_ranks():
    """PipelinedVerifier with launch groups of several batches over 2 ranks: records of [rank][batch] must be
    regrouped into per-batch global transcripts, partials folded per batch, groups pipelined in fixed order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in res:
        assert got == want, (rank, got, want)
