"""Open-ended differential fuzz of the entry points against the CPU oracle (a longer-running sibling of
tests/test_gpu_parity.py::test_mutation_fuzz_against_oracle).   python tools/fuzz_campaign.py [seconds] [seed] [threads]
threads > 1: that many host threads run the campaign CONCURRENTLY on the one shared settings handle (each with its own generator,
seed + thread index) - their small calls meet in the handle's small-call queue (csrc/capi_coalesce.hpp) and share launches, and
every answer must still be the oracle's for the caller's own input."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_data as G
import oracle_lib as O
from kzg_rs_amd import api
from kzg_rs_amd.api import Blob, Bytes32, Bytes48, KzgError, KzgProof, KzgSettings

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
G1_INF = bytes([0xC0]) + bytes(47)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = random.Random(seed)
st, ost = KzgSettings.load_trusted_setup_file(), O.Settings.mainnet()
tuples = G.valid_blob_tuples()
edge_fr = [R, R - 1, R + 1, (1 << 256) - 1, 0, 1, 2 * R]
edge_fp = [P, P - 1, P + 1, 0, 1]


def rand_point(rng=rng):
    k = rng.randrange(6)
    if k == 0:
        return G1_INF
    if k == 1:  # x with random flags
        return bytes([rng.randrange(256)]) + rng.randbytes(47)
    if k == 2:  # edge x values with the compression flag set
        v = rng.choice(edge_fp) % (1 << 381)
        b = bytearray(v.to_bytes(48, "big")); b[0] |= 0x80 | (0x20 if rng.randrange(2) else 0)
        return bytes(b)
    if k == 3:  # a valid subgroup point: a multiple of a vector's commitment
        return O.g1_mul(rng.choice(tuples)[1], rng.randrange(1, R).to_bytes(32, "big"))
    if k == 4:  # on the curve, probably outside the subgroup: a random x that decodes unchecked
        while True:
            b = bytearray(rng.randrange(P).to_bytes(48, "big")); b[0] |= 0x80
            try:
                O.g1_decompress(bytes(b)); return bytes(b)   # oracle decode checks the subgroup: accept either way
            except O.OracleError:
                return bytes(b)
    t = rng.choice(tuples); return t[rng.randrange(1, 3)]


def mutate(blob, c, p, rng=rng):
    kind = rng.randrange(12)
    blob, c, p = bytearray(blob), bytearray(c), bytearray(p)
    if kind == 0: c[rng.randrange(48)] ^= 1 << rng.randrange(8)
    elif kind == 1: p[rng.randrange(48)] ^= 1 << rng.randrange(8)
    elif kind == 2:
        i = rng.randrange(4096); blob[32 * i: 32 * i + 32] = (rng.choice(edge_fr) % (1 << 256)).to_bytes(32, "big")
    elif kind == 3: blob[rng.randrange(131072)] ^= 1 << rng.randrange(8)
    elif kind == 4: c[:] = rand_point(rng)
    elif kind == 5: p[:] = rand_point(rng)
    elif kind == 6: c, p = p, c
    elif kind == 7: blob[:] = bytes(131072)
    elif kind == 8:
        for _ in range(rng.randrange(1, 40)):
            i = rng.randrange(4096); blob[32 * i: 32 * i + 32] = rng.randrange(R).to_bytes(32, "big")
    return bytes(blob), bytes(c), bytes(p)


def res(fn):
    try:
        return fn()
    except (KzgError, O.OracleError):
        return None


# Deterministic prologue: proof tuples built so that the MSM's special cases happen in a known bucket.  With one tuple
# r^0 = 1, so B = C + z pi - y G has scalar 1 on C, z on pi and -y on G: z = 1 puts pi beside C in bucket 1 of the first
# window, y = r - 1 puts G there too.  C = pi / G gives P + P (the doubling branch of the bucket addition), C = -pi / -G
# gives P - P (the identity branch), with affine or Jacobian table entries depending on the process's layout.
def neg48(pt):
    b = bytearray(pt); b[0] ^= 0x20; return bytes(b)


G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
one, rm1, zero = (1).to_bytes(32, "big"), (R - 1).to_bytes(32, "big"), bytes(32)
pi0 = next(t[2] for t in tuples if t[2] != G1_INF)
special = []
for c in (pi0, neg48(pi0), G1_GEN, neg48(G1_GEN), G1_INF):
    for y in (one, rm1, zero):
        for prf in (pi0, G1_GEN, neg48(G1_GEN), G1_INF):
            special.append((c, one, y, prf))
for c, z, y, prf in special:
    want = res(lambda: O.verify_kzg_proof_batch([c], [z], [y], [prf], ost))
    got = res(lambda: KzgProof.verify_kzg_proof_batch([Bytes48(c)], [Bytes32(z)], [Bytes32(y)], [Bytes48(prf)], st))
    got1 = res(lambda: KzgProof.verify_kzg_proof(Bytes48(c), Bytes32(z), Bytes32(y), Bytes48(prf), st))
    want2 = res(lambda: O.verify_kzg_proof_batch([c, c], [z, z], [y, y], [prf, prf], ost))
    got2 = res(lambda: KzgProof.verify_kzg_proof_batch([Bytes48(c)] * 2, [Bytes32(z)] * 2, [Bytes32(y)] * 2, [Bytes48(prf)] * 2, st))
    if got != want or got1 != want or got2 != want2:
        print("MISMATCH special tuple c=%s y=%s pi=%s got=%r/%r/%r want=%r/%r" % (c.hex()[:8], y.hex()[-4:], prf.hex()[:8], got, got1, got2, want, want2)); sys.exit(1)
print("special bucket cases: %d tuples x 3 calls, no mismatch (%d true)" % (len(special), sum(1 for c, z, y, prf in special if res(lambda: O.verify_kzg_proof_batch([c], [z], [y], [prf], ost)))))

import threading
t_end = time.time() + budget
failures, lock = [], threading.Lock()
totals = {"cases": 0, True: 0, False: 0, None: 0}


def campaign(tid):
    rng = random.Random(seed + 7919 * tid)
    cases, counts = 0, {True: 0, False: 0, None: 0}

    def fail(msg):
        with lock:
            failures.append(msg)

    while time.time() < t_end and not failures:
        n = rng.choice([1, 1, 2, 3, 4, 7, 12])
        batch = [list(rng.choice(tuples)) for _ in range(n)]
        for _ in range(rng.randrange(0, 3)):
            k = rng.randrange(n); batch[k] = list(mutate(*batch[k], rng=rng))
        blobs, cs, ps = [list(x) for x in zip(*batch)]
        want = res(lambda: O.verify_blob_kzg_proof_batch(blobs, cs, ps, ost))
        got = res(lambda: KzgProof.verify_blob_kzg_proof_batch([Blob(b) for b in blobs], [Bytes48(c) for c in cs], [Bytes48(p) for p in ps], st))
        if got != want:
            return fail("MISMATCH batch seed=%d thread=%d case=%d n=%d got=%r want=%r" % (seed, tid, cases, n, got, want))
        counts[got] += 1
        # proof-tuple entry points on (C, z, y, pi) with random / edge scalars
        m = rng.choice([1, 2, 5])
        tc = [rand_point(rng) if rng.randrange(4) == 0 else rng.choice(tuples)[1] for _ in range(m)]
        tp = [rand_point(rng) if rng.randrange(4) == 0 else rng.choice(tuples)[2] for _ in range(m)]
        tz = [(rng.choice(edge_fr) % (1 << 256) if rng.randrange(5) == 0 else rng.randrange(R)).to_bytes(32, "big") for _ in range(m)]
        ty = [(rng.choice(edge_fr) % (1 << 256) if rng.randrange(5) == 0 else rng.randrange(R)).to_bytes(32, "big") for _ in range(m)]
        want = res(lambda: O.verify_kzg_proof_batch(tc, tz, ty, tp, ost))
        got = res(lambda: KzgProof.verify_kzg_proof_batch([Bytes48(x) for x in tc], [Bytes32(x) for x in tz], [Bytes32(x) for x in ty], [Bytes48(x) for x in tp], st))
        if got != want:
            return fail("MISMATCH proof batch seed=%d thread=%d case=%d got=%r want=%r" % (seed, tid, cases, got, want))
        want = res(lambda: O.verify_kzg_proof(tc[0], tz[0], ty[0], tp[0], ost))
        got = res(lambda: KzgProof.verify_kzg_proof(Bytes48(tc[0]), Bytes32(tz[0]), Bytes32(ty[0]), Bytes48(tp[0]), st))
        if got != want:
            return fail("MISMATCH single proof seed=%d thread=%d case=%d got=%r want=%r" % (seed, tid, cases, got, want))
        # the same tuples as m INDEPENDENT proofs through one call: entry i = verify_kzg_proof of tuple i
        wants = [res(lambda: O.verify_kzg_proof(tc[i], tz[i], ty[i], tp[i], ost)) for i in range(m)]
        gots = api.verify_kzg_proofs(tc, tz, ty, tp, st)
        if gots != wants:
            return fail("MISMATCH independent proofs seed=%d thread=%d case=%d got=%r want=%r" % (seed, tid, cases, gots, wants))
        cases += 1
    with lock:
        totals["cases"] += cases
        for k in (True, False, None):
            totals[k] += counts[k]


if n_threads <= 1:
    campaign(0)
else:
    ths = [threading.Thread(target=campaign, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
if failures:
    print(failures[0]); sys.exit(1)
q = st.small_queue_stats()
print("fuzz campaign seed=%d%s: %d cases x 4 entry points, no mismatch; blob-batch outcomes %s; small-call queue: %d requests in %d launches (largest %d)"
      % (seed, "" if n_threads <= 1 else " threads=%d on one handle" % n_threads, totals["cases"], {k: totals[k] for k in (True, False, None)},
         q["requests"], q["launches"], q["max_items"]))
