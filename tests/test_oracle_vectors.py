"""Pins the CPU oracle against every golden vector the reference's own tests hold
(SURVEY.md 8c): the 175 c-kzg-4844 mainnet vectors (strict: null <=> error), the two scalar
KATs the reference asserts (src/kzg_proof.rs:739-753, :755-778), and the derived anchors of
SURVEY.md 10 / 10.2 / 10.3.  CPU only."""
import pytest

import golden_data as G
import oracle_lib as O


@pytest.fixture(scope="module")
def settings():
    return O.Settings.mainnet()


def _expect(fn, expected):
    try:
        got = fn()
    except O.OracleError:
        got = None
    assert got == expected


def test_constants_anchor():
    a = G.kat()["anchors"]
    frR, frR2, frinv, fpR, fpR2, fpinv = O.constants()
    assert frR.hex() == a["fr_R"] and frR2.hex() == a["fr_R2"] and "%016x" % frinv == a["fr_inv64"]
    assert fpR.hex() == a["fp_R"] and fpR2.hex() == a["fp_R2"] and "%016x" % fpinv == a["fp_inv64"]


def test_sha256_known_answers():
    import hashlib
    for msg in [b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 63, b"a" * 64, b"a" * 65, bytes(range(256)) * 17]:
        assert O.sha256(msg) == hashlib.sha256(msg).digest()


def test_roots_and_g2_anchor(settings):
    a = G.kat()["anchors"]
    assert [settings.root(i).hex() for i in range(4)] == a["roots_of_unity_first4"]
    ts = open(G.os.path.join(O.ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    assert settings.g2(0).hex() == ts[2 + 4096]  # decompress -> compress round trip, = standard generator
    assert settings.g2(1).hex() == ts[2 + 4096 + 1]
    t = a["tau_g2"]
    g2 = settings.g2(1)
    assert g2[48:].hex() == t["x_c0"] and (bytes([g2[0] & 0x1F]) + g2[1:48]).hex() == t["x_c1"]
    xy, inf = O.g1_decompress(bytes.fromhex(a["commitment_fb324"]["compressed"]))
    assert not inf and xy[:48].hex() == a["commitment_fb324"]["x"] and xy[48:].hex() == a["commitment_fb324"]["y"]


def test_verify_kzg_proof_vectors(settings):
    cases = G.vectors()["verify_kzg_proof"]
    assert len(cases) == 122
    for c in cases:
        f = [bytes.fromhex(c[k]) for k in ("commitment", "z", "y", "proof")]
        if [len(x) for x in f] != [48, 32, 32, 48]:
            assert c["output"] is None  # Bytes48/Bytes32::from_slice fails: src/dtypes.rs:20-25
            continue
        _expect(lambda: O.verify_kzg_proof(*f, settings), c["output"])


def test_verify_kzg_proof_batch_from_single_vectors(settings):
    """verify_kzg_proof_batch (src/kzg_proof.rs:399-444) has no vector file of its own; it must agree with the
    single-proof vectors: all true tuples as one batch -> true, any false tuple mixed in -> false, each tuple alone
    as a 1-element batch -> its own expected output (r^0 = 1 makes the equations identical)."""
    good, bad = [], []
    for c in G.vectors()["verify_kzg_proof"]:
        f = [bytes.fromhex(c[k]) for k in ("commitment", "z", "y", "proof")]
        if [len(x) for x in f] != [48, 32, 32, 48]:
            continue
        _expect(lambda: O.verify_kzg_proof_batch(*[[x] for x in f], settings), c["output"])
        if c["output"] is not None:
            (good if c["output"] else bad).append(f)
    assert len(good) == 54 and len(bad) == 48
    cols = lambda ts: [list(c) for c in zip(*ts)]
    assert O.verify_kzg_proof_batch(*cols(good), settings) is True
    for b in bad[::5]:
        assert O.verify_kzg_proof_batch(*cols(good[:9] + [b] + good[9:20]), settings) is False
    assert O.verify_kzg_proof_batch([], [], [], [], settings) is True


def test_verify_blob_kzg_proof_vectors(settings):
    cases = G.vectors()["verify_blob_kzg_proof"]
    assert len(cases) == 29
    for c in cases:
        blob, cm, pf = G.blob(c["blob"]), bytes.fromhex(c["commitment"]), bytes.fromhex(c["proof"])
        if (len(blob), len(cm), len(pf)) != (131072, 48, 48):
            assert c["output"] is None
            continue
        _expect(lambda: O.verify_blob_kzg_proof(blob, cm, pf, settings), c["output"])


@pytest.mark.parametrize("be", [False, True])
def test_verify_blob_kzg_proof_batch_vectors(settings, be):
    cases = G.vectors()["verify_blob_kzg_proof_batch"]
    assert len(cases) == 24
    for c in cases:
        blobs = [G.blob(b) for b in c["blobs"]]
        cs = [bytes.fromhex(x) for x in c["commitments"]]
        ps = [bytes.fromhex(x) for x in c["proofs"]]
        if any(len(b) != 131072 for b in blobs) or any(len(x) != 48 for x in cs + ps):
            assert c["output"] is None
            continue
        if not (len(blobs) == len(cs) == len(ps)):
            assert c["output"] is None  # src/kzg_proof.rs:491-501 (lives in the caller of the C API)
            continue
        _expect(lambda: O.verify_blob_kzg_proof_batch(blobs, cs, ps, settings, be=be), c["output"])


def test_kat_compute_challenge():
    k = G.kat()["compute_challenge"]
    c = G.case("verify_blob_kzg_proof", k["case"])
    assert O.compute_challenge(G.blob(c["blob"]), bytes.fromhex(c["commitment"])).hex() == k["z"]


def test_kat_evaluate_polynomial(settings):
    k = G.kat()["evaluate_polynomial_in_evaluation_form"]
    c = G.case("verify_blob_kzg_proof", k["case"])
    assert O.evaluate_polynomial_in_evaluation_form(G.blob(c["blob"]), bytes.fromhex(k["z"]), settings).hex() == k["y"]


def test_zy_table(settings):
    for suffix, (z, y) in G.kat()["zy_table"].items():
        c = G.case("verify_blob_kzg_proof", suffix)
        blob = G.blob(c["blob"])
        zz = O.compute_challenge(blob, bytes.fromhex(c["commitment"]))
        assert zz.hex() == z
        assert O.evaluate_polynomial_in_evaluation_form(blob, zz, settings).hex() == y


def test_evaluate_at_root_of_unity(settings):
    """src/kzg_proof.rs:109-111: z equal to a root returns the polynomial value itself."""
    c = G.case("verify_blob_kzg_proof", "correct_proof_19b3f3f8c98ea31e")
    blob = G.blob(c["blob"])
    for i in (0, 1, 2, 77, 4095):
        assert O.evaluate_polynomial_in_evaluation_form(blob, settings.root(i), settings) == blob[32 * i: 32 * i + 32]


def test_batch_intermediates(settings):
    for e in G.kat()["batch_intermediates"]:
        c = G.case("verify_blob_kzg_proof_batch", e["case"])
        blobs = [G.blob(b) for b in c["blobs"]]
        cs = [bytes.fromhex(x) for x in c["commitments"]]
        ps = [bytes.fromhex(x) for x in c["proofs"]]
        for conv, be in (("le", False), ("be", True)):
            ok, zs, ys, r, A, B = O.verify_blob_kzg_proof_batch_ex(blobs, cs, ps, settings, be=be)
            assert ok == c["output"]
            assert r.hex() == e[conv]["r"] and A.hex() == e[conv]["A"] and B.hex() == e[conv]["B"]


def test_multithreaded_batch_matches(settings):
    tuples = G.valid_blob_tuples()
    blobs, cs, ps = zip(*tuples)
    a = O.verify_blob_kzg_proof_batch_ex(blobs, cs, ps, settings, nthreads=1)
    b = O.verify_blob_kzg_proof_batch_ex(blobs, cs, ps, settings, nthreads=4)
    assert a == b and a[0] is True
