"""Pure-Python big-int model of BLS12-381 + the KZG verification protocol.

TEST INFRASTRUCTURE ONLY (part of oracle/): slow, naive, written from the public curve
parameters and from the reference's protocol flow. Used to (a) validate the pairing programs on the
CPU (tests/test_slp_pairing.py) and (b) cross-check the C oracle's intermediates and constants on
small cases (tests/test_constants.py).  Never imported by the product path or its build: the
constants the HIP code embeds come from tools/gen_constants.py over tools/bls_params.py.

Protocol flow follows /root/reference/src/kzg_proof.rs (cited per function).
Curve definitions: p, r, x = -0xd201000000010000, E: y^2 = x^3 + 4,
E': y^2 = x^3 + 4(u+1) over Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(u+1)),
Fp12 = Fp6[w]/(w^2-v).
"""
import hashlib

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000  # |x|, x is negative

# ---------------------------------------------------------------- Fp2 (a + b u), u^2 = -1


def f2(a, b=0):
    return (a % P, b % P)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)  # u + 1


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_scale(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return ((a[0] * n) % P, (-a[1] * n) % P)


def f2_pow(a, e):
    out = F2_ONE
    while e:
        if e & 1:
            out = f2_mul(out, a)
        a = f2_sqr(a)
        e >>= 1
    return out


def f2_sqrt(a):
    """Square root in Fp2 or None.  (p^2 = 9 mod 16 style generic Tonelli not needed:
    use the norm trick.)"""
    if a == F2_ZERO:
        return F2_ZERO
    # a = (x + y u)^2 ;  norm n = a0^2 + a1^2 must be a square in Fp
    n = (a[0] * a[0] + a[1] * a[1]) % P
    s = pow(n, (P + 1) // 4, P)
    if s * s % P != n:
        return None
    inv2 = pow(2, -1, P)
    for sign in (1, -1):
        t = (a[0] + sign * s) * inv2 % P
        x = pow(t, (P + 1) // 4, P)
        if x * x % P == t and x != 0:
            y = a[1] * pow(2 * x, -1, P) % P
            c = (x, y)
            if f2_sqr(c) == a:
                return c
    return None


# ---------------------------------------------------------------- Fp12 as polynomials in w over Fp2, w^6 = xi
# element = list of 6 Fp2 coefficients c[k] of w^k.
# Relation to the tower used by the C/HIP code ((c0 + c1 v + c2 v^2) + (c3 + c4 v + c5 v^2) w,
# v = w^2):  tower index (i, j) [coefficient of v^j w^i]  <->  w^(2j + i).


def f12_one():
    return [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    for k in range(10, 5, -1):
        t[k - 6] = f2_add(t[k - 6], f2_mul(t[k], XI))
    return t[:6]


def f12_pow(a, e):
    out = f12_one()
    while e:
        if e & 1:
            out = f12_mul(out, a)
        a = f12_mul(a, a)
        e >>= 1
    return out


def f12_conj(a):
    """a^(p^6): w -> -w."""
    return [a[k] if k % 2 == 0 else f2_neg(a[k]) for k in range(6)]


def f12_frobenius(a):
    """a^p: coefficient-wise conjugate times gamma_k = xi^(k (p-1)/6)."""
    return [f2_mul(f2_conj(a[k]), f2_pow(XI, k * (P - 1) // 6)) for k in range(6)]


# ---------------------------------------------------------------- curves (affine, None = infinity)


def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 4) % P == 0


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    return (x3, (lam * (a[0] - x3) - a[1]) % P)


def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def g1_mul(a, k):
    out = None
    k %= R if k >= 0 else R  # scalars are taken mod r only for subgroup points; keep plain int otherwise
    while k:
        if k & 1:
            out = g1_add(out, a)
        a = g1_add(a, a)
        k >>= 1
    return out


def g1_mul_int(a, k):
    """Multiply by a plain non-negative integer (no reduction mod r) - for subgroup checks."""
    out = None
    while k:
        if k & 1:
            out = g1_add(out, a)
        a = g1_add(a, a)
        k >>= 1
    return out


B2 = f2_scale(XI, 4)


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_scale(f2_sqr(a[0]), 3), f2_inv(f2_scale(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_neg(a):
    return None if a is None else (a[0], f2_neg(a[1]))


def g2_mul(a, k):
    out = None
    k %= R
    while k:
        if k & 1:
            out = g2_add(out, a)
        a = g2_add(a, a)
        k >>= 1
    return out


# ---------------------------------------------------------------- (de)compression (ZCash / IETF format)


def g1_decompress(b, check_subgroup=True):
    """48 bytes -> affine point / None (infinity); raises ValueError when rejected."""
    if len(b) != 48:
        raise ValueError("length")
    c, inf, sort = b[0] >> 7 & 1, b[0] >> 6 & 1, b[0] >> 5 & 1
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    if not c:
        raise ValueError("compression flag clear")
    if inf:
        if sort or x:
            raise ValueError("bad infinity encoding")
        return None
    if x >= P:
        raise ValueError("x >= p")
    y2 = (x * x * x + 4) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        raise ValueError("not on curve")
    if (y > (P - 1) // 2) != bool(sort):
        y = P - y
    pt = (x, y)
    if check_subgroup and g1_mul_int(pt, R) is not None:
        raise ValueError("not in subgroup")
    return pt


def g1_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def g2_decompress(b):
    if len(b) != 96:
        raise ValueError("length")
    c, inf, sort = b[0] >> 7 & 1, b[0] >> 6 & 1, b[0] >> 5 & 1
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    if not c:
        raise ValueError("compression flag clear")
    if inf:
        if sort or x0 or x1:
            raise ValueError("bad infinity")
        return None
    if x0 >= P or x1 >= P:
        raise ValueError("x >= p")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise ValueError("not on curve")
    # lexicographically largest: compare c1 first, then c0
    def largest(t):
        return t[1] > (P - 1) // 2 or (t[1] == 0 and t[0] > (P - 1) // 2)

    if largest(y) != bool(sort):
        y = f2_neg(y)
    return (x, y)


def g2_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), y = pt
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80
    if y[1] > (P - 1) // 2 or (y[1] == 0 and y[0] > (P - 1) // 2):
        b[0] |= 0x20
    return bytes(b)


G1_GEN = g1_decompress(
    bytes.fromhex(
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    ),
    check_subgroup=False,
)
G2_GEN = g2_decompress(
    bytes.fromhex(
        "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    )
)

# ---------------------------------------------------------------- pairing (affine Miller loop, naive final exp)


def _line(T, Q, Pt):
    """Line through twist points T, Q (tangent if equal) evaluated at Pt in G1, scaled by w^3:
    l*w^3 = (lam*x1 - y1) - lam*xP * w^2 + yP * w^3   (pairing section)."""
    x1, y1 = T
    if T == Q:
        lam = f2_mul(f2_scale(f2_sqr(x1), 3), f2_inv(f2_scale(y1, 2)))
    else:
        lam = f2_mul(f2_sub(Q[1], y1), f2_inv(f2_sub(Q[0], x1)))
    xP, yP = Pt
    c = [F2_ZERO] * 6
    c[0] = f2_sub(f2_mul(lam, x1), y1)
    c[2] = f2_neg(f2_scale(lam, xP))
    c[3] = (yP % P, 0)
    return c


def miller_loop(pairs):
    """prod f_{|x|,Q}(P) over (P, Q) pairs, conjugated because x < 0. Infinity pairs skipped."""
    f = f12_one()
    live = [(Pt, Q) for Pt, Q in pairs if Pt is not None and Q is not None]
    Ts = [Q for _, Q in live]
    for bit in bin(X_ABS)[3:]:
        f = f12_mul(f, f)
        for k, (Pt, Q) in enumerate(live):
            f = f12_mul(f, _line(Ts[k], Ts[k], Pt))
            Ts[k] = g2_add(Ts[k], Ts[k])
        if bit == "1":
            for k, (Pt, Q) in enumerate(live):
                f = f12_mul(f, _line(Ts[k], Q, Pt))
                Ts[k] = g2_add(Ts[k], Q)
    return f12_conj(f)


def final_exponentiation(f):
    return f12_pow(f, (P**12 - 1) // R)


def pairings_verify(a1, a2, b1, b2):
    """reference src/pairings.rs:5-9:  e(a1,a2) == e(b1,b2)."""
    f = miller_loop([(g1_neg(a1), a2), (b1, b2)])
    return final_exponentiation(f) == f12_one()


# ---------------------------------------------------------------- KZG protocol

FIELD_ELEMENTS_PER_BLOB = 4096
OMEGA = 0x564C0A11A0F704F4FC3E8ACFE0F8245F0AD1347B378FBF96E206DA11A5D36306  # SCALE2_ROOT_OF_UNITY[12]


def bitrev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2)


def roots_of_unity():
    """reference build.rs:131-170 + :89-105: powers of omega, bit-reversal permuted."""
    pw = [1]
    for _ in range(4095):
        pw.append(pw[-1] * OMEGA % R)
    return [pw[bitrev(i, 12)] for i in range(4096)]


_ROOTS = None


def roots():
    global _ROOTS
    if _ROOTS is None:
        _ROOTS = roots_of_unity()
    return _ROOTS


def scalar_from_be_canonical(b):
    """src/kzg_proof.rs:27-43."""
    v = int.from_bytes(b, "big")
    if len(b) != 32 or v >= R:
        raise ValueError("non canonical scalar")
    return v


def blob_to_polynomial(blob):
    """src/dtypes.rs:48-57."""
    if len(blob) != 131072:
        raise ValueError("blob length")
    return [scalar_from_be_canonical(blob[32 * i : 32 * i + 32]) for i in range(4096)]


def compute_challenge(blob, commitment_bytes):
    """src/kzg_proof.rs:46-72 (+ :74-91: digest as big-endian integer mod r)."""
    t = b"FSBLOBVERIFY_V1_" + (0).to_bytes(8, "big") + (4096).to_bytes(8, "big") + blob + commitment_bytes
    assert len(t) == 131152
    return int.from_bytes(hashlib.sha256(t).digest(), "big") % R


def evaluate_polynomial_in_evaluation_form(poly, z):
    """src/kzg_proof.rs:94-133."""
    w = roots()
    for i in range(4096):
        if z == w[i]:
            return poly[i]
    out = 0
    for i in range(4096):
        out += pow(z - w[i], -1, R) * w[i] % R * poly[i]
    out %= R
    out = out * pow(4096, -1, R) % R
    return out * (pow(z, 4096, R) - 1) % R


def verify_kzg_proof_impl(C, z, y, Pi, tau_g2):
    """src/kzg_proof.rs:203-223: e(C - yG, G2) == e(pi, [tau]G2 - [z]G2)."""
    x_minus_z = g2_add(tau_g2, g2_neg(g2_mul(G2_GEN, z)))
    p_minus_y = g1_add(C, g1_neg(g1_mul(G1_GEN, y)))
    return pairings_verify(p_minus_y, G2_GEN, Pi, x_minus_z)


def verify_kzg_proof(cb, zb, yb, pb, tau_g2):
    """src/kzg_proof.rs:353-397. Raises ValueError for Err(..)."""
    z = scalar_from_be_canonical(zb)
    y = scalar_from_be_canonical(yb)
    C = g1_decompress(cb)
    Pi = g1_decompress(pb)
    return verify_kzg_proof_impl(C, z, y, Pi, tau_g2)


def verify_blob_kzg_proof(blob, cb, pb, tau_g2):
    """src/kzg_proof.rs:446-470."""
    C = g1_decompress(cb)
    poly = blob_to_polynomial(blob)
    Pi = g1_decompress(pb)
    z = compute_challenge(blob, cb)
    y = evaluate_polynomial_in_evaluation_form(poly, z)
    return verify_kzg_proof_impl(C, z, y, Pi, tau_g2)


def compute_r(cbs, zs, ys, pbs, endian="little"):
    """src/kzg_proof.rs:291-348 (z, y serialised little-endian there: quirk Q1)."""
    n = len(cbs)
    t = b"RCKZGBATCH___V1_" + (4096).to_bytes(8, "big") + n.to_bytes(8, "big")
    for i in range(n):
        t += cbs[i] + zs[i].to_bytes(32, endian) + ys[i].to_bytes(32, endian) + pbs[i]
    return int.from_bytes(hashlib.sha256(t).digest(), "big") % R


def verify_blob_kzg_proof_batch(blobs, cbs, pbs, tau_g2, endian="little", want_intermediates=False):
    """src/kzg_proof.rs:472-525 + :399-444."""
    if len(blobs) == 0:
        return True
    if len(blobs) == 1:
        return verify_blob_kzg_proof(blobs[0], cbs[0], pbs[0], tau_g2)
    if len(blobs) != len(cbs) or len(blobs) != len(pbs):
        raise ValueError("length mismatch")
    Cs = [g1_decompress(c) for c in cbs]
    Ps = [g1_decompress(p) for p in pbs]
    zs, ys = [], []
    for i, blob in enumerate(blobs):
        poly = blob_to_polynomial(blob)
        z = compute_challenge(blob, cbs[i])
        zs.append(z)
        ys.append(evaluate_polynomial_in_evaluation_form(poly, z))
    r = compute_r(cbs, zs, ys, pbs, endian)
    A = None
    B = None
    rp = 1
    for i in range(len(blobs)):
        A = g1_add(A, g1_mul(Ps[i], rp))
        cmy = g1_add(Cs[i], g1_neg(g1_mul(G1_GEN, ys[i])))
        B = g1_add(B, g1_mul(cmy, rp))
        B = g1_add(B, g1_mul(Ps[i], rp * zs[i] % R))
        rp = rp * r % R
    ok = pairings_verify(A, tau_g2, B, G2_GEN)
    if want_intermediates:
        return ok, r, A, B
    return ok
