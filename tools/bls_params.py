"""BLS12-381 parameters and the few integer helpers the build-time generators need (tools/gen_constants.py).

Self-contained on purpose: the PRODUCT build must not import anything from oracle/ (the oracle is the checker).  The
numbers are the published curve parameters; the generators are given by their affine coordinates (the ZCash-encoded
generator points, decompressed) and verified to lie on their curves at import time.  tests/test_constants.py checks
every array this feeds into csrc/constants.inc against values recomputed in the test and against the oracle.
"""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000  # |x|; the curve parameter x is negative
# SCALE2_ROOT_OF_UNITY[12] of the reference (src/consts.rs:90-95): a primitive 4096th root of unity in Fr
OMEGA = 0x564C0A11A0F704F4FC3E8ACFE0F8245F0AD1347B378FBF96E206DA11A5D36306

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
# (x.c0, x.c1), (y.c0, y.c1) over Fp2 = Fp[u]/(u^2 + 1)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)


def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


assert (G1_GEN[1] ** 2 - G1_GEN[0] ** 3 - 4) % P == 0, "G1 generator not on y^2 = x^3 + 4"
_x3 = _f2mul(_f2mul(G2_GEN[0], G2_GEN[0]), G2_GEN[0])
_y2 = _f2mul(G2_GEN[1], G2_GEN[1])
assert ((_y2[0] - _x3[0] - 4) % P, (_y2[1] - _x3[1] - 4) % P) == (0, 0), "G2 generator not on y^2 = x^3 + 4(u + 1)"
assert pow(OMEGA, 4096, R) == 1 and pow(OMEGA, 2048, R) == R - 1


def g1_add(a, b):
    """Affine addition on E: y^2 = x^3 + 4 (None = the point at infinity)."""
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


def g1_mul_int(a, k):
    """[k]a for a plain non-negative integer k (double-and-add)."""
    out = None
    while k:
        if k & 1:
            out = g1_add(out, a)
        a = g1_add(a, a)
        k >>= 1
    return out
