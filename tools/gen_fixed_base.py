"""Fixed-base tables of the two generators for the one-proof path (kzg_rs_amd/slp/gen_pairing.py SCALARS):
    kzg_rs_amd/data/fixed_base.bin =  G2 table | G1 table
    G2: [32 windows][256 digits] x (X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1), entry (w, d) = [d 2^(8w)] G2  homogeneous with Z = 1,
        the identity (0 : 1 : 0) for d = 0
    G1: [32][256] x (X Y Z), entry (w, d) = [d 2^(8w)] G1
every coordinate a canonical 12x32-bit Montgomery element (x 2^384 mod p, little-endian words) - the instance-input format of
the latency programs, so that the selection by the digits of a scalar is a plain copy.  Build-time constants of the curve
(the generators are given by coordinates in tools/bls_params.py); nothing here depends on a trusted setup.
    python tools/gen_fixed_base.py"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import bls_params as bp  # noqa: E402

P = bp.P
OUT = os.path.join(os.path.dirname(HERE), "kzg_rs_amd", "data", "fixed_base.bin")
WINDOWS, DIGITS = 32, 256


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, -a[1] * n % P)


class Fp1:  # field operations for the affine point arithmetic below: over Fp, and over Fp2 = Fp[u]/(u^2 + 1)
    mul = staticmethod(lambda a, b: a * b % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    inv = staticmethod(lambda a: pow(a, -1, P))
    small = staticmethod(lambda k, a: k * a % P)
    coords = staticmethod(lambda a: [a])


class Fp2:
    mul = staticmethod(f2_mul)
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P))
    inv = staticmethod(f2_inv)
    small = staticmethod(lambda k, a: (k * a[0] % P, k * a[1] % P))
    coords = staticmethod(lambda a: [a[0], a[1]])


def add(F, p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if y1 != y2:
            return None
        lam = F.mul(F.small(3, F.mul(x1, x1)), F.inv(F.small(2, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def mont_words(v):
    v = v * (1 << 384) % P
    return struct.pack("<12I", *[(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)])


def table(F, gen, one, zero):
    out = bytearray()
    base = gen
    for w in range(WINDOWS):
        acc = None
        for d in range(DIGITS):
            X, Y, Z = (zero, one, zero) if acc is None else (acc[0], acc[1], one)
            for c in (X, Y, Z):
                for v in F.coords(c):
                    out += mont_words(v)
            acc = add(F, acc, base)
        base = acc  # [256] base = the next window's base
    return bytes(out)


def main():
    blob = table(Fp2, bp.G2_GEN, (1, 0), (0, 0)) + table(Fp1, bp.G1_GEN, 1, 0)
    assert len(blob) == WINDOWS * DIGITS * (6 + 3) * 48
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "wb") as f:
        f.write(blob)
    print(OUT, len(blob))


if __name__ == "__main__":
    main()
