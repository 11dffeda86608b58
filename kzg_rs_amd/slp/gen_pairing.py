"""Build the two SLP programs of the pairing check (see trace.py / schedule.py).

PREP   (once per KzgSettings):  Q in G2 (affine)  ->  the 68 Miller-loop line coefficient triples
        of Q ("G2Prepared" in the reference: src/pairings.rs:6 G2Prepared::from).
VERIFY (once per pairing check): A, B in G1 (Jacobian) + the prepared lines of [tau]G2 and of the
        G2 generator  ->  6 Fp values that are all zero  iff  e(A, [tau]G2) == e(B, G2)
        (reference src/pairings.rs:5-9 with a1 = A, a2 = g2_points[1], b1 = B, b2 = generator,
        as called from src/kzg_proof.rs:436-441).

Mathematics (checked numerically against an independent model in tests/test_slp_pairing.py):
  * untwist psi(x', y') = (x'/w^2, y'/w^3); a line through psi(T), psi(Q) evaluated at P and
    multiplied by w^3 is  (lam x1 - y1) - lam xP w^2 + yP w^3  (lam = twist slope in Fp2), i.e. the
    sparse Fp12 element  c0 + c1 v + c4 v w.  Factors from proper subfields are killed by the
    final exponentiation, so lines are scaled freely by Fp2 elements and by ZP^3, which lets P
    stay in Jacobian coordinates (no inversion) and makes an identity P (Z = 0) contribute a
    killed factor, exactly like skipping the pair.
  * the check  f^((p^12-1)/r) == 1  is done inversion-free:  with E = (p^2+1) * 3(p^4-p^2+1)/r,
    f^((p^6-1)E) = conj(f^E)/f^E, so the pairing product is 1  iff  f^E lies in Fp6;  the negative
    parts of the x-chain are moved to a denominator D and multiplied back as conj(D) (their
    quotient differs by the norm D*conj(D), which is in Fp6).

    python -m kzg_rs_amd.slp.gen_pairing            # writes kzg_rs_amd/data/slp_*.bin
"""
import os
import sys

from . import trace
from .schedule import schedule
from .schedule2 import schedule2
from .trace import F2, F6, F12, Graph, f12_one, f2_const, P

X_ABS = 0xD201000000010000
N_LINES = 68  # 63 doublings + 5 additions
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "data")


def x_bits():
    return bin(X_ABS)[3:]  # below the top bit, MSB first


def build_prep(test_q=None):
    """Inputs: x.c0 x.c1 y.c0 y.c1 of Q (affine, on the twist).  Outputs: 68 x (c0, c1, c4) Fp2."""
    g = Graph()
    tq = test_q or ((0, 0), (0, 0))
    qx = F2(g.inp(tq[0][0]), g.inp(tq[0][1]))
    qy = F2(g.inp(tq[1][0]), g.inp(tq[1][1]))
    b3 = f2_const(g, (12, 12))  # 3 b' = 12 (1 + u)
    X, Y, Z = qx, qy, F2(g.const(1), g.const(0))
    lines = []
    for bit in x_bits():
        # doubling: homogeneous projective, a = 0  (EFD dbl-2007-bl); line c0 = Y^2 - 3b'Z^2, c1 = -3X^2, c4 = 2YZ
        XX, YY, ZZ = X.sqr(), Y.sqr(), Z.sqr()
        w = XX.dbl() + XX
        s = (Y * Z).dbl()
        lines.append((YY - b3 * ZZ, -w, s))
        ss = s.sqr()
        sss = s * ss
        Rr = Y * s
        RR = Rr.sqr()
        B = (X * Rr).dbl()
        h = w.sqr() - B.dbl()
        X, Y, Z = h * s, w * (B - h) - RR.dbl(), sss
        if bit == "1":
            # mixed addition T + Q (EFD madd-1998-cmo); line c0 = u x2 - v y2, c1 = -u, c4 = v
            u = qy * Z - Y
            v = qx * Z - X
            lines.append((u * qx - v * qy, -u, v))
            uu, vv = u.sqr(), v.sqr()
            vvv = v * vv
            Rr = vv * X
            A = uu * Z - vvv - Rr.dbl()
            X, Y, Z = v * A, u * (Rr - A) - vvv * Y, vvv * Z
    assert len(lines) == N_LINES
    for c0, c1, c4 in lines:
        for c in (c0, c1, c4):
            g.output(c.c0)
            g.output(c.c1)
    return g


def exp_x(a):
    acc = a
    for bit in x_bits():
        acc = acc.sqr()
        if bit == "1":
            acc = acc * a
    return acc


def build_verify(test_inputs=None, test_prep=None, group=1):
    """Inputs (per instance): A.X A.Y A.Z B.X B.Y B.Z (Jacobian, Montgomery on the GPU).
    Settings inputs: prepared lines of Q1 = [tau]G2 then of Q2 = G2 generator (2 x 408 Fp).
    Outputs: the 6 Fp coefficients of the w-odd half of s; all zero <=> e(A,Q1) == e(B,Q2).
    group: Miller-loop bits taken together.  One bit is f <- f^2 T_i (T_i = the product of the bit's lines): two dependent
    Fp12 levels.  k bits at once are f <- f^(2^k) M with M = (..(T_1^2 T_2)^2 ..)^2 T_k, which depends on the lines only and
    is computed beside the main chain: k + 1 dependent levels instead of 2k, for a few more products (the latency program
    takes group = 4: 432 product levels instead of 480; the throughput program keeps the fewest products)."""
    g = Graph()
    ti = test_inputs or [0, 1, 0, 0, 1, 0]
    AX, AY, AZ, BX, BY, BZ = [g.inp(v) for v in ti]
    tp = test_prep or [0] * (2 * N_LINES * 6)
    prep = [g.inp(v) for v in tp]

    def line_coeffs(k, i):
        base = (k * N_LINES + i) * 6
        return [F2(prep[base + 2 * j], prep[base + 2 * j + 1]) for j in range(3)]

    # pair 1: (-A, Q1); pair 2: (B, Q2)
    pts = [(AX, -AY, AZ), (BX, BY, BZ)]
    scal = []
    for (PX, PY, PZ) in pts:
        z2 = PZ * PZ
        scal.append((z2 * PZ, PX * PZ, PY))  # ZP^3, XP*ZP, YP
    zero2 = F2(g.const(0), g.const(0))

    def line(k, i):
        c0, c1, c4 = line_coeffs(k, i)
        z3, xz, y = scal[k]
        return F12(F6(c0.mul_fp(z3), c1.mul_fp(xz), zero2), F6(zero2, c4.mul_fp(y), zero2))

    f = f12_one(g)
    i = 0
    if group <= 1:
        for bit in x_bits():
            f = f.sqr()
            f = f * (line(0, i) * line(1, i))
            i += 1
            if bit == "1":
                f = f * (line(0, i) * line(1, i))
                i += 1
    else:
        factors = []  # T_i per bit
        for bit in x_bits():
            t = line(0, i) * line(1, i)
            i += 1
            if bit == "1":
                t = t * (line(0, i) * line(1, i))
                i += 1
            factors.append(t)
        for pos in range(0, len(factors), group):
            grp = factors[pos: pos + group]
            m = grp[0]
            for t in grp[1:]:
                m = m.sqr() * t
            for _ in grp:
                f = f.sqr()
            f = f * m
    assert i == N_LINES
    # inversion-free final test
    u = f.frobenius(2, g) * f
    a = exp_x(u) * u
    v = exp_x(a) * a
    vX = exp_x(v)
    vX2 = exp_x(vX)
    vX3 = exp_x(vX2)
    N = vX2.frobenius(1, g) * v.frobenius(3, g) * vX
    D = v.frobenius(1, g) * vX3 * vX.frobenius(2, g)
    u3 = u.sqr() * u
    s = N * u3 * D.conj()
    for c in (s.c1.c0, s.c1.c1, s.c1.c2):
        g.output(c.c0)
        g.output(c.c1)
    return g


LATENCY_LANES = 256  # four wavefronts, one per SIMD of a CU: every product level of the schoolbook towers fits one step
# (measured on MI355X: 1.18 ms with 128 lanes, 1.20 with 192, 1.15 with 256 - the program is bound by step latency)


def build_verify_latency(test_inputs=None, test_prep=None):
    """The same check traced with the schoolbook tower formulas (trace.TOWER: 2.2x the products, a quarter of the
    dependent additions) and four Miller-loop bits per step of the main chain - the graph the latency scheduler
    (schedule2.py) is given."""
    old = trace.TOWER
    trace.TOWER = "schoolbook"
    try:
        return build_verify(test_inputs, test_prep, group=4)
    finally:
        trace.TOWER = old


def main(lanes=64):
    os.makedirs(DATA, exist_ok=True)
    for name, graph, n_inst in (("prep", build_prep(), 4), ("verify", build_verify(), 6)):
        blob, stats = schedule(graph, lanes=lanes, n_instance_inputs=n_inst)
        path = os.path.join(DATA, "slp_%s.bin" % name)
        with open(path, "wb") as f:
            f.write(blob)
        print(name, stats)
    # VERIFY once more for a single check at a time: several wavefronts per instance, radix-2^29 lazy arithmetic
    blob, stats = schedule2(build_verify_latency(), lanes=LATENCY_LANES, n_instance_inputs=6)
    with open(os.path.join(DATA, "slp_verify2.bin"), "wb") as f:
        f.write(blob)
    print("verify2", stats)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
