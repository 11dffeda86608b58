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


# ---------------------------------------------------------------- ONE proof at a time: the reference's own equation
# KzgProof::verify_kzg_proof (src/kzg_proof.rs:353-397) checks  e(C - [y]G, G2) == e(pi, [tau]G2 - [z]G2).  Its scalars z
# and y are known before any point is decoded, so both scalar multiplications and the 68 line triples of the per-call G2
# point Q = [tau]G2 - [z]G2 are computed BESIDE the point decode (program SCALARS), and what is left behind the decode is
# the pairing itself (program VERIFY3) - no MSM, no multiples tables.
#   SCALARS: inputs = 32 table entries of the G2 generator (8-bit windows of z: entry w = [digit_w 2^(8w)]G2, homogeneous
#            (X, Y, Z) over Fp2, the identity (0 : 1 : 0) for a zero digit - selected by a glue kernel, pure data movement),
#            32 entries of the G1 generator for y, and [tau]G2 (affine).  Sums by a tree of COMPLETE additions (Renes,
#            Costello, Batina 2016, algorithm 7, a = 0: identity operands and equal operands need no special case), then the
#            lines of Q from projective coordinates.  Outputs: 408 line coefficients, [y]G (X, Y, Z), Z of Q (2 Fp: zero
#            only if z = tau, when the caller takes the general path instead).
#   VERIFY3: inputs = pi (x, y, z) and C (x, y, z) homogeneous with z in {0, 1} (identity = (0 : 1 : 0)), [y]G, the 408
#            line coefficients of Q; settings inputs = the lines of [tau]G2 and of the generator as for VERIFY (only the
#            generator's are read).  B = C - [y]G by one complete addition in the graph; points stay homogeneous: a line
#            c0 + c1 x + c4 y is scaled by Z to c0 Z + c1 X + c4 Y (an identity contributes c4 Y w^3: killed).
N_WINDOWS = 32


def cadd(p, q, mul_b3):
    """Complete addition on y^2 = x^3 + b in homogeneous projective coordinates (RCB 2016, algorithm 7).  Works over F and F2
    values alike; mul_b3(x) = 3 b x."""
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    t0, t1, t2 = X1 * X2, Y1 * Y2, Z1 * Z2
    t3 = (X1 + Y1) * (X2 + Y2) - (t0 + t1)
    t4 = (Y1 + Z1) * (Y2 + Z2) - (t1 + t2)
    Y3 = (X1 + Z1) * (X2 + Z2) - (t0 + t2)
    t0 = t0.dbl() + t0
    t2 = mul_b3(t2)
    Z3 = t1 + t2
    t1 = t1 - t2
    Y3 = mul_b3(Y3)
    X3 = t3 * t1 - t4 * Y3
    Y3 = Y3 * t0 + t1 * Z3
    Z3 = Z3 * t4 + t0 * t3
    return X3, Y3, Z3


def tree_sum(pts, mul_b3):
    while len(pts) > 1:
        pts = [cadd(pts[i], pts[i + 1], mul_b3) if i + 1 < len(pts) else pts[i] for i in range(0, len(pts), 2)]
    return pts[0]


def lines_of_projective(g, Q):
    """build_prep's Miller-loop lines for Q = (X : Y : Z) homogeneous over Fp2 (Z any non-zero value): the running point
    starts at Q itself; the doubling is RCB 2016 algorithm 9 (two product levels and one multiplication by the constant
    3 b' between them, against four levels of dbl-2007-bl: this chain of 63 doublings IS the program's running time); the
    five additions are general projective additions (EFD add-1998-cmo-2), and an addition's line
    c0 = u x2 - v y2, c1 = -u, c4 = v is scaled by Z2 to  u X2 - v Y2,  -u Z2,  v Z2.
    The tangent's line at T = (X : Y : Z), scaled by 2 Y Z^2... : c0 = Y^2 - 3 b' Z^2, c1 = -3 X^2, c4 = 2 Y Z (as build_prep)."""
    QX, QY, QZ = Q
    twelve = g.const(12)
    b3 = lambda x: x.mul_xi().mul_fp(twelve)  # 3 b' = 12 (1 + u): a product step by a constant (as additions it is two more
    # linear steps per doubling - 549 steps instead of 502 for the same estimated time)
    X, Y, Z = QX, QY, QZ
    lines = []
    for bit in x_bits():
        t0, t1, t2, xy, XX = Y.sqr(), Y * Z, Z.sqr(), X * Y, X.sqr()
        bz = b3(t2)
        lines.append((t0 - bz, -(XX.dbl() + XX), t1.dbl()))
        z8 = t0.dbl().dbl().dbl()
        X3 = bz * z8
        Y3 = t0 + bz
        Z3 = t1 * z8
        t0 = t0 - (bz.dbl() + bz)
        X, Y, Z = (t0 * xy).dbl(), X3 + t0 * Y3, Z3
        if bit == "1":
            Y1Z2, X1Z2, Z1Z2 = Y * QZ, X * QZ, Z * QZ
            u = QY * Z - Y1Z2
            v = QX * Z - X1Z2
            lines.append((u * QX - v * QY, -(u * QZ), v * QZ))
            uu, vv = u.sqr(), v.sqr()
            vvv = v * vv
            Rr = vv * X1Z2
            A = uu * Z1Z2 - vvv - Rr.dbl()
            X, Y, Z = v * A, u * (Rr - A) - vvv * Y1Z2, vvv * Z1Z2
    assert len(lines) == N_LINES
    return lines


def build_scalars(test_inputs=None):
    """Inputs: 32 x (X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1) entries of the G2 table, 32 x (X Y Z) entries of the G1 table, then
    [tau]G2 affine (x.c0 x.c1 y.c0 y.c1): 292 values.  Outputs, in the order VERIFY3 reads its inputs 6..416: X Y Z of
    sum(G1 entries), the 408 line coefficients of Q = [tau]G2 - sum(G2 entries); then Z.c0 Z.c1 of Q."""
    old = trace.TOWER
    trace.TOWER = "schoolbook"
    try:
        g = Graph()
        ti = test_inputs or ([0, 0, 1, 0, 0, 0] * N_WINDOWS + [0, 1, 0] * N_WINDOWS + [0, 0, 0, 0])
        it = iter(ti)
        nxt = lambda: g.inp(next(it))
        e2 = []
        for _ in range(N_WINDOWS):
            c = [nxt() for _ in range(6)]
            e2.append((F2(c[0], c[1]), F2(c[2], c[3]), F2(c[4], c[5])))
        e1 = [tuple(nxt() for _ in range(3)) for _ in range(N_WINDOWS)]
        tau = [nxt() for _ in range(4)]
        twelve = g.const(12)
        b3_g2 = lambda x: x.mul_xi().mul_fp(twelve)  # 3 b' = 12 (1 + u)
        b3_g1 = lambda x: x * twelve                 # 3 b  = 12
        zg2 = tree_sum(e2, b3_g2)
        one2 = F2(g.const(1), g.const(0))
        Q = cadd((F2(tau[0], tau[1]), F2(tau[2], tau[3]), one2), (zg2[0], -zg2[1], zg2[2]), b3_g2)
        yg = tree_sum(e1, b3_g1)
        for c in yg:
            g.output(c)
        for c0, c1, c4 in lines_of_projective(g, Q):
            for c in (c0, c1, c4):
                g.output(c.c0)
                g.output(c.c1)
        g.output(Q[2].c0)
        g.output(Q[2].c1)
        return g
    finally:
        trace.TOWER = old


def build_verify3(test_inputs=None, test_prep=None):
    """Inputs (per instance): pi (X Y Z), C (X Y Z) - homogeneous, Z in {0, 1} - then [y]G (X Y Z, any scale), the 408 line
    coefficients of Q, and Z.c0 Z.c1 of Q (= SCALARS' outputs in their order).  Settings inputs: the prepared lines of
    [tau]G2 and of the generator (2 x 408 Fp; only the generator's are used).  Outputs: 6 Fp values, all zero <=>
    e(pi, Q) == e(C - [y]G, G2); then Z.c0, Z.c1 of Q passed through (both zero: Q is the identity, z = tau - the lines
    are then meaningless and the caller takes the general path)."""
    old = trace.TOWER
    trace.TOWER = "schoolbook"
    try:
        g = Graph()
        ti = test_inputs or ([0, 1, 0, 0, 1, 0, 0, 1, 0] + [0] * (N_LINES * 6) + [1, 0])
        inp = [g.inp(v) for v in ti]
        tp = test_prep or [0] * (2 * N_LINES * 6)
        prep = [g.inp(v) for v in tp]
        twelve = g.const(12)
        A = (inp[0], -inp[1], inp[2])  # pair 1 is (-pi, Q)
        yG = (inp[6], -inp[7], inp[8])
        B = cadd((inp[3], inp[4], inp[5]), yG, lambda x: x * twelve)
        qlines = inp[9: 9 + N_LINES * 6]
        zq = inp[9 + N_LINES * 6:]
        assert len(zq) == 2

        def coeffs(k, i):
            src, base = (qlines, i * 6) if k == 0 else (prep, (N_LINES + i) * 6)
            return [F2(src[base + 2 * j], src[base + 2 * j + 1]) for j in range(3)]

        zero2 = F2(g.const(0), g.const(0))
        pts = [A, B]

        def line(k, i):
            c0, c1, c4 = coeffs(k, i)
            X, Y, Z = pts[k]
            return F12(F6(c0.mul_fp(Z), c1.mul_fp(X), zero2), F6(zero2, c4.mul_fp(Y), zero2))

        factors, i = [], 0
        for bit in x_bits():
            t = line(0, i) * line(1, i)
            i += 1
            if bit == "1":
                t = t * (line(0, i) * line(1, i))
                i += 1
            factors.append(t)
        assert i == N_LINES
        f = f12_one(g)
        group = 4
        for pos in range(0, len(factors), group):
            grp = factors[pos: pos + group]
            m = grp[0]
            for t in grp[1:]:
                m = m.sqr() * t
            for _ in grp:
                f = f.sqr()
            f = f * m
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
        for c in zq:
            g.output(c)
        return g
    finally:
        trace.TOWER = old


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
    # ONE verify_kzg_proof at a time: the scalar multiplications + the lines of the per-call G2 point, and the pairing behind them
    for name, graph, n_inst in (("scalars", build_scalars(), N_WINDOWS * 9 + 4), ("verify3", build_verify3(), 9 + N_LINES * 6 + 2)):
        blob, stats = schedule2(graph, lanes=LATENCY_LANES, n_instance_inputs=n_inst, out_values=name == "scalars")
        with open(os.path.join(DATA, "slp_%s.bin" % name), "wb") as f:
            f.write(blob)
        print(name, stats)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
