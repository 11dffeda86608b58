"""Straight-line-program (SLP) tracer for Fp arithmetic.

The pairing check is ~25 000 Fp multiplications with a dependency depth of a few hundred.  One
GPU lane would need ~30 ms for it (an Fp product has ~1.5 us latency when a wave is alone on
its SIMD), so instead the Fp-level data-flow graph is recorded ONCE here, scheduled into
wave-wide steps (schedule.py) and executed by a small LDS-resident interpreter kernel
(csrc/slp.hpp): every step all lanes perform one Fp operation on operands held in LDS slots.

This module records the graph.  Every traced value carries a shadow Python integer so that the
traced algorithm can be checked numerically while it is being recorded (self-test) and so that
tests can compare the emitted program, run by the reference interpreter in schedule.py, with an
independent model.
"""

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB

# op kinds
IN, CONST, MUL, ADD, SUB = "in", "const", "mul", "add", "sub"

# Tower multiplication formulas.  "karatsuba": fewest Fp products (3 per Fp2, 6 Fp2 per Fp6, 3 Fp6 per Fp12 product) at
# the price of ~11 dependent additions between two product levels - right for the one-wavefront interpreter whose
# throughput matters (many instances side by side).  "schoolbook": 4 / 9 / 4 products, no pre-additions and short sums
# after them (4 dependent binary additions per level, 2 after 4-input fusion) - right for the latency interpreter
# (csrc/slp2.hpp), where the dependency depth is the running time and lanes are plentiful.
TOWER = "karatsuba"


class Graph:
    def __init__(self):
        self.kind = []  # per node
        self.a = []
        self.b = []
        self.val = []  # shadow value (plain integer mod P)
        self.cse = {}
        self.consts = {}  # value -> node
        self.inputs = []  # node ids in declaration order
        self.outputs = []

    def _new(self, kind, a, b, val):
        self.kind.append(kind)
        self.a.append(a)
        self.b.append(b)
        self.val.append(val % P)
        return len(self.kind) - 1

    def inp(self, val=0):
        n = self._new(IN, len(self.inputs), -1, val)
        self.inputs.append(n)
        return F(self, n)

    def const(self, val):
        val %= P
        if val not in self.consts:
            self.consts[val] = self._new(CONST, -1, -1, val)
        return F(self, self.consts[val])

    def _is_const(self, n, v):
        return self.kind[n] == CONST and self.val[n] == v

    def op(self, kind, x, y):
        a, b = x.n, y.n
        if kind == MUL:
            if self._is_const(a, 0) or self._is_const(b, 0):
                return self.const(0)
            if self._is_const(a, 1):
                return y
            if self._is_const(b, 1):
                return x
            if a > b:
                a, b = b, a
            val = self.val[a] * self.val[b]
        elif kind == ADD:
            if self._is_const(a, 0):
                return y
            if self._is_const(b, 0):
                return x
            if a > b:
                a, b = b, a
            val = self.val[a] + self.val[b]
        else:
            if self._is_const(b, 0):
                return x
            if a == b:
                return self.const(0)
            val = self.val[a] - self.val[b]
        if self.kind[a] == CONST and self.kind[b] == CONST:
            return self.const(val)
        key = (kind, a, b)
        n = self.cse.get(key)
        if n is None:
            n = self._new(kind, a, b, val)
            self.cse[key] = n
        return F(self, n)

    def output(self, f):
        self.outputs.append(f.n)


class F:
    """A traced Fp value."""

    __slots__ = ("g", "n")

    def __init__(self, g, n):
        self.g = g
        self.n = n

    @property
    def v(self):
        return self.g.val[self.n]

    def __add__(self, o):
        return self.g.op(ADD, self, o)

    def __sub__(self, o):
        return self.g.op(SUB, self, o)

    def __mul__(self, o):
        return self.g.op(MUL, self, o)

    def __neg__(self):
        return self.g.op(SUB, self.g.const(0), self)

    def dbl(self):
        return self + self


# ---------------------------------------------------------------- towers over traced values
# Fp2 = Fp[u]/(u^2+1); Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u; Fp12 = Fp6[w]/(w^2 - v)


class F2:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1

    def __add__(self, o):
        return F2(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return F2(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return F2(-self.c0, -self.c1)

    def dbl(self):
        return F2(self.c0.dbl(), self.c1.dbl())

    def conj(self):
        return F2(self.c0, -self.c1)

    def __mul__(self, o):
        if TOWER == "schoolbook":
            return F2(self.c0 * o.c0 - self.c1 * o.c1, self.c0 * o.c1 + self.c1 * o.c0)
        # Karatsuba: 3 Fp products
        t0 = self.c0 * o.c0
        t1 = self.c1 * o.c1
        m = (self.c0 + self.c1) * (o.c0 + o.c1)
        return F2(t0 - t1, m - t0 - t1)

    def sqr(self):
        m = self.c0 * self.c1
        if TOWER == "schoolbook":
            return F2(self.c0 * self.c0 - self.c1 * self.c1, m.dbl())
        return F2((self.c0 + self.c1) * (self.c0 - self.c1), m.dbl())

    def mul_fp(self, k):
        return F2(self.c0 * k, self.c1 * k)

    def mul_xi(self):
        return F2(self.c0 - self.c1, self.c0 + self.c1)

    @property
    def v(self):
        return (self.c0.v, self.c1.v)


def f2_const(g, v):
    return F2(g.const(v[0]), g.const(v[1]))


def f2_mul_const(a, cv, g):
    """a * constant Fp2 value cv (python tuple); cheaper forms when cv is in Fp or is 1."""
    if cv == (1, 0):
        return a
    if cv[1] == 0:
        return a.mul_fp(g.const(cv[0]))
    return a * f2_const(g, cv)


class F6:
    __slots__ = ("c0", "c1", "c2")

    def __init__(self, c0, c1, c2):
        self.c0, self.c1, self.c2 = c0, c1, c2

    def __add__(self, o):
        return F6(self.c0 + o.c0, self.c1 + o.c1, self.c2 + o.c2)

    def __sub__(self, o):
        return F6(self.c0 - o.c0, self.c1 - o.c1, self.c2 - o.c2)

    def __neg__(self):
        return F6(-self.c0, -self.c1, -self.c2)

    def dbl(self):
        return F6(self.c0.dbl(), self.c1.dbl(), self.c2.dbl())

    def mul_v(self):
        return F6(self.c2.mul_xi(), self.c0, self.c1)

    def __mul__(self, o):
        a0, a1, a2, b0, b1, b2 = self.c0, self.c1, self.c2, o.c0, o.c1, o.c2
        if TOWER == "schoolbook":
            return F6(a0 * b0 + (a1 * b2 + a2 * b1).mul_xi(), a0 * b1 + a1 * b0 + (a2 * b2).mul_xi(), a0 * b2 + a1 * b1 + a2 * b0)
        v0, v1, v2 = a0 * b0, a1 * b1, a2 * b2
        c0 = ((a1 + a2) * (b1 + b2) - v1 - v2).mul_xi() + v0
        c1 = (a0 + a1) * (b0 + b1) - v0 - v1 + v2.mul_xi()
        c2 = (a0 + a2) * (b0 + b2) - v0 - v2 + v1
        return F6(c0, c1, c2)


class F12:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1

    def __mul__(self, o):
        if TOWER == "schoolbook":
            return F12(self.c0 * o.c0 + (self.c1 * o.c1).mul_v(), self.c0 * o.c1 + self.c1 * o.c0)
        t0 = self.c0 * o.c0
        t1 = self.c1 * o.c1
        m = (self.c0 + self.c1) * (o.c0 + o.c1) - t0 - t1
        return F12(t0 + t1.mul_v(), m)

    def sqr(self):
        ab = self.c0 * self.c1
        if TOWER == "schoolbook":
            return F12(self.c0 * self.c0 + (self.c1 * self.c1).mul_v(), ab.dbl())
        m = (self.c0 + self.c1) * (self.c0 + self.c1.mul_v()) - ab - ab.mul_v()
        return F12(m, ab.dbl())

    def conj(self):
        return F12(self.c0, -self.c1)

    def coeffs(self):
        """Fp2 coefficients of w^0..w^5 (tower (c_i . c_j) is w^(2j+i))."""
        return [self.c0.c0, self.c1.c0, self.c0.c1, self.c1.c1, self.c0.c2, self.c1.c2]

    @staticmethod
    def from_coeffs(c):
        return F12(F6(c[0], c[2], c[4]), F6(c[1], c[3], c[5]))

    def frobenius(self, k, g):
        """a^(p^k): coefficient-wise conj^k, times gamma_{j,k} = xi^(j (p^k - 1)/6)."""
        out = []
        for j, c in enumerate(self.coeffs()):
            cc = c.conj() if (k & 1) else c
            out.append(f2_mul_const(cc, frob_gamma(j, k), g))
        return F12.from_coeffs(out)


def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = _f2mul(r, a)
        a = _f2mul(a, a)
        e >>= 1
    return r


_gamma_cache = {}


def frob_gamma(j, k):
    if (j, k) not in _gamma_cache:
        _gamma_cache[(j, k)] = _f2pow((1, 1), j * (P**k - 1) // 6)
    return _gamma_cache[(j, k)]


def f12_one(g):
    z = F2(g.const(0), g.const(0))
    return F12(F6(F2(g.const(1), g.const(0)), z, z), F6(z, z, z))
