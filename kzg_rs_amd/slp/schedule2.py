"""Latency scheduler for a traced Fp graph (trace.Graph): the binary program of the multi-wave interpreter
csrc/slp2.hpp, and a reference interpreter of that format (exact integer semantics of the kernel, used by the CPU tests).

Why a second format.  The one-wave programs of schedule.py are bound by dependency DEPTH, not by work: between two
product levels of the pairing lie ~11 dependent additions, each a whole interpreter step (LDS round trip, carry chain,
conditional subtraction): 5 253 of the 5 755 steps of the verification program are additions.  This scheduler attacks
the depth:
  * values live in radix 2^29 (14 limbs, csrc/fp29.hpp) and are LAZY: nothing is reduced outside a product; every
    value carries a static bound (a multiple of p) computed here, and a subtraction adds a constant 2^e p chosen from
    that bound, with limbs pre-biased so that no limb borrows;
  * a linear step is 4-ary:  dst = +-x1 +- x2 +- x3 +- x4 + BIAS  (the binary add/sub DAG is re-associated: an operand
    that is itself an addition is expanded in place while the operand count stays <= 4);
  * a product step takes pre-added operands:  dst = (a1 + a2) * (b1 + b2)  (an addition that feeds a product is
    inlined into it);
  * with the tracer's "schoolbook" tower formulas (trace.TOWER) two product levels are 2 linear steps apart instead
    of 11, for 2.2x the products - lanes are plentiful: the program is scheduled for several wavefronts.

Binary format (little-endian u32 words):
  [0] magic 0x32504c53 ("SLP2")  [1] lanes  [2] n_slots  [3] n_steps  [4] n_const  [5] n_inputs
  [6] n_settings_inputs  [7] n_outputs  [8] n_load_steps (the LOAD steps are the first steps of the program)
  [9] output mode: 0 = the kernel writes, per output, "is it 0 mod p" (all-zero words / a 1) - what a check needs;
                   1 = it writes the VALUES as canonical 12x32 Montgomery elements - the instance-input format, so that one
                       program's outputs are the next one's inputs (SCALARS -> VERIFY3)
  [10..15] reserved
  const pool : n_const * 16 words  (14 limbs of 29 bits + 2 zero words: field constants in Montgomery form, radix
               2^406, or raw limb patterns - the subtraction biases)
  out slots  : n_outputs words, zero-padded to a multiple of 4 (the descriptors are read as 16-byte vectors)
  descriptors: n_steps * lanes * 4 words
       w0 = s0 | s1 << 16     w1 = s2 | s3 << 16     w2 = dst | bias_slot << 16
       w3 = kind << 30 | active << 29 | src << 26 | neg_mask << 16 | load_index (LOAD)
     kind 0 LIN : dst = sum_i (-1)^neg_i slot[s_i] + slot[bias_slot], carries propagated (limbs 0..12 < 2^29)
     kind 1 MUL : dst = (slot[s0] + slot[s1]) * (slot[s2] + slot[s3]) * 2^-406 mod p, below 2p
     kind 2 LOAD: dst = const pool[load_index] (src 0) | instance input (src 1) | settings input (src 2)
  Slot 0 is the constant zero (absent operands point at it); it is never written.
"""
import heapq
import struct

from .trace import ADD, CONST, IN, MUL, SUB, P

MAGIC2 = 0x32504C53
R406 = 1 << 406
MASK29 = (1 << 29) - 1
K_LIN, K_MUL, K_LOAD = 0, 1, 2
SRC_CONST, SRC_INST, SRC_SET = 0, 1, 2
MAX_PRODUCT_BOUND = 1 << 25  # a b < 2^406 p  <=  bound(a) bound(b) <= 2^25  (p < 2^381)


def limbs29(v):
    assert 0 <= v < (1 << 406)
    return [(v >> (29 * i)) & MASK29 for i in range(14)]


def bias_limbs(e):
    """2^e p with limbs 0..12 boosted by 2^31 (borrowed from the limb above): a limb-wise subtraction of up to three
    normalised values from it never borrows."""
    c = limbs29((1 << e) * P)
    b = [c[0] + (1 << 31)] + [c[i] + (1 << 31) - 4 for i in range(1, 13)] + [c[13] - 4]
    assert sum(x << (29 * i) for i, x in enumerate(b)) == (1 << e) * P and all(0 <= x < (1 << 32) for x in b)
    assert min(b[:13]) >= 3 * MASK29
    return b


class Plan:
    """The rewritten graph: LOAD / LIN / MUL operations over the nodes of a trace.Graph."""

    def __init__(self, g, n_instance_inputs, K=4):
        self.g, self.n_in = g, n_instance_inputs
        n = len(g.kind)
        islin = lambda x: g.kind[x] in (ADD, SUB)
        live = [False] * n
        st = list(g.outputs)
        while st:
            x = st.pop()
            if live[x]:
                continue
            live[x] = True
            if g.kind[x] in (MUL, ADD, SUB):
                st += [g.a[x], g.b[x]]
        h = [0] * n  # binary linear height above the last product (expansion heuristic)
        for x in range(n):
            if live[x] and islin(x):
                h[x] = 1 + max(h[g.a[x]] if islin(g.a[x]) else 0, h[g.b[x]] if islin(g.b[x]) else 0)
        self.lin, self.mul = {}, {}  # node -> [(sign, node)] | ((a1, a2|None), (b1, b2|None))
        need = []

        def want(x):
            if islin(x) and x not in self.lin:
                self.lin[x] = None
                need.append(x)

        for x in range(n):
            if live[x] and g.kind[x] == MUL:
                ops = []
                for o in (g.a[x], g.b[x]):
                    if g.kind[o] == ADD:  # an addition feeding a product is inlined into it
                        ops.append((g.a[o], g.b[o]))
                        want(g.a[o])
                        want(g.b[o])
                    else:
                        ops.append((o, None))
                        want(o)
                self.mul[x] = tuple(ops)
        for o in g.outputs:
            want(o)
        while need:
            x = need.pop()
            terms = [(1, g.a[x]), (1 if g.kind[x] == ADD else -1, g.b[x])]
            while len(terms) < K:
                best = None
                for i, (s, t) in enumerate(terms):
                    if not islin(t):
                        continue
                    new_neg = sum(1 for ss, _ in terms if ss < 0) - (1 if s < 0 else 0) + (2 if (s < 0 and g.kind[t] == ADD) else 1 if (s < 0 or g.kind[t] == SUB) else 0)
                    if new_neg > 3:
                        continue
                    if best is None or h[t] > h[terms[best][1]]:
                        best = i
                if best is None:
                    break
                s, t = terms.pop(best)
                terms += [(s, g.a[t]), (s if g.kind[t] == ADD else -s, g.b[t])]
            assert sum(1 for s, _ in terms if s < 0) <= 3
            terms.sort(key=lambda st_: st_[0] < 0)  # positives first (cosmetic)
            self.lin[x] = terms
            for _, t in terms:
                want(t)
        # sources that must be materialised in a slot
        self.src = set()
        for x, terms in self.lin.items():
            self.src.update(t for _, t in terms if g.kind[t] in (IN, CONST))
        for x, ops in self.mul.items():
            self.src.update(t for op in ops for t in op if t is not None and g.kind[t] in (IN, CONST))
        self.src.update(o for o in g.outputs if g.kind[o] in (IN, CONST))
        # ---- bounds (multiples of p) in topological (= id) order; biases
        self.bound = {}
        self.bias_e = {}  # lin node -> e (bias 2^e p) or None
        for x in sorted(self.src):
            self.bound[x] = 2 if g.kind[x] == IN else 1  # inputs arrive through a conversion product (< 2p); constants are canonical
        for x in sorted(list(self.lin) + list(self.mul)):
            if x in self.mul:
                (a1, a2), (b1, b2) = self.mul[x]
                ba = self.bound[a1] + (self.bound[a2] if a2 is not None else 0)
                bb = self.bound[b1] + (self.bound[b2] if b2 is not None else 0)
                assert ba * bb <= MAX_PRODUCT_BOUND, ("product operands out of range", x, ba, bb)
                self.bound[x] = 2
            else:
                pos = sum(self.bound[t] for s, t in self.lin[x] if s > 0)
                neg = sum(self.bound[t] for s, t in self.lin[x] if s < 0)
                e = None
                if neg:
                    e = max(0, (neg - 1).bit_length())  # 2^e >= neg
                    assert (1 << e) >= neg
                self.bias_e[x] = e
                self.bound[x] = pos + ((1 << e) if e is not None else 0)
                assert self.bound[x] < (1 << 24)
        self.max_bound = max(self.bound.values())


def schedule2(g, lanes=256, n_instance_inputs=None, K=4, margin=None, cost_mul=560, cost_lin=170, cost_load=60, out_values=False):
    """-> (program bytes, statistics).  margin: an operation is not started while its height (remaining critical path) is
    more than `margin` below the most urgent ready operation - keeps early-computable values (the line evaluations of
    all Miller iterations) from occupying LDS slots for the whole program."""
    if n_instance_inputs is None:
        n_instance_inputs = len(g.inputs)
    plan = Plan(g, n_instance_inputs, K)
    n = len(g.kind)
    ops = {}  # op id -> (kind, node)   (op ids: node id for LIN / MUL / LOAD of a source; ("b", e) for a bias constant)
    deps, users = {}, {}
    for x in plan.src:
        ops[x] = K_LOAD
        deps[x] = []
    for e in sorted({e for e in plan.bias_e.values() if e is not None}):
        ops[("b", e)] = K_LOAD
        deps[("b", e)] = []
    for x, terms in plan.lin.items():
        ops[x] = K_LIN
        d = {t for _, t in terms}
        if plan.bias_e[x] is not None:
            d.add(("b", plan.bias_e[x]))
        deps[x] = sorted(d, key=str)
    for x, opnds in plan.mul.items():
        ops[x] = K_MUL
        deps[x] = sorted({t for op in opnds for t in op if t is not None})
    for o, ds in deps.items():
        for d in ds:
            users.setdefault(d, []).append(o)
    cost = {K_LIN: cost_lin, K_MUL: cost_mul, K_LOAD: cost_load}
    # heights by reverse topological order: node ids are topological; bias loads have no deps
    order = [o for o in ops if not isinstance(o, tuple)]
    order.sort()
    height = {}
    for o in reversed(order):
        height[o] = cost[ops[o]] + max([height[u] for u in users.get(o, [])] + [0])
    for o in ops:
        if isinstance(o, tuple):
            height[o] = cost[K_LOAD] + max([height[u] for u in users.get(o, [])] + [0])
    pending = {o: len(ds) for o, ds in deps.items()}
    ready = {K_LIN: [], K_MUL: [], K_LOAD: []}
    seq = 0

    def push(o):
        nonlocal seq
        seq += 1
        heapq.heappush(ready[ops[o]], (-height[o], seq, o))

    for o, c in pending.items():
        if c == 0:
            push(o)
    if margin is None:
        margin = 6 * (cost_mul + 2 * cost_lin)
    steps = []
    while any(ready.values()):
        hmax = max(-ready[k][0][0] for k in ready if ready[k])
        # loads first (cheap, and nothing waits on vmcnt in the other step kinds), then every ready linear step, then products
        # (all loads are ready at the start and are issued at once, whatever their urgency: a handful of steps that wait
        # on global memory, and ~850 slots = 54 KB of LDS for the whole program instead of a load step every few levels)
        kind = K_LOAD if ready[K_LOAD] else None
        if kind is None:
            for k in (K_LIN, K_MUL):
                if ready[k] and -ready[k][0][0] >= hmax - margin:
                    kind = k
                    break
        if kind is None:
            kind = max((k for k in ready if ready[k]), key=lambda k: -ready[k][0][0])
        batch = []
        while ready[kind] and len(batch) < lanes and (kind == K_LOAD or not batch or -ready[kind][0][0] >= hmax - margin):
            batch.append(heapq.heappop(ready[kind])[2])
        steps.append((kind, batch))
        for o in batch:
            for u in users.get(o, []):
                pending[u] -= 1
                if pending[u] == 0:
                    push(u)
    assert all(c == 0 for c in pending.values()), "unscheduled operations (cycle?)"
    # ---- slot allocation (slot 0 = zero)
    last_use = {}
    for si, (_, batch) in enumerate(steps):
        for o in batch:
            for d in deps[o]:
                last_use[d] = si
    out_set = set(g.outputs)
    slot, free, n_slots, release_at = {}, [], 1, {}
    peak = 0
    for si, (_, batch) in enumerate(steps):
        for o in batch:
            if free:
                s = heapq.heappop(free)
            else:
                s = n_slots
                n_slots += 1
            slot[o] = s
            if o not in out_set:
                release_at.setdefault(max(last_use.get(o, si), si), []).append(s)
        for s in release_at.pop(si, []):
            heapq.heappush(free, s)
    assert n_slots < 65536
    # ---- constants: field constants (Montgomery, radix 2^406) then the biases (raw limbs)
    const_nodes = sorted(x for x in plan.src if g.kind[x] == CONST)
    const_index = {x: i for i, x in enumerate(const_nodes)}
    bias_es = sorted({e for e in plan.bias_e.values() if e is not None})
    for i, e in enumerate(bias_es):
        const_index[("b", e)] = len(const_nodes) + i
    n_in = n_instance_inputs
    n_set = len(g.inputs) - n_instance_inputs
    n_load_steps = sum(1 for k, _ in steps if k == K_LOAD)
    assert all(k == K_LOAD for k, _ in steps[:n_load_steps]), "the interpreter runs the LOAD steps as a prefix"
    words = [MAGIC2, lanes, n_slots, len(steps), len(const_index), n_in, n_set, len(g.outputs), n_load_steps, 1 if out_values else 0] + [0] * 6
    for x in const_nodes:
        words += limbs29(g.val[x] * R406 % P) + [0, 0]
    for e in bias_es:
        words += bias_limbs(e) + [0, 0]
    words += [slot[o] for o in g.outputs] + [0] * (-len(g.outputs) % 4)
    for kind, batch in steps:
        for li in range(lanes):
            if li >= len(batch):
                words += [0, 0, 0, kind << 30]
                continue
            o = batch[li]
            w3 = kind << 30 | 1 << 29
            if kind == K_LOAD:
                if isinstance(o, tuple) or g.kind[o] == CONST:
                    src, idx = SRC_CONST, const_index[o]
                else:
                    idx = g.a[o]
                    src, idx = (SRC_INST, idx) if idx < n_in else (SRC_SET, idx - n_in)
                assert idx < 65536
                words += [0, 0, slot[o], w3 | src << 26 | idx]
            elif kind == K_MUL:
                (a1, a2), (b1, b2) = plan.mul[o]
                s = [slot[a1], slot[a2] if a2 is not None else 0, slot[b1], slot[b2] if b2 is not None else 0]
                words += [s[0] | s[1] << 16, s[2] | s[3] << 16, slot[o], w3]
            else:
                terms = plan.lin[o]
                s = [slot[t] for _, t in terms] + [0] * (4 - len(terms))
                neg = sum(1 << i for i, (sg, _) in enumerate(terms) if sg < 0)
                e = plan.bias_e[o]
                bs = slot[("b", e)] if e is not None else 0
                words += [s[0] | s[1] << 16, s[2] | s[3] << 16, slot[o] | bs << 16, w3 | neg << 16]
    blob = struct.pack("<%dI" % len(words), *words)
    stats = {
        "lanes": lanes, "slots": n_slots, "lds_bytes": 64 * n_slots, "steps": len(steps),
        "mul_steps": sum(1 for k, _ in steps if k == K_MUL), "lin_steps": sum(1 for k, _ in steps if k == K_LIN),
        "load_steps": sum(1 for k, _ in steps if k == K_LOAD),
        "mul_ops": sum(len(b) for k, b in steps if k == K_MUL), "lin_ops": sum(len(b) for k, b in steps if k == K_LIN),
        "consts": len(const_index), "max_bound_p": plan.max_bound, "bytes": len(blob),
        "est_cycles": sum(cost[k] for k, _ in steps),
    }
    return blob, stats


def parse2(blob):
    w = struct.unpack("<%dI" % (len(blob) // 4), blob)
    assert w[0] == MAGIC2
    lanes, n_slots, n_steps, n_const, n_in, n_set, n_out = w[1:8]
    p = 16
    consts = [list(w[p + 16 * i: p + 16 * i + 14]) for i in range(n_const)]
    p += 16 * n_const
    outs = list(w[p: p + n_out])
    p += n_out + (-n_out % 4)
    desc = w[p: p + 4 * lanes * n_steps]
    assert p + 4 * lanes * n_steps == len(w)
    return dict(lanes=lanes, n_slots=n_slots, n_steps=n_steps, consts=consts, n_in=n_in, n_set=n_set, outs=outs, desc=desc)


def run_reference2(blob, inputs, settings_inputs=()):
    """Reference interpreter with the KERNEL's integer semantics at the value level: slots hold non-negative integers
    (x 2^406 Montgomery representatives, lazily reduced); a linear step is an exact integer sum (the limb-wise
    pre-conditions of the kernel - at most three subtrahends, every operand normalised, the result below 2^409 - are
    asserted), a product step is the exact Montgomery quotient (a b + m p) / 2^406.  inputs / settings_inputs: plain
    integers mod p.  Returns the outputs as plain integers mod p."""
    pr = parse2(blob)
    lanes, desc = pr["lanes"], pr["desc"]
    assert len(inputs) == pr["n_in"] and len(settings_inputs) == pr["n_set"]
    pinv = (-pow(P, -1, R406)) % R406
    slots = [None] * pr["n_slots"]
    slots[0] = 0

    def mont(a, b):
        t = a * b
        assert t < R406 * P, "product operands out of range"
        m = (t * pinv) % R406
        r = (t + m * P) >> 406
        assert r < 2 * P
        return r

    for s in range(pr["n_steps"]):
        writes, reads = [], set()
        kind0 = desc[4 * s * lanes + 3] >> 30
        for li in range(lanes):
            w0, w1, w2, w3 = desc[4 * (s * lanes + li): 4 * (s * lanes + li) + 4]
            assert w3 >> 30 == kind0, "mixed step"
            if not (w3 >> 29) & 1:
                continue
            dst = w2 & 0xFFFF
            src = [w0 & 0xFFFF, w0 >> 16, w1 & 0xFFFF, w1 >> 16]
            if kind0 == K_LOAD:
                which, idx = (w3 >> 26) & 7, w3 & 0xFFFF
                if which == SRC_CONST:
                    r = sum(x << (29 * i) for i, x in enumerate(pr["consts"][idx]))
                elif which == SRC_INST:
                    r = inputs[idx] % P * R406 % P
                else:
                    r = settings_inputs[idx] % P * R406 % P
            elif kind0 == K_MUL:
                reads.update(src)
                v = [slots[x] for x in src]
                assert all(x is not None for x in v), "read of an unwritten slot"
                r = mont(v[0] + v[1], v[2] + v[3])
            else:
                neg, bs = (w3 >> 16) & 15, w2 >> 16
                reads.update(src + [bs])
                v = [slots[x] for x in src]
                assert all(x is not None for x in v) and slots[bs] is not None, "read of an unwritten slot"
                assert bin(neg).count("1") <= 3 and (neg == 0) == (bs == 0)
                r = sum(-x if (neg >> i) & 1 else x for i, x in enumerate(v)) + slots[bs]
                assert 0 <= r < (1 << 409), "linear step out of range"
            writes.append((dst, r))
        dsts = [d for d, _ in writes]
        assert len(set(dsts)) == len(dsts), "two lanes write one slot"
        assert 0 not in dsts and not (reads & set(dsts)), "a slot is read and written in the same step"
        for d, r in writes:
            slots[d] = r
    rinv = pow(R406, -1, P)
    return [slots[o] * rinv % P for o in pr["outs"]]
