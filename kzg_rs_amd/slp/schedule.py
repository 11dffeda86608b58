"""Schedule a traced Fp graph (trace.Graph) into wave-wide steps, allocate LDS slots, emit the
binary program the HIP interpreter (csrc/slp.hpp) runs, and provide a reference interpreter
for that binary format (used by the tests to validate programs without a GPU).

Machine model: LANES lanes; every step is homogeneous - either a MUL step (each active lane
does one Montgomery product) or a LIN step (each active lane does one add/sub or loads one
constant / input into a slot).  Operands live in LDS slots of 48 bytes.  A value's slot is
recycled only after the step of its last use, so within a step no lane writes a slot another
lane reads: one barrier per step is enough.

Binary format (little-endian u32 words):
  [0] magic 0x31504c53 ("SLP1")   [1] lanes   [2] n_slots   [3] n_steps
  [4] n_const   [5] n_inputs   [6] n_settings_inputs   [7] n_outputs   [8..15] reserved
  const pool : n_const * 12 words   (Montgomery form, R = 2^384)
  out slots  : n_outputs words
  step kinds : n_steps words        (0 = LIN (add/sub), 1 = MUL, 2 = LOAD)
  descriptors: n_steps * lanes * 2 words:  w0 = a | b << 16  (LOAD*: w0 = index),
                                           w1 = dst | op << 16 | step_kind << 30
  ops: 0 NOP, 1 ADD, 2 SUB, 3 MUL, 4 LOADC (const pool), 5 LOADI (per-instance input),
       6 LOADS (per-settings input table)
"""
import heapq
import struct

from .trace import ADD, CONST, IN, MUL, SUB, P

MAGIC = 0x31504C53
OP_NOP, OP_ADD, OP_SUB, OP_MUL, OP_LOADC, OP_LOADI, OP_LOADS = range(7)
R384 = 1 << 384


def schedule(g, lanes=64, n_instance_inputs=None, mul_cost=12, lin_cost=1):
    """n_instance_inputs: the first that many declared inputs are per-instance (LOADI); the rest
    are per-settings (LOADS)."""
    n = len(g.kind)
    if n_instance_inputs is None:
        n_instance_inputs = len(g.inputs)
    # ---- liveness (DCE) and consumers
    live = [False] * n
    stack = list(g.outputs)
    while stack:
        x = stack.pop()
        if live[x]:
            continue
        live[x] = True
        if g.kind[x] in (MUL, ADD, SUB):
            stack.append(g.a[x])
            stack.append(g.b[x])
    users = [[] for _ in range(n)]
    for x in range(n):
        if live[x] and g.kind[x] in (MUL, ADD, SUB):
            users[g.a[x]].append(x)
            if g.b[x] != g.a[x]:
                users[g.b[x]].append(x)
    # ---- height = longest path to an output (priority)
    height = [0] * n
    for x in range(n - 1, -1, -1):
        if not live[x]:
            continue
        h = 0
        for u in users[x]:
            h = max(h, height[u])
        height[x] = h + (mul_cost if g.kind[x] == MUL else lin_cost)
    is_src = [g.kind[x] in (IN, CONST) for x in range(n)]
    # pending = number of distinct non-source operands not yet computed
    pending = [0] * n
    for x in range(n):
        if live[x] and not is_src[x]:
            ops = {g.a[x], g.b[x]}
            pending[x] = sum(1 for o in ops if not is_src[o])
    done = [False] * n
    loaded = [False] * n  # sources materialised in a slot
    ready_mul, ready_lin = [], []  # heaps of (-height, node)
    wanted_loads = []  # sources to load (heap)
    load_queued = [False] * n
    waiting_on_load = {}  # source -> [nodes]

    def operands(x):
        return {g.a[x], g.b[x]}

    def try_ready(x):
        """x has all computed operands; make sure its source operands are loaded."""
        missing = [o for o in operands(x) if is_src[o] and not loaded[o]]
        if missing:
            for o in missing:
                waiting_on_load.setdefault(o, []).append(x)
                if not load_queued[o]:
                    load_queued[o] = True
                    heapq.heappush(wanted_loads, (-height[x], o))
            return
        heapq.heappush(ready_mul if g.kind[x] == MUL else ready_lin, (-height[x], x))

    for x in range(n):
        if live[x] and not is_src[x] and pending[x] == 0:
            try_ready(x)
    # outputs that are sources themselves
    for o in g.outputs:
        if is_src[o] and not load_queued[o]:
            load_queued[o] = True
            heapq.heappush(wanted_loads, (0, o))

    steps = []  # (kind, [node...])  kind 0 lin / 1 mul; for loads the node is the source node

    def complete(nodes):
        for x in nodes:
            if is_src[x]:
                loaded[x] = True
                for w in waiting_on_load.pop(x, []):
                    if all((not is_src[o]) or loaded[o] for o in operands(w)):
                        heapq.heappush(ready_mul if g.kind[w] == MUL else ready_lin, (-height[w], w))
            else:
                done[x] = True
                for u in users[x]:
                    pending[u] -= 1
                    if pending[u] == 0:
                        try_ready(u)

    while ready_mul or ready_lin or wanted_loads:
        # all available linear work first (cheap steps)
        while ready_lin or wanted_loads:
            # loads get steps of their own (kind 2): only those steps wait on global memory, so the
            # interpreter's descriptor prefetch is never drained by an add/sub step
            while wanted_loads:
                batch = []
                while len(batch) < lanes and wanted_loads:
                    batch.append(heapq.heappop(wanted_loads)[1])
                steps.append((2, batch))
                complete(batch)
            if ready_lin:
                batch = []
                while len(batch) < lanes and ready_lin:
                    batch.append(heapq.heappop(ready_lin)[1])
                steps.append((0, batch))
                complete(batch)
        if ready_mul:
            batch = []
            while len(batch) < lanes and ready_mul:
                batch.append(heapq.heappop(ready_mul)[1])
            steps.append((1, batch))
            complete(batch)
    for o in g.outputs:
        assert done[o] or loaded[o], "output not computed"
    # ---- slot allocation
    last_use = {}
    for si, (_, batch) in enumerate(steps):
        for x in batch:
            if not is_src[x]:
                for o in operands(x):
                    last_use[o] = si
    out_set = set(g.outputs)
    slot = {}
    free = []
    n_slots = 0
    release_at = {}
    for si, (_, batch) in enumerate(steps):
        for x in batch:
            if free:
                s = heapq.heappop(free)
            else:
                s = n_slots
                n_slots += 1
            slot[x] = s
            if x not in out_set:
                lu = last_use.get(x, si)
                release_at.setdefault(lu, []).append(s) if lu > si else release_at.setdefault(si, []).append(s)
        for s in release_at.pop(si, []):
            heapq.heappush(free, s)
    assert n_slots < 65536
    # ---- emit
    const_nodes = sorted(x for x in range(n) if live[x] and g.kind[x] == CONST and load_queued[x])
    const_index = {x: i for i, x in enumerate(const_nodes)}
    n_in = n_instance_inputs
    n_set = len(g.inputs) - n_instance_inputs
    words = [MAGIC, lanes, n_slots, len(steps), len(const_nodes), n_in, n_set, len(g.outputs)] + [0] * 8
    for x in const_nodes:
        v = g.val[x] * R384 % P
        words += [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
    words += [slot[o] for o in g.outputs]
    words += [k for k, _ in steps]
    for kind, batch in steps:
        kbit = kind << 30  # step kind (2 bits) replicated into every lane's descriptor (no separate dependent load)
        for li in range(lanes):
            if li >= len(batch):
                words += [0, OP_NOP << 16 | kbit]
                continue
            x = batch[li]
            k = g.kind[x]
            if k == CONST:
                words += [const_index[x], slot[x] | OP_LOADC << 16 | kbit]
            elif k == IN:
                idx = g.a[x]
                if idx < n_in:
                    words += [idx, slot[x] | OP_LOADI << 16 | kbit]
                else:
                    words += [idx - n_in, slot[x] | OP_LOADS << 16 | kbit]
            else:
                op = {MUL: OP_MUL, ADD: OP_ADD, SUB: OP_SUB}[k]
                words += [slot[g.a[x]] | slot[g.b[x]] << 16, slot[x] | op << 16 | kbit]
    blob = struct.pack("<%dI" % len(words), *words)
    stats = {
        "lanes": lanes,
        "slots": n_slots,
        "steps": len(steps),
        "mul_steps": sum(1 for k, _ in steps if k == 1),
        "lin_steps": sum(1 for k, _ in steps if k == 0),
        "load_steps": sum(1 for k, _ in steps if k == 2),
        "mul_ops": sum(len(b) for k, b in steps if k == 1),
        "lin_ops": sum(len(b) for k, b in steps if k == 0),
        "consts": len(const_nodes),
        "bytes": len(blob),
    }
    return blob, stats


def parse(blob):
    w = struct.unpack("<%dI" % (len(blob) // 4), blob)
    assert w[0] == MAGIC
    lanes, n_slots, n_steps, n_const, n_in, n_set, n_out = w[1:8]
    p = 16
    consts = []
    for _ in range(n_const):
        v = sum(w[p + i] << (32 * i) for i in range(12))
        consts.append(v)
        p += 12
    outs = list(w[p : p + n_out])
    p += n_out
    kinds = list(w[p : p + n_steps])
    p += n_steps
    desc = w[p : p + 2 * lanes * n_steps]
    return dict(lanes=lanes, n_slots=n_slots, n_steps=n_steps, consts=consts, n_in=n_in, n_set=n_set, outs=outs,
                kinds=kinds, desc=desc)


def run_reference(blob, inputs, settings_inputs=()):
    """Reference interpreter of the binary format.  inputs / settings_inputs: plain integers mod P.
    Returns the output values (plain integers).  Slots hold plain integers here (the GPU holds
    Montgomery form; the mapping is a bijection so results agree)."""
    pr = parse(blob)
    rinv = pow(R384, -1, P)
    consts = [c * rinv % P for c in pr["consts"]]
    slots = [None] * pr["n_slots"]
    lanes, desc = pr["lanes"], pr["desc"]
    assert len(inputs) == pr["n_in"] and len(settings_inputs) == pr["n_set"]
    for s in range(pr["n_steps"]):
        writes = []
        reads = set()
        for li in range(lanes):
            w0, w1 = desc[2 * (s * lanes + li)], desc[2 * (s * lanes + li) + 1]
            op, dst = (w1 >> 16) & 0x3FFF, w1 & 0xFFFF
            assert (w1 >> 30) == pr["kinds"][s]
            if op == OP_NOP:
                continue
            if op in (OP_LOADC, OP_LOADI, OP_LOADS):
                assert pr["kinds"][s] == 2, "load outside a load step"
            if op == OP_LOADC:
                r = consts[w0]
            elif op == OP_LOADI:
                r = inputs[w0] % P
            elif op == OP_LOADS:
                r = settings_inputs[w0] % P
            else:
                a, b = slots[w0 & 0xFFFF], slots[w0 >> 16]
                reads.add(w0 & 0xFFFF)
                reads.add(w0 >> 16)
                assert a is not None and b is not None, "read of an unwritten slot"
                assert (op == OP_MUL) == (pr["kinds"][s] == 1), "op kind does not match step kind"
                assert pr["kinds"][s] != 2, "arithmetic op in a load step"
                r = (a * b if op == OP_MUL else a + b if op == OP_ADD else a - b) % P
            writes.append((dst, r))
        dsts = [d for d, _ in writes]
        assert len(set(dsts)) == len(dsts), "two lanes write one slot"
        assert not (reads & set(dsts)), "a slot is read and written in the same step"
        for d, r in writes:
            slots[d] = r
    return [slots[o] for o in pr["outs"]]
