"""Python model of integration/rust/base-hip (hip.rs StageBatch, primitive.rs eval_circuit / bytes_gate / xtime_many,
state.rs): the same control flow, function by function, emitting the tfa_gate lists the Rust code would submit.
TEST INFRASTRUCTURE: the Rust sources cannot be compiled in this image, so the tests drive this model instead -
against the C++ mirror's circuit dump, against a clear execution of the emitted gates, and (GPU) through
tfa_stage_submit.  Keep it in step with the Rust files when either changes.

A ciphertext is ("E", slot, negated) or ("T", bool); an engine object receives submit(list of gate dicts)."""
from oracle import boolean_tree_oracle as B

AND, NAND, OR, NOR, XOR, XNOR, MUX, NOT, COPY = range(9)


def trivial(b):
    return ("T", bool(b))


def not_(ct):                                   # ServerKey::not
    return ("T", not ct[1]) if ct[0] == "T" else ("E", ct[1], not ct[2])


def _clear(op, a, b):
    return {AND: a & b, NAND: not (a & b), OR: a | b, NOR: not (a | b), XOR: a ^ b, XNOR: not (a ^ b)}[op]


class StageBatch:                               # hip.rs StageBatch
    def __init__(self, engine):
        self.engine, self.pending, self.gates = engine, [], []

    def _ready(self, ct):
        self.pending.append(ct)
        return len(self.pending) - 1

    def gate(self, op, a, b):
        if a[0] == "T" and b[0] == "T":
            return self._ready(trivial(_clear(op, a[1], b[1])))
        if a[0] == "T" or b[0] == "T":
            x, y = (a, b[1]) if b[0] == "T" else (b, a[1])
            f0, f1 = _clear(op, False, y), _clear(op, True, y)
            return self._ready(trivial(f0) if f0 == f1 else x if f1 else not_(x))
        self.pending.append(None)
        t = len(self.pending) - 1
        self.gates.append(dict(op=op, in0=a[1], in1=b[1], in2=0, out=t, neg=int(a[2]) | int(b[2]) << 1))
        return t

    def mux(self, c, t, e):
        if c[0] == "T":
            return self._ready(t if c[1] else e)
        if t[0] == "T":
            return self.gate(OR, c, e) if t[1] else self.gate(AND, not_(c), e)
        if e[0] == "T":
            return self.gate(OR, t, not_(c)) if e[1] else self.gate(AND, c, t)
        self.pending.append(None)
        k = len(self.pending) - 1
        self.gates.append(dict(op=MUX, in0=c[1], in1=t[1], in2=e[1], out=k, neg=int(c[2]) | int(t[2]) << 1 | int(e[2]) << 2))
        return k

    def run(self):
        first = self.engine.alloc(len(self.gates))
        slot_of = {}
        for k, g in enumerate(self.gates):
            slot_of[g["out"]] = first + k
            g["out"] = first + k
        if self.gates:
            self.engine.submit(self.gates)
        return [p if p is not None else ("E", slot_of[k], False) for k, p in enumerate(self.pending)]


def leaf(byte, op):                             # primitive.rs leaf(): byte = 8 ciphertexts, MSB first
    if op < 2:
        return trivial(op == 0)
    bit = byte[7 - (op - 2) // 2]
    return bit if op % 2 == 0 else not_(bit)


def eval_circuit(stages, roots, bytes_, engine):  # primitive.rs eval_circuit
    done = [dict() for _ in bytes_]
    for b, byte in enumerate(bytes_):
        for e in stages[0]:
            done[b][e] = leaf(byte, e[1])
    for stage in stages[1:]:
        batch, tickets = StageBatch(engine), []
        for b, byte in enumerate(bytes_):
            for e in stage:
                if e[0] == B.MUX:
                    t = batch.mux(leaf(byte, e[1]), done[b][e[2]], done[b][e[3]])
                else:
                    t = batch.gate({B.AND: AND, B.OR: OR, B.XOR: XOR}[e[0]], done[b][e[1]], done[b][e[2]])
                tickets.append((b, e, t))
        results = batch.run()
        for b, e, t in tickets:
            done[b][e] = results[t]
    return [[done[b][r] for r in roots] for b in range(len(bytes_))]


def bytes_gate(op, pairs, engine):              # primitive.rs bytes_gate
    batch = StageBatch(engine)
    tickets = [[batch.gate(op, a, b) for a, b in zip(x, y)] for x, y in pairs]
    results = batch.run()
    return [[results[t] for t in ts] for ts in tickets]


def xtime_many(bytes_, engine):                 # primitive.rs xtime_many
    batch, tickets = StageBatch(engine), []
    for byte in bytes_:
        c = byte[0]
        shifted = list(byte[1:]) + [trivial(False)]
        tickets.append([batch.mux(c, not_(x) if 0x1B & (1 << (7 - i)) else x, x) for i, x in enumerate(shifted)])
    results = batch.run()
    return [[results[t] for t in ts] for ts in tickets]


SWAPS = [(1, 4), (2, 8), (3, 12), (6, 9), (7, 13), (11, 14)]


def transpose(v):
    v = list(v)
    for a, b in SWAPS:
        v[a], v[b] = v[b], v[a]
    return v


class State:                                    # state.rs State
    def __init__(self, data):
        self.data = data                        # 16 bytes, transposed layout

    def sub_bytes(self, circuit, engine):
        self.data = eval_circuit(circuit[0], circuit[1], self.data, engine)

    @staticmethod
    def _xor_layer(lhs, rhs_index, rhs, engine):
        return bytes_gate(XOR, [(x, rhs[i]) for x, i in zip(lhs, rhs_index)], engine)

    def mix_columns(self, engine):
        d = self.data
        first = [8, 9, 10, 11, 8, 9, 10, 11, 0, 1, 2, 3, 0, 1, 2, 3]
        y = bytes_gate(XOR, [(d[i], d[4 + i]) for i in first], engine)
        y = self._xor_layer(y, [4, 5, 6, 7, 0, 1, 2, 3, 12, 13, 14, 15, 8, 9, 10, 11], d, engine)
        doubled = xtime_many(d, engine)
        y = self._xor_layer(y, list(range(16)), doubled, engine)
        self.data = self._xor_layer(y, [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3], doubled, engine)

    def inv_mix_columns(self, engine):
        d = self.data
        y = bytes_gate(XOR, [(d[i], d[(i + 4) % 16]) for i in range(16)], engine)
        t0 = bytes_gate(XOR, [(d[i], d[i + 8]) for i in range(8)], engine)
        for _ in range(2):
            t0 = xtime_many(t0, engine)
        t1 = bytes_gate(XOR, [(y[i], y[8 + i]) for i in range(4)], engine)
        y = xtime_many(y, engine)
        y = self._xor_layer(y, [0, 1, 2, 3, 4, 5, 6, 7] * 2, t0, engine)
        x = self._xor_layer(d, list(range(16)), y, engine)
        x = self._xor_layer(x, [i % 4 for i in range(16)], t1, engine)
        for _ in range(3):
            t1 = xtime_many(t1, engine)
        self.data = self._xor_layer(x, [i % 4 for i in range(16)], t1, engine)

    def xor_key(self, key_bytes, engine):       # xor_key_enc / xor_state (bytes in the transposed layout)
        self.data = self._xor_layer(self.data, list(range(16)), key_bytes, engine)

    def shift_rows(self):
        for row in range(1, 4):
            r = self.data[4 * row:4 * row + 4]
            self.data[4 * row:4 * row + 4] = r[row:] + r[:row]


class ClearEngine:
    """Executes tfa_gate lists on booleans with the semantics of tfa_stage_submit: every gate reads the slots as they
    were before the stage; bit k of `neg` negates operand k."""

    def __init__(self):
        self.slots, self.stages = [], []

    def alloc(self, n):
        first = len(self.slots)
        self.slots.extend([None] * n)
        return first

    def put(self, bits):
        first = self.alloc(len(bits))
        self.slots[first:] = [bool(b) for b in bits]
        return [("E", first + k, False) for k in range(len(bits))]

    def get(self, ct):
        return ct[1] if ct[0] == "T" else self.slots[ct[1]] != ct[2]

    def submit(self, gates):
        self.stages.append([dict(g) for g in gates])
        before = list(self.slots)
        for g in gates:
            a = before[g["in0"]] != bool(g["neg"] & 1)
            b = before[g["in1"]] != bool(g["neg"] & 2)
            if g["op"] == MUX:
                c = before[g["in2"]] != bool(g["neg"] & 4)
                self.slots[g["out"]] = b if a else c
            else:
                self.slots[g["out"]] = _clear(g["op"], a, b)
            assert a is not None and b is not None


def byte_bits(v):
    return [bool(v & (1 << i)) for i in range(7, -1, -1)]


def bits_byte(bits):
    return sum(int(b) << (7 - i) for i, b in enumerate(bits))


def circuit(table):
    roots = B.generate_reduced_bool_expr(table)
    return B.staged(roots), roots
