"""Python model of integration/rust/base-hip: hip.rs (SlotRef, ServerKey / Engine with its recording mode, StageBatch),
primitive.rs (eval_circuit / bytes_gate / xtime_many), state.rs, key_schedule.rs, and the callers in modes (ECB, CTR):
the same control flow, function by function, emitting the tfa_gate lists the Rust code would submit.
TEST INFRASTRUCTURE: the Rust sources cannot be compiled in this image, so the tests drive this model instead -
against the C++ mirror's circuit dump, against a clear execution of the emitted gates, by stage counts, and (GPU) through
tfa_stage_submit.  tests/test_rust_source_matches_model.py compares the gate calls and index tables of both texts.

A ciphertext is ("E", SlotRef, negated) or ("T", bool).  A ServerKey owns slot allocation, levels and flushing; its
backend executes: submit(list of gate dicts), write(slots, bits), read(slots) -> bits."""
import heapq

from oracle import boolean_tree_oracle as B

AND, NAND, OR, NOR, XOR, XNOR, MUX, NOT, COPY = range(9)
AUTO_FLUSH_GATES = 2_000_000


def rebalance_xor_chains(levels, dead):         # hip.rs rebalance_xor_chains
    """Chains of xor / xnor gates whose intermediate results are unobservable (slot in `dead`: no handle left; read by exactly
    one recorded gate, itself an xor / xnor) are rebuilt, with the same number of xor gates on the same slots, as the tree of
    least depth for the levels at which their operands become available; every gate is levelled again.  Operand negations
    and xnors are gathered into one negation at the root."""
    gates = [g for lv in levels for g in lv]    # level order is a topological order
    producer = {g["out"]: i for i, g in enumerate(gates)}
    uses, user = [0] * len(gates), [-1] * len(gates)

    def operands(g):
        return [g["in0"], g["in1"]] + ([g["in2"]] if g["op"] == MUX else [])

    def is_xor(g):
        return g["op"] in (XOR, XNOR)

    for i, g in enumerate(gates):
        for s in operands(g):
            p = producer.get(s, -1)
            if p >= 0:
                uses[p] += 1
                user[p] = i
    absorbed = [is_xor(g) and g["out"] in dead and uses[i] == 1 and is_xor(gates[user[i]]) for i, g in enumerate(gates)]
    if not any(absorbed):
        return levels
    level, out = {}, []

    def put(g, lvl):
        while len(out) < lvl:
            out.append([])
        out[lvl - 1].append(g)
        level[g["out"]] = lvl

    for i, g in enumerate(gates):
        if absorbed[i]:
            continue
        if not (is_xor(g) and any(producer.get(s, -1) >= 0 and absorbed[producer[s]] for s in operands(g))):
            put(g, 1 + max(level.get(s, 0) for s in operands(g)))
            continue
        heap, spare, parity, stack = [], [], 0, [i]
        while stack:
            x = gates[stack.pop()]
            parity ^= int(x["op"] == XNOR) ^ (x["neg"] & 1) ^ ((x["neg"] >> 1) & 1)
            for s in (x["in0"], x["in1"]):
                p = producer.get(s, -1)
                if p >= 0 and absorbed[p]:
                    spare.append(s)
                    stack.append(p)
                else:
                    heap.append((level.get(s, 0), len(heap), s))
        seq = len(heap)
        heapq.heapify(heap)
        while len(heap) > 2:
            a, b = heapq.heappop(heap), heapq.heappop(heap)
            slot, lvl = spare.pop(), max(a[0], b[0]) + 1
            put(dict(op=XOR, in0=a[2], in1=b[2], in2=0, out=slot, neg=0), lvl)
            heapq.heappush(heap, (lvl, seq, slot))
            seq += 1
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        put(dict(op=XOR, in0=a[2], in1=b[2], in2=0, out=g["out"], neg=parity), max(a[0], b[0]) + 1)
    return out


def trivial(b):
    return ("T", bool(b))


def not_(ct):                                   # ServerKey::not
    return ("T", not ct[1]) if ct[0] == "T" else ("E", ct[1], not ct[2])


def _clear(op, a, b):
    return {AND: a & b, NAND: not (a & b), OR: a | b, NOR: not (a | b), XOR: a ^ b, XNOR: not (a ^ b)}[op]


class SlotRef:                                  # hip.rs SlotRef: the slot returns to `released` when the last handle goes
    __slots__ = ("index", "level", "epoch", "released")

    def __init__(self, index, level, epoch, released):
        self.index, self.level, self.epoch, self.released = index, level, epoch, released

    def __del__(self):
        self.released.append(self.index)


class ServerKey:                                # hip.rs ServerKey + Engine
    def __init__(self, backend, recording=True):
        self.backend = backend
        self.next_fresh, self.free, self.released = 0, [], []
        self.recording, self.levels, self.pending_gates, self.epoch = recording, [], 0, 0
        self.xor_rebalance = True
        self.stats = dict(gates=0, stages=0, flushes=0, max_stage_width=0)

    def alloc(self):                            # Engine::alloc
        if not self.recording:
            self.free.extend(self.released)
            del self.released[:]
        if self.free:
            return self.free.pop()
        self.next_fresh += 1
        return self.next_fresh - 1

    def _submitted(self, gates):
        self.backend.submit(gates)
        self.stats["gates"] += len(gates)
        self.stats["stages"] += 1
        self.stats["max_stage_width"] = max(self.stats["max_stage_width"], len(gates))

    def flush(self):                            # Engine::flush
        levels, self.levels = self.levels, []
        if self.xor_rebalance and self.pending_gates:
            levels = rebalance_xor_chains(levels, set(self.released))
        for lv in levels:
            if lv:
                self._submitted(lv)
        self.pending_gates = 0
        self.stats["flushes"] += 1
        self.epoch += 1
        self.free.extend(self.released)
        del self.released[:]

    def set_xor_rebalance(self, on):
        self.xor_rebalance = on

    def set_recording(self, on):
        self.flush()
        self.recording = on

    def handle(self, index, level, epoch):
        return ("E", SlotRef(index, level, epoch, self.released), False)

    def batch(self):
        return StageBatch(self)

    # ClientKey::encrypt_many / decrypt_many
    def encrypt_many(self, bits):
        slots = [self.alloc() for _ in bits]
        self.backend.write(slots, bits)
        return [self.handle(s, 0, self.epoch) for s in slots]

    def decrypt_many(self, cts):
        enc = [(i, c) for i, c in enumerate(cts) if c[0] == "E"]
        out = [c[0] == "T" and c[1] for c in cts]
        if not enc:
            return out
        self.flush()
        for (i, c), b in zip(enc, self.backend.read([c[1].index for _, c in enc])):
            out[i] = bool(b) != c[2]
        return out


class StageBatch:                               # hip.rs StageBatch
    def __init__(self, key):
        self.key, self.pending, self.gates, self.levels, self.epochs, self.held = key, [], [], [], [], []

    def _ready(self, ct):
        self.pending.append(ct)
        return len(self.pending) - 1

    def _file(self, code, ins):
        self.pending.append(None)
        ticket = len(self.pending) - 1
        now = self.key.epoch
        g = dict(op=code, in0=0, in1=0, in2=0, out=ticket, neg=0)
        level = 0
        for k, s in enumerate(ins):
            g["in%d" % k] = s[1].index
            g["neg"] |= int(s[2]) << k
            if s[1].epoch == now:
                level = max(level, s[1].level)
            self.held.append(s[1])
        self.gates.append(g)
        self.levels.append(level)
        self.epochs.append(now)
        return ticket

    def gate(self, op, a, b):
        if a[0] == "T" and b[0] == "T":
            return self._ready(trivial(_clear(op, a[1], b[1])))
        if a[0] == "T" or b[0] == "T":
            x, y = (a, b[1]) if b[0] == "T" else (b, a[1])
            f0, f1 = _clear(op, False, y), _clear(op, True, y)
            return self._ready(trivial(f0) if f0 == f1 else x if f1 else not_(x))
        return self._file(op, [a, b])

    def mux(self, c, t, e):
        if c[0] == "T":
            return self._ready(t if c[1] else e)
        if t[0] == "T":
            return self.gate(OR, c, e) if t[1] else self.gate(AND, not_(c), e)
        if e[0] == "T":
            return self.gate(OR, t, not_(c)) if e[1] else self.gate(AND, c, t)
        return self._file(MUX, [c, t, e])

    def run(self):
        key = self.key
        slot_of = {}
        if key.recording and key.pending_gates + len(self.gates) > AUTO_FLUSH_GATES:
            key.flush()
        now = key.epoch
        for k, g in enumerate(self.gates):
            ticket = g["out"]
            g["out"] = key.alloc()
            level = (1 + (self.levels[k] if self.epochs[k] == now else 0)) if key.recording else 0
            slot_of[ticket] = key.handle(g["out"], level, now)
            if key.recording:
                while len(key.levels) < level:
                    key.levels.append([])
                key.levels[level - 1].append(g)
                key.pending_gates += 1
        if not key.recording and self.gates:
            key._submitted(self.gates)
        del self.held[:]
        return [p if p is not None else slot_of[k] for k, p in enumerate(self.pending)]


def leaf(byte, op):                             # primitive.rs leaf(): byte = 8 ciphertexts, MSB first
    if op < 2:
        return trivial(op == 0)
    bit = byte[7 - (op - 2) // 2]
    return bit if op % 2 == 0 else not_(bit)


def eval_circuit(stages, roots, bytes_, sk):    # primitive.rs eval_circuit
    done = [dict() for _ in bytes_]
    for b, byte in enumerate(bytes_):
        for e in stages[0]:
            done[b][e] = leaf(byte, e[1])
    for stage in stages[1:]:
        batch, tickets = sk.batch(), []
        for b, byte in enumerate(bytes_):
            for e in stage:
                if e[0] == B.AND:
                    t = batch.gate(AND, done[b][e[1]], done[b][e[2]])
                elif e[0] == B.OR:
                    t = batch.gate(OR, done[b][e[1]], done[b][e[2]])
                elif e[0] == B.XOR:
                    t = batch.gate(XOR, done[b][e[1]], done[b][e[2]])
                else:
                    assert e[0] == B.MUX
                    t = batch.mux(leaf(byte, e[1]), done[b][e[2]], done[b][e[3]])
                tickets.append((b, e, t))
        results = batch.run()
        for b, e, t in tickets:
            done[b][e] = results[t]
    return [[done[b][r] for r in roots] for b in range(len(bytes_))]


def bytes_gate(op, pairs, sk):                  # primitive.rs bytes_gate
    batch = sk.batch()
    tickets = [[batch.gate(op, a, b) for a, b in zip(x, y)] for x, y in pairs]
    results = batch.run()
    return [[results[t] for t in ts] for ts in tickets]


def xtime_many(bytes_, sk):                     # primitive.rs xtime_many
    batch, tickets = sk.batch(), []
    for byte in bytes_:
        c = byte[0]
        shifted = list(byte[1:]) + [trivial(False)]
        tickets.append([batch.mux(c, not_(x) if 0x1B & (1 << (7 - i)) else x, x) for i, x in enumerate(shifted)])
    results = batch.run()
    return [[results[t] for t in ts] for ts in tickets]


SWAPS = [(1, 4), (2, 8), (3, 12), (6, 9), (7, 13), (11, 14)]
RCON = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36]


def transpose(v):
    v = list(v)
    for a, b in SWAPS:
        v[a], v[b] = v[b], v[a]
    return v


def byte_bits(v):
    return [bool(v & (1 << i)) for i in range(7, -1, -1)]


def bits_byte(bits):
    return sum(int(b) << (7 - i) for i, b in enumerate(bits))


def trivial_byte(v):                            # FHEByte::trivial_clear
    return [trivial(b) for b in byte_bits(v)]


def circuit(table):
    roots = B.generate_reduced_bool_expr(table)
    return B.staged(roots), roots


_CIRCUITS = {}


def s_box_circuit():
    if "s" not in _CIRCUITS:
        _CIRCUITS["s"] = circuit(B.S_BOX)
    return _CIRCUITS["s"]


def inv_s_box_circuit():
    if "i" not in _CIRCUITS:
        _CIRCUITS["i"] = circuit(B.INV_S_BOX)
    return _CIRCUITS["i"]


class State:                                    # state.rs State
    def __init__(self, data):
        self.data = data                        # 16 bytes, transposed layout

    @classmethod
    def from_u8_enc(cls, block, sk):
        return cls(transpose([sk.encrypt_many(byte_bits(v)) for v in block]))

    def decrypt_to_u8(self, sk):
        return bytes(transpose([bits_byte(sk.decrypt_many(byte)) for byte in self.data]))

    def sub_bytes(self, sk):
        c = s_box_circuit()
        self.data = eval_circuit(c[0], c[1], self.data, sk)

    def inv_sub_bytes(self, sk):
        c = inv_s_box_circuit()
        self.data = eval_circuit(c[0], c[1], self.data, sk)

    @staticmethod
    def xor_layer(lhs, rhs_index, rhs, sk):
        return bytes_gate(XOR, [(x, rhs[i]) for x, i in zip(lhs, rhs_index)], sk)

    def mix_columns(self, sk):
        d = self.data
        first = [8, 9, 10, 11, 8, 9, 10, 11, 0, 1, 2, 3, 0, 1, 2, 3]
        y = bytes_gate(XOR, [(d[i], d[4 + i]) for i in first], sk)
        second = [4, 5, 6, 7, 0, 1, 2, 3, 12, 13, 14, 15, 8, 9, 10, 11]
        y = self.xor_layer(y, second, d, sk)
        doubled = xtime_many(d, sk)
        same = list(range(16))
        y = self.xor_layer(y, same, doubled, sk)
        next_row = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3]
        self.data = self.xor_layer(y, next_row, doubled, sk)

    def inv_mix_columns(self, sk):
        d = self.data
        y = bytes_gate(XOR, [(d[i], d[(i + 4) % 16]) for i in range(16)], sk)
        t0 = bytes_gate(XOR, [(d[i], d[i + 8]) for i in range(8)], sk)
        for _ in range(2):
            t0 = xtime_many(t0, sk)
        t1 = bytes_gate(XOR, [(y[i], y[8 + i]) for i in range(4)], sk)
        y = xtime_many(y, sk)
        t0_index = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7]
        y = self.xor_layer(y, t0_index, t0, sk)
        same = list(range(16))
        cols = [i % 4 for i in range(16)]
        x = self.xor_layer(d, same, y, sk)
        x = self.xor_layer(x, cols, t1, sk)
        for _ in range(3):
            t1 = xtime_many(t1, sk)
        self.data = self.xor_layer(x, cols, t1, sk)

    def xor_key_enc(self, key, sk):             # also xor_state: bytes in the transposed layout
        same = list(range(16))
        self.data = self.xor_layer(self.data, same, key.data, sk)

    xor_state = xor_key_enc

    def shift_rows(self):
        for row in range(1, 4):
            r = self.data[4 * row:4 * row + 4]
            self.data[4 * row:4 * row + 4] = r[row:] + r[:row]

    def inv_shift_rows(self):
        for row in range(1, 4):
            r = self.data[4 * row:4 * row + 4]
            self.data[4 * row:4 * row + 4] = r[-row:] + r[:-row]


class Key:                                      # key_schedule.rs Key
    def __init__(self, data):
        self.data = data

    @classmethod
    def from_u8_enc(cls, key, sk):
        return cls(transpose([sk.encrypt_many(byte_bits(v)) for v in key]))

    def decrypt_to_u8(self, sk):
        return bytes(transpose([bits_byte(sk.decrypt_many(byte)) for byte in self.data]))

    def generate_next_key_in_place(self, rcon, sk):
        rot = [7, 11, 15, 3]
        circ = s_box_circuit()
        temp = eval_circuit(circ[0], circ[1], [self.data[i] for i in rot], sk)
        round_constant = trivial_byte(rcon)
        temp[0] = bytes_gate(XOR, [(temp[0], round_constant)], sk)[0]
        col0 = [0, 4, 8, 12]
        for i, b in zip(col0, bytes_gate(XOR, [(self.data[i], temp[i // 4]) for i in col0], sk)):
            self.data[i] = b
        for c in range(3):
            dst = [c + 1, c + 5, c + 9, c + 13]
            for i, b in zip(dst, bytes_gate(XOR, [(self.data[i], self.data[i - 1]) for i in dst], sk)):
                self.data[i] = b

    def generate_round_keys(self, sk):
        keys = [Key(list(self.data))]
        for i in range(10):
            key = Key(list(keys[i].data))
            key.generate_next_key_in_place(RCON[i], sk)
            keys.append(key)
        return keys


def ecb_encrypt(state, keys, sk):               # modes/src/ecb.rs:16-32, unchanged caller
    state.xor_key_enc(keys[0], sk)
    for i in range(1, 10):
        state.sub_bytes(sk)
        state.shift_rows()
        state.mix_columns(sk)
        state.xor_key_enc(keys[i], sk)
    state.sub_bytes(sk)
    state.shift_rows()
    state.xor_key_enc(keys[10], sk)


def ecb_decrypt(state, keys, sk):               # modes/src/ecb.rs:34-50
    state.xor_key_enc(keys[10], sk)
    for i in range(9, 0, -1):
        state.inv_shift_rows()
        state.inv_sub_bytes(sk)
        state.xor_key_enc(keys[i], sk)
        state.inv_mix_columns(sk)
    state.inv_shift_rows()
    state.inv_sub_bytes(sk)
    state.xor_key_enc(keys[0], sk)


def ctr_encrypt(counters, states, keys, sk):    # modes/src/ctr.rs:26-35: every counter through ECB, then the xor
    stream = [State(list(c.data)) for c in counters]
    for s in stream:
        ecb_encrypt(s, keys, sk)
    for s, k in zip(states, stream):
        s.xor_state(k, sk)


class ClearEngine:
    """Backend that executes tfa_gate lists on booleans with the semantics of tfa_stage_submit: every gate reads the
    slots as they were before the stage; bit k of `neg` negates operand k."""

    def __init__(self):
        self.slots, self.stages = [], []

    def _ensure(self, n):
        if n > len(self.slots):
            self.slots.extend([None] * (n - len(self.slots)))

    def write(self, slots, bits):
        self._ensure(max(slots) + 1)
        for s, b in zip(slots, bits):
            self.slots[s] = bool(b)

    def read(self, slots):
        return [self.slots[s] for s in slots]

    def submit(self, gates):
        self.stages.append([dict(g) for g in gates])
        self._ensure(max(g["out"] for g in gates) + 1)
        before = list(self.slots)
        for g in gates:
            a = before[g["in0"]] != bool(g["neg"] & 1)
            b = before[g["in1"]] != bool(g["neg"] & 2)
            assert before[g["in0"]] is not None and before[g["in1"]] is not None
            if g["op"] == MUX:
                assert before[g["in2"]] is not None
                c = before[g["in2"]] != bool(g["neg"] & 4)
                self.slots[g["out"]] = b if a else c
            else:
                self.slots[g["out"]] = _clear(g["op"], a, b)
