"""CPU oracle for the circuit layer: a plain-Python restatement of the reference's boolean
expression IR and S-box circuit generator.  TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h).

Follows /root/reference/base/src/boolean_tree.rs:
  Operand order/negation      :14-61, 507-517
  Not for BooleanExpr         :130-144
  mux simplification rules    :148-185
  operand ordering            :187-208
  reduce_mux                  :211-251
  stage / to_hashset          :298-332
  total order                 :519-562
and /root/reference/base/src/sbox.rs:6-61 (tables, bit planes, generate_reduced_bool_expr).

Pinned by the reference's own shape tests (boolean_tree.rs:734-780) and by exhaustive
evaluation against both S-box tables (sbox.rs:113-166 checks the same property under FHE).

Encoding: an Operand is an int (True=0, False=1, Bit_k=2+2k, NotBit_k=3+2k), so `x ^ 1` negates.
An expression is a tuple whose first field is the enum discriminant, so Python's tuple order is
exactly the reference's `Ord`:  (0, operand) | (1, l, r)=And | (2, l, r)=Or | (3, l, r)=Xor |
(4, sel, l, r)=Mux.
"""

TRUE, FALSE = 0, 1
OPERAND, AND, OR, XOR, MUX = 0, 1, 2, 3, 4


def bit(k):
    return 2 + 2 * k


def not_bit(k):
    return 3 + 2 * k


def operand(o):
    return (OPERAND, o)


def from_bool(b):
    return operand(TRUE if b else FALSE)


T = operand(TRUE)
F = operand(FALSE)


def _ordered(kind, a, b):
    return (kind, a, b) if a < b else (kind, b, a)


def neg(e):
    k = e[0]
    if k == OPERAND:
        return (OPERAND, e[1] ^ 1)
    if k == AND:
        return _ordered(OR, neg(e[1]), neg(e[2]))
    if k == OR:
        return _ordered(AND, neg(e[1]), neg(e[2]))
    if k == XOR:
        return _ordered(XOR, neg(e[1]), e[2])
    return (MUX, e[1], neg(e[2]), neg(e[3]))


def _mux_left_true(sel, right):
    if right == T:
        return T
    if right == F:
        return operand(sel)
    return _ordered(OR, operand(sel), right)


def _mux_left_false(sel, right):
    if right == T:
        return operand(sel ^ 1)
    if right == F:
        return F
    return _ordered(AND, operand(sel ^ 1), right)


def _mux_left_x(sel, left, right):
    if right == T:
        return _mux_left_true(sel ^ 1, left)
    if right == F:
        return _mux_left_false(sel ^ 1, left)
    if left == right:
        return left
    if left == neg(right):
        return _ordered(XOR, operand(sel ^ 1), left)
    return (MUX, sel, left, right)


def mux(sel, left, right):
    if left == T:
        return _mux_left_true(sel, right)
    if left == F:
        return _mux_left_false(sel, right)
    return _mux_left_x(sel, left, right)


def reduce_mux(items):
    n = len(items)
    assert n & (n - 1) == 0 if n else True, "Input was not a power of 2!"
    if n == 0:
        return T
    if n == 1:
        return items[0]
    size_log_2 = n.bit_length()
    acc = list(items)
    for k in range(size_log_2 - 1):
        acc = [mux(bit(k), acc[i], acc[i + 1]) for i in range(0, len(acc), 2)]
    assert len(acc) == 1
    return acc[0]


_stage_memo = {}


def stage(e):
    if e[0] == OPERAND:
        return 0
    s = _stage_memo.get(e)
    if s is None:
        s = max(stage(e[-2]), stage(e[-1])) + 1
        _stage_memo[e] = s
    return s


def to_set(e, out):
    if e in out:
        return
    out.add(e)
    if e[0] != OPERAND:
        to_set(e[-2], out)
        to_set(e[-1], out)


def evaluate_clear(e, x, memo=None):
    """Clear evaluation with input byte x; Bit_k is bit k of x (LSB = Bit0)."""
    k = e[0]
    if k == OPERAND:
        o = e[1]
        if o == TRUE:
            return 1
        if o == FALSE:
            return 0
        v = (x >> ((o - 2) >> 1)) & 1
        return v ^ (o & 1)
    if memo is not None and e in memo:
        return memo[e]
    if k == MUX:
        s = evaluate_clear(operand(e[1]), x)
        r = evaluate_clear(e[2], x, memo) if s else evaluate_clear(e[3], x, memo)
    else:
        a, b = evaluate_clear(e[1], x, memo), evaluate_clear(e[2], x, memo)
        r = (a & b) if k == AND else (a | b) if k == OR else (a ^ b)
    if memo is not None:
        memo[e] = r
    return r


S_BOX = bytes.fromhex(
    "637c777bf26b6fc53001672bfed7ab76ca82c97dfa5947f0add4a2af9ca472c0b7fd9326363ff7cc34a5e5f171d83115"
    "04c723c31896059a071280e2eb27b27509832c1a1b6e5aa0523bd6b329e32f8453d100ed20fcb15b6acbbe394a4c58cf"
    "d0efaafb434d338545f9027f503c9fa851a3408f929d38f5bcb6da2110fff3d2cd0c13ec5f974417c4a77e3d645d1973"
    "60814fdc222a908846eeb814de5e0bdbe0323a0a4906245cc2d3ac629195e479e7c8376d8dd54ea96c56f4ea657aae08"
    "ba78252e1ca6b4c6e8dd741f4bbd8b8a703eb5664803f60e613557b986c11d9ee1f8981169d98e949b1e87e9ce5528df"
    "8ca1890dbfe6426841992d0fb054bb16"
)
INV_S_BOX = bytes(S_BOX.index(i) for i in range(256))


def bit_x_s_box(data, position):
    # sbox.rs:45-50: the table is walked in reverse
    return [bool(v & (1 << position)) for v in reversed(data)]


def generate_reduced_bool_expr(data):
    # sbox.rs:53-61: bit planes 7..0, so root i is output bit 7-i (MSB first)
    return [reduce_mux([from_bool(b) for b in bit_x_s_box(data, pos)]) for pos in range(7, -1, -1)]


def staged(roots):
    """Unique sub-expressions of the roots, bucketed by stage and sorted in canonical order."""
    nodes = set()
    for r in roots:
        to_set(r, nodes)
    buckets = [[] for _ in range(max(stage(e) for e in nodes) + 1)]
    for e in sorted(nodes):
        buckets[stage(e)].append(e)
    return buckets


_OPN = ["True", "False"] + [f"{p}Bit{k}" for k in range(8) for p in ("", "Not")]


def to_string(e):
    """Debug form identical to Rust's derived formatting minus the Operand(..)/Box wrappers."""
    k = e[0]
    if k == OPERAND:
        return _OPN[e[1]]
    if k == MUX:
        return f"Mux({_OPN[e[1]]},{to_string(e[2])},{to_string(e[3])})"
    return f"{('And', 'Or', 'Xor')[k - 1]}({to_string(e[1])},{to_string(e[2])})"


# ---------------------------------------------------------------------------------------------
# Clear AES-128 (FIPS-197), used as the known-answer generator for the mode tests, the role the
# `aes` crate plays in /root/reference/tfhe-aes/src/main.rs:136-141,165-171.
RCON = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36]


def key_expansion_clear(key):
    # /root/reference/base/src/key_schedule.rs:183-210
    rk = [bytes(key)]
    for rnd in range(1, 11):
        prev = rk[-1]
        t = [S_BOX[prev[13]], S_BOX[prev[14]], S_BOX[prev[15]], S_BOX[prev[12]]]
        t[0] ^= RCON[rnd - 1]
        cur = bytearray(16)
        for i in range(4):
            cur[i] = prev[i] ^ t[i]
        for i in range(4, 16):
            cur[i] = prev[i] ^ cur[i - 4]
        rk.append(bytes(cur))
    return rk


def _xtime(v):
    return ((v << 1) ^ 0x1B) & 0xFF if v & 0x80 else v << 1


def _mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a = _xtime(a)
        b >>= 1
    return r


def aes_encrypt_block(rk, block):
    s = bytearray(a ^ b for a, b in zip(block, rk[0]))
    for rnd in range(1, 11):
        s = bytearray(S_BOX[v] for v in s)
        s = bytearray(s[(4 * c + r + 4 * r) % 16] for c in range(4) for r in range(4))  # ShiftRows, column-major
        if rnd < 10:
            o = bytearray(16)
            for c in range(4):
                col = s[4 * c:4 * c + 4]
                for r in range(4):
                    o[4 * c + r] = _mul(col[r], 2) ^ _mul(col[(r + 1) % 4], 3) ^ col[(r + 2) % 4] ^ col[(r + 3) % 4]
            s = o
        s = bytearray(a ^ b for a, b in zip(s, rk[rnd]))
    return bytes(s)


def aes_decrypt_block(rk, block):
    s = bytearray(a ^ b for a, b in zip(block, rk[10]))
    for rnd in range(9, -1, -1):
        s = bytearray(s[(4 * c + r - 4 * r) % 16] for c in range(4) for r in range(4))  # InvShiftRows
        s = bytearray(INV_S_BOX[v] for v in s)
        s = bytearray(a ^ b for a, b in zip(s, rk[rnd]))
        if rnd > 0:
            o = bytearray(16)
            for c in range(4):
                col = s[4 * c:4 * c + 4]
                for r in range(4):
                    o[4 * c + r] = (_mul(col[r], 14) ^ _mul(col[(r + 1) % 4], 11) ^ _mul(col[(r + 2) % 4], 13)
                                    ^ _mul(col[(r + 3) % 4], 9))
            s = o
    return bytes(s)
