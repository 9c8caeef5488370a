"""Circuit layer: the reference's own shape tests (/root/reference/base/src/boolean_tree.rs:734-780),
its level tests in the clear (:782-992 check the same property through FHE), the S-box circuits
(/root/reference/base/src/sbox.rs:113-166) and the gate bills, for the Python oracle and for the
C++ host mirror (which must produce the identical staged circuit)."""
import collections
import random

import pytest

from oracle import boolean_tree_oracle as B
from tfhe_aes_amd import host as H

T, F = True, False
SHAPES = [([T], "True"), ([T, T], "True"), ([T, T, T, T], "True"), ([T, F], "Bit0"), ([T, F, T, F], "Bit0"),
          ([T, F, F, F], "And(Bit0,Bit1)"), ([], "True")]


@pytest.mark.parametrize("table,expected", SHAPES)
def test_reduce_mux_shapes(table, expected):
    assert B.to_string(B.reduce_mux([B.from_bool(b) for b in table])) == expected
    assert H.reduce_mux(table) == expected


def test_reduce_mux_rejects_ragged_input():
    with pytest.raises(AssertionError):
        B.reduce_mux([B.T, B.F, B.T])
    with pytest.raises(H.HostError, match="power of 2"):
        H.reduce_mux([T, F, T])


def clear_mux_eval(table, x):
    """Fold of the truth table with selector bit k at level k (boolean_tree.rs:703-732)."""
    acc, k = list(table), 0
    while len(acc) > 1:
        s = (x >> k) & 1
        acc = [acc[i] if s else acc[i + 1] for i in range(0, len(acc), 2)]
        k += 1
    return acc[0]


@pytest.mark.parametrize("levels", [0, 1, 2, 5, 7, 8])
def test_reduce_mux_evaluates_like_the_fold(levels):
    rng = random.Random(levels)
    tables = [[rng.random() < 0.5 for _ in range(1 << levels)] for _ in range(8)]
    if levels <= 2:  # all truth tables at the small levels, as the reference does
        tables = [[bool((m >> i) & 1) for i in range(1 << levels)] for m in range(1 << (1 << levels))]
    for tab in tables:
        e = B.reduce_mux([B.from_bool(b) for b in tab])
        for x in range(256):
            assert bool(B.evaluate_clear(e, x)) == clear_mux_eval(tab, x)
        assert H.reduce_mux(tab) == B.to_string(e)


@pytest.mark.parametrize("name,table", [("sbox", B.S_BOX), ("inv_sbox", B.INV_S_BOX)], ids=["sbox", "inv_sbox"])
def test_sbox_circuits_exhaustive(name, table):
    roots = B.generate_reduced_bool_expr(table)
    for x in range(256):
        memo, v = {}, 0
        for i, r in enumerate(roots):
            v |= B.evaluate_clear(r, x, memo) << (7 - i)
        assert v == table[x]


def test_sbox_tables():
    assert B.S_BOX[0x01] == 0x7C and B.S_BOX[0x53] == 0xED and B.S_BOX[0x00] == 0x63
    assert all(B.INV_S_BOX[B.S_BOX[x]] == x for x in range(256))


EXPECTED = {  # stage histogram and gate kinds (And, Or, Xor, Mux) of the two circuits
    "sbox": ([8, 25, 133, 127, 64, 32, 16, 8], {1: 24, 2: 24, 3: 13, 4: 344}),
    "inv_sbox": ([8, 29, 133, 123, 64, 32, 16, 8], {1: 19, 2: 26, 3: 12, 4: 348}),
}


@pytest.mark.parametrize("name,table", [("sbox", B.S_BOX), ("inv_sbox", B.INV_S_BOX)], ids=["sbox", "inv_sbox"])
def test_gate_bill_and_host_mirror_identity(name, table):
    roots = B.generate_reduced_bool_expr(table)
    stages = B.staged(roots)
    hist, kinds = EXPECTED[name]
    assert [len(s) for s in stages] == hist
    got = collections.Counter(e[0] for s in stages[1:] for e in s)
    assert dict(got) == kinds
    pbs = sum(2 if e[0] == B.MUX else 1 for s in stages[1:] for e in s)
    assert pbs == (749 if name == "sbox" else 753) and sum(hist[1:]) == 405
    assert all(B.stage(r) == 7 and r[0] == B.MUX for r in roots)
    leaves = {e[1] for e in stages[0]}
    assert leaves == {B.bit(k) for k in range(4)} | {B.not_bit(k) for k in range(4)}
    # mux selectors are never negated
    assert all(e[1] % 2 == 0 for s in stages for e in s if e[0] == B.MUX)
    expected_dump = "".join(f"{i}\t{B.to_string(e)}\n" for i, s in enumerate(stages) for e in s)
    expected_dump += "".join(f"root\t{B.to_string(r)}\n" for r in roots)
    assert H.sbox_circuit_dump(name == "inv_sbox") == expected_dump


def test_first_stage_nodes_of_sbox():
    st1 = [B.to_string(e) for e in B.staged(B.generate_reduced_bool_expr(B.S_BOX))[1]]
    assert st1[:5] == ["And(Bit0,Bit1)", "And(Bit0,NotBit1)", "And(Bit0,NotBit2)", "And(NotBit0,Bit1)", "And(NotBit0,NotBit1)"]
    assert st1[-5:] == ["Mux(Bit2,Bit0,Bit1)", "Mux(Bit2,Bit0,NotBit1)", "Mux(Bit2,NotBit0,Bit1)", "Mux(Bit2,NotBit0,NotBit1)", "Mux(Bit2,Bit1,Bit0)"]


def test_key_expansion_clear_matches_oracle():
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    assert H.key_expansion_clear(key) == B.key_expansion_clear(key)
