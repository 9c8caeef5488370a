"""The gate lists the Rust binding (integration/rust/base-hip) submits, checked through its Python model
(tests/rust_binding_model.py): which gates per stage (against the C++ mirror's circuit dump and the reference's
bills, /root/reference/base/src/primitive.rs:250-265, state.rs:52-118), what they compute (clear execution of the
tfa_gate semantics against the S-box tables and FIPS-197), and on the GPU through tfa_stage_submit."""
import collections

import numpy as np
import pytest

from oracle import boolean_tree_oracle as B
from tests import rust_binding_model as M
from tfhe_aes_amd import host as H


@pytest.mark.parametrize("name,table", [("sbox", B.S_BOX), ("inv_sbox", B.INV_S_BOX)], ids=["sbox", "inv_sbox"])
def test_sub_byte_gate_lists_match_the_host_mirror(name, table):
    circ = M.circuit(table)
    eng = M.ClearEngine()
    out = M.eval_circuit(circ[0], circ[1], [eng.put(M.byte_bits(0x01))], eng)[0]
    assert M.bits_byte([eng.get(c) for c in out]) == table[0x01]          # 0x01 -> 0x7c (primitive.rs:438-452)
    assert len(eng.stages) == 7
    # stage s of the model = the expressions of height s of the C++ mirror's circuit, in its order
    dump = [ln.split("\t") for ln in H.sbox_circuit_dump(name == "inv_sbox").splitlines()]
    for s, gates in enumerate(eng.stages, start=1):
        want = [e for st, e in dump if st == str(s)]
        assert len(gates) == len(want)
        kinds = ["Mux" if g["op"] == M.MUX else {M.AND: "And", M.OR: "Or", M.XOR: "Xor"}[g["op"]] for g in gates]
        assert kinds == [e.split("(")[0] for e in want]
    assert sum(len(g) for g in eng.stages) == 405
    assert sum(2 if g["op"] == M.MUX else 1 for st in eng.stages for g in st) == (749 if name == "sbox" else 753)
    # no gate reads a slot written in its own stage, no slot is written twice
    for st in eng.stages:
        outs = [g["out"] for g in st]
        assert len(set(outs)) == len(outs)
        assert not set(outs) & {g[k] for g in st for k in ("in0", "in1", "in2")}


@pytest.mark.parametrize("table", [B.S_BOX, B.INV_S_BOX], ids=["sbox", "inv_sbox"])
def test_sub_byte_gate_lists_compute_the_table(table):
    circ = M.circuit(table)
    for x in range(256):
        eng = M.ClearEngine()
        out = M.eval_circuit(circ[0], circ[1], [eng.put(M.byte_bits(x))], eng)[0]
        assert M.bits_byte([eng.get(c) for c in out]) == table[x]


def test_state_round_gate_bill_and_values():
    """One full round on a state (FIPS-197 App. B round 1): 7 S-box submits for 16 bytes, MixColumns = 512 xors + 112
    muxes in 5 stages, AddRoundKey 128 xors (state.rs:52-118, 217-224)."""
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    rk = B.key_expansion_clear(key)
    block = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
    eng = M.ClearEngine()
    st = M.State([eng.put(M.byte_bits(v)) for v in M.transpose(bytes(a ^ b for a, b in zip(block, rk[0])))])
    circ = M.circuit(B.S_BOX)
    st.sub_bytes(circ, eng)
    assert [len(s) for s in eng.stages] == [400, 2128, 2032, 1024, 512, 256, 128]
    st.shift_rows()
    n0 = len(eng.stages)
    st.mix_columns(eng)
    mc = eng.stages[n0:]
    assert len(mc) == 5
    ops = collections.Counter(g["op"] for s in mc for g in s)
    assert ops == {M.XOR: 512, M.MUX: 112}
    key_bytes = [eng.put(M.byte_bits(v)) for v in M.transpose(rk[1])]
    st.xor_key(key_bytes, eng)
    assert len(eng.stages[-1]) == 128
    got = bytes(M.transpose([M.bits_byte([eng.get(c) for c in byte]) for byte in st.data]))
    assert got.hex() == "a49c7ff2689f352b6b5bea43026a5049"       # FIPS-197 App. B, start of round 2
    # and back: the inverse layers
    st.xor_key(key_bytes, eng)
    n0 = len(eng.stages)
    st.inv_mix_columns(eng)
    ops = collections.Counter(g["op"] for s in eng.stages[n0:] for g in s)
    assert ops[M.MUX] == 7 * (8 + 8 + 16 + 4 * 3)
    got = bytes(M.transpose([M.bits_byte([eng.get(c) for c in byte]) for byte in st.data]))
    assert got.hex() == "d4bf5d30e0b452aeb84111f11e2798e5"       # after ShiftRows of round 1


def test_trivial_operands_never_reach_the_engine():
    eng = M.ClearEngine()
    x = eng.put([True])[0]
    batch = M.StageBatch(eng)
    t = [batch.gate(M.XOR, x, M.trivial(True)), batch.gate(M.AND, x, M.trivial(False)), batch.gate(M.OR, M.trivial(False), x),
         batch.mux(M.trivial(True), x, M.trivial(False)), batch.mux(x, M.trivial(True), M.trivial(False)), batch.gate(M.NAND, x, M.trivial(True))]
    res = batch.run()
    assert eng.stages == []
    assert [eng.get(res[k]) for k in t] == [False, False, True, True, True, False]


@pytest.mark.gpu
def test_sub_byte_gate_lists_on_the_gpu():
    """SubBytes(0x01) = 0x7c and InvSubBytes(0x7c) = 0x01 with the model's gate lists submitted through the C ABI."""
    from tfhe_aes_amd import engine as E

    p = E.default_params()
    lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 11)
    ctx = E.Context(p, 0)
    ctx.keys_upload(bsk, ksk)
    ctx.pool_reserve(4096)

    class GpuEngine:
        def __init__(self):
            self.next = 0

        def alloc(self, n):
            first = self.next
            self.next += n
            return first

        def put(self, bits):
            first = self.alloc(len(bits))
            ctx.pool_write(np.arange(first, first + len(bits)), E.client_encrypt(p, lwe_sk, 99 + first, np.array(bits, np.uint8)))
            return [("E", first + k, False) for k in range(len(bits))]

        def submit(self, gates):
            g = E.make_gates([x["op"] for x in gates], [x["in0"] for x in gates], [x["in1"] for x in gates], [x["in2"] for x in gates],
                             [x["out"] for x in gates], [x["neg"] for x in gates])
            ctx.stage_submit(g)

        def get_byte(self, byte):
            raw = E.client_decrypt(p, lwe_sk, ctx.pool_read(np.array([c[1] for c in byte])))
            return M.bits_byte([bool(b) != c[2] for b, c in zip(raw, byte)])

    eng = GpuEngine()
    fwd, inv = M.circuit(B.S_BOX), M.circuit(B.INV_S_BOX)
    out = M.eval_circuit(fwd[0], fwd[1], [eng.put(M.byte_bits(0x01))], eng)[0]
    assert eng.get_byte(out) == 0x7C
    back = M.eval_circuit(inv[0], inv[1], [out], eng)[0]
    assert eng.get_byte(back) == 0x01
