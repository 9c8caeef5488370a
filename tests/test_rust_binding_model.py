"""The gate lists the Rust binding (integration/rust/base-hip) submits, checked through its Python model
(tests/rust_binding_model.py): which gates per stage (against the C++ mirror's circuit dump and the reference's
bills, /root/reference/base/src/primitive.rs:250-265, state.rs:52-118, key_schedule.rs:90-131), what they compute (clear
execution of the tfa_gate semantics against the S-box tables and FIPS-197), how a recording key shares stages between
blocks (/root/reference/modes/src/ctr.rs:26-35), how slots are recycled, and on the GPU through tfa_stage_submit."""
import collections

import numpy as np
import pytest

from oracle import boolean_tree_oracle as B
from tests import rust_binding_model as M
from tfhe_aes_amd import host as H

FIPS_KEY = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
FIPS_PT = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
FIPS_CT = "3925841d02dc09fbdc118597196a0b32"


def eager():
    eng = M.ClearEngine()
    return eng, M.ServerKey(eng, recording=False)


def byte_of(sk, byte):
    return M.bits_byte(sk.decrypt_many(byte))


@pytest.mark.parametrize("name,table", [("sbox", B.S_BOX), ("inv_sbox", B.INV_S_BOX)], ids=["sbox", "inv_sbox"])
def test_sub_byte_gate_lists_match_the_host_mirror(name, table):
    circ = M.circuit(table)
    eng, sk = eager()
    out = M.eval_circuit(circ[0], circ[1], [sk.encrypt_many(M.byte_bits(0x01))], sk)[0]
    assert byte_of(sk, out) == table[0x01]          # 0x01 -> 0x7c (primitive.rs:438-452)
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
        eng, sk = eager()
        out = M.eval_circuit(circ[0], circ[1], [sk.encrypt_many(M.byte_bits(x))], sk)[0]
        assert byte_of(sk, out) == table[x]


def test_state_round_gate_bill_and_values():
    """One full round on a state (FIPS-197 App. B round 1), eager key: 7 S-box submits for 16 bytes, MixColumns = 512
    xors + 112 muxes in 5 stages, AddRoundKey 128 xors (state.rs:52-118, 217-224)."""
    rk = B.key_expansion_clear(FIPS_KEY)
    eng, sk = eager()
    st = M.State.from_u8_enc(bytes(a ^ b for a, b in zip(FIPS_PT, rk[0])), sk)
    st.sub_bytes(sk)
    assert [len(s) for s in eng.stages] == [400, 2128, 2032, 1024, 512, 256, 128]
    st.shift_rows()
    n0 = len(eng.stages)
    st.mix_columns(sk)
    mc = eng.stages[n0:]
    assert len(mc) == 5
    ops = collections.Counter(g["op"] for s in mc for g in s)
    assert ops == {M.XOR: 512, M.MUX: 112}
    key1 = M.Key.from_u8_enc(rk[1], sk)
    st.xor_key_enc(key1, sk)
    assert len(eng.stages[-1]) == 128
    assert st.decrypt_to_u8(sk).hex() == "a49c7ff2689f352b6b5bea43026a5049"       # FIPS-197 App. B, start of round 2
    # and back: the inverse layers
    st.xor_key_enc(key1, sk)
    n0 = len(eng.stages)
    st.inv_mix_columns(sk)
    ops = collections.Counter(g["op"] for s in eng.stages[n0:] for g in s)
    assert ops[M.MUX] == 7 * (8 + 8 + 16 + 4 * 3)
    assert st.decrypt_to_u8(sk).hex() == "d4bf5d30e0b452aeb84111f11e2798e5"       # after ShiftRows of round 1


def test_trivial_operands_never_reach_the_engine():
    eng, sk = eager()
    x = sk.encrypt_many([True])[0]
    batch = sk.batch()
    t = [batch.gate(M.XOR, x, M.trivial(True)), batch.gate(M.AND, x, M.trivial(False)), batch.gate(M.OR, M.trivial(False), x),
         batch.mux(M.trivial(True), x, M.trivial(False)), batch.mux(x, M.trivial(True), M.trivial(False)), batch.gate(M.NAND, x, M.trivial(True))]
    res = batch.run()
    assert eng.stages == []
    assert sk.decrypt_many([res[k] for k in t]) == [False, False, True, True, True, False]


def test_recording_key_levels_one_block_like_the_host_mirror():
    """A recording key files every gate at 1 + its deepest operand's level: one ECB block (ecb.rs:16-32, unchanged caller)
    takes the 117 stages and 71 824 gates of the C++ mirror's lazy submitter, against 126 submits on an eager key, and
    decrypts to the FIPS-197 Appendix B ciphertext; the inverse cipher (ecb.rs:34-50) takes 144 stages and returns the block.
    With the re-association of unobservable xor chains (the default, as in the C++ mirror) the same gates take 99 / 135."""
    for rebalance, enc_stages, dec_stages in ((False, 117, 144), (True, 99, 135)):
        eng = M.ClearEngine()
        sk = M.ServerKey(eng)
        sk.set_xor_rebalance(rebalance)
        keys = [M.Key.from_u8_enc(k, sk) for k in B.key_expansion_clear(FIPS_KEY)]
        st = M.State.from_u8_enc(FIPS_PT, sk)
        M.ecb_encrypt(st, keys, sk)
        assert eng.stages == []                                   # nothing runs before the first decrypt
        assert st.decrypt_to_u8(sk).hex() == FIPS_CT
        assert (sk.stats["stages"], sk.stats["gates"]) == (enc_stages, 71824)
        assert sum(2 if g["op"] == M.MUX else 1 for s in eng.stages for g in s) == 127872
        if rebalance:   # MixColumns + AddRoundKey: three levels (496, 240, 128 bootstraps) instead of five (352, 128, 128, 128, 128)
            assert [sum(2 if g["op"] == M.MUX else 1 for g in s) for s in eng.stages[:11]] == [128, 480, 3680, 3984, 2048, 1024, 512, 256, 496, 240, 128]
        M.ecb_decrypt(st, keys, sk)
        assert st.decrypt_to_u8(sk) == FIPS_PT
        assert sk.stats["stages"] - enc_stages == dec_stages
        assert sk.stats["gates"] - 71824 == 75604
    eng2, eager_key = eager()
    keys2 = [M.Key.from_u8_enc(k, eager_key) for k in B.key_expansion_clear(FIPS_KEY)]
    st2 = M.State.from_u8_enc(FIPS_PT, eager_key)
    M.ecb_encrypt(st2, keys2, eager_key)
    assert len(eng2.stages) == 126 and st2.decrypt_to_u8(eager_key).hex() == FIPS_CT


def test_xor_chain_rebalance_keeps_every_observable_value():
    """rebalance_xor_chains on random circuits (boolean backend): xor / xnor chains with negated operands, intermediates
    that are dropped, kept or shared by two consumers, ands and muxes in between.  Every kept handle reads the clear
    evaluation, the gates are as many as recorded, every stage's gates read only slots written by earlier stages, and the
    stages are fewer than without the re-association."""
    import random
    counts = {}
    for rebalance in (False, True):
        rnd = random.Random(4711)
        stages = gates = 0
        for trial in range(40):
            eng = M.ClearEngine()
            sk = M.ServerKey(eng)
            sk.set_xor_rebalance(rebalance)
            vals = [rnd.random() < 0.5 for _ in range(12)]
            live = list(zip(sk.encrypt_many(vals), vals))
            kept = []
            for step in range(150):
                pick = lambda: live[rnd.randrange(max(0, len(live) - 6), len(live))] if rnd.randrange(3) else live[rnd.randrange(len(live))]
                a, b = pick(), pick()
                if rnd.randrange(4) == 0:
                    a = (M.not_(a[0]), not a[1])
                kind = rnd.randrange(10)
                z = sk.batch()
                if kind < 6:
                    op = M.XOR if rnd.randrange(4) else M.XNOR
                    t, want = z.gate(op, a[0], b[0]), (a[1] != b[1]) == (op == M.XOR)
                elif kind < 8:
                    t, want = z.gate(M.AND, a[0], b[0]), a[1] and b[1]
                else:
                    c = pick()
                    t, want = z.mux(c[0], a[0], b[0]), (a[1] if c[1] else b[1])
                r = (z.run()[t], want)
                del z
                if rnd.randrange(5) == 0:
                    kept.append(r)
                live.append(r)
                if len(live) > 14 and rnd.randrange(2):
                    del live[rnd.randrange(12, len(live) - 1)]
            kept.extend(live[12:])
            del live, a, b, r
            got = sk.decrypt_many([ct for ct, _ in kept])
            assert got == [want for _, want in kept]
            stages += sk.stats["stages"]
            gates += sk.stats["gates"]
        counts[rebalance] = (stages, gates)
    assert counts[True][1] == counts[False][1]
    assert counts[True][0] < counts[False][0]


def test_two_ctr_blocks_share_stages():
    """modes/src/ctr.rs:26-35 unchanged on a recording key: the blocks' gates meet in the same stages, 118 for two blocks
    (117 of ECB + the final xor), not 2 x 118; twice the gates of one block; SP 800-38A F.5.1 values.  With the xor chains
    re-associated (default) the final xor joins the last AddRoundKey's tree: 99 stages."""
    key = FIPS_KEY
    ctr0 = int("f0f1f2f3f4f5f6f7f8f9fafbfcfdfeff", 16)
    pts = [bytes.fromhex("6bc1bee22e409f96e93d7e117393172a"), bytes.fromhex("ae2d8a571e03ac9c9eb76fac45af8e51")]
    want = ["874d6191b620e3261bef6864990db6ce", "9806f66b7970fdff8617187bb9fffdff"]
    for rebalance, n_stages in ((False, 118), (True, 99)):
        eng = M.ClearEngine()
        sk = M.ServerKey(eng)
        sk.set_xor_rebalance(rebalance)
        keys = [M.Key.from_u8_enc(k, sk) for k in B.key_expansion_clear(key)]
        counters = [M.State.from_u8_enc((ctr0 + b).to_bytes(16, "big"), sk) for b in range(2)]
        states = [M.State.from_u8_enc(p, sk) for p in pts]
        M.ctr_encrypt(counters, states, keys, sk)
        del counters
        assert [s.decrypt_to_u8(sk).hex() for s in states] == want
        assert sk.stats["stages"] == n_stages == len(eng.stages)
        assert sk.stats["gates"] == 2 * (71824 + 128)
        assert sk.stats["max_stage_width"] == 2 * 2128


def test_online_key_expansion_takes_110_stages():
    """key_schedule.rs:162-171 on a recording key: 10 rounds x (7 S-box stages + 4 xor layers) = 110 stages, 17 480 gates,
    31 240 bootstraps, the four S-boxes of a round in the same seven stages; all eleven round keys as in FIPS-197 A.1."""
    eng = M.ClearEngine()
    sk = M.ServerKey(eng)
    k0 = M.Key.from_u8_enc(FIPS_KEY, sk)
    rks = k0.generate_round_keys(sk)
    assert [k.decrypt_to_u8(sk) for k in rks] == B.key_expansion_clear(FIPS_KEY)
    assert rks[10].decrypt_to_u8(sk).hex() == "d014f9a8c9ee2589e13f0cc8b6630ca6"
    assert (sk.stats["stages"], sk.stats["gates"]) == (110, 17480)
    assert sum(2 if g["op"] == M.MUX else 1 for s in eng.stages for g in s) == 31240
    assert [len(s) for s in eng.stages[:11]] == [100, 532, 508, 256, 128, 64, 32, 32, 32, 32, 32]
    # eager key: the same gates, one submit per StageBatch::run
    eng2, ek = eager()
    M.Key.from_u8_enc(FIPS_KEY, ek).generate_round_keys(ek)
    assert len(eng2.stages) == 110 and sum(len(s) for s in eng2.stages) == 17480


def test_slots_are_recycled_after_a_flush():
    """A slot returns to the free list when its last handle is dropped and is handed out again after the next flush: three
    blocks one after the other need no more pool than one (the bump allocator this replaces grew by 72 k slots a block)."""
    eng = M.ClearEngine()
    sk = M.ServerKey(eng)
    keys = [M.Key.from_u8_enc(k, sk) for k in B.key_expansion_clear(FIPS_KEY)]
    high = []
    for _ in range(3):
        st = M.State.from_u8_enc(FIPS_PT, sk)
        M.ecb_encrypt(st, keys, sk)
        assert st.decrypt_to_u8(sk).hex() == FIPS_CT
        del st
        high.append(sk.next_fresh)
    assert high[0] < 11 * 128 + 128 + 71824 + 1 and high[1] == high[2] <= high[0] + 128
    # a released slot is not reused while a recorded gate may still read it
    x, y = sk.encrypt_many([True, False])
    z = sk.batch()
    t = z.gate(M.XOR, x, y)
    out = z.run()[t]
    ix = x[1].index
    del x, z
    w = sk.encrypt_many([True])[0]
    assert w[1].index != ix                                   # the xor is recorded, not run: slot ix is still its operand
    assert sk.decrypt_many([out]) == [True]
    v = sk.encrypt_many([False])[0]
    assert v[1].index == ix                                   # after the flush it is free again
    # an eager key reuses at once: the stream is ordered
    eng2, ek = eager()
    a, b = ek.encrypt_many([True, True])
    r = ek.batch()
    t = r.gate(M.AND, a, b)
    res = r.run()[t]
    ia = a[1].index
    del a, r
    c = ek.encrypt_many([False])[0]
    assert c[1].index == ia and ek.decrypt_many([res]) == [True]


class GpuBackend:
    """Backend of the model on the real engine: gate lists go through tfa_stage_submit, bits through the client routines."""

    def __init__(self, slots=1 << 17, seed=11):
        from tfhe_aes_amd import engine as E
        self.E, self.p = E, E.default_params()
        self.lwe_sk, _, bsk, ksk = E.client_keygen(self.p, seed)
        self.ctx = E.Context(self.p, 0)
        self.ctx.keys_upload(bsk, ksk)
        self.ctx.pool_reserve(slots)
        self.writes, self.stages = 0, 0

    def write(self, slots, bits):
        self.writes += 1
        self.ctx.pool_write(np.array(slots), self.E.client_encrypt(self.p, self.lwe_sk, 99 + self.writes, np.array(bits, np.uint8)))

    def read(self, slots):
        return [bool(b) for b in self.E.client_decrypt(self.p, self.lwe_sk, self.ctx.pool_read(np.array(slots)))]

    def submit(self, gates):
        self.stages += 1
        self.ctx.stage_submit(self.E.make_gates([x["op"] for x in gates], [x["in0"] for x in gates], [x["in1"] for x in gates],
                                                [x["in2"] for x in gates], [x["out"] for x in gates], [x["neg"] for x in gates]))


@pytest.mark.gpu
def test_sub_byte_gate_lists_on_the_gpu():
    """SubBytes(0x01) = 0x7c and InvSubBytes(0x7c) = 0x01 with the model's gate lists submitted through the C ABI."""
    sk = M.ServerKey(GpuBackend(4096), recording=False)
    fwd, inv = M.circuit(B.S_BOX), M.circuit(B.INV_S_BOX)
    out = M.eval_circuit(fwd[0], fwd[1], [sk.encrypt_many(M.byte_bits(0x01))], sk)[0]
    assert byte_of(sk, out) == 0x7C
    back = M.eval_circuit(inv[0], inv[1], [out], sk)[0]
    assert byte_of(sk, back) == 0x01


@pytest.mark.gpu
def test_key_schedule_gate_lists_on_the_gpu():
    """The model's key-schedule gate lists (key_schedule.rs on the recording key) through tfa_stage_submit: 110 stages,
    round key 10 of FIPS-197 A.1, and every other round key."""
    be = GpuBackend(1 << 16)
    sk = M.ServerKey(be)
    rks = M.Key.from_u8_enc(FIPS_KEY, sk).generate_round_keys(sk)
    assert rks[10].decrypt_to_u8(sk).hex() == "d014f9a8c9ee2589e13f0cc8b6630ca6"
    assert be.stages == 110
    assert [k.decrypt_to_u8(sk) for k in rks] == B.key_expansion_clear(FIPS_KEY)
