"""The host mirror (FHEByte / State / Key / modes) on the HIP engine, written after the reference's
own tests: /root/reference/base/src/primitive.rs:387-470, sbox.rs:113-166, state.rs:321-477,
key_schedule.rs:220-246, modes/src/{ecb,cbc,ofb,ctr}.rs tests, plus the SP 800-38A vectors."""
import json
import os

import numpy as np
import pytest

from oracle import boolean_tree_oracle as B
from tfhe_aes_amd import host as H

pytestmark = pytest.mark.gpu
V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aes_vectors.json")))
h = bytes.fromhex


def test_gate_api_truth_tables_and_trivials(hkeys):
    k = hkeys
    e = {b: k.encrypt(b) for b in (False, True)}
    t = {b: k.trivial_encrypt(b) for b in (False, True)}
    fns = {"and": (k.and_, lambda a, b: a and b), "or": (k.or_, lambda a, b: a or b), "xor": (k.xor, lambda a, b: a != b),
           "nand": (k.nand, lambda a, b: not (a and b)), "nor": (k.nor, lambda a, b: not (a or b)), "xnor": (k.xnor, lambda a, b: a == b)}
    results = []
    for name, (fn, clear) in fns.items():
        for a in (False, True):
            for b in (False, True):
                results.append((fn(e[a], e[b]), clear(a, b)))
                r = fn(e[a], t[b])  # a trivial operand never costs a bootstrap
                results.append((r, clear(a, b)))
                results.append((fn(t[a], t[b]), clear(a, b)))
                assert fn(t[a], t[b]).is_trivial()
    for c in (False, True):
        for a in (False, True):
            for b in (False, True):
                results.append((k.mux(e[c], e[a], e[b]), a if c else b))
                results.append((k.mux(e[c], t[a], e[b]), a if c else b))
                results.append((k.mux(e[c], e[a], t[b]), a if c else b))
                results.append((k.mux(t[c], e[a], e[b]), a if c else b))
                results.append((k.mux(e[c], k.not_(e[a]), e[b]), (not a) if c else b))
    before = k.host_stats()["gates"]
    k.flush()
    assert k.host_stats()["stages"] >= 1 and k.host_stats()["gates"] > before
    for ct, exp in results:
        assert k.decrypt(ct) == bool(exp)
    assert k.decrypt(k.not_(e[True])) is False and k.decrypt(k.not_(t[False])) is True


def test_xor_and_bytes(hkeys):  # primitive.rs:387-436
    x, y = H.FHEByte.from_u8_enc(0b10101010, hkeys), H.FHEByte.from_u8_enc(0b11001100, hkeys)
    assert x.xor(y).decrypt_to_u8() == 0b01100110
    assert x.and_(y).decrypt_to_u8() == 0b10001000
    assert x.or_(y).decrypt_to_u8() == 0b11101110
    assert x.not_().decrypt_to_u8() == 0b01010101
    assert x.xor(H.FHEByte.from_u8_clear(0xFF, hkeys)).decrypt_to_u8() == 0b01010101


def test_sub_byte(hkeys):  # primitive.rs:438-452
    x = H.FHEByte.from_u8_enc(0x01, hkeys)
    before = hkeys.host_stats()
    y = x.sub_byte()
    assert y.decrypt_to_u8() == 0x7C
    after = hkeys.host_stats()
    assert after["gates"] - before["gates"] == 405 and after["stages"] - before["stages"] == 7
    assert y.inv_sub_byte().decrypt_to_u8() == 0x01


@pytest.mark.timeout(300)
@pytest.mark.parametrize("inverse", [False, True])
def test_all_sbox_values(hkeys, inverse):  # sbox.rs:113-166, all 256 inputs in one batched evaluation
    table = B.INV_S_BOX if inverse else B.S_BOX
    xs = [H.FHEByte.from_u8_enc(v, hkeys) for v in range(256)]
    before = hkeys.host_stats()
    ys = [x.inv_sub_byte() if inverse else x.sub_byte() for x in xs]
    hkeys.flush()
    after = hkeys.host_stats()
    assert after["stages"] - before["stages"] == 7  # 256 bytes share the 7 stages
    assert after["gates"] - before["gates"] == 256 * 405
    assert [y.decrypt_to_u8() for y in ys] == list(table)


def test_mul_gf_2_all_values(hkeys):  # primitive.rs:454-470
    xs = [H.FHEByte.from_u8_enc(v, hkeys) for v in range(256)]
    before = hkeys.host_stats()["gates"]
    ys = [x.mul_x_gf2() for x in xs]
    hkeys.flush()
    assert hkeys.host_stats()["gates"] - before == 256 * 7  # 7 muxes per byte, the LSB is free
    clear = lambda v: ((v << 1) ^ 0x1B) & 0xFF if v & 0x80 else (v << 1)
    assert [y.decrypt_to_u8() for y in ys] == [clear(v) for v in range(256)]


def test_state_round_functions(hkeys):  # state.rs:321-438, FIPS-197 Appendix B round 1
    v = V["fips197_b"]
    s = H.State.from_u128_enc(int(v["round1_start"], 16), hkeys)
    assert s.decrypt_to_u8().hex() == v["round1_start"]
    s.sub_bytes()
    assert s.decrypt_to_u8().hex() == v["round1_sub_bytes"]
    s.shift_rows()
    assert s.decrypt_to_u8().hex() == v["round1_shift_rows"]
    s.mix_columns()
    assert s.decrypt_to_u8().hex() == v["round1_mix_columns"]
    s.inv_mix_columns()
    assert s.decrypt_to_u8().hex() == v["round1_shift_rows"]
    s.inv_shift_rows()
    assert s.decrypt_to_u8().hex() == v["round1_sub_bytes"]
    s.inv_sub_bytes()
    assert s.decrypt_to_u8().hex() == v["round1_start"]
    s.xor_key_clear(h(v["key"]))
    assert s.decrypt_to_u8() == bytes(a ^ b for a, b in zip(h(v["round1_start"]), h(v["key"])))
    assert s.decrypt_to_u128() == int.from_bytes(s.decrypt_to_u8(), "big")


@pytest.mark.timeout(300)
def test_key_schedule_online(hkeys):  # key_schedule.rs:220-246 prints; here the round keys are asserted
    key = h(V["fips197_b"]["key"])
    rks = H.Key.from_u128_enc(int.from_bytes(key, "big"), hkeys).generate_round_keys()
    expected = B.key_expansion_clear(key)
    assert [rk.decrypt_to_u8() for rk in rks] == expected
    assert expected[10].hex() == V["fips197_b"]["round_key_10"]
    clear = H.Key.from_u8_clear(key, hkeys).generate_round_keys()  # trivial key: no bootstrap at all
    assert [rk.decrypt_to_u8() for rk in clear] == expected


# Dependent stages of one block.  As recorded (set_xor_rebalance(False)): 117 / 144.  With the default re-association of
# unobservable xor chains (MixColumns + AddRoundKey: three levels instead of five): 99 / 135.
ECB_ENCRYPT_STAGES_AS_RECORDED = 117
ECB_DECRYPT_STAGES_AS_RECORDED = 144
ECB_ENCRYPT_STAGES = 99
ECB_DECRYPT_STAGES = 135


def offline_round_keys(key, hkeys):
    return [H.Key.from_u8_enc(rk, hkeys) for rk in H.key_expansion_clear(key)]


@pytest.mark.timeout(300)
def test_ecb_once(hkeys):  # modes/src/ecb.rs:62-89, FIPS-197 Appendix B
    v = V["fips197_b"]
    ecb = H.ECB(offline_round_keys(h(v["key"]), hkeys))
    state = H.State.from_u128_enc(int(v["plaintext"], 16), hkeys)
    # the batch submitter levels the recorded gates by data dependence: 117 stages where the reference's loop
    # order has 126 (ecb.rs:16-32); the inverse cipher levels to 144 where it has 198 (ecb.rs:34-50, state.rs:120-215);
    # re-associating the xor chains (the default) takes the same gates in 99 / 135 stages
    try:
        for rebalance, enc_stages, dec_stages in ((False, ECB_ENCRYPT_STAGES_AS_RECORDED, ECB_DECRYPT_STAGES_AS_RECORDED),
                                                  (True, ECB_ENCRYPT_STAGES, ECB_DECRYPT_STAGES)):
            hkeys.set_xor_rebalance(rebalance)
            state = H.State.from_u128_enc(int(v["plaintext"], 16), hkeys)
            before = hkeys.host_stats()
            ecb.encrypt(state)
            assert state.decrypt_to_u128() == int(v["ciphertext"], 16)
            after = hkeys.host_stats()
            assert after["gates"] - before["gates"] == 71824
            assert after["stages"] - before["stages"] == enc_stages
            before = after
            ecb.decrypt(state)
            assert state.decrypt_to_u128() == int(v["plaintext"], 16)
            after = hkeys.host_stats()
            assert after["gates"] - before["gates"] == 75604
            assert after["stages"] - before["stages"] == dec_stages
    finally:
        hkeys.set_xor_rebalance(True)


@pytest.mark.timeout(300)
def test_ecb_twice_same_result(hkeys):  # modes/src/ecb.rs:91-114
    v = V["fips197_c1"]
    ecb = H.ECB(offline_round_keys(h(v["key"]), hkeys))
    s0 = H.State.from_u8_enc(h(v["plaintext"]), hkeys)
    s1 = s0.clone()
    ecb.encrypt_many([s0, s1])
    assert s0.decrypt_to_u8().hex() == s1.decrypt_to_u8().hex() == v["ciphertext"]


@pytest.mark.timeout(600)
def test_sp800_38a_cbc_ofb_ctr(hkeys):  # modes/src/{cbc,ofb,ctr}.rs tests + NIST known answers
    v = V["sp800_38a"]
    rk = offline_round_keys(h(v["key"]), hkeys)
    mk = lambda: [H.State.from_u8_enc(h(p), hkeys) for p in v["plaintext"]]
    iv = H.State.from_u8_enc(h(v["iv"]), hkeys)
    blocks = mk()
    cbc = H.CBC(rk, iv, 4)
    cbc.encrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["cbc"]
    cbc.decrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["plaintext"]
    blocks = mk()
    ofb = H.OFB(rk, iv, 4)
    ofb.encrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["ofb"]
    ofb.decrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["plaintext"]
    c0 = int(v["ctr_init"], 16)
    counters = [H.State.from_u128_enc((c0 + i) % (1 << 128), hkeys) for i in range(4)]
    blocks = mk()
    ctr = H.CTR(rk, counters, 4)
    ctr.encrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["ctr"]
    ctr.decrypt(blocks)
    assert [b.decrypt_to_u8().hex() for b in blocks] == v["plaintext"]
    with pytest.raises(H.HostError):
        H.CTR(rk, counters[:3], 4)


def test_state_read_import_roundtrip(hkeys):
    data = bytes(range(16))
    s = H.State.from_u8_enc(data, hkeys)
    s.xor_key_clear(bytes([0xFF] * 16))
    lwe = s.read()
    assert lwe.shape == (128, hkeys.params.lwe_dimension + 1)
    t = H.State.import_lwe(lwe, hkeys)
    assert t.decrypt_to_u8() == bytes(b ^ 0xFF for b in data)


def test_states_read_into_device_memory(hkeys):
    """tfh_states_read_device: the rows a collective would carry, written on the GPU, equal the host read word for word
    (negated handles and trivial bits included)."""
    import torch
    a = H.State.from_u8_enc(bytes(range(16)), hkeys)
    b = H.State.from_u8_enc(bytes(range(16, 32)), hkeys)
    b.xor_key_clear(bytes([0x5A] * 16))           # negation flags on half of the bits
    a.xor_state(b)                                # bootstrapped outputs
    out = torch.zeros((2, 128, hkeys.params.lwe_dimension + 1), dtype=torch.int32, device="cuda:0")
    H.read_states_device([a, b], out)
    got = out.cpu().numpy().view(np.uint32)
    assert (got[0] == a.read()).all() and (got[1] == b.read()).all()
    assert H.State.import_lwe(got[1], hkeys).decrypt_to_u8() == bytes(x ^ 0x5A for x in range(16, 32))


@pytest.mark.timeout(900)
def test_config5_64_ctr_blocks_sharded_over_8_ranks(hkeys):
    """BASELINE configs[4] on the one GPU at hand: 64 independent CTR blocks (SP 800-38A F.5.1 counters extended),
    cut by shard.block_partition(64, 8, r) exactly as 8 GPUs would cut them, every rank's 8 blocks run as one batch
    (ctr.rs:26-35), the shards exported on the device and reassembled in block order; every block against a clear AES,
    the first four against the standard."""
    import torch
    from tfhe_aes_amd import shard
    v = V["sp800_38a"]
    key = h(v["key"])
    rk = offline_round_keys(key, hkeys)
    c0 = int(v["ctr_init"], 16)
    rng = np.random.default_rng(64)
    n_blocks, world = 64, 8
    pts = [h(v["plaintext"][b]) if b < 4 else rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for b in range(n_blocks)]
    words = hkeys.params.lwe_dimension + 1
    shards = []
    for r in range(world):
        mine = shard.block_partition(n_blocks, world, r)
        assert len(mine) == 8
        counters = [H.State.from_u128_enc((c0 + b) % (1 << 128), hkeys) for b in mine]
        states = [H.State.from_u8_enc(pts[b], hkeys) for b in mine]
        before = hkeys.host_stats()["stages"]
        H.CTR(rk, counters, len(mine)).encrypt(states)
        local = torch.empty((len(mine), 128, words), dtype=torch.int32, device="cuda:0")
        H.read_states_device(states, local)
        # 8 blocks share the stages; the final xor with the plaintext joins the last AddRoundKey's xor tree: no stage of its own
        assert hkeys.host_stats()["stages"] - before == ECB_ENCRYPT_STAGES
        shards.append(local)
    # what all_gather_into_tensor + the reorder of shard.gather_blocks_device produce: rank-major shards -> block order
    allbuf = torch.cat(shards)
    b = torch.arange(n_blocks, device="cuda:0")
    out = allbuf[(b % world) * 8 + b // world].cpu().numpy().view(np.uint32)
    got = [H.State.import_lwe(out[i], hkeys).decrypt_to_u8() for i in range(n_blocks)]
    exp = [bytes(x ^ y for x, y in zip(B.aes_encrypt_block(B.key_expansion_clear(key), ((c0 + i) % (1 << 128)).to_bytes(16, "big")), pts[i])) for i in range(n_blocks)]
    assert got == exp
    assert [x.hex() for x in got[:4]] == v["ctr"]


@pytest.mark.timeout(300)
def test_xor_chain_rebalance_random_circuits(hkeys):
    """ServerKey's re-association of unobservable xor chains (tfhe_boolean.cpp: rebalance_xor_chains) on random circuits: xor /
    xnor chains with negated operands, intermediates that are dropped (re-associated), kept (observable: must keep their
    value) or shared by two consumers, other gates and muxes in between.  Every kept handle decrypts to the clear evaluation,
    the gate count is the recorded one, and the stages are never more than without the re-association."""
    k = hkeys
    rng = np.random.default_rng(20240607)
    stages = {}
    try:
        for rebalance in (False, True):
            k.set_xor_rebalance(rebalance)
            rng = np.random.default_rng(77)
            before = k.host_stats()
            checked = 0
            for trial in range(6):
                vals = [bool(b) for b in rng.integers(0, 2, 12)]
                live = [(k.encrypt(v), v) for v in vals]                     # (handle, clear value)
                kept = []
                for step in range(120):
                    kind = rng.integers(0, 10)
                    pick = lambda: live[rng.integers(max(0, len(live) - 6), len(live))] if rng.integers(0, 3) else live[rng.integers(0, len(live))]
                    a, b = pick(), pick()
                    if rng.integers(0, 4) == 0:
                        a = (k.not_(a[0]), not a[1])
                    if kind < 6:
                        r = (k.xor(a[0], b[0]), a[1] != b[1]) if rng.integers(0, 4) else (k.xnor(a[0], b[0]), a[1] == b[1])
                    elif kind < 8:
                        r = (k.and_(a[0], b[0]), a[1] and b[1])
                    else:
                        c = pick()
                        r = (k.mux(c[0], a[0], b[0]), a[1] if c[1] else b[1])
                    if rng.integers(0, 5) == 0:
                        kept.append(r)                                       # observable for good
                    live.append(r)
                    if len(live) > 14 and rng.integers(0, 2):                # drop an older handle: what it fed may re-associate
                        del live[rng.integers(12, len(live) - 1)]
                kept.extend(live[12:])
                del live
                for ct, want in kept:
                    assert k.decrypt(ct) == want
                    checked += 1
            after = k.host_stats()
            stages[rebalance] = (after["stages"] - before["stages"], after["gates"] - before["gates"])
            assert checked > 100
    finally:
        k.set_xor_rebalance(True)
    assert stages[True][1] == stages[False][1]           # same gates
    assert stages[True][0] <= stages[False][0]           # never more dependent stages
    # a plain chain of eight operands, intermediates dropped, one operand negated, one link an xnor: 7 gates in 3 stages
    # instead of 7; with the middle of the chain kept by the caller the two halves re-associate on their own (2 + 2 stages)
    bits = [True, False, True, True, False, False, True, False]
    for keep_middle, want_stages in ((False, 3), (True, 4)):
        cts = [k.encrypt(b) for b in bits]
        k.flush()
        before = k.host_stats()
        acc, clear, middle = cts[0], bits[0], None
        for i in range(1, 8):
            rhs = k.not_(cts[i]) if i == 2 else cts[i]
            acc = k.xnor(acc, rhs) if i == 5 else k.xor(acc, rhs)
            clear = (clear != (bits[i] != (i == 2))) != (i == 5)
            if i == 4 and keep_middle:
                middle = (acc, clear)
        assert k.decrypt(acc) == clear
        if middle:
            assert k.decrypt(middle[0]) == middle[1]
        after = k.host_stats()
        assert after["gates"] - before["gates"] == 7 and after["stages"] - before["stages"] == want_stages
