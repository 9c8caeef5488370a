"""The clear AES of the oracle against FIPS-197 / SP 800-38A: pins every known answer the
encrypted tests rely on (the role the `aes` crate plays in the reference's main.rs:136-171)."""
import json
import os

from oracle import boolean_tree_oracle as B

V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aes_vectors.json")))
h = bytes.fromhex


def xor(a, b):
    return bytes(x ^ y for x, y in zip(a, b))


def test_fips197_appendix_b_and_c1():
    for name in ("fips197_b", "fips197_c1"):
        v = V[name]
        rk = B.key_expansion_clear(h(v["key"]))
        ct = B.aes_encrypt_block(rk, h(v["plaintext"]))
        assert ct.hex() == v["ciphertext"]
        assert B.aes_decrypt_block(rk, ct).hex() == v["plaintext"]
    assert B.key_expansion_clear(h(V["fips197_b"]["key"]))[10].hex() == V["fips197_b"]["round_key_10"]


def test_round1_intermediate_states():
    v = V["fips197_b"]
    s = h(v["round1_start"])
    sb = bytes(B.S_BOX[x] for x in s)
    assert sb.hex() == v["round1_sub_bytes"]
    sr = bytes(sb[(4 * c + r + 4 * r) % 16] for c in range(4) for r in range(4))
    assert sr.hex() == v["round1_shift_rows"]


def test_sp800_38a_modes():
    v = V["sp800_38a"]
    rk = B.key_expansion_clear(h(v["key"]))
    pt = [h(x) for x in v["plaintext"]]
    assert [B.aes_encrypt_block(rk, p).hex() for p in pt] == v["ecb"]
    prev, out = h(v["iv"]), []
    for p in pt:
        prev = B.aes_encrypt_block(rk, xor(p, prev))
        out.append(prev.hex())
    assert out == v["cbc"]
    ks, out = h(v["iv"]), []
    for p in pt:
        ks = B.aes_encrypt_block(rk, ks)
        out.append(xor(p, ks).hex())
    assert out == v["ofb"]
    ctr0, out = int.from_bytes(h(v["ctr_init"]), "big"), []
    for i, p in enumerate(pt):
        out.append(xor(p, B.aes_encrypt_block(rk, ((ctr0 + i) % (1 << 128)).to_bytes(16, "big"))).hex())
    assert out == v["ctr"]
