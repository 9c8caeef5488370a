"""Soak run: ROUNDS x 65536 random two-input gates, each round feeding on the outputs of the two rounds before it;
counts results that decrypt wrongly (expected 0).  usage: python3 tools/soak.py [ROUNDS=256] [GATES_PER_ROUND=65536]
(narrow rounds exercise the one-bootstrap-per-workgroup and four-wave kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from tfhe_aes_amd import engine as E

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
p = E.default_params()
lwe_sk, _, bsk, ksk = E.client_keygen(p, 77)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk); ctx.pool_reserve(3 * G)
rng = np.random.default_rng(12)
bits = rng.integers(0, 2, 2 * G).astype(np.uint8)
ctx.pool_write(np.arange(2 * G), E.client_encrypt(p, lwe_sk, 4343, bits))
x, y = bits[:G].copy(), bits[G:].copy()
src_a, src_b, dst = 0, G, 2 * G
failures, t0 = 0, time.time()
for rnd in range(ROUNDS):
    ops = rng.integers(0, 6, G)  # AND NAND OR NOR XOR XNOR
    perm = rng.permutation(G)
    ctx.stage_submit(E.make_gates(ops, src_a + np.arange(G), src_b + perm, None, dst + np.arange(G)))
    ctx.sync()
    a, b = x, y[perm]
    exp = np.select([ops == 0, ops == 1, ops == 2, ops == 3, ops == 4, ops == 5], [a & b, 1 - (a & b), a | b, 1 - (a | b), a ^ b, 1 - (a ^ b)]).astype(np.uint8)
    got = E.client_decrypt(p, lwe_sk, ctx.pool_read(dst + np.arange(G)))
    failures += int((got != exp).sum())
    x, y = exp, x
    src_a, src_b, dst = dst, src_a, src_b
    if rnd % 16 == 15:
        print(f"round {rnd + 1}/{ROUNDS}: {(rnd + 1) * G} gates, {failures} failures, {time.time() - t0:.0f} s", flush=True)
print(f"total {ROUNDS * G} gates, {failures} decryption failures")
sys.exit(1 if failures else 0)
