import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from oracle.pyoracle import Oracle
from tfhe_aes_amd import engine as E
o = Oracle(); p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = o.keygen(77)
bsk_f = o.bsk_to_fourier(bsk)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk); ctx.pool_reserve(4096)
rng = np.random.default_rng(0)
tot_bad = 0; tot_mis = 0
for rep in range(60):
    G = int(rng.integers(1, 17))
    a = rng.integers(0, 2, G).astype(np.uint8); b = rng.integers(0, 2, G).astype(np.uint8)
    ca = o.encrypt_bits(lwe_sk, 1000 + rep, 0, a); cb = o.encrypt_bits(lwe_sk, 1000 + rep, 1, b)
    base = int(rng.integers(0, 1000))
    ctx.pool_write(base + np.arange(G), ca); ctx.pool_write(base + 100 + np.arange(G), cb)
    ops = rng.integers(0, 6, G)
    ctx.stage_submit(E.make_gates(ops, base + np.arange(G), base + 100 + np.arange(G), None, base + 200 + np.arange(G)))
    ctx.sync()
    out = ctx.pool_read(base + 200 + np.arange(G))
    ref = o.gates(bsk, bsk_f, ksk, ops.astype(np.uint8), ca, cb)
    mis = int((out != ref).any(axis=1).sum())
    exp = np.select([ops == 0, ops == 1, ops == 2, ops == 3, ops == 4, ops == 5], [a & b, 1 - (a & b), a | b, 1 - (a | b), a ^ b, 1 - (a ^ b)])
    bad = int((o.decrypt_bits(lwe_sk, out) != exp).sum())
    tot_bad += bad; tot_mis += mis
    if mis or bad:
        print("rep", rep, "G", G, "mismatch rows", np.nonzero((out != ref).any(axis=1))[0], "bad", bad, "ops", ops, flush=True)
        # diagnose: PBS only
        for g in np.nonzero((out != ref).any(axis=1))[0]:
            u1, _ = o.precombine(int(ops[g]), ca[g], cb[g])
            big_gpu = ctx.bootstrap_batch(u1[None])[0]
            big_ref = o.pbs(bsk, bsk_f, u1)
            print("   standalone pbs equal:", (big_gpu == big_ref).all(), " ks of ref equal to out:", (o.keyswitch(ksk, big_ref) == out[g]).all(),
                  "ndiff words", int((out[g] != ref[g]).sum()), flush=True)
print("total mismatching gates", tot_mis, "decrypt errors", tot_bad)
