"""Bit-exactness of an experiment build on the GPU: 64 bootstraps through TFHE_AES_HIP_LIB's wide kernels (8-wave and 4-wave
mapping) against the CPU oracle's exact engine.  usage: TFHE_AES_HIP_LIB=tools/exp/lib_x.so python3 tools/check_exp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from oracle.pyoracle import Oracle
from tfhe_aes_amd import engine as E
o = Oracle()
p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = o.keygen(0x5446484541455321)
bad = 0
for waves in ("8", "4"):
    os.environ["TFA_PBS_WAVES"] = waves
    ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk)
    G = 64
    rng = np.random.default_rng(3)
    a, b = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
    ca, cb = o.encrypt_bits(lwe_sk, 11, 0, a), o.encrypt_bits(lwe_sk, 12, 0, b)
    ctx.pool_reserve(3 * G)
    ctx.pool_write(np.arange(G), ca); ctx.pool_write(np.arange(G, 2 * G), cb)
    ops = np.where(np.arange(G) % 2 == 0, E.NAND, E.XOR)
    ctx.stage_submit(E.make_gates(ops, np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G)))
    ctx.sync()
    out = ctx.pool_read(np.arange(2 * G, 3 * G))
    ref = o.gates(bsk, None, ksk, ops[:4].astype(np.uint8), ca[:4], cb[:4])      # exact-integer engine (no Fourier key given)
    dec = o.decrypt_bits(lwe_sk, out)
    exp = np.where(ops == E.NAND, 1 - (a & b), a ^ b)
    nbad = int((dec != exp).sum())
    same = bool((ref == out[:4]).all())
    print("waves", waves, "wrong decryptions", nbad, "first 4 outputs equal the exact oracle:", same)
    bad += nbad + (0 if same else 1)
    ctx.close()
sys.exit(1 if bad else 0)
