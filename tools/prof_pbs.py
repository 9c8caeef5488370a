"""Small fixed workload for rocprofv3: G NAND/XOR gates, one stage, run twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from tfhe_aes_amd import engine as E
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 1)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk); ctx.pool_reserve(3 * G)
bits = np.random.default_rng(0).integers(0, 2, 2 * G).astype(np.uint8)
ctx.pool_write(np.arange(2 * G), E.client_encrypt(p, lwe_sk, 5, bits))
gates = E.make_gates(np.where(np.arange(G) % 2 == 0, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))
ctx.set_profiling(True)
for _ in range(2):
    ctx.stage_submit(gates)
ctx.sync()
st = ctx.stats()
print("pbs ms/launch", st.pbs_kernel_ms / st.pbs_launches, "ks ms/launch", st.ks_kernel_ms / st.ks_launches, "PBS/s", G / (st.pbs_kernel_ms / st.pbs_launches) * 1e3)
