"""Kernel time of one launch of G bootstraps cut off after a given number of blind-rotation steps: the slope is the
time per step, the intercept the fixed cost of a launch (prologue, first key fetches, uneven finish)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from tfhe_aes_amd import engine as E
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 1)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk)
bits = np.random.default_rng(0).integers(0, 2, G).astype(np.uint8)
cts = E.client_encrypt(p, lwe_sk, 5, bits)
ctx.set_profiling(True)
ctx.blind_rotate_batch(cts, 805)
for steps in (1, 100, 200, 400, 805, 805):
    ctx.reset_stats()
    ctx.blind_rotate_batch(cts, steps)
    st = ctx.stats()
    print(f"G={G} steps={steps:4d} kernel {st.pbs_kernel_ms:8.3f} ms")
