"""Times the narrow-stage mappings: G gates per stage (G <= CUs: one workgroup per CU; <= 2 CUs: two), several stages."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from tfhe_aes_amd import engine as E
p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 1)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk)
for G in [int(x) for x in sys.argv[1:]] or [128, 256, 512]:
    ctx.pool_reserve(3 * G)
    bits = np.random.default_rng(0).integers(0, 2, 2 * G).astype(np.uint8)
    ctx.pool_write(np.arange(2 * G), E.client_encrypt(p, lwe_sk, 5, bits))
    gates = E.make_gates(np.where(np.arange(G) % 2 == 0, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))
    for _ in range(2): ctx.stage_submit(gates)
    ctx.sync(); ctx.reset_stats(); ctx.set_profiling(True)
    for _ in range(6): ctx.stage_submit(gates)
    ctx.sync(); st = ctx.stats(); ctx.set_profiling(False)
    out = E.client_decrypt(p, lwe_sk, ctx.pool_read(np.arange(2 * G, 3 * G)))
    ok = (out == np.where(np.arange(G) % 2 == 0, 1 - (bits[:G] & bits[G:]), bits[:G] ^ bits[G:])).all()
    print(f"G={G}: pbs {st.pbs_kernel_ms / st.pbs_launches:.3f} ms per stage, ks {st.ks_kernel_ms / st.ks_launches:.3f} ms, correct {ok}")
