"""Phase breakdown of pbs_kernel from the diagnostic build (make -C tfhe-aes_amd diag)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from tfhe_aes_amd import engine as E
E.LIB_PATH = os.path.join(ROOT, "tfhe-aes_amd", "libtfhe_aes_hip_diag.so")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = E.default_params()
lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 1)
ctx = E.Context(p, 0); ctx.keys_upload(bsk, ksk); ctx.pool_reserve(3 * G)
bits = np.random.default_rng(0).integers(0, 2, 2 * G).astype(np.uint8)
ctx.pool_write(np.arange(2 * G), E.client_encrypt(p, lwe_sk, 5, bits))
gates = E.make_gates(np.where(np.arange(G) % 2 == 0, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))
lib = E.load_library()
buf = (C.c_ulonglong * 32)()
ctx.stage_submit(gates); ctx.sync(); lib.tfa_debug_read_stamps(ctx._ctx, buf)
ctx.set_profiling(True)
ctx.stage_submit(gates); ctx.sync(); lib.tfa_debug_read_stamps(ctx._ctx, buf)
st = ctx.stats()
names = ["loop overhead", "rotate+decompose", "digit convert", "forward FFT", "ring wait+barrier", "key MAC+reduce", "inverse FFT", "round+accumulate"]
tot = sum(buf[i] for i in range(8))
print(f"G={G} waves={os.environ.get('TFA_PBS_WAVES','auto')} pbs {st.pbs_kernel_ms:.2f} ms; stamps total {tot} ticks ({tot/805:.0f} per step; s_memtime ticks at 100 MHz)")
names += ["(acquire) pre", "(acquire) own-DMA/LDS wait", "(acquire) barrier", "(acquire) refill issue"]
tot = sum(buf[i] for i in range(12))
for i, n in enumerate(names):
    print(f"  {n:28s} {buf[i]/805:9.1f} ticks/step  {100.0*buf[i]/tot:5.1f}%")
