"""First-light check of the HIP engine against the CPU oracle (run on the GPU box)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
import numpy as np
from oracle.pyoracle import Oracle
from tfhe_aes_amd import engine as E

o = Oracle(); p = E.default_params()
t = time.time(); lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, 0x5446484541455321); print("keygen %.2fs" % (time.time() - t), flush=True)
o_keys = o.keygen(0x5446484541455321)
print("client keygen == oracle keygen:", all((a == b).all() for a, b in zip((lwe_sk, glwe_sk, bsk, ksk), o_keys)), flush=True)
ctx = E.Context(p, 0)
t = time.time(); ctx.keys_upload(bsk, ksk); print("keys_upload %.2fs" % (time.time() - t), flush=True)
rng = np.random.default_rng(5)
# keyswitch bit-exact
big = rng.integers(0, 2**32, (5, ctx.big + 1), dtype=np.uint64).astype(np.uint32)
ks_gpu = ctx.keyswitch_batch(big)
ks_cpu = np.stack([o.keyswitch(ksk, b) for b in big])
print("keyswitch bit-exact:", (ks_gpu == ks_cpu).all(), flush=True)
# blind rotation prefix
G = 4
bits = rng.integers(0, 2, G).astype(np.uint8)
cts = E.client_encrypt(p, lwe_sk, 11, bits)
for steps in (0, 1, 2, 8, 805):
    acc_gpu = ctx.blind_rotate_batch(cts, steps)
    bad = 0; maxd = 0
    for g in range(G if steps < 805 else 2):
        acc_cpu = o.blind_rotate(bsk, None, cts[g], steps)
        d = (acc_gpu[g].astype(np.int64) - acc_cpu.astype(np.int64) + 2**31) % 2**32 - 2**31
        maxd = max(maxd, int(np.abs(d).max())); bad += int((d != 0).sum())
    print(f"blind_rotate steps={steps}: mismatching words {bad}, max |diff| {maxd}", flush=True)
# gates through the pool
G = 1024
a = rng.integers(0, 2, G).astype(np.uint8); b = rng.integers(0, 2, G).astype(np.uint8); c = rng.integers(0, 2, G).astype(np.uint8)
ca, cb, cc = (E.client_encrypt(p, lwe_sk, s, x) for s, x in ((21, a), (22, b), (23, c)))
ctx.pool_reserve(4 * G)
ctx.pool_write(np.arange(G), ca); ctx.pool_write(np.arange(G, 2 * G), cb); ctx.pool_write(np.arange(2 * G, 3 * G), cc)
ops = rng.integers(0, 7, G)
neg = rng.integers(0, 8, G)
gates = E.make_gates(ops, np.arange(G), np.arange(G, 2 * G), np.arange(2 * G, 3 * G), np.arange(3 * G, 4 * G), neg)
ctx.set_profiling(True)
t = time.time(); ctx.stage_submit(gates); ctx.sync(); dt = time.time() - t
out = ctx.pool_read(np.arange(3 * G, 4 * G))
na, nb, nc = a ^ (neg & 1), b ^ ((neg >> 1) & 1), c ^ ((neg >> 2) & 1)
exp = np.select([ops == 0, ops == 1, ops == 2, ops == 3, ops == 4, ops == 5, ops == 6],
                [na & nb, 1 - (na & nb), na | nb, 1 - (na | nb), na ^ nb, 1 - (na ^ nb), np.where(na == 1, nb, nc)]).astype(np.uint8)
got = E.client_decrypt(p, lwe_sk, out)
st = ctx.stats()
print(f"{G} mixed gates: wrong {int((got != exp).sum())}; wall {dt*1e3:.1f} ms; pbs {st.pbs_count} in {st.pbs_kernel_ms:.2f} ms "
      f"({st.pbs_count/st.pbs_kernel_ms*1e3:.0f} PBS/s); ks {st.ks_count} in {st.ks_kernel_ms:.2f} ms", flush=True)
# oracle comparison on 16 of them (bit-exact expected unless an FFT rounding differs)
idx = np.arange(16)
def negate(ct, m): return np.where(m[:, None] == 1, (0 - ct.astype(np.int64)) % 2**32, ct).astype(np.uint32)
oa, ob, oc = negate(ca[idx], neg[idx] & 1), negate(cb[idx], (neg[idx] >> 1) & 1), negate(cc[idx], (neg[idx] >> 2) & 1)
bsk_f = o.bsk_to_fourier(bsk)
ref = o.gates(bsk, bsk_f, ksk, ops[idx].astype(np.uint8), oa, ob, oc)
print("gate outputs identical to oracle (fft engine) on 16 gates:", int((ref == out[idx]).all(axis=1).sum()), "/ 16", flush=True)
# larger batch timing
G = 16384
ctx.pool_reserve(4 * G)
bits2 = rng.integers(0, 2, 2 * G).astype(np.uint8)
ct2 = E.client_encrypt(p, lwe_sk, 31, bits2)
ctx.pool_write(np.arange(2 * G), ct2)
gates = E.make_gates(np.where(np.arange(G) % 2 == 0, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))
ctx.reset_stats()
t = time.time(); ctx.stage_submit(gates); ctx.sync(); dt = time.time() - t
st = ctx.stats()
out = ctx.pool_read(np.arange(2 * G, 3 * G)); got = E.client_decrypt(p, lwe_sk, out)
x, y = bits2[:G], bits2[G:]
exp = np.where(np.arange(G) % 2 == 0, 1 - (x & y), x ^ y)
print(f"{G} NAND/XOR: wrong {int((got != exp).sum())}; wall {dt*1e3:.1f} ms -> {G/dt:.0f} gates/s; pbs kernel {st.pbs_kernel_ms:.1f} ms, ks kernel {st.ks_kernel_ms:.1f} ms", flush=True)
