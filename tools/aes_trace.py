"""One AES-128 ECB block through the host mirror; with TFA_TRACE=1 the engine prints every stage's width."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
from tfhe_aes_amd import host as H
keys = H.gen_keys(7, 0)
rk = [H.Key.from_u8_enc(k, keys) for k in H.key_expansion_clear(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"))]
st = H.State.from_u8_enc(bytes.fromhex("3243f6a8885a308d313198a2e0370734"), keys)
warm = st.clone(); warm.sub_bytes(); keys.flush()
print("== timed block ==", file=sys.stderr, flush=True)
t = time.perf_counter(); H.ECB(rk).encrypt(st); keys.flush(); dt = time.perf_counter() - t
print("ecb block s", dt, st.decrypt_to_u8().hex())
