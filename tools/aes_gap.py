"""One AES-128 ECB block: wall time against the sum of the kernels' times (what the host and the launch gaps cost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))
from tfhe_aes_amd import host as H
keys = H.gen_keys(7, 0)
rk = [H.Key.from_u8_enc(k, keys) for k in H.key_expansion_clear(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"))]
ecb = H.ECB(rk)
for prof in (False, True):
    state = H.State.from_u8_enc(bytes.fromhex("3243f6a8885a308d313198a2e0370734"), keys)
    warm = state.clone(); warm.sub_bytes(); keys.flush()
    keys.set_profiling(prof); keys.reset_engine_stats()
    t = time.perf_counter(); ecb.encrypt(state); keys.flush(); dt = time.perf_counter() - t
    st = keys.engine_stats()
    print(f"profiling={prof}: wall {dt*1e3:.1f} ms; kernels pbs {st.pbs_kernel_ms:.1f} ms in {st.pbs_launches} launches, ks {st.ks_kernel_ms:.1f} ms in {st.ks_launches}")
    assert state.decrypt_to_u8().hex() == "3925841d02dc09fbdc118597196a0b32"
