"""BASELINE configs[0] on the CPU: one encrypted byte through the staged S-box circuit, stage by stage, every gate of a
stage in one (OpenMP-parallel) oracle call: the loop of FHEByte::sub_byte (/root/reference/base/src/primitive.rs:209-274)
restated over the oracle, with the reference's own expectation 0x01 -> 0x7c (primitive.rs:438-452)."""
import time

import numpy as np

from oracle import boolean_tree_oracle as B
from oracle.subbyte_oracle import sub_byte_on_oracle

def test_sub_byte_0x01_gives_0x7c_on_the_cpu_path(oracle, okeys, obsk_f):
    lwe_sk = okeys[0]
    x = 0x01
    bits = np.array([(x >> k) & 1 for k in range(8)], np.uint8)
    cts = oracle.encrypt_bits(lwe_sk, 77, 0, bits)
    t = time.perf_counter()
    out, n_gates = sub_byte_on_oracle(oracle, okeys, obsk_f, cts, B.S_BOX)
    dt = time.perf_counter() - t
    got = oracle.decrypt_bits(lwe_sk, np.stack(out))
    assert sum(int(b) << (7 - i) for i, b in enumerate(got)) == 0x7C
    assert n_gates == 405
    print(f"SubBytes on the CPU oracle: {dt:.2f} s with {oracle.num_threads()} threads")
