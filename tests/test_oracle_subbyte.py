"""BASELINE configs[0] on the CPU: one encrypted byte through the staged S-box circuit, stage by stage, every gate of a
stage in one (OpenMP-parallel) oracle call: the loop of FHEByte::sub_byte (/root/reference/base/src/primitive.rs:209-274)
restated over the oracle, with the reference's own expectation 0x01 -> 0x7c (primitive.rs:438-452)."""
import time

import numpy as np

from oracle import boolean_tree_oracle as B
from oracle import pyoracle as P

KIND_TO_OP = {1: P.AND, 2: P.OR, 3: P.XOR, 4: P.MUX}


def sub_byte_on_oracle(o, keys, bsk_f, bits_lsb_first, table):
    """bits_lsb_first: [8][n+1] ciphertexts of bit k of the byte.  Returns the 8 output ciphertexts, MSB first."""
    lwe_sk, glwe_sk, bsk, ksk = keys
    roots = B.generate_reduced_bool_expr(table)
    stages = B.staged(roots)
    val = {}
    for e in stages[0]:                       # leaves: Bit_k = 2 + 2k, NotBit_k = 3 + 2k (negation = LWE negation)
        k, negated = (e[1] - 2) // 2, (e[1] - 2) % 2
        ct = bits_lsb_first[k]
        val[e] = (np.uint32(0) - ct) if negated else ct
    n_gates = 0
    for st in stages[1:]:
        ops = np.array([KIND_TO_OP[e[0]] for e in st], np.uint8)
        def operand_ct(code):                 # a mux selector is a bare operand code
            ct = bits_lsb_first[(code - 2) // 2]
            return (np.uint32(0) - ct) if (code - 2) % 2 else ct
        ins = [np.stack([(operand_ct(e[1]) if e[0] == 4 else val[e[1]]) for e in st]),
               np.stack([val[e[2]] for e in st]),
               np.stack([(val[e[3]] if e[0] == 4 else val[e[2]]) for e in st])]
        with np.errstate(over="ignore"):
            out = o.gates(bsk, bsk_f, ksk, ops, ins[0], ins[1], ins[2])
        for e, ct in zip(st, out):
            val[e] = ct
        n_gates += len(st)
    return [val[r] for r in roots], n_gates


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
