"""FHEByte::sub_byte on the CPU oracle: the staged S-box loop of the reference
(/root/reference/base/src/primitive.rs:209-274) restated over oracle gates, every gate of a stage in one
(OpenMP-parallel) oracle call, the stand-in for the reference's rayon pool (primitive.rs:253-262).

TEST INFRASTRUCTURE ONLY: imported by tests/ and the cpu_baseline leg of bench.py.  The product never imports this module.
"""
import numpy as np

from . import boolean_tree_oracle as B
from . import pyoracle as P

KIND_TO_OP = {1: P.AND, 2: P.OR, 3: P.XOR, 4: P.MUX}


def sub_byte_on_oracle(o, keys, bsk_f, bits_lsb_first, table):
    """bits_lsb_first: [8][n+1] ciphertexts of bit k of the byte (primitive.rs:210 reverses the MSB-first deque).
    Returns the 8 output ciphertexts, MSB first (primitive.rs:268-273), and the number of gates evaluated."""
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
