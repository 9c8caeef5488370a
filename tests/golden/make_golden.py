"""Generates tests/golden/tfhe_golden.json with the CPU oracle (never with the reference, which
cannot be built here and holds no ciphertext-level fixture).  Keys are regenerated from the seed at
test time; the fixture stores hashes and a few words so it stays small."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as P  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    o = P.Oracle()
    seed, enc_seed = 0x5446484541455321, 0x600D
    lwe_sk, glwe_sk, bsk, ksk = o.keygen(seed)
    ops = [P.AND, P.NAND, P.XOR, P.MUX, P.OR, P.NOR, P.XNOR, P.MUX]
    bits = np.array([[1, 1, 0, 1, 0, 0, 1, 0], [1, 0, 1, 1, 0, 1, 1, 0], [0, 0, 0, 0, 1, 1, 0, 1]], np.uint8)
    ins = [o.encrypt_bits(lwe_sk, enc_seed, s, bits[s]) for s in range(3)]
    out_exact = o.gates(bsk, None, ksk, np.array(ops, np.uint8), *ins)
    out_fft = o.gates(bsk, o.bsk_to_fourier(bsk), ksk, np.array(ops, np.uint8), *ins)
    assert (out_exact == out_fft).all()
    ks_in = [0x12345678, 0x9ABCDEF0, 0xFFFFFFFF, 0x80000000, 0x00010000, 0x0000FFFF, 0x7FFFFFFF, 0xDEADBEEF]
    big = np.array(ks_in + [0] * (o.params.k * o.params.N + 1 - 8), np.uint32)
    gold = {
        "_generator": "tests/golden/make_golden.py (CPU oracle, exact engine == fft engine)",
        "seed": seed, "enc_seed": enc_seed, "ops": ops, "input_bits": bits.reshape(-1).tolist(),
        "lwe_sk_sha256": sha(lwe_sk), "glwe_sk_sha256": sha(glwe_sk), "bsk_sha256": sha(bsk), "ksk_sha256": sha(ksk),
        "inputs_sha256": sha(np.stack(ins)), "outputs_sha256": sha(out_exact), "output_bodies": [int(x) for x in out_exact[:, -1]],
        "output_bits": [int(x) for x in o.decrypt_bits(lwe_sk, out_exact)],
        "ks_input_first8": ks_in, "ks_output_sha256": sha(o.keyswitch(ksk, big)),
    }
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tfhe_golden.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print(json.dumps({k: v for k, v in gold.items() if "sha" not in k}))


if __name__ == "__main__":
    main()
