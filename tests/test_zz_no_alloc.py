"""tfa_stage_submit_noalloc (job lists and big-LWE rows in the library's static device array) against
tfa_stage_submit, through the C ABI on cuda:0: bit-identical output ciphertexts for the same gates, on two calls in a
row (the second smaller, so it runs on the first call's leftovers in the shared array), plus the size limit."""
import numpy as np
import pytest

from tfhe_aes_amd import engine as E

pytestmark = pytest.mark.gpu


def _stage(rng, n_gates, base_out, n_in):
    ops = rng.integers(0, 9, n_gates)           # every opcode, NOT and COPY included
    in0, in1, in2 = (rng.integers(0, n_in, n_gates) for _ in range(3))
    neg = rng.integers(0, 8, n_gates)
    return E.make_gates(ops, in0, in1, in2, base_out + np.arange(n_gates), neg)


def test_noalloc_matches_stage_submit_on_two_calls_in_a_row(gpu_ctx, oracle, okeys):
    lwe_sk = okeys[0]
    rng = np.random.default_rng(77)
    n_in = 512
    sizes = (700, 90)                            # wide (matrix-core keyswitch, 8-wave bootstraps), then narrow (latency kernels)
    total = sum(sizes)
    bits = rng.integers(0, 2, n_in).astype(np.uint8)
    gpu_ctx.pool_reserve(n_in + 2 * total)
    gpu_ctx.pool_write(np.arange(n_in), oracle.encrypt_bits(lwe_sk, 501, 0, bits))
    off = n_in
    stages = []
    for n in sizes:
        ref_gates = _stage(rng, n, off, n_in)
        new_gates = ref_gates.copy()
        new_gates["out"] = ref_gates["out"] + total
        stages.append((ref_gates, new_gates))
        off += n
    for ref_gates, _ in stages:                  # the existing entry point first ...
        gpu_ctx.stage_submit(ref_gates)
    gpu_ctx.sync()
    for _, new_gates in stages:                  # ... then the new one, twice in a row without a sync in between
        gpu_ctx.stage_submit_noalloc(new_gates)
    gpu_ctx.sync()
    for ref_gates, new_gates in stages:
        ref = gpu_ctx.pool_read(ref_gates["out"])
        got = gpu_ctx.pool_read(new_gates["out"])
        assert (got == ref).all()                # bit-exact, word for word
        # and the results are right, not just equal: decrypt against the clear gates
        a = bits[ref_gates["in0"]] ^ (ref_gates["neg"] & 1)
        b = bits[ref_gates["in1"]] ^ ((ref_gates["neg"] >> 1) & 1)
        c = bits[ref_gates["in2"]] ^ ((ref_gates["neg"] >> 2) & 1)
        op = ref_gates["op"]
        exp = np.select([op == 0, op == 1, op == 2, op == 3, op == 4, op == 5, op == 6, op == 7, op == 8],
                        [a & b, 1 - (a & b), a | b, 1 - (a | b), a ^ b, 1 - (a ^ b), np.where(a == 1, b, c), 1 - a, a]).astype(np.uint8)
        assert (oracle.decrypt_bits(lwe_sk, got) == exp).all()


def test_noalloc_rejects_stages_beyond_its_static_scratch(gpu_ctx):
    gpu_ctx.pool_reserve(3 * E.NOALLOC_MAX_GATES + 8)
    n = E.NOALLOC_MAX_GATES + 1
    with pytest.raises(E.TfaError, match="exceed the static scratch"):
        gpu_ctx.stage_submit_noalloc(E.make_gates(np.full(n, E.AND), np.zeros(n), np.ones(n), None, 2 + np.arange(n)))
    n = E.NOALLOC_MAX_PBS // 2 + 1               # MUX = two bootstraps: within the gate limit, beyond the bootstrap limit
    with pytest.raises(E.TfaError, match="bootstraps exceed the static scratch"):
        gpu_ctx.stage_submit_noalloc(E.make_gates(np.full(n, E.MUX), np.zeros(n), np.ones(n), np.ones(n), 2 + np.arange(n)))
    gpu_ctx.stage_submit_noalloc(np.zeros(0, E.GATE_DTYPE))  # empty stage: no-op
