"""Parity of the HIP kernels with the CPU oracle, through the C ABI (ctypes), on cuda:0.
Integer stages (gate combination, modulus switch, rotation, decomposition, sample extract,
keyswitch) must be bit-exact.  The blind rotation runs an fp64 FFT: its products are rounded to
integers and the rounding error of the transform stays far below 1/2, so its outputs are expected
to be bit-identical to the exact-integer oracle too; the hard bound asserted is the FFT noise
bound below, and the number of bit-identical results is asserted to be nearly all of them."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as P
from tfhe_aes_amd import engine as E

pytestmark = pytest.mark.gpu

# |phase(gpu) - phase(exact oracle)| allowed on a bootstrap output, in torus units of 2^-32.  One
# flipped decomposition digit moves the phase by about 2^12; the blind-rotation noise itself has a
# standard deviation of about 2^21, so 2^16 is the "FFT noise bound": far below the scheme's noise.
FFT_NOISE_BOUND = 1 << 16


def sdiff(a, b):
    return (a.astype(np.int64) - b.astype(np.int64) + 2**31) % 2**32 - 2**31


def test_keyswitch_bit_exact(gpu_ctx, oracle, okeys):
    rng = np.random.default_rng(1)
    big = rng.integers(0, 2**32, (37, gpu_ctx.big + 1), dtype=np.uint64).astype(np.uint32)
    big[0] = 0
    big[1] = 0xFFFFFFFF
    big[2] = 0x80000000
    big[3] = 0x00010000  # exactly half of the last representable step: exercises the rounding tie
    got = gpu_ctx.keyswitch_batch(big)
    ref = np.stack([oracle.keyswitch(okeys[3], b) for b in big])
    assert (got == ref).all()


def test_keyswitch_tile_sizes_bit_exact(gpu_ctx, oracle, okeys):
    """Batch sizes from one gate to many gate tiles: the matrix-core keyswitch with every split of the input
    coefficients that the engine picks (24-fold for one tile down to unsplit; partial sums combined with integer atomics)."""
    rng = np.random.default_rng(2)
    for count in (1, 37, 300, 1100, 2500, 8200):
        big = rng.integers(0, 2**32, (count, gpu_ctx.big + 1), dtype=np.uint64).astype(np.uint32)
        got = gpu_ctx.keyswitch_batch(big)
        for i in (0, count // 2, count - 1):
            assert (got[i] == oracle.keyswitch(okeys[3], big[i])).all()
        # linearity in the body: the mask part does not depend on b
        big2 = big[:1].copy()
        big2[0, -1] += 12345
        assert (sdiff(gpu_ctx.keyswitch_batch(big2)[0], got[0])[:-1] == 0).all()


def test_keyswitch_vector_path_bit_exact(oracle, okeys):
    """The vector-ALU keyswitch kernels (other parameter sets in production; the default set always takes the
    matrix-core kernel), every tile width with and without the split over input coefficients: pinned with TFA_KS_VALU=1."""
    import os
    os.environ["TFA_KS_VALU"] = "1"
    try:
        ctx = E.Context(E.default_params(), 0)
    finally:
        del os.environ["TFA_KS_VALU"]
    try:
        ctx.keys_upload(okeys[2], okeys[3])
        rng = np.random.default_rng(3)
        for count in (1, 37, 300, 2500, 8200):
            big = rng.integers(0, 2**32, (count, ctx.big + 1), dtype=np.uint64).astype(np.uint32)
            got = ctx.keyswitch_batch(big)
            for i in (0, count // 3, count - 1):
                assert (got[i] == oracle.keyswitch(okeys[3], big[i])).all()
    finally:
        ctx.close()


def test_keyswitch_matrix_core_tile_edges(gpu_ctx, oracle, okeys):
    """Matrix-core keyswitch at its tile edges: exactly one tile of 128 gates short of / past a boundary, extreme
    inputs (all-ones rounds up into the top digit and wraps), every row checked on the smallest batch."""
    rng = np.random.default_rng(4)
    for count in (256, 257, 383, 385):
        big = rng.integers(0, 2**32, (count, gpu_ctx.big + 1), dtype=np.uint64).astype(np.uint32)
        big[0] = 0xFFFFFFFF
        big[1] = 0x80000000
        big[2] = 0x7FFFFFFF
        big[3] = 0x00010000
        big[count - 1, ::7] = 0xFFFF0000
        got = gpu_ctx.keyswitch_batch(big)
        rows = range(count) if count == 256 else (0, 1, 2, 3, 127, 128, 255, 256, count - 1)
        for i in rows:
            assert (got[i] == oracle.keyswitch(okeys[3], big[i])).all(), (count, i)


def test_blind_rotation_prefix_bit_exact(gpu_ctx, oracle, okeys):
    """Accumulator after 0, 1, 2 and 9 external products against the exact-integer oracle."""
    cts = oracle.encrypt_bits(okeys[0], 77, 0, np.array([0, 1, 1], np.uint8))
    cts[2, 5] = 0  # a mask word that modulus-switches to 0: the skipped-iteration path
    for steps in (0, 1, 2, 9):
        got = gpu_ctx.blind_rotate_batch(cts, steps)
        for g in range(3):
            ref = oracle.blind_rotate(okeys[2], None, cts[g], steps)
            # every word equal; on failure the message carries the size of the disagreement (the FFT noise bound is
            # the fallback diagnostic, not the bar)
            assert (got[g] == ref).all(), (steps, g, int(np.abs(sdiff(got[g], ref)).max()), float((got[g] == ref).mean()))


def test_bootstrap_against_exact_oracle(gpu_ctx, oracle, okeys):
    lwe_sk, glwe_sk, bsk, ksk = okeys
    bits = np.array([0, 1], np.uint8)
    cts = oracle.encrypt_bits(lwe_sk, 78, 0, bits)
    got = gpu_ctx.bootstrap_batch(cts)
    for g in range(2):
        ref = oracle.pbs(bsk, None, cts[g])
        d = (oracle.phase(glwe_sk, got[g]) - oracle.phase(glwe_sk, ref) + 2**31) % 2**32 - 2**31
        assert abs(d) <= FFT_NOISE_BOUND
        assert (got[g] == ref).all(), "expected bit-identical to the exact engine"
        assert (oracle.phase(glwe_sk, got[g]) < 2**31) == bool(bits[g])


def test_bootstrap_batch_against_fft_oracle(gpu_ctx, oracle, okeys, obsk_f):
    lwe_sk, glwe_sk, bsk, ksk = okeys
    bits = np.random.default_rng(9).integers(0, 2, 48).astype(np.uint8)
    cts = oracle.encrypt_bits(lwe_sk, 79, 0, bits)
    got = gpu_ctx.bootstrap_batch(cts)
    identical = 0
    for g in range(48):
        ref = oracle.pbs(bsk, obsk_f, cts[g])
        d = (oracle.phase(glwe_sk, got[g]) - oracle.phase(glwe_sk, ref) + 2**31) % 2**32 - 2**31
        assert abs(d) <= FFT_NOISE_BOUND
        identical += int((got[g] == ref).all())
        # refreshed noise: the output sits within 2^-6 of +-1/8
        ph = oracle.phase(glwe_sk, got[g])
        centre = (1 << 29) if bits[g] else (7 << 29)
        assert abs((ph - centre + 2**31) % 2**32 - 2**31) < 2**26
    assert identical >= 46


def test_every_kernel_variant_bit_exact(variant_ctx, oracle, okeys):
    """Each mapping of the bootstrap onto the GPU (chosen per launch from the stage width in production)
    against the exact-integer oracle: accumulator after a few external products, whole bootstraps on a
    batch that leaves the last workgroup partly idle, and agreement between gates run through the pool."""
    lwe_sk, glwe_sk, bsk, ksk = okeys
    bits = np.random.default_rng(21).integers(0, 2, 13).astype(np.uint8)
    cts = oracle.encrypt_bits(lwe_sk, 90, 0, bits)
    cts[4, 7] = 0
    for steps in (0, 1, 3):
        got = variant_ctx.blind_rotate_batch(cts, steps)
        for g in (0, 4, 12):
            assert (got[g] == oracle.blind_rotate(bsk, None, cts[g], steps)).all(), (steps, g)
    got = variant_ctx.bootstrap_batch(cts)
    for g in (0, 5, 12):
        assert (got[g] == oracle.pbs(bsk, None, cts[g])).all()
    for g in range(13):
        assert (oracle.phase(glwe_sk, got[g]) < 2**31) == bool(bits[g])
    # 600 gates through the pool: wider than one workgroup per CU for the latency variants
    G = 600
    rng = np.random.default_rng(22)
    a, b, c = (rng.integers(0, 2, G).astype(np.uint8) for _ in range(3))
    variant_ctx.pool_reserve(4 * G)
    for k, x in enumerate((a, b, c)):
        variant_ctx.pool_write(np.arange(k * G, (k + 1) * G), oracle.encrypt_bits(lwe_sk, 91, k, x))
    ops = rng.integers(0, 7, G)
    variant_ctx.stage_submit(E.make_gates(ops, np.arange(G), np.arange(G, 2 * G), np.arange(2 * G, 3 * G), np.arange(3 * G, 4 * G)))
    variant_ctx.sync()
    out = variant_ctx.pool_read(np.arange(3 * G, 4 * G))
    assert (oracle.decrypt_bits(lwe_sk, out) == clear_gate(ops, a, b, c)).all()


def test_batch_split_into_full_rounds_and_tail(gpu_ctx, oracle, okeys):
    """A batch just over a whole round of the 8-wave kernel (8 bootstraps x 256 CUs) runs as that round plus a
    separately mapped tail: outputs on both sides of the cut, for the extracted LWE and for the raw accumulator."""
    lwe_sk, glwe_sk, bsk, ksk = okeys
    count = 2048 + 37
    bits = np.random.default_rng(31).integers(0, 2, count).astype(np.uint8)
    cts = oracle.encrypt_bits(lwe_sk, 95, 0, bits)
    got = gpu_ctx.bootstrap_batch(cts)
    for g in (0, 2047, 2048, count - 1):
        assert (got[g] == oracle.pbs(bsk, None, cts[g])).all(), g
    assert ((np.array([oracle.phase(glwe_sk, x) for x in got[2040:2060]]) < 2**31) == bits[2040:2060].astype(bool)).all()
    acc = gpu_ctx.blind_rotate_batch(cts, 2)
    for g in (2047, 2048, count - 1):
        assert (acc[g] == oracle.blind_rotate(bsk, None, cts[g], 2)).all(), g


def clear_gate(op, a, b, c):
    return np.select([op == 0, op == 1, op == 2, op == 3, op == 4, op == 5, op == 6],
                     [a & b, 1 - (a & b), a | b, 1 - (a | b), a ^ b, 1 - (a ^ b), np.where(a == 1, b, c)]).astype(np.uint8)


def test_gates_match_the_committed_golden_fixture(gpu_ctx, oracle, okeys):
    """tests/golden/tfhe_golden.json (eight gates of all kinds, made by tests/golden/make_golden.py with the exact
    engine): the GPU's output ciphertexts hash to the committed value, word for word."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tfhe_golden.json")))
    lwe_sk = okeys[0]
    bits = np.array(gold["input_bits"], np.uint8).reshape(3, -1)
    ins = [oracle.encrypt_bits(lwe_sk, gold["enc_seed"], s, bits[s]) for s in range(3)]  # inputs: the fixture's streams
    assert hashlib.sha256(np.stack(ins).tobytes()).hexdigest() == gold["inputs_sha256"]
    G = bits.shape[1]
    gpu_ctx.pool_reserve(4 * G)
    for k, ct in enumerate(ins):
        gpu_ctx.pool_write(np.arange(k * G, (k + 1) * G), ct)
    gpu_ctx.stage_submit(E.make_gates(np.array(gold["ops"]), np.arange(G), np.arange(G, 2 * G), np.arange(2 * G, 3 * G), np.arange(3 * G, 4 * G)))
    gpu_ctx.sync()
    out = gpu_ctx.pool_read(np.arange(3 * G, 4 * G))
    assert [int(x) for x in out[:, -1]] == gold["output_bodies"]
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == gold["outputs_sha256"]
    assert list(E.client_decrypt(E.default_params(), lwe_sk, out)) == gold["output_bits"]


def test_mixed_gates_with_negations(gpu_ctx, oracle, okeys, obsk_f):
    lwe_sk, _, bsk, ksk = okeys
    G = 2048
    rng = np.random.default_rng(10)
    a, b, c = (rng.integers(0, 2, G).astype(np.uint8) for _ in range(3))
    ca, cb, cc = (oracle.encrypt_bits(lwe_sk, 80, s, x) for s, x in enumerate((a, b, c)))
    gpu_ctx.pool_reserve(4 * G)
    gpu_ctx.pool_write(np.arange(G), ca)
    gpu_ctx.pool_write(np.arange(G, 2 * G), cb)
    gpu_ctx.pool_write(np.arange(2 * G, 3 * G), cc)
    ops, neg = rng.integers(0, 7, G), rng.integers(0, 8, G)
    gpu_ctx.stage_submit(E.make_gates(ops, np.arange(G), np.arange(G, 2 * G), np.arange(2 * G, 3 * G), np.arange(3 * G, 4 * G), neg))
    gpu_ctx.sync()
    out = gpu_ctx.pool_read(np.arange(3 * G, 4 * G))
    exp = clear_gate(ops, a ^ (neg & 1), b ^ ((neg >> 1) & 1), c ^ ((neg >> 2) & 1))
    assert (oracle.decrypt_bits(lwe_sk, out) == exp).all()
    # word-for-word against the EXACT-integer oracle on a sample (negation = LWE negation of the operand): every
    # sampled row identical.  (The oracle's own fp64-FFT engine is a different rounding sequence and may differ from
    # the exact engine in a digit now and then; the GPU is held to the exact one.)
    idx = np.arange(0, G, G // 16)[:16]
    flip = lambda ct, m: np.where(m[:, None] == 1, (0 - ct.astype(np.int64)) % 2**32, ct).astype(np.uint32)
    ref = oracle.gates(bsk, None, ksk, ops[idx].astype(np.uint8), flip(ca[idx], neg[idx] & 1), flip(cb[idx], (neg[idx] >> 1) & 1),
                       flip(cc[idx], (neg[idx] >> 2) & 1))
    same = (ref == out[idx]).all(axis=1)
    diag = [int((oracle.phase(lwe_sk, ref[i]) - oracle.phase(lwe_sk, out[idx[i]]) + 2**31) % 2**32 - 2**31) for i in np.nonzero(~same)[0]]
    assert same.all(), f"rows differing from the exact oracle: {np.nonzero(~same)[0].tolist()}, phase differences {diag} (FFT noise bound {FFT_NOISE_BOUND})"


def test_not_copy_and_chained_stages(gpu_ctx, oracle, okeys):
    lwe_sk = okeys[0]
    bits = np.array([1, 0, 1, 1], np.uint8)
    gpu_ctx.pool_reserve(64)
    gpu_ctx.pool_write(np.arange(4), oracle.encrypt_bits(lwe_sk, 81, 0, bits))
    g1 = E.make_gates([E.NOT, E.COPY, E.XOR], [0, 1, 2], [0, 0, 3], None, [10, 11, 12])
    g2 = E.make_gates([E.AND, E.MUX], [10, 12], [11, 10], [0, 11], [20, 21])  # stage 2 consumes stage 1
    gpu_ctx.stage_submit(g1)
    gpu_ctx.stage_submit(g2)
    gpu_ctx.sync()
    got = oracle.decrypt_bits(lwe_sk, gpu_ctx.pool_read([10, 11, 12, 20, 21]))
    n0, c1, x23 = 1 - bits[0], bits[1], bits[2] ^ bits[3]
    assert list(got) == [n0, c1, x23, n0 & c1, (n0 if x23 else c1)]
    raw = gpu_ctx.pool_read([0, 10])
    assert (raw[1] == (0 - raw[0].astype(np.int64)) % 2**32).all()  # NOT is an exact LWE negation


def test_in_place_gates_read_the_pool_as_it_was_before_the_stage(gpu_ctx, oracle, okeys):
    """The *_assign forms (primitive.rs:96-133: xor_assign / and_assign / or_assign on the left operand): a gate may
    write a slot that it or another gate of the same stage reads; all of them see the value from before the stage."""
    lwe_sk = okeys[0]
    bits = np.array([1, 0, 1, 1, 0, 1], np.uint8)
    gpu_ctx.pool_reserve(32)
    gpu_ctx.pool_write(np.arange(6), oracle.encrypt_bits(lwe_sk, 83, 0, bits))
    g = E.make_gates([E.XOR, E.AND, E.OR, E.MUX, E.NOT, E.COPY],
                     [0, 0, 2, 3, 5, 1],       # in0 (MUX: selector)
                     [1, 1, 0, 0, 0, 0],       # in1
                     [0, 0, 0, 4, 0, 0],       # in2
                     [0, 1, 2, 3, 5, 10])      # out: gates 0-3 overwrite operands that later gates of the stage read
    gpu_ctx.stage_submit(g)
    gpu_ctx.sync()
    got = oracle.decrypt_bits(lwe_sk, gpu_ctx.pool_read([0, 1, 2, 3, 5, 10]))
    b = bits
    assert list(got) == [b[0] ^ b[1], b[0] & b[1], b[2] | b[0], (b[0] if b[3] else b[4]), 1 - b[5], b[1]]


def test_stage_hazards_are_rejected(gpu_ctx, oracle, okeys):
    """Two gates writing one slot, and a NOT/COPY writing a slot that another NOT/COPY of the stage reads, are refused
    before anything is launched (the pool keeps its contents); the legal neighbours of those cases still run."""
    lwe_sk = okeys[0]
    bits = np.array([1, 0, 1, 0], np.uint8)
    gpu_ctx.pool_reserve(32)
    gpu_ctx.pool_write(np.arange(4), oracle.encrypt_bits(lwe_sk, 84, 0, bits))
    before = gpu_ctx.pool_read(np.arange(12))
    with pytest.raises(E.TfaError, match="two gates of one stage write slot 8"):
        gpu_ctx.stage_submit(E.make_gates([E.AND, E.XOR], [0, 1], [1, 2], None, [8, 8]))
    with pytest.raises(E.TfaError, match="two gates of one stage write slot 9"):
        gpu_ctx.stage_submit(E.make_gates([E.MUX, E.COPY], [0, 1], [1, 0], [2, 0], [9, 9]))
    with pytest.raises(E.TfaError, match="another NOT/COPY of the stage reads"):
        gpu_ctx.stage_submit(E.make_gates([E.NOT, E.COPY], [0, 8], [0, 0], None, [8, 9]))      # chained: COPY reads what NOT writes
    with pytest.raises(E.TfaError, match="another NOT/COPY of the stage reads"):
        gpu_ctx.stage_submit(E.make_gates([E.NOT, E.COPY], [0, 0], [0, 0], None, [0, 9]))      # in place under a second reader
    with pytest.raises(E.TfaError, match="another NOT/COPY of the stage reads"):
        gpu_ctx.stage_submit_noalloc(E.make_gates([E.COPY, E.NOT], [1, 2], [0, 0], None, [2, 1]))  # a swap through NOT/COPY
    gpu_ctx.sync()
    assert (gpu_ctx.pool_read(np.arange(12)) == before).all()
    # legal: NOT in place (sole reader), a bootstrapped gate reading a slot that a NOT of the same stage overwrites
    gpu_ctx.stage_submit(E.make_gates([E.NOT, E.AND, E.COPY], [0, 0, 1], [0, 2, 0], None, [0, 8, 9]))
    gpu_ctx.sync()
    got = oracle.decrypt_bits(lwe_sk, gpu_ctx.pool_read([0, 8, 9]))
    assert list(got) == [1 - bits[0], bits[0] & bits[2], bits[1]]


def test_keyswitch_other_parameters_vector_path(oracle):
    """The vector keyswitch is generic in (n, base, levels): n = 700 (not 805), 2^4 x 4 levels (not 2^3 x 5), where
    the matrix-core kernel does not apply, at every tile width, against the oracle under the same parameters; and a
    dimension beyond the kernel's 806 output columns is refused at context creation."""
    from oracle.pyoracle import Oracle
    o = Oracle()
    o.params.lwe_dimension, o.params.ks_base_log, o.params.ks_level = 700, 4, 4
    keys = o.keygen(0x1234)
    p = E.default_params()
    p.lwe_dimension, p.ks_base_log, p.ks_level = 700, 4, 4
    ctx = E.Context(p, 0)
    try:
        ctx.keys_upload(keys[2], keys[3])
        rng = np.random.default_rng(6)
        for count in (1, 37, 300, 2500, 8200):
            big = rng.integers(0, 2**32, (count, ctx.big + 1), dtype=np.uint64).astype(np.uint32)
            big[0, :4] = (0, 0xFFFFFFFF, 0x80000000, 0x00008000)   # the last: a rounding tie at 16 representable bits
            got = ctx.keyswitch_batch(big)
            assert got.shape == (count, 701)
            for i in (0, count // 2, count - 1):
                assert (got[i] == o.keyswitch(keys[3], big[i])).all(), (count, i)
    finally:
        ctx.close()
    p.lwe_dimension = 806
    with pytest.raises(E.TfaError):
        E.Context(p, 0)


def test_two_contexts_with_different_keys_run_interleaved(gpu_ctx, oracle, okeys):
    """Two contexts on one GPU (a caller serving two key sets): submits interleave on their own streams and no state
    is shared, so each decrypts under its own key only."""
    other_keys = oracle.keygen(0x5446484541455321 + 17)
    other = E.Context(E.default_params(), 0)
    try:
        other.keys_upload(other_keys[2], other_keys[3])
        rng = np.random.default_rng(5)
        n = 300  # a width that takes the two-workgroups-per-CU mapping
        expect, ctxs = [], [(gpu_ctx, okeys[0]), (other, other_keys[0])]
        for seed, (c, sk) in enumerate(ctxs):
            bits = rng.integers(0, 2, 2 * n).astype(np.uint8)
            c.pool_reserve(3 * n)
            c.pool_write(np.arange(2 * n), oracle.encrypt_bits(sk, 90 + seed, 0, bits))
            expect.append(bits[:n] & bits[n:])
        gates = E.make_gates(np.full(n, E.AND), np.arange(n), np.arange(n, 2 * n), None, np.arange(2 * n, 3 * n))
        for _ in range(2):  # both streams busy at once, twice (the second pass overwrites with the same results)
            for c, _sk in ctxs:
                c.stage_submit(gates)
        for (c, sk), exp in zip(ctxs, expect):
            c.sync()
            assert (oracle.decrypt_bits(sk, c.pool_read(np.arange(2 * n, 3 * n))) == exp).all()
    finally:
        other.close()


def test_error_paths(gpu_ctx, okeys):
    lib = E.load_library()
    fresh = E.Context(E.default_params(), 0)
    g = E.make_gates([E.AND], [0], [1], None, [2])
    with pytest.raises(E.TfaError, match="before tfa_keys_upload"):
        fresh.stage_submit(g)
    fresh.close()
    gpu_ctx.pool_reserve(16)
    bad = E.make_gates([E.AND], [0], [1], None, [2**31])
    with pytest.raises(E.TfaError, match="out of range"):
        gpu_ctx.stage_submit(bad)
    with pytest.raises(E.TfaError, match="opcode"):
        gpu_ctx.stage_submit(E.make_gates([99], [0], [1], None, [2]))
    gpu_ctx.stage_submit(np.zeros(0, E.GATE_DTYPE))  # empty stage is a no-op
    assert gpu_ctx.pool_read(np.zeros(0, np.uint32)).shape == (0, gpu_ctx.n + 1)
    assert lib.tfa_pool_write(gpu_ctx._ctx, None, 0, None) == -1


@pytest.mark.timeout(300)
def test_config2_65536_nand_xor_gates(gpu_ctx, oracle, okeys):
    """BASELINE config 2: 65536 independent NAND/XOR gates of fresh encryptions, one stage submit."""
    lwe_sk = okeys[0]
    G = 65536
    bits = np.random.default_rng(11).integers(0, 2, 2 * G).astype(np.uint8)
    cts = E.client_encrypt(E.default_params(), lwe_sk, 4242, bits)
    gpu_ctx.pool_reserve(3 * G)
    gpu_ctx.pool_write(np.arange(2 * G), cts)
    even = np.arange(G) % 2 == 0
    gpu_ctx.stage_submit(E.make_gates(np.where(even, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G)))
    gpu_ctx.sync()
    got = oracle.decrypt_bits(lwe_sk, gpu_ctx.pool_read(np.arange(2 * G, 3 * G)))
    x, y = bits[:G], bits[G:]
    assert (got == np.where(even, 1 - (x & y), x ^ y)).all()


@pytest.mark.timeout(600)
def test_zero_decryption_failures_over_a_million_gates(gpu_ctx, oracle, okeys):
    """16 x 65536 gates, each round feeding on the previous round's outputs (so bootstrapped, not
    only fresh, noise is exercised); every one of the 1,048,576 results must decrypt correctly."""
    lwe_sk = okeys[0]
    G = 65536
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 2, 2 * G).astype(np.uint8)
    gpu_ctx.pool_reserve(3 * G)
    gpu_ctx.pool_write(np.arange(2 * G), E.client_encrypt(E.default_params(), lwe_sk, 4343, bits))
    x, y = bits[:G].copy(), bits[G:].copy()
    failures = 0
    src_a, src_b, dst = 0, G, 2 * G
    for rnd in range(16):
        ops = rng.integers(0, 6, G)
        perm = rng.permutation(G)
        gpu_ctx.stage_submit(E.make_gates(ops, src_a + np.arange(G), src_b + perm, None, dst + np.arange(G)))
        gpu_ctx.sync()
        exp = clear_gate(ops, x, y[perm], y)
        got = oracle.decrypt_bits(lwe_sk, gpu_ctx.pool_read(dst + np.arange(G)))
        failures += int((got != exp).sum())
        x, y = exp, x  # next round: a = this output, b = previous a
        src_a, src_b, dst = dst, src_a, src_b
    assert failures == 0


@pytest.mark.gpu
def test_profile_launches_records_every_timed_launch(gpu_ctx, oracle, okeys):
    """tfa_profile_launches: one record per bootstrap / keyswitch launch of the stages submitted while profiling is on, in
    order, with the stage's unit counts (a MUX is two bootstraps and one keyswitch) and the mapping that ran it; cleared by
    tfa_reset_stats; nothing is recorded while profiling is off."""
    from tfhe_aes_amd import engine as E
    ctx, p = gpu_ctx, E.default_params()
    lwe_sk = okeys[0]
    widths = [100, 300, 1000, 2500]
    G = max(widths)
    ctx.pool_reserve(4 * G)
    bits = np.random.default_rng(5).integers(0, 2, 3 * G).astype(np.uint8)
    ctx.pool_write(np.arange(3 * G), oracle.encrypt_bits(lwe_sk, 41, 0, bits))
    ctx.reset_stats()
    ctx.set_profiling(True)
    for w in widths:
        ops = np.where(np.arange(w) % 2 == 0, E.MUX, E.XOR)
        ctx.stage_submit(E.make_gates(ops, np.arange(w), np.arange(G, G + w), np.arange(2 * G, 2 * G + w), np.arange(3 * G, 3 * G + w)))
    rec = ctx.profile_launches()
    ctx.set_profiling(False)
    pbs, ks = rec[rec["kind"] == 0], rec[rec["kind"] == 1]
    assert list(ks["units"]) == widths
    assert list(pbs["units"]) == [w + (w + 1) // 2 for w in widths]
    assert (rec["ms"] > 0).all()
    # mapping: 1 / 2 = one bootstrap per workgroup (narrow stages), 4 / 8 = one per wave, 8 | tail << 8 = full rounds plus a tail
    assert all((m & 0xFF) in (1, 2, 4, 8) for m in pbs["mapping"]) and (pbs["mapping"][0] & 0xFF) in (1, 2)
    st = ctx.stats()
    assert st.pbs_launches == len(widths) and abs(st.pbs_kernel_ms - float(pbs["ms"].sum())) < 1e-3 * st.pbs_kernel_ms + 1e-3
    ctx.stage_submit(E.make_gates([E.XOR], [0], [1], None, [3 * G]))      # profiling off: no new record
    assert len(ctx.profile_launches()) == len(rec)
    ctx.reset_stats()
    assert len(ctx.profile_launches()) == 0
