"""The bootstrap kernels extract both decomposition digits of (sign * v - acc) with a shortened integer sequence
(rotated_digits in tfhe-aes_amd/csrc/pbs_kernel.h).  This restates that sequence in numpy and checks it against the
digit rule written out literally (closest representable, then digits from the least significant end with
carry = res > B/2 or (res == B/2 and bit 9 of the remaining state), the rule the CPU oracle implements in
oracle/tfhe_oracle.c: orc_decompose) on random inputs and on every rounding / carry boundary."""
import numpy as np

B_LOG, LEVELS = 10, 2


def literal_rule(x):
    x = x.astype(np.uint64)
    shift = 32 - B_LOG * LEVELS
    state = ((x >> np.uint64(shift - 1)) + np.uint64(1)) >> np.uint64(1)
    state &= np.uint64((1 << (B_LOG * LEVELS)) - 1)
    digits = []
    for _ in range(LEVELS):
        res = state & np.uint64(1023)
        state = state >> np.uint64(B_LOG)
        carry = ((((res - np.uint64(1)) | state) & res) >> np.uint64(B_LOG - 1)) & np.uint64(1)  # tfhe-rs: ((res - 1) | state) & res, bit 9
        state = state + carry
        digits.append(res.astype(np.int64) - (carry.astype(np.int64) << B_LOG))
    return digits[1], digits[0]  # (d1 pairs with q / 2^10, d2 with q / 2^20)


def kernel_sequence(v, jj, acc):
    v, acc = v.astype(np.uint32), acc.astype(np.uint32)
    neg = np.where((jj >> 9) & 1, np.uint32(0xFFFFFFFF), np.uint32(0))
    with np.errstate(over="ignore"):
        y = (v ^ neg) - (acc + neg - np.uint32(2048))
        s, b19 = y >> np.uint32(12), y >> np.uint32(31)
        u = s + np.uint32(511) + b19
        d2 = (s - (u & np.uint32(0xFFFFFC00))).astype(np.int32)
        d1 = (((u + np.uint32(511 << 10)) >> np.uint32(10)) & np.uint32(1023)).astype(np.int32) - 511
    return d1.astype(np.int64), d2.astype(np.int64)


def asm_sequence(v, r, m, acc):
    """the narrow-stage loops' version (tools/pbs_asm/lat_gen.py; the wide loop's until it took fp_sequence above, and still
    its rotation addressing and sign handling: rotate_decompose in tools/pbs_asm/gen.py): negated accumulator in registers and
    rotation buffer, wrap / sign taken from one mask per lane; r = t - a, m = coefficient group, jj = r + 16 m"""
    n_src, n_own = np.uint32(0) - v.astype(np.uint32), np.uint32(0) - acc.astype(np.uint32)
    r = r.astype(np.uint32)
    with np.errstate(over="ignore"):
        e = ((~r) >> np.uint32(4)) & np.uint32(31)
        fw = np.uint32(0xFFFFFFFE) << e                                   # bit m: the index wrapped past 511
        s0 = np.where((r >> np.uint32(9)) & np.uint32(1), np.uint32(0xFFFFFFFF), np.uint32(0))
        fs = ~(fw ^ s0)
        wrapped = (fw >> m.astype(np.uint32)) & np.uint32(1)
        idx = (r & np.uint32(511)) + np.uint32(16) * m.astype(np.uint32) - np.uint32(512) * wrapped
        sg = np.where((fs >> m.astype(np.uint32)) & np.uint32(1), np.uint32(0xFFFFFFFF), np.uint32(0))
        c = (sg & np.uint32(1)) + n_own + np.uint32(2048)
        y = (n_src ^ sg) + c
        s, b19 = y >> np.uint32(12), y >> np.uint32(31)
        u = s + b19 + np.uint32(511)
        f = ((np.uint32(1023) - u) >> np.uint32(10)) & np.uint32(1023)    # the loop produces -d1 and negates in the products
        nd1 = np.where(f >> np.uint32(9), f | np.uint32(0xFFFFFC00), f).astype(np.int32)
        d2 = (s - (u & np.uint32(0xFFFFFC00))).astype(np.int32)
    return -nd1.astype(np.int64), d2.astype(np.int64), idx


def fp_sequence(x):
    """the hand-scheduled wide loop's digits ("fp_digits" in tools/pbs_asm/gen.py): the biased, masked difference becomes the
    low mantissa word of the double 2^40 + w 2^-12, V = that double - (2^40 + 2^19 - 2^9) is the rounded difference in units
    of 2^12 (an integer in [-2^19 + 512, 2^19 + 512)), t = V (1 - 2^-21) / 1024 + 1.5 * 2^52 rounds V / 1024 to the nearest
    integer with ties toward zero (one IEEE rounding), and the loop keeps -d1 = 1.5 * 2^52 - t and d2 = V + 1024 (-d1);
    x = sign * v - acc"""
    with np.errstate(over="ignore"):
        w = (x.astype(np.uint32) + np.uint32((0x800 + 0x80000000 - 0x200000) & 0xFFFFFFFF)) & np.uint32(0xFFFFF000)
    d = (np.uint64(0x4270000000000000) | w.astype(np.uint64)).view(np.float64)
    v = d - (2.0 ** 40 + 2.0 ** 19 - 2.0 ** 9)
    magic = 1.5 * 2.0 ** 52
    t = v * ((1.0 - 2.0 ** -21) / 1024.0) + magic     # the product is exact (42 significant bits), so this is the fused result
    neg_d1 = magic - t
    d2 = v + 1024.0 * neg_d1
    assert (v == np.rint(v)).all() and (neg_d1 == np.rint(neg_d1)).all()
    return (-neg_d1).astype(np.int64), d2.astype(np.int64)


def test_fp_sequence_every_state():
    """all 2^20 rounded states (with the discarded low bits at their extremes and around the rounding point), random words"""
    hi = np.arange(1 << 20, dtype=np.uint64) << np.uint64(12)
    for low in (0, 1, 0x7FF, 0x800, 0x801, 0xFFF):
        x = ((hi + np.uint64(low)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        e1, e2 = literal_rule(x)
        g1, g2 = fp_sequence(x)
        assert (e1 == g1).all() and (e2 == g2).all()
        assert g1.min() == -511 and g1.max() == 512 and g2.min() == -512 and g2.max() == 512
    x = np.random.default_rng(13).integers(0, 1 << 32, 1 << 21, dtype=np.uint64).astype(np.uint32)
    e1, e2 = literal_rule(x)
    g1, g2 = fp_sequence(x)
    assert (e1 == g1).all() and (e2 == g2).all()


def test_asm_sequence():
    rng = np.random.default_rng(12)
    n = 1 << 21
    v = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    acc = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    r = rng.integers(-1023, 16, n).astype(np.int32)                       # t - a, t in [0, 16), a in [0, 1024)
    m = rng.integers(0, 32, n).astype(np.int32)
    for rr, mm in ((r, m), (np.repeat(np.arange(-1023, 16, dtype=np.int32), 32), np.tile(np.arange(32, dtype=np.int32), 1039))):
        k = len(rr)
        jj = rr + 16 * mm
        with np.errstate(over="ignore"):
            x = np.where((jj >> 9) & 1, np.uint32(0) - v[:k], v[:k]) - acc[:k]
        e1, e2 = literal_rule(x)
        g1, g2, idx = asm_sequence(v[:k], rr, mm, acc[:k])
        assert (e1 == g1).all() and (e2 == g2).all()
        assert (idx == (jj & 511).astype(np.uint32)).all()                # the address it reads is the rotated index


def check(v, jj, acc):
    with np.errstate(over="ignore"):
        x = np.where((jj >> 9) & 1, np.uint32(0) - v, v) - acc
    e1, e2 = literal_rule(x)
    g1, g2 = kernel_sequence(v, jj, acc)
    assert (e1 == g1).all() and (e2 == g2).all()
    # the digits recompose to the closest multiple of q / 2^20, mod 2^32
    closest = ((x.astype(np.uint64) + np.uint64(2048)) >> np.uint64(12) << np.uint64(12)) & np.uint64(0xFFFFFFFF)
    assert (((g1 << 22) + (g2 << 12)) % (1 << 32) == closest.astype(np.int64)).all()
    assert (np.abs(g1) <= 512).all() and (np.abs(g2) <= 512).all()


def test_random_inputs():
    rng = np.random.default_rng(11)
    n = 1 << 21
    check(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), rng.integers(-1023, 513, n).astype(np.int32),
          rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))


def test_boundaries():
    hi = np.arange(1 << 12, dtype=np.uint64)[:, None, None] << np.uint64(20)
    mid = np.array([0, 0x7FF, 0x800, 0x801, 0xFFF, 0x80000, 0x7FFFF, 0x80800, 0x7F800, 0xFF800], dtype=np.uint64)[None, :, None]
    lo = np.arange(-3, 4, dtype=np.int64).astype(np.uint64)[None, None, :]
    x = ((hi + mid + lo) & np.uint64(0xFFFFFFFF)).astype(np.uint32).ravel()
    zero = np.zeros_like(x)
    check(x, np.full(x.shape, 5, np.int32), zero)
    with np.errstate(over="ignore"):
        check(np.uint32(0) - x, np.full(x.shape, 517, np.int32), zero)       # negated operand
        check(x + np.uint32(77), np.full(x.shape, -700, np.int32), zero + np.uint32(77))
    edge = np.concatenate([np.arange(0, 0x3000, dtype=np.uint64), np.arange(0xFFFFD000, 1 << 32, dtype=np.uint64)]).astype(np.uint32)
    check(edge, np.full(edge.shape, 3, np.int32), np.zeros_like(edge))
