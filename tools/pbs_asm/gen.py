#!/usr/bin/env python3
"""Generator of the hand-scheduled blind-rotation loop (gfx950 assembly) of pbs_asm_kernel.

Emits tfhe-aes_amd/csrc/pbs_asm_loop.inc: one inline-asm body holding the whole step loop of the 8-wave bootstrap kernel
(rotation, signed decomposition, two forward negacyclic transforms, multiply-accumulate with the key ring, inverse
transform, rounding), with every register chosen here and every LDS wait derived by place_waits().  The arithmetic is that of
csrc/pbs_kernel.h (same lane mapping, key layout and twiddle table; its own LDS layout: K_* below), with the constant twiddle
scalings folded into fused multiply-adds, the accumulator kept negated and the rotation buffer interleaved by polynomial.  tools/pbs_asm/emu.py executes the emitted text on the CPU against the exact oracle.

Register map (VGPR):  v2 rotation buffer of this wave + 4 g | v3 t | v4-7 exchange store addresses |
v8-11 exchange row-read addresses | v16 twiddle row address | v17 lane*16 | v18 key read address | v19-31 scratch |
v32-63 accumulator (negated) | v64-127 array A (level-1 data, then the output spectrum) | v128-191 array B (level-2 data; the
inverse transform's twiddles) | v192-223 key words / twiddles / rotated coefficients | v224-255 temporaries.
v0 and s0-s7 are left to the compiler.  Arrays keep "home slots": element k of array X has its real part in
X + 2 k and its imaginary part in X + 32 + 2 k; butterflies read their inputs from home slots and write their outputs
back to the slots they consumed, so only intermediates use the temporaries and no register move is ever needed.
"""
import math
import os
import sys

# ---- geometry shared with pbs_kernel.h (PbsGeom<8,1>, quad lanes, swizzled exchange buffer) ----
K_TR_GROUP = 260
K_WAVE_LDS = 8320
K_A_BYTES = 1664
WAVES = int(os.environ.get("PBS_ASM_WAVES", "8"))     # bootstraps per workgroup: 8 (two waves per SIMD) or 4 (one per SIMD, for 1024-wide stages)
assert WAVES in (4, 8)
ROWS_PER_WAVE = 16 // WAVES                          # LDS-DMA rows of a key chunk each wave fetches
# schedule choices that can be switched off for measurement (PBS_ASM_OPT=comma list replaces the default set)
OPT = set(f for f in os.environ.get("PBS_ASM_OPT", "mac_il,tw_spread,inv_split,lanesum,ls_pf2,wait_merge,dma_off,fp_digits,ring2").split(",") if f)
# "lanesum": the sum over the four polynomials stays inside a lane.  After the second forward pass the four lanes of a quad
# exchange frequency groups (in-place DPP moves) so that lane x holds, for ALL four polynomials and both levels, the four bins
# whose high index digit is x; the products then accumulate in registers (no reduce-scatter of partial sums: 48 DPP moves and
# 96 additions fewer per step), and a second exchange hands output column c to the lanes of polynomial c.  A lane's second-pass
# transform must deliver its bins rotated by its polynomial number for that: the rotation is a factor i^(g t) on the inputs,
# folded into the inter-pass twiddles, which the table therefore holds in all four quarter-turn rotations.
LANESUM = "lanesum" in OPT
# "ring2": a chunk's barrier is followed by the load of the chunk TWO ahead, into the slot of the chunk two back (whose every
# read has been consumed by then), instead of three ahead into the slot of the previous chunk: the key reads still in flight at
# a barrier (the previous chunk's last words) then need not be drained in front of it.
RING_AHEAD = 2 if "ring2" in OPT else 3
FP_DIGITS = "fp_digits" in OPT
K_TW_OFFSET = WAVES * (K_WAVE_LDS + K_A_BYTES)   # [exchange / rotation buffers][a_tilde arrays][twiddles][key ring]
K_TW_ROW_BYTES = 1024 if LANESUM else 256       # twiddle table [element][t]: 16 bytes each, or 4 rotations x 16 bytes
K_RING_OFFSET = K_TW_OFFSET + 16 * K_TW_ROW_BYTES
K_CHUNK_BYTES = 16384
K_LDS_BYTES = K_RING_OFFSET + 4 * K_CHUNK_BYTES
STEP_KEY_BYTES = 131072
MAGIC = 6755399441055744.0  # 1.5 * 2^52
# "fp_digits": both digits of a coefficient come out of four fp64 instructions instead of seven integer ones and two
# conversions.  x = (sign * acc[src] - acc + 2^11 + 2^31 - 2^21) & ~0xfff is written into the low word of a register pair whose
# high word holds 0x42700000, so the pair reads as the double 2^40 + x 2^-12; V = pair - (2^40 + 2^19 - 2^9) is the rounded
# difference in units of 2^12, an integer in [-2^19 + 512, 2^19 + 512), the window in which the top digit is the tfhe-rs one,
# [-511, 512].  The low digit's rule (carry when the residue exceeds B/2, or equals B/2 and the state's top bit is set) is
# "V / 1024 to the nearest integer, ties toward zero": t = V (1 - 2^-21) / 1024 + 1.5 * 2^52 (exact product, one rounding at a
# unit of 1; the factor moves a tie by at least 2^-22 and nothing by 2^-10 or more), -d1 = 1.5 * 2^52 - t, d2 = V + 1024 (-d1),
# all exact.
DIGIT_BIAS = (0x800 + 0x80000000 - 0x200000) & 0xFFFFFFFF
DIGIT_HI = 0x42700000
DIGIT_K = 2.0 ** 40 + 2.0 ** 19 - 2.0 ** 9
DIGIT_C = (1.0 - 2.0 ** -21) / 1024.0


def kperm(p):
    return 4 * (p & 3) + (p >> 2)


# ---- scalar registers ----
S_KP = 8        # s[8:9]  key pointer of the current step for this wave's DMA rows
S_STEPS = 10    # steps left
S_ACUR = 11     # LDS byte address of a_tilde[i + 1]
S_A = 12        # current rotation amount
S_M0ROW = 13    # LDS byte address of this wave's first DMA row in ring slot 0
S_TMP = 14      # s[14:15] scratch pair (DMA row pointer)
S_C2048 = 16
S_C511 = 17
S_ROTM0 = 100    # LDS byte address of this wave's rotation buffer (for the M0-relative stores)
S_CONST0 = 18   # fp64 constants from here, in pairs


def const_reg(index):
    """first scalar register of constant pair `index`; s[32:33] are left alone (the compiler reserves s32 as stack pointer
    and warns about an asm statement that clobbers it)"""
    r = S_CONST0 + 2 * index
    return r + 2 if r >= 32 else r

V_ROTBASE, V_T = 2, 3          # v2 = rotation buffer + 4 g
V_TRWR = [4, 5, 6, 7]            # exchange store address for element groups r >> 2
V_TRRD = [8, 9, 10, 11]          # exchange row-read address for words (w & 3), offset 64 * (w >> 2)
V_KBASE = 12                     # LDS base + lane * 16: key words are read at v12 + ring offset
V_TW, V_LANE16, V_KADDR = 16, 17, 18
V_TWF = 13                       # lanesum: twiddle row address with this lane's rotation (g t) & 3 added (both directions:
                                 # the inverse multiplies after its exchange, in the lane whose t is the rotated index)
V_AREAD = 19                     # next step's rotation amount (u16 read from LDS)
V_SCR = list(range(20, 30))      # scratch ints; also v[20:21], v[22:23] as fp64 scratch of the twiddle products
V_MAGIC = 30                     # v[30:31] = 1.5 * 2^52
V_ACC = 32
R_A, R_B, R_K, R_T = 64, 128, 192, 224

AUTO_WAITS = os.environ.get("PBS_ASM_MANUAL_WAITS", "") == ""   # waits and DPP wait states placed by place_waits()

INLINE = {1.0: "1.0", 0.5: "0.5", 2.0: "2.0", 4.0: "4.0"}


def dreg(r):
    assert r % 2 == 0
    return "v[%d:%d]" % (r, r + 1)


def qreg(r):
    assert r % 4 == 0
    return "v[%d:%d]" % (r, r + 3)


class Consts:
    def __init__(self):
        self.vals = []
        self.idx = {}

    def get(self, value):
        key = float(value)
        assert key > 0
        if key not in self.idx:
            self.idx[key] = len(self.vals)
            self.vals.append(key)
        r = const_reg(self.idx[key])
        assert r + 1 < S_ROTM0, "out of scalar registers for constants"
        return "s[%d:%d]" % (r, r + 1)


class V:
    """A double in an even-aligned VGPR pair with a pending sign; `temp` marks registers owned by the temporary pool."""
    __slots__ = ("reg", "neg", "temp")

    def __init__(self, reg, neg=False, temp=False):
        self.reg, self.neg, self.temp = reg, neg, temp

    def __neg__(self):
        return V(self.reg, not self.neg, self.temp)

    def s(self, flip=False):
        return ("-" if (self.neg != flip) else "") + dreg(self.reg)


class C:
    __slots__ = ("r", "i")

    def __init__(self, r, i):
        self.r, self.i = r, i

    def times_i(self, s):      # multiply by s * i, s = +-1: renaming only
        return C(-self.i, self.r) if s > 0 else C(self.i, -self.r)


class Pool:
    def __init__(self, regs):
        self.free = list(regs)
        self.low_water = len(self.free)

    def get(self):
        assert self.free, "temporary pool exhausted"
        r = self.free.pop(0)
        self.low_water = min(self.low_water, len(self.free))
        return r

    def put(self, reg):
        assert reg not in self.free
        self.free.append(reg)


class Stream:
    def __init__(self, gen, pool):
        self.g, self.pool, self.ins = gen, pool, []

    def e(self, s):
        self.ins.append(s)

    def cst(self, val):
        val = float(val)
        return INLINE[val] if val in INLINE else self.g.consts.get(val)

    def _dst(self, dst):
        if dst is None:
            return V(self.pool.get(), temp=True)
        return V(dst)

    def kill(self, *vs):
        for v in vs:
            if v.temp:
                self.pool.put(v.reg)

    def add(self, a, b, dst=None):
        d = self._dst(dst)
        self.e("v_add_f64 %s, %s, %s" % (dreg(d.reg), a.s(), b.s()))
        return d

    def mul(self, a, b, dst=None):
        d = self._dst(dst)
        self.e("v_mul_f64 %s, %s, %s" % (dreg(d.reg), a.s(), b.s()))
        return d

    def mulc(self, a, cval, dst=None):
        d = self._dst(dst)
        self.e("v_mul_f64 %s, %s, %s" % (dreg(d.reg), a.s(flip=cval < 0), self.cst(abs(cval))))
        return d

    def fma(self, a, b, c, dst=None):
        d = self._dst(dst)
        self.e("v_fma_f64 %s, %s, %s, %s" % (dreg(d.reg), a.s(), b.s(), c.s()))
        return d

    def fmac(self, cval, b, c, dst=None):   # cval * b + c
        if abs(cval) == 1.0:
            return self.add(b if cval > 0 else -b, c, dst)
        d = self._dst(dst)
        self.e("v_fma_f64 %s, %s, %s, %s" % (dreg(d.reg), b.s(flip=cval < 0), self.cst(abs(cval)), c.s()))
        return d


def rot_tan(st, z, tau, S):
    """z * (1 + i S tau); the imaginary part stays in place, the real part moves to a temporary"""
    t = tau if S > 0 else -tau
    nr = st.fmac(-t, z.i, z.r)
    ni = st.fmac(t, z.r, z.i, dst=(None if z.i.temp else z.i.reg))
    if z.i.temp:
        st.kill(z.i)
    st.kill(z.r)
    return C(nr, ni)


def dft4(st, z, S, out):
    """out: four (re_reg, im_reg) destinations (the home slots of the positions consumed)"""
    z0, z1, z2, z3 = z
    t0 = C(st.add(z0.r, z2.r), st.add(z0.i, z2.i))
    t1 = C(st.add(z0.r, -z2.r), st.add(z0.i, -z2.i))
    st.kill(z0.r, z0.i, z2.r, z2.i)
    t2 = C(st.add(z1.r, z3.r), st.add(z1.i, z3.i))
    d = C(st.add(z1.r, -z3.r), st.add(z1.i, -z3.i))
    st.kill(z1.r, z1.i, z3.r, z3.i)
    t3 = d.times_i(S)
    o0 = C(st.add(t0.r, t2.r, out[0][0]), st.add(t0.i, t2.i, out[0][1]))
    o2 = C(st.add(t0.r, -t2.r, out[2][0]), st.add(t0.i, -t2.i, out[2][1]))
    st.kill(t0.r, t0.i, t2.r, t2.i)
    o1 = C(st.add(t1.r, t3.r, out[1][0]), st.add(t1.i, t3.i, out[1][1]))
    o3 = C(st.add(t1.r, -t3.r, out[3][0]), st.add(t1.i, -t3.i, out[3][1]))
    st.kill(t1.r, t1.i, d.r, d.i)
    return [o0, o1, o2, o3]


def dft4_scaled(st, z, c1, c2, q, S, out, c0=None):
    """DFT4 of (c0) z0, c1 z1, c2 z2, c3 z3 with q = c3 / c1; the scalings ride on the multiply-adds."""
    z0, z1, z2, z3 = z
    if c0 is not None and c0 != 1.0:
        nr, ni = st.mulc(z0.r, c0), st.mulc(z0.i, c0)
        st.kill(z0.r, z0.i)
        z0 = C(nr, ni)
    t0 = C(st.fmac(c2, z2.r, z0.r), st.fmac(c2, z2.i, z0.i))
    t1 = C(st.fmac(-c2, z2.r, z0.r), st.fmac(-c2, z2.i, z0.i))
    st.kill(z0.r, z0.i, z2.r, z2.i)
    t2 = C(st.fmac(q, z3.r, z1.r), st.fmac(q, z3.i, z1.i))
    d = C(st.fmac(-q, z3.r, z1.r), st.fmac(-q, z3.i, z1.i))
    st.kill(z1.r, z1.i, z3.r, z3.i)
    o0 = C(st.fmac(c1, t2.r, t0.r, out[0][0]), st.fmac(c1, t2.i, t0.i, out[0][1]))
    o2 = C(st.fmac(-c1, t2.r, t0.r, out[2][0]), st.fmac(-c1, t2.i, t0.i, out[2][1]))
    st.kill(t0.r, t0.i, t2.r, t2.i)
    sc = c1 if S > 0 else -c1
    o1 = C(st.fmac(-sc, d.i, t1.r, out[1][0]), st.fmac(sc, d.r, t1.i, out[1][1]))       # t1 + S i c1 d
    o3 = C(st.fmac(sc, d.i, t1.r, out[3][0]), st.fmac(-sc, d.r, t1.i, out[3][1]))
    st.kill(t1.r, t1.i, d.r, d.i)
    return [o0, o1, o2, o3]


C1, S1, R = 0.9238795325112867, 0.3826834323650898, 0.7071067811865476
T1, T3 = S1 / C1, C1 / S1
KC32 = [math.cos(math.pi * n / 32) for n in range(16)]
KS32 = [math.sin(math.pi * n / 32) for n in range(16)]


class Arr:
    """A complex array of 16 in home slots of region `base`; pos[k] = slot holding logical element k."""

    def __init__(self, base):
        self.base = base
        self.pos = list(range(16))

    def slot(self, p):
        return (self.base + 2 * p, self.base + 32 + 2 * p)

    def val(self, k):
        r, i = self.slot(self.pos[k])
        return C(V(r), V(i))

    def val_at(self, p):
        """the value in position p, whatever logical element that is"""
        r, i = self.slot(p)
        return C(V(r), V(i))


class Loop:
    def __init__(self):
        self.consts = Consts()
        self.out = []
        self.pool = Pool(range(R_T, R_T + 32, 2))

    def e(self, s):
        self.out.append(s)

    def lds_wait(self, n):
        self.e("s_waitcnt lgkmcnt(%d)" % n)

    # ---- register DFT16 in place (port of dft16m of pbs_kernel.h) ---------------------------------------------------
    def dft16(self, X, S, twist):
        """Transforms array X over its logical index; afterwards logical output k sits in slot 4 (k & 3) + (k >> 2) of
        the slots the inputs occupied."""
        st = Stream(self, self.pool)
        inp = [X.val(k) for k in range(16)]
        slots = [X.slot(X.pos[k]) for k in range(16)]    # slot of logical input k
        u = [None] * 16                                  # u[4 k1 + n2], with the slot it lives in
        uslot = [None] * 16
        for n2 in range(4):
            ns = [n2, 4 + n2, 8 + n2, 12 + n2]
            grp = [inp[n] for n in ns]
            outs = [slots[n] for n in ns]
            if not twist:
                o = dft4(st, grp, S, outs)
            else:
                grp = [rot_tan(st, grp[k], KS32[ns[k]] / KC32[ns[k]], 1) if ns[k] > 0 else grp[k] for k in range(4)]
                o = dft4_scaled(st, grp, KC32[4 + n2], KC32[8 + n2], KC32[12 + n2] / KC32[4 + n2], 1, outs, c0=(KC32[n2] if n2 > 0 else None))
            for k1 in range(4):
                u[4 * k1 + n2] = o[k1]
                uslot[4 * k1 + n2] = outs[k1]
        newpos = [None] * 16

        def place(k1, o, outs):
            for k2 in range(4):
                k = k1 + 4 * k2
                r, _ = outs[k2]
                newpos[k] = (r - X.base) // 2
        # k1 = 0
        outs = uslot[0:4]
        place(0, dft4(st, u[0:4], S, outs), outs)
        # k1 = 1: e = 1, 2, 3
        outs = uslot[4:8]
        g = [u[4], rot_tan(st, u[5], T1, S), rot_tan(st, u[6], 1.0, S), rot_tan(st, u[7], T3, S)]
        place(1, dft4_scaled(st, g, C1, R, S1 / C1, S, outs), outs)
        # k1 = 2: e = 2, 4 (a quarter turn), 6
        outs = uslot[8:12]
        g = [u[8], rot_tan(st, u[9], 1.0, S), u[10].times_i(S), rot_tan(st, u[11], -1.0, S)]
        place(2, dft4_scaled(st, g, R, 1.0, -1.0, S, outs), outs)
        # k1 = 3: e = 3, 6, 9
        outs = uslot[12:16]
        g = [u[12], rot_tan(st, u[13], T3, S), rot_tan(st, u[14], -1.0, S), rot_tan(st, u[15], T1, S)]
        place(3, dft4_scaled(st, g, S1, -R, -C1 / S1, S, outs), outs)
        X.pos = newpos
        assert sorted(newpos) == list(range(16))
        assert len(self.pool.free) == 16, "temporaries leaked"
        self.out.extend(st.ins)

    # ---- rotation + decomposition -------------------------------------------------------------------------------------
    def rotate_decompose(self):
        """d = acc * X^a - acc, both digits of every coefficient as doubles: level 1 -> array A, level 2 -> array B
        (element n1 real part = coefficient 16 n1 + t, imaginary part = coefficient 16 n1 + t + 256).

        Registers and rotation buffer hold the NEGATED accumulator n = -acc (the buffer was written by the previous
        step's tail, or by the prologue).  With r = t - a, coefficient 16 m + t of acc * X^a is sign * acc[(r + 16 m) & 511],
        sign = -1 where bit 9 of r + 16 m is set.  Along m a lane's index wraps past 511 at most once, at
        m = ((511 - (r & 511)) >> 4) + 1, and the sign changes there and only there: one mask per lane and step (bit m = wrapped)
        gives the read address (base - 2048 bytes once wrapped) and the sign of every coefficient with a bit-field extract
        each, instead of recomputing r + 16 m twice per coefficient."""
        e = self.e
        fw, fs, a0, tmp = V_SCR[0], V_SCR[1], V_SCR[2], V_SCR[3]
        e("v_subrev_u32 v%d, s%d, v%d" % (tmp, S_A, V_T))                # r = t - a
        e("v_not_b32 v%d, v%d" % (fw, tmp))
        e("v_bfe_u32 v%d, v%d, 4, 5" % (fw, fw))                          # (511 - (r & 511)) >> 4
        e("v_lshlrev_b32 v%d, v%d, -2" % (fw, fw))               # bit m set: index wrapped
        e("v_bfe_i32 v%d, v%d, 9, 1" % (fs, tmp))
        e("v_xnor_b32 v%d, v%d, v%d" % (fs, fw, fs))                      # bit m set: sign = +1, the negated n[src] enters
        e("v_and_b32 v%d, 0x1ff, v%d" % (tmp, tmp))
        e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (a0, tmp, V_ROTBASE))    # buffer word 4 index + g
        fast = "int_fast" in OPT
        if fast:
            # Kept for the record, not in the default set.  Alone in a loop, v_add / v_sub / v_and / v_xor / v_lshrrev issue in
            # about 2.4 cycles at two waves per SIMD and every three-operand integer form (v_bfe, v_add3, v_xad, v_lshl_add)
            # in about 4.9, like an fp64 instruction; this variant trades each long form for two or three short ones
            # (97 instructions more per step).  Inside this loop it is 1.3 % SLOWER (215.2 k against 218.0 k bootstraps/s in
            # one GPU call): between fp64 instructions a short form costs a full issue slot, so only the count matters.
            w13, whi = V_SCR[4], V_SCR[5]
            e("v_lshlrev_b32 v%d, 13, v%d" % (w13, fw))                   # bit 13 + m = wrapped(m), m <= 18
            e("v_lshrrev_b32 v%d, 6, v%d" % (whi, fw))                    # bit 13 + (m - 19) = wrapped(m), m >= 19
        for m in range(32):
            if fast:
                if m == 0:
                    e("v_and_b32 v%d, 0x2000, v%d" % (tmp, w13))
                else:
                    e("v_lshrrev_b32 v%d, %d, v%d" % (tmp, m if m <= 18 else m - 19, w13 if m <= 18 else whi))
                    e("v_and_b32 v%d, 0x2000, v%d" % (tmp, tmp))
                e("v_sub_u32 v%d, v%d, v%d" % (tmp, a0, tmp))               # a0 - 8192 once wrapped
            else:
                e("v_bfe_i32 v%d, v%d, %d, 1" % (tmp, fw, m))
                e("v_lshl_add_u32 v%d, v%d, 13, v%d" % (tmp, tmp, a0))      # a0 - 8192 once wrapped
            e("ds_read_b32 v%d, v%d offset:%d" % (R_K + m, tmp, 256 * m))
        if FP_DIGITS:
            self.digits_fp(fs)
            return
        # the m-th read has returned when at most 31 - m younger ones are in flight (the counter saturates at 15)
        sg, c, y, s_, u, w = V_SCR[3], V_SCR[4], V_SCR[5], V_SCR[6], V_SCR[7], V_SCR[8]
        for m in range(32):
            self.lds_wait(min(15, 31 - m))
            v = R_K + m
            # y = sign * acc[src] - acc + 2^11 = (n[src] ^ sg) - sg + n + 2^11 with sg = ~0 where sign = +1
            e("v_bfe_i32 v%d, v%d, %d, 1" % (sg, fs, m))
            if fast:
                e("v_xor_b32 v%d, v%d, v%d" % (y, v, sg))
                e("v_sub_u32 v%d, v%d, v%d" % (y, y, sg))
                e("v_add_u32 v%d, v%d, v%d" % (y, y, V_ACC + m))
                e("v_add_u32 v%d, 0x800, v%d" % (y, y))
                e("v_lshrrev_b32 v%d, 12, v%d" % (s_, y))
                e("v_lshrrev_b32 v%d, 31, v%d" % (u, y))
                e("v_add_u32 v%d, v%d, v%d" % (u, s_, u))
                e("v_add_u32 v%d, 0x1ff, v%d" % (u, u))                     # s + bit 19 of the state + 511
            else:
                e("v_bfe_u32 v%d, v%d, %d, 1" % (c, fs, m))
                e("v_add3_u32 v%d, v%d, v%d, s%d" % (c, c, V_ACC + m, S_C2048))
                e("v_xad_u32 v%d, v%d, v%d, v%d" % (y, v, sg, c))
                e("v_lshrrev_b32 v%d, 12, v%d" % (s_, y))
                e("v_lshrrev_b32 v%d, 31, v%d" % (u, y))
                e("v_add3_u32 v%d, v%d, v%d, s%d" % (u, s_, u, S_C511))      # s + bit 19 of the state + 511
            # -d1: the top digit is the value in [-511, 512] congruent to u >> 10 mod 1024, so its negative is the
            # sign-extended 10-bit field of (1023 - u) >> 10 = -(u >> 10); array A carries the sign to the products
            e("v_sub_u32 v%d, 0x3ff, v%d" % (w, u))
            e("v_bfe_i32 v%d, v%d, 10, 10" % (w, w))
            e("v_cvt_f64_i32 %s, v%d" % (dreg(R_A + 2 * m), w))
            e("v_and_b32 v%d, 0xfffffc00, v%d" % (u, u))
            e("v_sub_u32 v%d, v%d, v%d" % (u, s_, u))                    # d2
            e("v_cvt_f64_i32 %s, v%d" % (dreg(R_B + 2 * m), u))

    def digits_fp(self, fs):
        """digits of the 32 rotated coefficients in R_K (see FP_DIGITS above), four coefficients at a time so that no
        instruction follows the one it depends on; array A <- -d1, array B <- d2"""
        e = self.e
        G_ = int(os.environ.get("PBS_ASM_FPG", "0"))
        if G_:
            # measurement variant: groups of G_ coefficients, no temporaries for the doubles: the pair (x, DIGIT_HI) is array B's
            # slot itself (its high word written per coefficient), V replaces it in place, t sits in array A's slot
            sg = [R_T + j for j in range(G_)]
            cy = [R_T + 16 + j for j in range(G_)]
            kreg, creg, c1024 = self.consts.get(DIGIT_K), self.consts.get(DIGIT_C), self.consts.get(1024.0)
            for m0 in range(0, 32, G_):
                ms = range(m0, m0 + G_)
                for j, m in enumerate(ms):
                    self.lds_wait(min(15, 31 - m))
                    e("v_bfe_i32 v%d, v%d, %d, 1" % (sg[j], fs, m))
                    e("v_bfe_u32 v%d, v%d, %d, 1" % (cy[j], fs, m))
                for j, m in enumerate(ms):
                    e("v_add3_u32 v%d, v%d, v%d, s%d" % (cy[j], cy[j], V_ACC + m, S_C2048))
                for j, m in enumerate(ms):
                    e("v_xad_u32 v%d, v%d, v%d, v%d" % (cy[j], R_K + m, sg[j], cy[j]))
                for j, m in enumerate(ms):
                    e("v_and_b32 v%d, s%d, v%d" % (R_B + 2 * m, S_C511, cy[j]))
                    e("v_mov_b32 v%d, 0x%x" % (R_B + 2 * m + 1, DIGIT_HI))
                for m in ms:
                    e("v_add_f64 %s, %s, -%s" % (dreg(R_B + 2 * m), dreg(R_B + 2 * m), kreg))
                for m in ms:
                    e("v_fma_f64 %s, %s, %s, %s" % (dreg(R_A + 2 * m), dreg(R_B + 2 * m), creg, dreg(V_MAGIC)))
                for m in ms:
                    e("v_add_f64 %s, %s, -%s" % (dreg(R_A + 2 * m), dreg(V_MAGIC), dreg(R_A + 2 * m)))
                for m in ms:
                    e("v_fma_f64 %s, %s, %s, %s" % (dreg(R_B + 2 * m), dreg(R_A + 2 * m), c1024, dreg(R_B + 2 * m)))
            return
        D = [R_T + 2 * j for j in range(4)]              # (x, DIGIT_HI) pairs
        Vv = [R_T + 8 + 2 * j for j in range(4)]
        Tt = [R_T + 16 + 2 * j for j in range(4)]
        sg = [R_T + 24 + j for j in range(4)]
        cy = [R_T + 28 + j for j in range(4)]
        for j in range(4):
            e("v_mov_b32 v%d, 0x%x" % (D[j] + 1, DIGIT_HI))
        kreg, creg, c1024 = self.consts.get(DIGIT_K), self.consts.get(DIGIT_C), self.consts.get(1024.0)
        for m0 in range(0, 32, 4):
            for j in range(4):
                self.lds_wait(min(15, 31 - (m0 + j)))
                e("v_bfe_i32 v%d, v%d, %d, 1" % (sg[j], fs, m0 + j))
                e("v_bfe_u32 v%d, v%d, %d, 1" % (cy[j], fs, m0 + j))
            for j in range(4):
                e("v_add3_u32 v%d, v%d, v%d, s%d" % (cy[j], cy[j], V_ACC + m0 + j, S_C2048))
            for j in range(4):
                e("v_xad_u32 v%d, v%d, v%d, v%d" % (cy[j], R_K + m0 + j, sg[j], cy[j]))
            for j in range(4):
                e("v_and_b32 v%d, s%d, v%d" % (D[j], S_C511, cy[j]))
            for j in range(4):
                e("v_add_f64 %s, %s, -%s" % (dreg(Vv[j]), dreg(D[j]), kreg))
            for j in range(4):
                e("v_fma_f64 %s, %s, %s, %s" % (dreg(Tt[j]), dreg(Vv[j]), creg, dreg(V_MAGIC)))
            for j in range(4):
                e("v_add_f64 %s, %s, -%s" % (dreg(R_A + 2 * (m0 + j)), dreg(V_MAGIC), dreg(Tt[j])))
            for j in range(4):
                e("v_fma_f64 %s, %s, %s, %s" % (dreg(R_B + 2 * (m0 + j)), dreg(R_A + 2 * (m0 + j)), c1024, dreg(Vv[j])))

    # ---- twiddles -------------------------------------------------------------------------------------------------------
    def tw_issue(self, k0, count, region):
        for k in range(count):
            addr = V_TWF if LANESUM else V_TW
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(region + 4 * k), addr, (k0 + k) * K_TW_ROW_BYTES))

    def tw_mul(self, X, k0, count, region, S):
        """X[k0 .. k0+count) *= (c + i S s) in place, entry j of `region` = (c, s) of element k0 + j"""
        e = self.e
        # "tw_il": two elements at a time on two pairs of scratch registers, so that a multiply-add follows the product it
        # consumes by four instructions instead of two
        width = 2 if "tw_il" in OPT else 1
        assert count % width == 0
        for j0 in range(0, count, width):
            muls, fmas = [], []
            for w in range(width):
                j = j0 + w
                m1, m2 = V(V_SCR[4 * w]), V(V_SCR[4 * w + 2])
                c, s = V(region + 4 * j), V(region + 4 * j + 2)
                z = X.val(k0 + j)
                muls.append("v_mul_f64 %s, %s, %s" % (dreg(m1.reg), z.i.s(), s.s()))
                muls.append("v_mul_f64 %s, %s, %s" % (dreg(m2.reg), z.r.s(), s.s()))
                fmas.append("v_fma_f64 %s, %s, %s, %s" % (dreg(z.r.reg), z.r.s(), c.s(), m1.s(flip=S > 0)))
                fmas.append("v_fma_f64 %s, %s, %s, %s" % (dreg(z.i.reg), z.i.s(), c.s(), m2.s(flip=S < 0)))
            for ln in muls + fmas:
                e(ln)

    def tw_mul_first(self, X, count, region, scratch):
        """Phase 1 of X[k] *= (c - i s) (the inverse transform's twiddles), from the real parts alone:
        scratch[k] = re s, re <- re c"""
        for j in range(count):
            c, s_ = V(region + 4 * j), V(region + 4 * j + 2)
            z = X.val(j)
            self.e("v_mul_f64 %s, %s, %s" % (dreg(scratch + 2 * j), z.r.s(), s_.s()))
            self.e("v_mul_f64 %s, %s, %s" % (dreg(z.r.reg), z.r.s(), c.s()))

    def tw_mul_second(self, X, count, region, scratch):
        """Phase 2, once the imaginary parts have arrived: re <- re c + im s, im <- im c - re s"""
        for j in range(count):
            c, s_ = V(region + 4 * j), V(region + 4 * j + 2)
            z = X.val(j)
            self.e("v_fma_f64 %s, %s, %s, %s" % (dreg(z.r.reg), z.i.s(), s_.s(), z.r.s()))
            self.e("v_fma_f64 %s, %s, %s, -%s" % (dreg(z.i.reg), z.i.s(), c.s(), dreg(scratch + 2 * j)))

    # ---- exchange -------------------------------------------------------------------------------------------------------
    def exchange_plane(self, X, plane):
        """16 x 16 exchange of one real plane: element r of this lane -> [r][t], row t back into the home slots in
        natural order.  Only issues; the caller waits."""
        if "write2" in OPT:     # two elements of a four-group per store: half the store instructions, and 1.4 % SLOWER in the
            # kernel (223.4 k against 226.5 k bootstraps/s in one GPU call): the paired store occupies the LDS path longer
            # than the two single ones; kept for the record, not in the default set
            for r in range(0, 16, 2):
                r0, r1 = X.slot(X.pos[r])[plane], X.slot(X.pos[r + 1])[plane]
                self.e("ds_write2_b64 v%d, %s, %s offset0:%d offset1:%d" % (V_TRWR[r >> 2], dreg(r0), dreg(r1), r * 16, (r + 1) * 16))
        else:
            for r in range(16):
                reg = X.slot(X.pos[r])[plane]
                self.e("ds_write_b64 v%d, %s offset:%d" % (V_TRWR[r >> 2], dreg(reg), r * 128))
        base = X.base + 32 * plane
        for w in range(8):
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(base + 4 * w), V_TRRD[w & 3], 64 * (w >> 2)))

    def exchange(self, X):
        self.exchange_plane(X, 0)
        self.exchange_plane(X, 1)
        X.pos = list(range(16))

    # ---- key ring ---------------------------------------------------------------------------------------------------------
    def acquire(self, q):
        """chunk q of the current step: own DMA rows landed, barrier, refill the slot freed by the previous chunk with
        chunk q + 3 (the key buffer is padded, so the prefetch needs no end-of-key test), key read address of chunk q"""
        e = self.e
        e("s_waitcnt vmcnt(%d)" % ((RING_AHEAD - 1) * ROWS_PER_WAVE))       # own rows of chunk q: only the younger chunks' rows stay in flight
        e("s_barrier")
        qq = q + RING_AHEAD
        if LANESUM:      # the key is stored in the order the ring takes it: [step][chunk][row][lane]
            off = qq * K_CHUNK_BYTES if qq < 8 else STEP_KEY_BYTES + (qq - 8) * K_CHUNK_BYTES
        else:
            off = qq * 8192 if qq < 8 else STEP_KEY_BYTES + (qq - 8) * 8192
        slot = qq % 4
        if "dma_off" in OPT:
            # one row pointer and one M0 per chunk: the instruction's offset moves the global address AND the LDS address of
            # an LDS-DMA load, so the wave's further rows (contiguous in the key, 1 KiB apart in the ring) need no scalar work;
            # the ring slot of the key reads sits in their offset field, so the read address register never changes
            assert LANESUM, "rows of a wave are contiguous only in the lane-local-sum key layout"
            e("s_add_u32 s%d, s%d, 0x%x" % (S_TMP, S_KP, off))
            e("s_addc_u32 s%d, s%d, 0" % (S_TMP + 1, S_KP + 1))
            e("s_add_u32 m0, s%d, 0x%x" % (S_M0ROW, slot * K_CHUNK_BYTES))
            e("s_nop 0")
            for j in range(ROWS_PER_WAVE):
                e("global_load_lds_dwordx4 v%d, s[%d:%d]%s" % (V_LANE16, S_TMP, S_TMP + 1, " offset:%d" % (1024 * j) if j else ""))
            self.ring_slot = q % 4
            return
        for j in range(ROWS_PER_WAVE):
            if j == 0:
                e("s_add_u32 s%d, s%d, 0x%x" % (S_TMP, S_KP, off))
            else:
                e("s_add_u32 s%d, s%d, 0x400" % (S_TMP, S_TMP))
            e("s_addc_u32 s%d, s%d, 0" % (S_TMP + 1, S_KP + 1 if j == 0 else S_TMP + 1))
            e("s_add_u32 m0, s%d, 0x%x" % (S_M0ROW, slot * K_CHUNK_BYTES + 1024 * j))
            e("s_nop 0")
            e("global_load_lds_dwordx4 v%d, s[%d:%d]" % (V_LANE16, S_TMP, S_TMP + 1))
        e("v_add_u32 v%d, 0x%x, v%d" % (V_KADDR, K_RING_OFFSET + (q % 4) * K_CHUNK_BYTES, V_KBASE))

    def key_issue(self, pos, pair):
        """key words of frequency position `pos`, columns 2 pair and 2 pair + 1, both levels"""
        kk = pos % 2
        for c in (2 * pair, 2 * pair + 1):
            for lv in (0, 1):
                self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(R_K + 16 * lv + 4 * c), V_KADDR, (kk * 8 + lv * 4 + c) * 1024))

    def mac(self, A, B, prefetched=False):
        """A[k2] <- sum over polynomials and levels of spectrum x key, for all 16 frequency groups (in key order)"""
        e = self.e
        T = [R_T + 2 * j for j in range(8)]          # pr[0..3], pi[0..3]
        pr, pi = T[0:4], T[4:8]
        if not prefetched:
            self.acquire(0)
            self.key_issue(0, 0)
            self.key_issue(0, 1)
        for pos in range(16):
            k2 = kperm(pos)
            more = pos + 1 < 16
            opens = more and (pos + 1) % 2 == 0
            f1, f2 = A.val(k2), B.val(k2)
            f1 = C(-f1.r, -f1.i)                      # array A is the spectrum of the NEGATED level-1 digits
            if opens:
                self.acquire((pos + 1) // 2)   # place_waits() drains the old chunk's reads before the barrier
            prod = []
            for pair in (0, 1):
                cols = (2 * pair, 2 * pair + 1)
                b0 = {c: (V(R_K + 4 * c), V(R_K + 4 * c + 2)) for c in cols}
                b1 = {c: (V(R_K + 16 + 4 * c), V(R_K + 16 + 4 * c + 2)) for c in cols}
                seq = [[] for _ in range(4)]         # four independent accumulation chains, interleaved
                for n, c in enumerate(cols):
                    x0, y0 = b0[c]
                    x1, y1 = b1[c]
                    seq[2 * n] = ["v_mul_f64 %s, %s, %s" % (dreg(pr[c]), f1.r.s(), x0.s()),
                                  "v_fma_f64 %s, %s, %s, %s" % (dreg(pr[c]), f1.i.s(flip=True), y0.s(), dreg(pr[c])),
                                  "v_fma_f64 %s, %s, %s, %s" % (dreg(pr[c]), f2.r.s(), x1.s(), dreg(pr[c])),
                                  "v_fma_f64 %s, %s, %s, %s" % (dreg(pr[c]), f2.i.s(flip=True), y1.s(), dreg(pr[c]))]
                    seq[2 * n + 1] = ["v_mul_f64 %s, %s, %s" % (dreg(pi[c]), f1.r.s(), y0.s()),
                                      "v_fma_f64 %s, %s, %s, %s" % (dreg(pi[c]), f1.i.s(), x0.s(), dreg(pi[c])),
                                      "v_fma_f64 %s, %s, %s, %s" % (dreg(pi[c]), f2.r.s(), y1.s(), dreg(pi[c])),
                                      "v_fma_f64 %s, %s, %s, %s" % (dreg(pi[c]), f2.i.s(), x1.s(), dreg(pi[c]))]
                prod.append([seq[ch][step] for step in range(4) for ch in range(4)])
            # sum over the four polynomials of the quad, scattered so that lane g ends with output polynomial g:
            # lane g's p[j] belongs to output j ^ g.  Two values from the neighbour at distance 2, one at distance 1.
            red = []
            def dpp(reg, ctrl):
                for h in (0, 1):
                    red.append("v_mov_b32_dpp v%d, v%d quad_perm:%s row_mask:0xf bank_mask:0xf" % (reg + h, reg + h, ctrl))
            X2, X1 = "[2,3,0,1]", "[1,0,3,2]"
            dpp(pr[2], X2)
            dpp(pi[2], X2)
            dpp(pr[3], X2)
            dpp(pi[3], X2)
            red.append("v_add_f64 %s, %s, %s" % (dreg(pr[0]), dreg(pr[0]), dreg(pr[2])))     # sa (real)
            red.append("v_add_f64 %s, %s, %s" % (dreg(pr[1]), dreg(pr[1]), dreg(pr[3])))     # sb (real)
            red.append("v_add_f64 %s, %s, %s" % (dreg(pi[0]), dreg(pi[0]), dreg(pi[2])))
            red.append("v_add_f64 %s, %s, %s" % (dreg(pi[1]), dreg(pi[1]), dreg(pi[3])))
            dpp(pr[1], X1)
            dpp(pi[1], X1)
            orr, oi = A.slot(A.pos[k2])
            red.append("v_add_f64 %s, %s, %s" % (dreg(orr), dreg(pr[0]), dreg(pr[1])))
            red.append("v_add_f64 %s, %s, %s" % (dreg(oi), dreg(pi[0]), dreg(pi[1])))
            # the next position's key words: a pair's registers are refilled while the other pair multiplies / the sums travel
            nxt0 = self.capture(self.key_issue, pos + 1, 0) if more else []
            nxt1 = self.capture(self.key_issue, pos + 1, 1) if more else []
            if "mac_il" in OPT:
                self.out.extend(prod[0])
                self.out.extend(self.spread(prod[1], nxt0))
                self.out.extend(self.spread(red, nxt1))
            else:
                self.out.extend(prod[0] + nxt0 + prod[1] + nxt1 + red)

    # ---- lanesum: quad exchanges and the in-lane multiply-accumulate ---------------------------------------------------------
    def quad_exchange(self, X, forward):
        """In place.  forward: lane x takes, from lane (x + d) & 3, the positions of slot set (4 - d) & 3; afterwards slot set j
        of lane x holds polynomial (x - j) & 3 at the bins with high digit x.  inverse: lane c takes slot set d from lane
        (c + d) & 3: output column c, bins with high digit (c + d) & 3, in slot set d.  Slot set j = positions j, 4 + j, 8 + j, 12 + j."""
        for d in (1, 2, 3):
            j = (4 - d) & 3 if forward else d
            ctrl = "[%d,%d,%d,%d]" % tuple((i + d) & 3 for i in range(4))
            for k_lo in range(4):
                re, im = X.slot(4 * k_lo + j)
                for reg in (re, re + 1, im, im + 1):
                    self.e("v_mov_b32_dpp v%d, v%d quad_perm:%s row_mask:0xf bank_mask:0xf" % (reg, reg, ctrl))

    def key_buffers(self):
        """register blocks of four key words: two in the key region, a third in the half of the temporaries the products
        leave free (it is loaded only after the last forward butterfly has returned its temporaries)"""
        return [R_K, R_K + 16, R_T + 16] if "ls_pf2" in OPT else [R_K, R_K + 16]

    def key_issue_ls(self, lv, j, buf):
        """key words [j][j' = 0..3] of the chunk (k_lo, lv) at hand, into key buffer `buf` (first register of sixteen, or a list
        of four aligned register quads)"""
        base = self.ring_slot * K_CHUNK_BYTES if "dma_off" in OPT else 0
        quads = buf if isinstance(buf, list) else [buf + 4 * jp for jp in range(4)]
        for jp in range(4):
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(quads[jp]), V_KADDR, base + (4 * j + jp) * 1024))

    def mac_lanesum(self, A, B):
        """A[position (k_lo, j')] <- sum over levels and slot sets j of spectrum[lv][(k_lo, j)] x key[k_lo][lv][j][j'].  Eight
        independent accumulation chains per bin (four output slots, real and imaginary), sixteen multiply-adds each; a term's
        four key words are fetched one (two with ls_pf2) terms ahead, under the products.  Chunk (k_lo, lv) of the ring = the
        sixteen key words [j][j'] of one level of one bin.  "ls_pf3": from the second bin on a fourth buffer (the level-2
        slots of the first bin, consumed by then) lets the words travel three terms ahead."""
        T = [R_T + 2 * n for n in range(8)]
        pr, pi = T[0:4], T[4:8]
        free = [[b + 4 * jp for jp in range(4)] for b in self.key_buffers()]
        terms = [(k_lo, lv, j) for k_lo in range(4) for lv in (0, 1) for j in range(4)]
        # chunk 0 was acquired and the first terms' words requested inside the second forward pass (see step())
        held, issued = {}, len(free) - 1
        for n in range(issued):
            held[n] = free.pop(0)
        for n, (k_lo, lv, j) in enumerate(terms):
            X = A if lv == 0 else B
            f = X.val_at(4 * k_lo + j)
            if lv == 0:
                f = C(-f.r, -f.i)                 # array A is the spectrum of the NEGATED level-1 digits
            first, last = (lv, j) == (0, 0), (lv, j) == (1, 3)
            if "ls_pf3" in OPT and n == 8:        # array B's first four positions have been consumed
                free.append([R_B, R_B + 4, R_B + 32, R_B + 36])
            # the key words of the terms ahead travel under this term's products; a new chunk is acquired first
            nxt = []
            while issued < len(terms) and free:
                nk, nlv, nj = terms[issued]
                if nj == 0:
                    nxt += self.capture(self.acquire, 2 * nk + nlv)
                held[issued] = free.pop(0)
                nxt += self.capture(self.key_issue_ls, nlv, nj, held[issued])
                issued += 1
            quads = held.pop(n)
            key = [(V(quads[jp]), V(quads[jp] + 2)) for jp in range(4)]
            ins = []
            for part in (0, 1):                   # part 0: products with the real part of the spectrum, part 1: imaginary
                for jp in range(4):
                    x, y = key[jp]
                    orr, oi = A.slot(4 * k_lo + jp)
                    dr = dreg(orr) if (last and part == 1) else dreg(pr[jp])
                    di = dreg(oi) if (last and part == 1) else dreg(pi[jp])
                    if first and part == 0:
                        ins.append("v_mul_f64 %s, %s, %s" % (dreg(pr[jp]), f.r.s(), x.s()))
                        ins.append("v_mul_f64 %s, %s, %s" % (dreg(pi[jp]), f.r.s(), y.s()))
                    elif part == 0:
                        ins.append("v_fma_f64 %s, %s, %s, %s" % (dreg(pr[jp]), f.r.s(), x.s(), dreg(pr[jp])))
                        ins.append("v_fma_f64 %s, %s, %s, %s" % (dreg(pi[jp]), f.r.s(), y.s(), dreg(pi[jp])))
                    else:
                        ins.append("v_fma_f64 %s, %s, %s, %s" % (dr, f.i.s(flip=True), y.s(), dreg(pr[jp])))
                        ins.append("v_fma_f64 %s, %s, %s, %s" % (di, f.i.s(), x.s(), dreg(pi[jp])))
            self.out.extend(self.spread(ins, nxt) if "mac_il" in OPT else ins + nxt)
            free.append(quads)

    # ---- the step ---------------------------------------------------------------------------------------------------------
    def capture(self, fn, *args):
        """the lines fn emits, as a list (not appended to the output)"""
        saved, self.out = self.out, []
        fn(*args)
        got, self.out = self.out, saved
        return got

    @staticmethod
    def spread(main, side, lo=0, hi=None):
        """main with the instructions of `side` (kept in order) inserted at even distances inside main[lo:hi]"""
        hi = len(main) if hi is None else hi
        if not side:
            return list(main)
        out, n, k = list(main[:lo]), hi - lo, 0
        for i in range(n):
            out.append(main[lo + i])
            while k < len(side) and (k + 1) * n <= (i + 1) * len(side):
                out.append(side[k])
                k += 1
        out.extend(side[k:])
        out.extend(main[hi:])
        return out

    def exchange_after_ready(self, main, X):
        """main (the transform that leaves X's elements in their home slots) with the stores of the exchange's first
        plane placed right after the instruction that completes each element, then the rest of the exchange"""
        out = list(main)
        last = {}
        for i, ln in enumerate(main):
            if ln.startswith("v_"):
                for reg in regs_of(ln.split(None, 1)[1].split(",")[0]):
                    last[reg] = i
        stores = []
        if "write2" in OPT:
            for r in range(0, 16, 2):
                r0, r1 = X.slot(X.pos[r])[0], X.slot(X.pos[r + 1])[0]
                at = max(last.get(("v", r0), -1), last.get(("v", r1), -1))
                stores.append((at, "ds_write2_b64 v%d, %s, %s offset0:%d offset1:%d" % (V_TRWR[r >> 2], dreg(r0), dreg(r1), r * 16, (r + 1) * 16)))
        for r in range(0 if "write2" not in OPT else 16, 16):
            reg = X.slot(X.pos[r])[0]
            stores.append((last.get(("v", reg), -1), "ds_write_b64 v%d, %s offset:%d" % (V_TRWR[r >> 2], dreg(reg), r * 128)))
        for at, ln in sorted(stores, key=lambda x: -x[0]):
            out.insert(at + 1, ln)
        return out

    def step(self):
        e = self.e
        A, B = Arr(R_A), Arr(R_B)
        self.rotate_decompose()
        # level 1: first pass, twiddles, then its exchange travels inside the first pass of level 2
        self.tw_issue(0, 8, R_K)
        self.dft16(A, 1, True)
        self.out.extend(self.spread(self.capture(self.tw_mul, A, 0, 8, R_K, 1), self.capture(self.tw_issue, 8, 8, R_T), 0, 32 if "tw_spread" in OPT else 0))
        self.tw_mul(A, 8, 8, R_T, 1)
        side = self.capture(self.exchange, A)       # twiddles 0-7 stay in R_K for level 2 (nothing else uses it until the key arrives)
        main = self.capture(self.dft16, B, 1, True)
        self.out.extend(self.spread(main, side))
        self.out.extend(self.spread(self.capture(self.tw_mul, B, 0, 8, R_K, 1), self.capture(self.tw_issue, 8, 8, R_T), 0, 32 if "tw_spread" in OPT else 0))
        self.tw_mul(B, 8, 8, R_T, 1)
        # level 2's exchange, the first chunk's barrier and the first key words travel inside the second pass of level 1
        if LANESUM:
            side = self.capture(self.exchange, B) + self.capture(self.acquire, 0) + self.capture(self.key_issue_ls, 0, 0, R_K)
            if len(self.key_buffers()) == 3:
                side += self.capture(self.key_issue_ls, 0, 1, R_K + 16)
            main = self.capture(self.dft16, A, 1, False)
            self.out.extend(self.spread(main, side))
            # level 1's quad exchange travels inside the second pass of level 2
            main = self.capture(self.dft16, B, 1, False)
            self.out.extend(self.spread(main, self.capture(self.quad_exchange, A, True)))
            self.quad_exchange(B, True)
            self.mac_lanesum(A, B)
            self.quad_exchange(A, False)
        else:
            side = self.capture(self.exchange, B) + self.capture(self.acquire, 0) + self.capture(self.key_issue, 0, 0) + self.capture(self.key_issue, 0, 1)
            main = self.capture(self.dft16, A, 1, False)
            self.out.extend(self.spread(main, side))
            self.dft16(B, 1, False)
            # spectrum x key
            self.mac(A, B, prefetched=True)
        # inverse transform of the sums (array A); array B is free and holds all sixteen twiddles
        e("v_mov_b32 v%d, s%d" % (V_SCR[0], S_ACUR))
        e("ds_read_u16 v%d, v%d" % (V_AREAD, V_SCR[0]))
        main = self.capture(self.dft16, A, -1, False)
        main = self.spread(main, self.capture(self.tw_issue, 0, 16, R_B), 0, 64 if "tw_spread" in OPT else 0)
        self.out.extend(self.exchange_after_ready(main, A))
        for r in range(8):
            e("ds_read_b128 %s, v%d offset:%d" % (qreg(A.base + 4 * r), V_TRRD[r & 3], 64 * (r >> 2)))
        self.exchange_plane(A, 1)
        A.pos = list(range(16))
        e("v_readfirstlane_b32 s%d, v%d" % (S_A, V_AREAD))
        # the twiddle products that need only the real parts run while the imaginary plane is on its way
        if "inv_split" in OPT:
            self.tw_mul_first(A, 16, R_B, R_K)
            self.tw_mul_second(A, 16, R_B, R_K)
        else:
            self.tw_mul(A, 0, 16, R_B, -1)
        self.dft16(A, -1, False)
        # untwist by e^{-i pi n1 / 32}, round to the torus, accumulate, and put the new accumulator into the rotation buffer
        st = Stream(self, self.pool)
        magic = V(V_MAGIC)
        # "tail_il": two elements at a time, stage by stage (untwist, scale and round, accumulate, store), so that no
        # instruction follows the one it depends on by fewer than four
        width = 2 if "tail_il" in OPT else 1
        for n0 in range(0, 16, width):
            zs = [A.val(n1) for n1 in range(n0, n0 + width)]
            rot = []
            for n1, z in zip(range(n0, n0 + width), zs):
                if n1 == 0:
                    rot.append(None)
                else:
                    tau = KS32[n1] / KC32[n1]
                    rot.append((st.fmac(tau, z.i, z.r), st.fmac(-tau, z.r, z.i)))      # (r + i i)(1 - i tau)
            rnd = []
            for n1, z, nrni in zip(range(n0, n0 + width), zs, rot):
                if nrni is None:
                    rnd.append((st.add(z.r, magic), st.add(z.i, magic)))
                else:
                    # round(c x) as an integer in the low mantissa word: one multiply-add with 1.5 * 2^52
                    rnd.append((st.fmac(KC32[n1], nrni[0], magic), st.fmac(KC32[n1], nrni[1], magic)))
                    st.kill(*nrni)
            for n1, (rr, ri) in zip(range(n0, n0 + width), rnd):
                for (val, m) in ((rr, n1), (ri, 16 + n1)):
                    st.e("v_sub_u32 v%d, v%d, v%d" % (V_ACC + m, V_ACC + m, val.reg))      # the negated accumulator
            for n1, (rr, ri) in zip(range(n0, n0 + width), rnd):
                for m in (n1, 16 + n1):
                    # buffer word 4 (16 m + t) + g = 64 m + lane: the store needs no address register
                    st.e("ds_write_addtid_b32 v%d offset:%d" % (V_ACC + m, 256 * m))
                st.kill(rr, ri)
        self.out.extend(st.ins)
        assert len(self.pool.free) == 16
        # M0 is shared with the ring's LDS-DMA (acquire() sets it before each of its loads): point it at the rotation
        # buffer in front of the first store of the tail
        k = next(i for i, ln in enumerate(self.out) if ln.startswith("ds_write_addtid_b32"))
        self.out[k:k] = ["s_mov_b32 m0, s%d" % S_ROTM0, "s_nop 0"]

    def build(self):
        e = self.e
        body = Loop.__new__(Loop)
        body.consts, body.out, body.pool = self.consts, [], self.pool
        body.step()
        # measurement only (PBS_ASM_PAD=class:count): `count` harmless extra instructions of one class spread evenly over the
        # step (they write registers nothing reads, so the results stay bit-exact): what one more instruction of that class
        # costs inside this loop
        pad = os.environ.get("PBS_ASM_PAD", "")
        if pad.startswith("dup:"):
            # every LDS instruction of one opcode issued twice in a row (same addresses, same data: harmless): its cost in situ
            opc = pad.split(":")[1]
            dup = []
            for ln in body.out:
                dup.append(ln)
                if ln.split()[0] == opc and ("exch" not in pad or any(", v%d " % r in ln + " " or ln.startswith(opc + " v%d," % r) for r in V_TRWR + V_TRRD)):
                    dup.append(ln)
            body.out = dup
        elif pad:
            kind, count = pad.split(":")
            count = int(count)
            forms = {"f64": ["v_add_f64 v[%d:%d], v[30:31], v[30:31]" % (r, r + 1) for r in (24, 26, 28)],
                     "fma": ["v_fma_f64 v[%d:%d], v[30:31], v[30:31], v[30:31]" % (r, r + 1) for r in (24, 26, 28)],
                     "int": ["v_bfe_i32 v%d, v3, 2, 1" % r for r in (24, 25, 26)],
                     "short": ["v_add_u32 v%d, v3, v3" % r for r in (24, 25, 26)],
                     "dpp": ["v_mov_b32_dpp v%d, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % r for r in (24, 25, 26)],
                     "lds": ["ds_read_b32 v%d, v17" % r for r in (24, 25, 26)],
                     "lds128": ["ds_read_b128 v[24:27], v%d offset:%d" % (V_KBASE, 1024 * k) for k in range(3)],     # conflict-free 16-byte reads
                     "vmem128": ["global_load_dwordx4 v[24:27], v%d, s[%d:%d]" % (V_LANE16, S_KP, S_KP + 1)] * 3,     # the same bytes from L1 / L2
                     "salu": ["s_add_u32 s97, s97, 1"] * 3}[kind]
            valu_at = [i for i, ln in enumerate(body.out) if ln.startswith("v_")]
            picks = [valu_at[(2 * k + 1) * len(valu_at) // (2 * count)] for k in range(count)]
            for k, at in reversed(list(enumerate(picks))):
                body.out.insert(at, forms[k % 3])
        # constants (known only after the body has been generated)
        import struct
        e("s_mov_b32 s%d, 0x%x" % (S_C2048, DIGIT_BIAS if FP_DIGITS else 0x800))
        e("s_mov_b32 s%d, 0x%x" % (S_C511, 0xfffff000 if FP_DIGITS else 0x1ff))
        e("v_readfirstlane_b32 s%d, v%d" % (S_ROTM0, V_ROTBASE))       # lane 0: g = 0, the buffer's first byte
        for j, val in enumerate(self.consts.vals):
            bits = struct.unpack("<Q", struct.pack("<d", val))[0]
            e("s_mov_b32 s%d, 0x%x" % (const_reg(j), bits & 0xFFFFFFFF))
            e("s_mov_b32 s%d, 0x%x" % (const_reg(j) + 1, bits >> 32))
        mbits = struct.unpack("<Q", struct.pack("<d", MAGIC))[0]
        e("v_mov_b32 v%d, 0x%x" % (V_MAGIC, mbits & 0xFFFFFFFF))
        e("v_mov_b32 v%d, 0x%x" % (V_MAGIC + 1, mbits >> 32))
        if "dma_off" in OPT:
            e("v_add_u32 v%d, 0x%x, v%d" % (V_KADDR, K_RING_OFFSET, V_KBASE))
        if "shift4" in OPT:      # measurement: the loop's instruction stream moved by four bytes (placement sensitivity)
            e("s_nop 0")
        e("L_step%=:")
        self.out.extend(place_waits(body.out) if AUTO_WAITS else body.out)
        e("s_add_u32 s%d, s%d, 0x%x" % (S_KP, S_KP, STEP_KEY_BYTES))
        e("s_addc_u32 s%d, s%d, 0" % (S_KP + 1, S_KP + 1))
        e("s_add_u32 s%d, s%d, 2" % (S_ACUR, S_ACUR))
        e("s_sub_u32 s%d, s%d, 1" % (S_STEPS, S_STEPS))
        e("s_cmp_lg_u32 s%d, 0" % S_STEPS)
        e("s_cbranch_scc1 L_step%=")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return self.out


_REG = __import__("re").compile(r"\b([vs])(?:\[(\d+):(\d+)\]|(\d+))")


def regs_of(text):
    """every VGPR / SGPR named in an operand string, as ("v", n) / ("s", n)"""
    out = set()
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(1), int(m.group(4))))
    return out


def lds_dst(ln):
    """destination registers of an LDS read (empty for everything else)"""
    if not ln.startswith("ds_read"):
        return set()
    return regs_of(ln.split(None, 1)[1].split(",")[0])


def place_waits(lines):
    """Drops the hand-placed LDS waits and DPP wait states and puts back exactly what the final instruction order needs.
    The LDS returns a wave's operations in issue order, so an instruction that touches the destination of a read still
    in flight waits until at most N younger operations are outstanding (N capped at 15, the counter's width); a
    barrier of the key ring waits for every key read issued before it (the chunk they came from is refilled after
    the barrier).  A DPP move needs two wait states after the vector instruction that wrote its source."""
    body = [ln for ln in lines if not (ln.startswith("s_waitcnt lgkmcnt(") or ln == "s_nop 1")]
    # "wait_merge": an s_waitcnt is an instruction like any other and takes an issue slot.  When reads that were issued long
    # ago (at least MERGE_AGE instructions back) are first used within the next MERGE_AHEAD instructions, one wait covers them
    # all: it waits for the youngest of them, which the in-order return makes a wait for every older one too.
    merge = "wait_merge" in OPT
    MERGE_AHEAD, MERGE_AGE = 16, 24
    touched_at = [regs_of(ln.split(None, 1)[1]) if (" " in ln and not ln.endswith(":")) else set() for ln in body]
    out, queue, recent = [], [], []          # queue: [dst regs, is key read, issue index] of LDS operations in flight, oldest first
    for pos, ln in enumerate(body):
        if ln.endswith(":"):
            assert not any(q[0] for q in queue), "LDS reads in flight across a label"
            out.append(ln)
            continue
        op = ln.split()[0]
        touched = touched_at[pos]
        need = -1
        for k, (dst, is_key, _) in enumerate(queue):
            if (dst & touched) or (is_key and op == "s_barrier" and RING_AHEAD == 3):
                need = k
        if op == "s_cbranch_scc1" or "vmcnt(0) lgkmcnt(0)" in ln:
            need = max([k for k, q in enumerate(queue) if q[0]], default=-1)
        if need >= 0:
            if merge:
                soon = set()
                for nxt in range(pos + 1, min(pos + 1 + MERGE_AHEAD, len(body))):
                    if body[nxt].endswith(":") or body[nxt].split()[0] in ("s_barrier", "s_cbranch_scc1"):
                        break
                    soon |= touched_at[nxt]
                for k in range(need + 1, len(queue)):
                    if queue[k][0] and (queue[k][0] & soon) and queue[k][2] <= pos - MERGE_AGE:
                        need = k
            n = min(len(queue) - 1 - need, 15)
            out.append("s_waitcnt lgkmcnt(%d)" % n)
            queue = queue[len(queue) - n:] if n else []
        if "_dpp" in op:
            src = regs_of(ln.split(None, 1)[1].split(",")[1].split()[0])
            for back, w in enumerate(reversed(recent[-2:])):
                if w & src:
                    out.append("s_nop %d" % (1 - back))
                    recent.append(set())
                    if back == 0:
                        recent.append(set())
                    break
        out.append(ln)
        if op.startswith("ds_"):
            queue.append((lds_dst(ln), op == "ds_read_b128" and ", v%d " % V_KADDR in ln + " ", pos))
        if op.startswith("v_"):
            recent.append(regs_of(ln.split(None, 1)[1].split(",")[0]))
        elif not op.startswith("s_waitcnt"):
            recent.append(set())
    return out


def clobbers(lines):
    import re
    v, s = set(), set()
    for ln in lines:
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", ln):
            v.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", ln):
            v.add(int(m.group(1)))
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", ln):
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", ln):
            s.add(int(m.group(1)))
    return sorted(v), sorted(s)


def experiment_filter(lines, flags):
    """Timing experiments (results become garbage): drop classes of instructions from the emitted loop.
    nolds: every ds_* instruction; nokey / notw / noexch / norot: the key, twiddle, exchange, rotation LDS traffic; nowait: the LDS waits; nobar: the barriers; nodma: the ring's LDS-DMA and its waits."""
    out = []
    tw_seen = 0
    for ln in lines:
        op = ln.split()[0]
        # rotation addressing / sign handling of the decomposition removed (what a wrap-free, pre-signed rotation buffer would save)
        if "noaddr" in flags and op == "v_lshl_add_u32" and ", 13, " in ln:
            continue
        if "noaddr" in flags and op == "v_bfe_i32" and ln.endswith(", 1") and ", v%d, " % V_SCR[0] in ln and "v%d," % V_SCR[3] in ln:
            continue
        if "nosign" in flags and op in ("v_bfe_i32", "v_bfe_u32") and ", v%d, " % V_SCR[1] in ln:
            continue
        if "nosign" in flags and op == "v_xad_u32":
            continue
        if "notw48" in flags and op == "v_mul_f64" and ("v[%d:%d]," % (V_SCR[0], V_SCR[0] + 1) in ln) and tw_seen < 48:
            tw_seen += 1      # one product of each of 48 inter-pass twiddle multiplications (what folding them into butterflies would save)
            continue
        if "nolds" in flags and op.startswith("ds_"):
            continue
        if "nokey" in flags and op == "ds_read_b128" and ", v%d " % V_KADDR in ln:
            continue
        if "notw" in flags and op == "ds_read_b128" and (", v%d " % V_TW in ln or ", v%d " % V_TWF in ln):
            continue
        if "twlane" in flags and op == "ds_read_b128" and ", v%d " % V_TW in ln:
            ln = ln.replace(", v%d " % V_TW, ", v%d " % V_LANE16)     # one distinct 16-byte word per lane instead of one per four lanes
            k = int(ln.split("offset:")[1])
            ln = ln.split("offset:")[0] + "offset:%d" % ((k // K_TW_ROW_BYTES) * 1024)
        if "noexch" in flags and (op == "ds_write_b64" or (op == "ds_read_b128" and any(", v%d " % r in ln for r in V_TRRD))):
            continue
        if "norot" in flags and op in ("ds_read_b32", "ds_write_b32"):
            continue
        if "nowait" in flags and op == "s_waitcnt" and "lgkmcnt" in ln and "vmcnt" not in ln:
            continue
        if "nobar" in flags and op == "s_barrier":
            continue
        if "nodma" in flags and (op.startswith("global_load_lds") or (op == "s_waitcnt" and ln.strip() == "s_waitcnt vmcnt(%d)" % ((RING_AHEAD - 1) * ROWS_PER_WAVE))):
            continue
        out.append(ln)
    return out


def main():
    g = Loop()
    lines = g.build()
    flags = [f for f in os.environ.get("PBS_ASM_EXP", "").split(",") if f]
    if flags:
        lines = experiment_filter(lines, flags)
    v, s = clobbers(lines)
    inputs_v = set([V_ROTBASE, V_T] + V_TRWR + V_TRRD + [V_KBASE, V_TW, V_LANE16] + ([V_TWF] if LANESUM else [])) | set(range(V_ACC, V_ACC + 32))
    inputs_s = set([S_KP, S_KP + 1, S_STEPS, S_ACUR, S_A, S_M0ROW])
    assert 0 not in v and not (set(range(0, 8)) & set(s)), "v0 / s0-s7 belong to the compiler"
    cl = ['"v%d"' % r for r in v if r not in inputs_v] + ['"s%d"' % r for r in s if r not in inputs_s]
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tfhe-aes_amd", "csrc", "pbs_asm_loop.inc")
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/pbs_asm/gen.py - do not edit.  %d instructions per blind-rotation step.\n" % len([x for x in lines if not x.endswith(":")]))
        f.write("#define TFA_PBS_ASM%s_BODY \\\n" % ("" if WAVES == 8 else "4"))
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("#define TFA_PBS_ASM%s_CLOBBERS %s\n" % ("" if WAVES == 8 else "4", ", ".join(cl + ['"memory"', '"scc"', '"vcc"'])))
        f.write("#define TFA_PBS_ASM%s_RING_AHEAD %d\n" % ("" if WAVES == 8 else "4", RING_AHEAD))      # chunks in flight when the loop starts
    stats = {}
    for ln in lines:
        op = ln.split()[0]
        stats[op] = stats.get(op, 0) + 1
    valu = sum(n for op, n in stats.items() if op.startswith("v_"))
    print("lines", len(lines), "VALU", valu, "f64", sum(n for op, n in stats.items() if "f64" in op), "ds", sum(n for op, n in stats.items() if op.startswith("ds_")),
          "consts", len(g.consts.vals), "temp low water", g.pool.low_water)
    return lines


if __name__ == "__main__":
    main()
