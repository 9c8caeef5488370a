#!/usr/bin/env python3
"""Generator of the hand-scheduled step loop of pbs_lat_asm_kernel (gfx950 assembly): ONE bootstrap per workgroup,
spread over its four waves, for stages too narrow to fill the GPU with one bootstrap per wave.

Mapping, data layouts and results are those of pbs_lat_kernel<1> (csrc/pbs_lat_kernel.h): wave w owns polynomial w of
the accumulator, lane L the coefficients L + 64 m; the size-256 transform runs over the 64 lanes as four radix-4 stages
with three hand-overs through wave-private LDS; the waves publish their spectra (one barrier per step, two buffers
alternating with the step parity) and wave w accumulates output polynomial w from all eight spectra and column w of
the key, which it loads straight from global memory into registers one step ahead.

What differs from the compiled kernel: every register and the order of the ~800 instructions of a step are chosen
here; the two levels of the forward transform are interleaved stage by stage so that one level's hand-over travels
under the other's butterfly; complex products are two multiplies and two multiply-adds; the twiddles of stages 1 and 2
are read from a small LDS table when needed (registers: 128 key words + 28 twiddle words + data); the waits come
from gen.place_waits().  tools/pbs_asm/lat_emu.py executes the emitted text on the CPU against the exact oracle.

Register map (VGPR): v2 rotation buffer of this wave | v3 L | v4-6 / v7, v28, v29 the lane's first element in hand-over
e's layout at stage e / e + 1 | v8 spectrum write base | v9-11 spectrum read bases of the polynomials (w + 1, 2, 3) & 3 |
v12 key offset (16 L) | v13 stage-twiddle table address | v14-17 the four spectrum addresses with the step parity added |
v18 next rotation amount | v19-27 scratch | v30-31 1.5 * 2^52 | v32-39 accumulator (negated) | v40-55 twist | v56-67
stage-0 twiddles | v68-83 / v84-99 data of level 1 / 2 as (re, im) quads; the spectra read from the other waves |
v100-115 output sums | v116-127 (+ whatever of v68-115 is free in a phase) temporaries | v128-255 key words
[level][k][q], k = polynomial (w + k) & 3.   SGPR: s10 steps left | s11 address of a_tilde[i + 1] | s12 rotation amount |
s13 parity offset | s14-15 scratch | s16, s17 the constants 2^11 and 511 | s20-31, s34-37 key pointers per (level, k) |
s38-39 twiddle table in global memory (preamble).  M0 = this wave's rotation buffer (ds_write_addtid_b32).
"""
import math
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen as G  # noqa: E402
from gen import C, V, dreg, qreg  # noqa: E402

# ---- geometry shared with pbs_lat_kernel.h / pbs_lat_asm_kernel.h ----
K_FFT_ELEMS = 320
K_FFT_BYTES = K_FFT_ELEMS * 16           # 5120
K_WAVE_BYTES = 2 * K_FFT_BYTES
K_RED_OFFSET = 4 * K_WAVE_BYTES          # 40960
K_SPEC_BYTES = 4 * 2 * 4 * 64 * 16       # 32768 per buffer
# PER_CU = 1: one workgroup per CU, two spectra buffers alternating with the step parity, one barrier per step.
# PER_CU = 2: two workgroups share a CU (two waves per SIMD hide each other's waits); one spectra buffer and a second barrier
# per step, so that both fit in the LDS.
PER_CU = int(os.environ.get("PBS_LAT_PER_CU", "1"))
K_SPEC_BUFFERS = 2 if PER_CU == 1 else 1
K_A_OFFSET = K_RED_OFFSET + K_SPEC_BUFFERS * K_SPEC_BYTES
K_TW_OFFSET = K_A_OFFSET + 1664          # stage-1 / stage-2 twiddles [s - 1][r][L] c64
K_LDS_BYTES = K_TW_OFFSET + 6 * 1024
STEP_KEY_BYTES = 2 * 4 * 4 * 4 * 64 * 16  # 131072
MAGIC = 6755399441055744.0

S_STEPS, S_ACUR, S_A, S_PAR, S_TMP = 10, 11, 12, 13, 14
S_C2048 = 16
S_C511 = 17
S_DK, S_DC, S_D1024 = 40, 42, 44           # fp64 constants of digits_fp: 2^40 + 2^19 - 2^9, (1 - 2^-21) / 1024, 1024
# Digits of the decomposition: 0 (shipped) the integer sequence; 1 / 2 the wide loop's floating-point digits (digits_fp), two /
# eight coefficients at a time.  Measured in one GPU call (ms per stage of 128 / 256 / 512 bootstraps): integer 2.29 / 2.38 /
# 3.95, two at a time 2.36 / 2.45 / 3.91, eight at a time 2.32 / 2.42 / 3.95: with one wave per SIMD the longer dependent
# chain of the fp64 form costs more than its 30 fewer instructions save, with two it is even.
FPD_MODE = int(os.environ.get("PBS_LAT_FPD", "0"))
FP_DIGITS = FPD_MODE != 0
S_TW = 38                                 # s[38:39] twiddle table in global memory (prologue only)


def key_sreg(j):
    """first scalar register of key pointer j = lv * 4 + k: s[20:31] and s[34:37]; s[32:33] are left alone (the compiler
    reserves s32 as stack pointer, and an asm statement that changes it would break any call the kernel ever makes)"""
    return 20 + 2 * j if j < 6 else 22 + 2 * j

V_ROTBASE, V_L = 2, 3
V_PA = [4, 5, 6]                          # lane's position in hand-over e's layout at stage e
V_PB = [7, 28, 29]                        # ... at stage e + 1
V_SPECWR, V_SPECRD, V_KOFF, V_TWADDR = 8, [9, 10, 11], 12, 13
V_SPECWR_CUR, V_SPECRD_CUR = 14, [15, 16, 17]
V_AREAD = 18
V_SCR = list(range(19, 30))
V_MAGIC = 30
V_ACC = 32
R_TWIST, R_TW0 = 40, 56
R_Z = [68, 84]
R_OUT = 100
R_T = 116
R_KEY = 128


# Hand-over buffers.  A lane holds, at stage s, the four elements whose base-4 index digit (3 - s) runs over the slots; its
# other three digits come from the lane number.  Every hand-over e (between stages e and e + 1, in either direction) has
# its own layout, a stride per digit in 16-byte units, chosen so that the ds_write_b128 of both stages (groups of 8
# contiguous lanes, 32 banks) and the ds_read_b128 of both stages (the instruction's four 16-lane groups, 64 banks) are all free of
# bank conflicts; no single layout does that for all four stages.
EXCH_STRIDES = [(64, 16, 4, 1), (16, 68, 4, 1), (80, 4, 17, 1)]      # (digit 3, 2, 1, 0)


def exch_pos(e, s, L, slot):
    """16-byte position, in hand-over e's layout, of slot `slot` of lane L at stage s (s = e or e + 1)"""
    assert s in (e, e + 1)
    lane_digits = [(L >> 4) & 3, (L >> 2) & 3, L & 3]
    lane_digits.insert(s, slot)                                      # digits (3, 2, 1, 0); stage s owns digit 3 - s
    return sum(d * st for d, st in zip(lane_digits, EXCH_STRIDES[e]))


def slot_stride_bytes(e, s):
    return 16 * EXCH_STRIDES[e][s]


class Quad:
    """A complex value in four consecutive registers (re lo, re hi, im lo, im hi)"""

    def __init__(self, reg):
        assert reg % 4 == 0
        self.reg = reg

    def c(self):
        return C(V(self.reg), V(self.reg + 2))


class Lat:
    def __init__(self):
        self.consts = G.Consts()
        self.out = []

    def e(self, s):
        self.out.append(s)

    def capture(self, fn, *args):
        saved, self.out = self.out, []
        fn(*args)
        got, self.out = self.out, saved
        return got

    # ---- complex product with a twiddle in a register quad: z <- z * (c + i S s), four instructions, in place ----
    def cmul(self, z, tw, S, scratch):
        e = self.e
        c, s_ = V(tw), V(tw + 2)
        m1, m2 = scratch, scratch + 2
        e("v_mul_f64 %s, %s, %s" % (dreg(m1), dreg(z + 2), s_.s()))
        e("v_mul_f64 %s, %s, %s" % (dreg(m2), dreg(z), s_.s()))
        e("v_fma_f64 %s, %s, %s, %s%s" % (dreg(z), dreg(z), c.s(), "-" if S > 0 else "", dreg(m1)))
        e("v_fma_f64 %s, %s, %s, %s%s" % (dreg(z + 2), dreg(z + 2), c.s(), "" if S > 0 else "-", dreg(m2)))

    def dft4(self, zbase, S, pool_regs):
        """radix-4 butterfly on the four quads at zbase, in place (output r in quad r)"""
        pool = G.Pool(pool_regs)
        st = G.Stream(self, pool)
        st.g = self
        z = [C(V(zbase + 4 * q), V(zbase + 4 * q + 2)) for q in range(4)]
        outs = [(zbase + 4 * q, zbase + 4 * q + 2) for q in range(4)]
        G.dft4(st, z, S, outs)
        self.out.extend(st.ins)

    # ---- hand-over between stages ----
    def ho_write(self, zbase, lv, s, e):
        reg = V_PA[e] if s == e else V_PB[e]
        for r in range(4):
            self.e("ds_write_b128 v%d, %s offset:%d" % (reg, qreg(zbase + 4 * r), lv * K_FFT_BYTES + r * slot_stride_bytes(e, s)))

    def ho_read(self, zbase, lv, s, e):
        reg = V_PA[e] if s == e else V_PB[e]
        for q in range(4):
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(zbase + 4 * q), reg, lv * K_FFT_BYTES + q * slot_stride_bytes(e, s)))

    def tw_read(self, s, regs):
        """the three twiddles of stage s (1 or 2) into three quads"""
        for r in range(3):
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(regs[r]), V_TWADDR, ((s - 1) * 3 + r) * 1024))

    # ---- the step ----
    def rotate_decompose(self):
        """d = acc * X^a - acc for this wave's polynomial (lane L: coefficients L + 64 m), both digits as doubles.  Same
        scheme as the wide loop (gen.py: rotate_decompose): registers and rotation buffer hold the negated accumulator, one
        mask per lane and step marks the coefficients whose index wrapped past 511 (read address - 2048 bytes, sign
        flipped), and the top digit is produced negated (mac_block flips the sign of the level-1 spectra)."""
        e = self.e
        fw, fs, a0, tmp = V_SCR[0], V_SCR[1], V_SCR[2], V_SCR[3]
        rk = [R_T + m for m in range(8)]
        e("v_subrev_u32 v%d, s%d, v%d" % (tmp, S_A, V_L))                # r = L - a
        e("v_not_b32 v%d, v%d" % (fw, tmp))
        e("v_bfe_u32 v%d, v%d, 6, 3" % (fw, fw))                          # (511 - (r & 511)) >> 6
        e("v_lshlrev_b32 v%d, v%d, -2" % (fw, fw))                        # bit m set: index wrapped
        e("v_bfe_i32 v%d, v%d, 9, 1" % (fs, tmp))
        e("v_xnor_b32 v%d, v%d, v%d" % (fs, fw, fs))                      # bit m set: sign = +1, the negated n[src] enters
        e("v_and_b32 v%d, 0x1ff, v%d" % (tmp, tmp))
        e("v_lshl_add_u32 v%d, v%d, 2, v%d" % (a0, tmp, V_ROTBASE))
        for m in range(8):
            e("v_bfe_i32 v%d, v%d, %d, 1" % (tmp, fw, m))
            e("v_lshl_add_u32 v%d, v%d, 11, v%d" % (tmp, tmp, a0))
            e("ds_read_b32 v%d, v%d offset:%d" % (rk[m], tmp, 256 * m))
        if FP_DIGITS:
            self.digits_fp(fs, rk)
            return
        sg, c, y, s_, u, w = V_SCR[3], V_SCR[4], V_SCR[5], V_SCR[6], V_SCR[7], V_SCR[8]
        for m in range(8):
            v = rk[m]
            q, part = m & 3, 0 if m < 4 else 2
            e("v_bfe_i32 v%d, v%d, %d, 1" % (sg, fs, m))
            e("v_bfe_u32 v%d, v%d, %d, 1" % (c, fs, m))
            e("v_add3_u32 v%d, v%d, v%d, s%d" % (c, c, V_ACC + m, S_C2048))
            e("v_xad_u32 v%d, v%d, v%d, v%d" % (y, v, sg, c))
            e("v_lshrrev_b32 v%d, 12, v%d" % (s_, y))
            e("v_lshrrev_b32 v%d, 31, v%d" % (u, y))
            e("v_add3_u32 v%d, v%d, v%d, s%d" % (u, s_, u, S_C511))
            e("v_sub_u32 v%d, 0x3ff, v%d" % (w, u))
            e("v_bfe_i32 v%d, v%d, 10, 10" % (w, w))                      # -d1
            e("v_cvt_f64_i32 %s, v%d" % (dreg(R_Z[0] + 4 * q + part), w))
            e("v_and_b32 v%d, 0xfffffc00, v%d" % (u, u))
            e("v_sub_u32 v%d, v%d, v%d" % (u, s_, u))                    # d2
            e("v_cvt_f64_i32 %s, v%d" % (dreg(R_Z[1] + 4 * q + part), u))

    def digits_fp(self, fs, rk):
        """Both digits of a coefficient from four fp64 instructions.  x = (sign * acc[src] - acc + 2^11 + 2^31 - 2^21) & ~0xfff is
        written into the low word of a register pair whose high word holds 0x42700000, so the pair reads as the double
        2^40 + x 2^-12; V = pair - (2^40 + 2^19 - 2^9) is the rounded difference in units of 2^12, an integer in
        [-2^19 + 512, 2^19 + 512), the window in which the top digit is the tfhe-rs one, [-511, 512].  The low digit's rule
        (carry when the residue exceeds B/2, or equals B/2 and the state's top bit is set) is "V / 1024 to the nearest integer,
        ties toward zero": t = V (1 - 2^-21) / 1024 + 1.5 * 2^52 (the product is exact, one rounding at a unit of 1; the factor
        moves a tie by at least 2^-22 and nothing by 2^-10 or more), -d1 = 1.5 * 2^52 - t, d2 = V + 1024 (-d1), all exact.
        Two coefficients at a time, so that no instruction follows the one it depends on."""
        e = self.e
        if FPD_MODE == 2:
            # all eight coefficients side by side, stage by stage, with no scratch register: the sign mask sits in the low word
            # of the coefficient's level-1 slot until t takes the slot, the biased difference and then the pair
            # (x, 0x42700000) in its level-2 slot, whose high word is written per coefficient
            z = [(R_Z[0] + 4 * (m & 3) + (0 if m < 4 else 2), R_Z[1] + 4 * (m & 3) + (0 if m < 4 else 2)) for m in range(8)]
            for m in range(8):
                e("v_bfe_i32 v%d, v%d, %d, 1" % (z[m][0], fs, m))
            for m in range(8):
                e("v_bfe_u32 v%d, v%d, %d, 1" % (z[m][1], fs, m))
            for m in range(8):
                e("v_add3_u32 v%d, v%d, v%d, s%d" % (z[m][1], z[m][1], V_ACC + m, S_C2048))
            for m in range(8):
                e("v_xad_u32 v%d, v%d, v%d, v%d" % (z[m][1], rk[m], z[m][0], z[m][1]))
            for m in range(8):
                e("v_and_b32 v%d, 0xfffff000, v%d" % (z[m][1], z[m][1]))
                e("v_mov_b32 v%d, 0x42700000" % (z[m][1] + 1))
            for m in range(8):
                e("v_add_f64 %s, %s, -s[%d:%d]" % (dreg(z[m][1]), dreg(z[m][1]), S_DK, S_DK + 1))
            for m in range(8):
                e("v_fma_f64 %s, %s, s[%d:%d], %s" % (dreg(z[m][0]), dreg(z[m][1]), S_DC, S_DC + 1, dreg(V_MAGIC)))
            for m in range(8):
                e("v_add_f64 %s, %s, -%s" % (dreg(z[m][0]), dreg(V_MAGIC), dreg(z[m][0])))                     # -d1
            for m in range(8):
                e("v_fma_f64 %s, %s, s[%d:%d], %s" % (dreg(z[m][1]), dreg(z[m][0]), S_D1024, S_D1024 + 1, dreg(z[m][1])))   # d2
            return
        D = [V_SCR[5], V_SCR[7]]                   # v[24:25], v[26:27]; the high words are set once, in front of the loop
        sg, cy = [V_SCR[3], R_T + 8], [V_SCR[4], R_T + 9]
        for m0 in range(0, 8, 2):
            ms = (m0, m0 + 1)
            for j, m in enumerate(ms):
                e("v_bfe_i32 v%d, v%d, %d, 1" % (sg[j], fs, m))
                e("v_bfe_u32 v%d, v%d, %d, 1" % (cy[j], fs, m))
            for j, m in enumerate(ms):
                e("v_add3_u32 v%d, v%d, v%d, s%d" % (cy[j], cy[j], V_ACC + m, S_C2048))
            for j, m in enumerate(ms):
                e("v_xad_u32 v%d, v%d, v%d, v%d" % (cy[j], rk[m], sg[j], cy[j]))
            for j, m in enumerate(ms):
                e("v_and_b32 v%d, 0xfffff000, v%d" % (D[j], cy[j]))
            z = [(dreg(R_Z[0] + 4 * (m & 3) + (0 if m < 4 else 2)), dreg(R_Z[1] + 4 * (m & 3) + (0 if m < 4 else 2))) for m in ms]
            for j in range(2):
                e("v_add_f64 %s, %s, -s[%d:%d]" % (z[j][1], dreg(D[j]), S_DK, S_DK + 1))
            for j in range(2):
                e("v_fma_f64 %s, %s, s[%d:%d], %s" % (z[j][0], z[j][1], S_DC, S_DC + 1, dreg(V_MAGIC)))
            for j in range(2):
                e("v_add_f64 %s, %s, -%s" % (z[j][0], dreg(V_MAGIC), z[j][0]))                     # -d1
            for j in range(2):
                e("v_fma_f64 %s, %s, s[%d:%d], %s" % (z[j][1], z[j][0], S_D1024, S_D1024 + 1, z[j][1]))   # d2

    def forward(self):
        """both levels, interleaved stage by stage: one level's hand-over travels under the other's butterfly.
        Temporaries: the output-sum registers (free until the products) for the butterflies, v116-127 for the three
        stage twiddles, v20-23 (the integer work is over) for the complex products."""
        pool = list(range(R_OUT, R_OUT + 16, 2))
        twq = [R_T, R_T + 4, R_T + 8]
        scr = 20
        self.tw_read(1, twq)                               # stage 1's twiddles are on their way through the twist and stage 0
        for lv in (0, 1):
            for q in range(4):
                self.cmul(R_Z[lv] + 4 * q, R_TWIST + 4 * q, 1, scr)
        for s in range(4):
            for lv in (0, 1):
                z = R_Z[lv]
                self.dft4(z, 1, pool)
                if s == 3:
                    continue
                for r in range(3):
                    tw = R_TW0 + 4 * r if s == 0 else twq[r]
                    self.cmul(z + 4 * (r + 1), tw, 1, scr)
                if s == 1 and lv == 1:
                    self.tw_read(2, twq)                   # stage 1 has used its twiddles: the registers take stage 2's
                self.ho_write(z, lv, s, s)
                self.ho_read(z, lv, s + 1, s)

    def publish_and_own_term(self):
        e = self.e
        for lv in (0, 1):
            for q in range(4):
                e("ds_write_b128 v%d, %s offset:%d" % (V_SPECWR_CUR, qreg(R_Z[lv] + 4 * q), (lv * 4 + q) * 1024))
        e("s_waitcnt vmcnt(0)")                            # the key words of this step (loaded during the previous one)
        # own polynomial (k = 0): its spectra are still in registers; level 1 initialises the sums
        self.mac_block(R_Z[0], 0, 0, first=True)

    def mac_block(self, f, lv, k, first):
        """out[q] += f[q] * key[lv][k][q] for the four q: eight independent chains, two rounds.  The level-1 spectra (lv = 0)
        are those of the negated digits: their sign is flipped here."""
        e = self.e
        p, n = ("-", "") if lv == 0 else ("", "-")
        for q in range(4):
            orr, oi = dreg(R_OUT + 4 * q), dreg(R_OUT + 4 * q + 2)
            fx, key = dreg(f + 4 * q), R_KEY + ((lv * 4 + k) * 4 + q) * 4
            if first:
                e("v_mul_f64 %s, %s%s, %s" % (orr, p, fx, dreg(key)))
                e("v_mul_f64 %s, %s%s, %s" % (oi, p, fx, dreg(key + 2)))
            else:
                e("v_fma_f64 %s, %s%s, %s, %s" % (orr, p, fx, dreg(key), orr))
                e("v_fma_f64 %s, %s%s, %s, %s" % (oi, p, fx, dreg(key + 2), oi))
        for q in range(4):
            orr, oi = dreg(R_OUT + 4 * q), dreg(R_OUT + 4 * q + 2)
            fy, key = dreg(f + 4 * q + 2), R_KEY + ((lv * 4 + k) * 4 + q) * 4
            e("v_fma_f64 %s, %s%s, %s, %s" % (orr, n, fy, dreg(key + 2), orr))
            e("v_fma_f64 %s, %s%s, %s, %s" % (oi, p, fy, dreg(key), oi))

    def spec_read(self, lv, k):
        for q in range(4):
            self.e("ds_read_b128 %s, v%d offset:%d" % (qreg(R_Z[lv] + 4 * q), V_SPECRD_CUR[k - 1], (lv * 4 + q) * 1024))

    def others(self):
        """the other three polynomials' spectra land in the registers of the own data (consumed by now), half a
        polynomial at a time: the next half is requested as soon as the products have released its registers"""
        self.spec_read(0, 1)
        self.mac_block(R_Z[1], 1, 0, first=False)          # the own level-2 term covers the first reads' latency
        self.spec_read(1, 1)
        for k in (1, 2, 3):
            self.mac_block(R_Z[0], 0, k, first=False)
            if k < 3:
                self.spec_read(0, k + 1)
            self.mac_block(R_Z[1], 1, k, first=False)
            if k < 3:
                self.spec_read(1, k + 1)

    def key_loads(self, k):
        """next step's key words of polynomial slot k (both levels) replace the ones just consumed"""
        e = self.e
        for lv in (0, 1):
            j = lv * 4 + k
            for q in range(4):
                e("global_load_dwordx4 %s, v%d, s[%d:%d] offset:%d" % (qreg(R_KEY + (j * 4 + q) * 4), V_KOFF, key_sreg(j), key_sreg(j) + 1, q * 1024))

    def inverse(self):
        e = self.e
        z = R_OUT
        pool = [68, 70, 72, 74, 76, 78, 80, 82]           # the spectra registers are free again
        scr = 84
        twq = [88, 92, 96]
        tw1 = [R_T, R_T + 4, R_T + 8]
        self.tw_read(2, twq)
        self.tw_read(1, tw1)
        for s in (3, 2, 1, 0):
            self.key_loads(3 - s)                          # a quarter of the next step's key before each stage
            if s < 3:
                for r in range(3):
                    tw = R_TW0 + 4 * r if s == 0 else (twq[r] if s == 2 else tw1[r])
                    self.cmul(z + 4 * (r + 1), tw, -1, scr)
            self.dft4(z, -1, pool)
            if s > 0:
                self.ho_write(z, 0, s, s - 1)
                self.ho_read(z, 0, s - 1, s - 1)
        for q in range(4):
            self.cmul(z + 4 * q, R_TWIST + 4 * q, -1, scr)

    def accumulate(self):
        e = self.e
        t = 68
        for q in range(4):
            for part, m in ((0, q), (2, q + 4)):
                e("v_add_f64 %s, %s, %s" % (dreg(t), dreg(R_OUT + 4 * q + part), dreg(V_MAGIC)))
                e("v_sub_u32 v%d, v%d, v%d" % (V_ACC + m, V_ACC + m, t))              # the negated accumulator
                e("ds_write_addtid_b32 v%d offset:%d" % (V_ACC + m, 256 * m))            # M0 = this wave's rotation buffer
                t += 2

    def step(self):
        e = self.e
        self.rotate_decompose()
        self.forward()
        # this step's parity: spectrum addresses
        e("v_add_u32 v%d, s%d, v%d" % (V_SPECWR_CUR, S_PAR, V_SPECWR))
        for k in range(3):
            e("v_add_u32 v%d, s%d, v%d" % (V_SPECRD_CUR[k], S_PAR, V_SPECRD[k]))
        self.publish_and_own_term()
        e(";drain")                                        # every wave's spectrum stores are done before any wave reads them
        e("s_barrier")
        self.others()
        if K_SPEC_BUFFERS == 1:
            e("s_barrier")                                 # every wave has read the spectra before the next step's are stored
        # next rotation amount
        e("v_mov_b32 v%d, s%d" % (V_SCR[0], S_ACUR))
        e("ds_read_u16 v%d, v%d" % (V_AREAD, V_SCR[0]))
        self.advance_key_pointers()
        self.inverse()
        e("v_readfirstlane_b32 s%d, v%d" % (S_A, V_AREAD))
        self.accumulate()

    def advance_key_pointers(self):
        for j in range(8):
            self.e("s_add_u32 s%d, s%d, 0x%x" % (key_sreg(j), key_sreg(j), STEP_KEY_BYTES))
            self.e("s_addc_u32 s%d, s%d, 0" % (key_sreg(j) + 1, key_sreg(j) + 1))

    def build(self):
        e = self.e
        body = Lat.__new__(Lat)
        body.consts, body.out = self.consts, []
        body.step()
        mbits = struct.unpack("<Q", struct.pack("<d", MAGIC))[0]
        if FP_DIGITS:
            e("s_mov_b32 s%d, 0x%x" % (S_C2048, (0x800 + 0x80000000 - 0x200000) & 0xFFFFFFFF))
            for reg, val in ((S_DK, 2.0 ** 40 + 2.0 ** 19 - 2.0 ** 9), (S_DC, (1.0 - 2.0 ** -21) / 1024.0), (S_D1024, 1024.0)):
                bits = struct.unpack("<Q", struct.pack("<d", val))[0]
                e("s_mov_b32 s%d, 0x%x" % (reg, bits & 0xFFFFFFFF))
                e("s_mov_b32 s%d, 0x%x" % (reg + 1, bits >> 32))
            e("v_mov_b32 v%d, 0x42700000" % (V_SCR[5] + 1))
            e("v_mov_b32 v%d, 0x42700000" % (V_SCR[7] + 1))
        else:
            e("s_mov_b32 s%d, 0x800" % S_C2048)
            e("s_mov_b32 s%d, 0x1ff" % S_C511)
        e("v_readfirstlane_b32 s%d, v%d" % (S_TMP, V_ROTBASE))         # nothing else in this loop uses M0
        e("s_mov_b32 m0, s%d" % S_TMP)
        e("v_mov_b32 v%d, 0x%x" % (V_MAGIC, mbits & 0xFFFFFFFF))
        e("v_mov_b32 v%d, 0x%x" % (V_MAGIC + 1, mbits >> 32))
        assert not self.consts.vals, "this loop folds no scalar constants (all its factors are per-lane twiddles)"
        # twist e^{i pi (L + 64 q) / 512} and the stage-0 twiddles e^{2 pi i L (r + 1) / 256} from the table in global memory
        # (s[36:37]: [0, 256) = e^{i pi n / 512}, [256, 512) = e^{2 pi i k / 256}); they stay in registers for the whole loop
        for q in range(4):
            e("global_load_dwordx4 %s, v%d, s[%d:%d] offset:%d" % (qreg(R_TWIST + 4 * q), V_KOFF, S_TW, S_TW + 1, 1024 * q))
        e("s_add_u32 s%d, s%d, 0x1000" % (S_TW, S_TW))
        e("s_addc_u32 s%d, s%d, 0" % (S_TW + 1, S_TW + 1))
        for r in range(3):
            t = V_SCR[r]
            if r == 0:
                e("v_mov_b32 v%d, v%d" % (t, V_L))
            elif r == 1:
                e("v_lshlrev_b32 v%d, 1, v%d" % (t, V_L))
            else:
                e("v_lshl_add_u32 v%d, v%d, 1, v%d" % (t, V_L, V_L))
            e("v_and_b32 v%d, 0xff, v%d" % (t, t))
            e("v_lshlrev_b32 v%d, 4, v%d" % (t, t))
            e("global_load_dwordx4 %s, v%d, s[%d:%d]" % (qreg(R_TW0 + 4 * r), t, S_TW, S_TW + 1))
        # the first step's key words
        for k in range(4):
            self.key_loads(k)
        e("s_waitcnt vmcnt(0)")                            # the twiddles are used right away
        e("L_lstep%=:")
        self.out.extend(place_waits_lat(body.out))
        if K_SPEC_BUFFERS == 2:
            e("s_xor_b32 s%d, s%d, 0x%x" % (S_PAR, S_PAR, K_SPEC_BYTES))
        e("s_add_u32 s%d, s%d, 2" % (S_ACUR, S_ACUR))
        e("s_sub_u32 s%d, s%d, 1" % (S_STEPS, S_STEPS))
        e("s_cmp_lg_u32 s%d, 0" % S_STEPS)
        e("s_cbranch_scc1 L_lstep%=")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return self.out


def place_waits_lat(lines):
    """gen.place_waits() with this kernel's conventions: the key loads (vmcnt) are waited for by the explicit
    s_waitcnt vmcnt(0) in front of the products and are not tracked; a barrier needs no LDS drain here (the
    spectra it publishes are stores)."""
    saved, saved_opt = G.V_KADDR, G.OPT
    G.V_KADDR = 1 << 20          # no register: nothing counts as a ring read
    # no merging of waits here: with one wave per SIMD a wait that is taken earlier than needed is time lost
    # (measured: 2.31 against 2.28 ms per round of 256, 3.93 against 3.88 ms per round of 512)
    G.OPT = G.OPT - {"wait_merge"}
    try:
        out = G.place_waits(list(lines))
    finally:
        G.V_KADDR, G.OPT = saved, saved_opt
    return ["s_waitcnt lgkmcnt(0)" if ln == ";drain" else ln for ln in out]


def experiment_filter(lines, flags):
    """timing experiments (results become garbage): nokey drops the key loads of the loop, nobar the barrier, nowait the
    LDS waits, noho the hand-overs between stages, nospec the reads of the other waves' spectra; duptw keeps the results and
    issues every twiddle read twice"""
    out, in_loop = [], False
    for ln in lines:
        op = ln.split()[0]
        if ln.endswith(":"):
            in_loop = True
        if in_loop and "nokey" in flags and op == "global_load_dwordx4":
            continue
        if "nobar" in flags and op == "s_barrier":
            continue
        if "nowait" in flags and ln.startswith("s_waitcnt lgkmcnt("):
            continue
        if "noho" in flags and op in ("ds_write_b128", "ds_read_b128") and any(", v%d " % r in ln + " " or ln.startswith("ds_write_b128 v%d," % r) for r in V_PA + V_PB):
            continue
        if "nospec" in flags and op == "ds_read_b128" and any(", v%d " % r in ln + " " for r in V_SPECRD_CUR):
            continue
        out.append(ln)
        # duptw (results stay exact): every stage-twiddle read issued twice, to price one more LDS instruction
        if in_loop and "duptw" in flags and op == "ds_read_b128" and ", v%d " % V_TWADDR in ln + " ":
            out.append(ln)
    return out


def main():
    g = Lat()
    lines = g.build()
    flags = [f for f in os.environ.get("PBS_ASM_EXP", "").split(",") if f]
    if flags:
        lines = experiment_filter(lines, flags)
    v, s = G.clobbers(lines)
    inputs_v = set([V_ROTBASE, V_L] + V_PA + V_PB + [V_SPECWR] + V_SPECRD + [V_KOFF, V_TWADDR]) | set(range(V_ACC, V_ACC + 8))
    inputs_s = set([S_STEPS, S_ACUR, S_A, S_PAR, S_TW, S_TW + 1]) | set(key_sreg(j) + h for j in range(8) for h in (0, 1))
    assert 0 not in v and not (set(range(0, 8)) & set(s)), "v0 / s0-s7 belong to the compiler"
    cl = ['"v%d"' % r for r in v if r not in inputs_v] + ['"s%d"' % r for r in s if r not in inputs_s]
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "..", "tfhe-aes_amd", "csrc", "pbs_lat_asm_loop.inc")
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/pbs_asm/lat_gen.py - do not edit.  %d instructions per blind-rotation step.\n" % len([x for x in lines if not x.endswith(":")]))
        f.write("#define TFA_LAT_ASM%s_BODY \\\n" % ("" if PER_CU == 1 else "2"))
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("#define TFA_LAT_ASM%s_CLOBBERS %s\n" % ("" if PER_CU == 1 else "2", ", ".join(cl + ['"memory"', '"scc"', '"vcc"'])))
    stats = {}
    for ln in lines:
        op = ln.split()[0]
        stats[op] = stats.get(op, 0) + 1
    print("lines", len(lines), "VALU", sum(n for op, n in stats.items() if op.startswith("v_")), "f64", sum(n for op, n in stats.items() if "f64" in op),
          "ds", sum(n for op, n in stats.items() if op.startswith("ds_")), "global", stats.get("global_load_dwordx4", 0))
    return lines


if __name__ == "__main__":
    main()
