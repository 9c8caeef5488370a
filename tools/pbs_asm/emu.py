#!/usr/bin/env python3
"""CPU execution of the generated blind-rotation loop (tools/pbs_asm/gen.py) against the exact-integer oracle.

Interprets the emitted gfx950 instruction text for one workgroup (8 waves x 64 lanes in lock step, numpy over lanes),
with the LDS, the key ring's LDS-DMA and the barrier modelled functionally (strict about waits: reads and DMA rows take effect only at the s_waitcnt that guarantees them).
TEST INFRASTRUCTURE: run by hand / by tests while the loop is being changed; nothing in the product imports it.

usage: emu.py [steps]   -> compares the accumulators of 8 bootstraps after `steps` external products with
                           oracle.blind_rotate (exact engine); exit status 0 when every word is equal.
"""
import os
import re
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen as G  # noqa: E402

W, L = G.WAVES, 64


class Machine:
    def __init__(self, key_bytes, lds_bytes=None):
        self.v = np.zeros((256, W, L), np.uint32)
        self.s = np.zeros((110, W), np.uint32)
        self.m0 = np.zeros(W, np.uint32)
        self.scc = np.zeros(W, bool)
        self.lds = np.zeros((lds_bytes or G.K_LDS_BYTES) + 64, np.uint8)
        self.key = key_bytes          # global memory image of the Fourier key; addresses are offsets into it
        self.count = {}
        # strict timing model: an LDS read delivers its registers, and an LDS-DMA row lands in the LDS, only at the
        # s_waitcnt that guarantees it (the latest moment the hardware allows); touching a register whose read is
        # still in flight is an error
        self.lds_queue = []           # per LDS operation in flight: None (a write) or (first register, words [k][W][L])
        self.dma_queue = []           # (destination byte indices, data)
        self.vm_queue = []            # global loads into registers in flight: (first register, words)

    # ---- operand access ----
    def dget(self, r):
        lo, hi = self.v[r].astype(np.uint64), self.v[r + 1].astype(np.uint64)
        return (lo | (hi << np.uint64(32))).view(np.float64)

    def dset(self, r, x):
        u = np.ascontiguousarray(x, np.float64).view(np.uint64)
        self.v[r] = (u & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        self.v[r + 1] = (u >> np.uint64(32)).astype(np.uint32)

    def src64(self, tok):
        neg = tok.startswith("-")
        if neg:
            tok = tok[1:]
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            x = self.dget(int(m.group(1)))
        else:
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
            if m:
                r = int(m.group(1))
                u = self.s[r].astype(np.uint64) | (self.s[r + 1].astype(np.uint64) << np.uint64(32))
                x = np.repeat(u.view(np.float64)[:, None], L, axis=1)
            else:
                x = np.full((W, L), float(tok))
        return -x if neg else x

    def src32(self, tok):
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return self.v[int(m.group(1))]
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return np.repeat(self.s[int(m.group(1))][:, None], L, axis=1)
        return np.full((W, L), int(tok, 0) & 0xFFFFFFFF, np.uint32)

    def ssrc(self, tok):
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return self.s[int(m.group(1))].copy()
        if tok == "m0":
            return self.m0.copy()
        return np.full(W, int(tok, 0) & 0xFFFFFFFF, np.uint32)

    def sdst(self, tok, val):
        if tok == "m0":
            self.m0 = val.astype(np.uint32)
        else:
            self.s[int(tok[1:])] = val.astype(np.uint32)

    # ---- LDS ----
    def lds_read(self, addr, nbytes):
        idx = addr.astype(np.int64)[..., None] + np.arange(nbytes)
        return self.lds[idx]

    def lds_write(self, addr, data):
        idx = addr.astype(np.int64)[..., None] + np.arange(data.shape[-1])
        self.lds[idx] = data

    def run(self, lines):
        labels = {ln[:-1]: k for k, ln in enumerate(lines) if ln.endswith(":")}
        pc = 0
        while pc < len(lines):
            ln = lines[pc]
            pc += 1
            if ln.endswith(":"):
                continue
            op, _, rest = ln.partition(" ")
            self.count[op] = self.count.get(op, 0) + 1
            a = [x.strip() for x in rest.split(",")] if rest else []
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m:
                    keep = int(m.group(1))
                    done, self.lds_queue = self.lds_queue[:len(self.lds_queue) - keep], self.lds_queue[len(self.lds_queue) - keep:] if keep else []
                    for item in done:
                        if item is not None:
                            for k, words in enumerate(item[1]):
                                self.v[item[0] + k] = words
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    keep = int(m.group(1))
                    done, self.dma_queue = self.dma_queue[:len(self.dma_queue) - keep], self.dma_queue[len(self.dma_queue) - keep:] if keep else []
                    for dst, data in done:
                        self.lds[dst] = data
                    # (a kernel uses either LDS-DMA rows or register loads, so one vmcnt serves both queues here)
                    done, self.vm_queue = self.vm_queue[:len(self.vm_queue) - keep], self.vm_queue[len(self.vm_queue) - keep:] if keep else []
                    for r0, words in done:
                        for k, wd in enumerate(words):
                            self.v[r0 + k] = wd
                continue
            if op == "s_barrier":
                if G.RING_AHEAD == 3:     # the slot refilled after this barrier is the one the youngest key reads came from
                    assert not any(item is not None and item[2] for item in self.lds_queue), "key reads in flight at a barrier: " + str(pc)
                continue
            if op == "s_nop":
                continue
            busy = set()
            for item in self.lds_queue:
                if item is not None:
                    busy.update(range(item[0], item[0] + len(item[1])))
            for r0, words in self.vm_queue:
                busy.update(range(r0, r0 + len(words)))
            if busy:
                used = {r for (kind, r) in G.regs_of(rest) if kind == "v"}
                if op.startswith("global_load_dwordx4"):
                    used = {r for (kind, r) in G.regs_of(a[1]) if kind == "v"}
                if op.startswith("ds_read"):     # a younger read may overwrite an older one's destination: the LDS returns in order
                    used = {r for (kind, r) in G.regs_of(a[1]) if kind == "v"}
                assert not (busy & used), "line %d touches v%s while its LDS read is in flight: %s" % (pc, sorted(busy & used), ln)
            if op == "s_mov_b32":
                self.sdst(a[0], self.ssrc(a[1]))
            elif op == "v_mov_b32":
                self.v[int(a[0][1:])] = self.src32(a[1])
            elif op in ("s_add_u32", "s_sub_u32"):
                x, y = self.ssrc(a[1]).astype(np.uint64), self.ssrc(a[2]).astype(np.uint64)
                r = x + y if op == "s_add_u32" else x - y
                self.scc = (r >> np.uint64(32)) != 0
                self.sdst(a[0], r & np.uint64(0xFFFFFFFF))
            elif op == "s_addc_u32":
                r = self.ssrc(a[1]).astype(np.uint64) + self.ssrc(a[2]).astype(np.uint64) + self.scc.astype(np.uint64)
                self.scc = (r >> np.uint64(32)) != 0
                self.sdst(a[0], r & np.uint64(0xFFFFFFFF))
            elif op == "s_cmp_lg_u32":
                self.scc = self.ssrc(a[0]) != self.ssrc(a[1])
            elif op == "s_cbranch_scc1":
                assert self.scc.all() or not self.scc.any(), "waves diverged"
                if self.scc.all():
                    pc = labels[a[0]]
            elif op == "v_add_u32":
                self.v[int(a[0][1:])] = self.src32(a[1]) + self.src32(a[2])
            elif op == "v_sub_u32":
                self.v[int(a[0][1:])] = self.src32(a[1]) - self.src32(a[2])
            elif op == "v_subrev_u32":
                self.v[int(a[0][1:])] = self.src32(a[2]) - self.src32(a[1])
            elif op == "v_and_b32":
                self.v[int(a[0][1:])] = self.src32(a[1]) & self.src32(a[2])
            elif op == "v_xor_b32":
                self.v[int(a[0][1:])] = self.src32(a[1]) ^ self.src32(a[2])
            elif op == "v_xnor_b32":
                self.v[int(a[0][1:])] = ~(self.src32(a[1]) ^ self.src32(a[2]))
            elif op == "v_not_b32":
                self.v[int(a[0][1:])] = ~self.src32(a[1])
            elif op == "v_xad_u32":
                self.v[int(a[0][1:])] = (self.src32(a[1]) ^ self.src32(a[2])) + self.src32(a[3])
            elif op == "v_lshl_add_u32":
                self.v[int(a[0][1:])] = (self.src32(a[1]) << self.src32(a[2])) + self.src32(a[3])
            elif op == "v_lshlrev_b32":
                self.v[int(a[0][1:])] = self.src32(a[2]) << self.src32(a[1])
            elif op == "v_lshrrev_b32":
                self.v[int(a[0][1:])] = self.src32(a[2]) >> self.src32(a[1])
            elif op == "v_add3_u32":
                self.v[int(a[0][1:])] = self.src32(a[1]) + self.src32(a[2]) + self.src32(a[3])
            elif op in ("v_bfe_u32", "v_bfe_i32"):
                x, off, wid = self.src32(a[1]), int(a[2]), int(a[3])
                f = (x >> np.uint32(off)) & np.uint32((1 << wid) - 1)
                if op == "v_bfe_i32":
                    f = np.where(f >> np.uint32(wid - 1), f | np.uint32((0xFFFFFFFF << wid) & 0xFFFFFFFF), f)
                self.v[int(a[0][1:])] = f
            elif op == "v_cvt_f64_i32":
                self.dset(int(re.match(r"v\[(\d+)", a[0]).group(1)), self.src32(a[1]).view(np.int32).astype(np.float64))
            elif op == "v_readfirstlane_b32":
                self.s[int(a[0][1:])] = self.v[int(a[1][1:])][:, 0]
            elif op == "v_add_f64":
                self.dset(int(re.match(r"v\[(\d+)", a[0]).group(1)), self.src64(a[1]) + self.src64(a[2]))
            elif op == "v_mul_f64":
                self.dset(int(re.match(r"v\[(\d+)", a[0]).group(1)), self.src64(a[1]) * self.src64(a[2]))
            elif op == "v_fma_f64":
                # not a fused operation here (numpy has none): the last bit may differ from the hardware, the rounded
                # torus values do not
                self.dset(int(re.match(r"v\[(\d+)", a[0]).group(1)), self.src64(a[1]) * self.src64(a[2]) + self.src64(a[3]))
            elif op == "v_mov_b32_dpp":
                m = re.match(r"v(\d+), v(\d+) quad_perm:\[(\d),(\d),(\d),(\d)\]", rest)
                d, s_ = int(m.group(1)), int(m.group(2))
                perm = [int(m.group(k)) for k in range(3, 7)]
                lanes = np.arange(L)
                src_lane = (lanes & ~3) | np.array(perm)[lanes & 3]
                self.v[d] = self.v[s_][:, src_lane]
            elif op in ("ds_read_b32", "ds_read_b128", "ds_read_u16"):
                dst, addr_tok = a[0], a[1].split()[0]
                off = int(re.search(r"offset:(\d+)", rest).group(1)) if "offset:" in rest else 0
                addr = self.src32(addr_tok) + np.uint32(off)
                if op == "ds_read_u16":
                    b = self.lds_read(addr, 2)
                    self.lds_queue.append((int(dst[1:]), [b[..., 0].astype(np.uint32) | (b[..., 1].astype(np.uint32) << 8)], False))
                else:
                    n = 4 if op == "ds_read_b32" else 16
                    assert (addr % n == 0).all(), "misaligned LDS read"
                    b = self.lds_read(addr, n).reshape(W, L, n // 4, 4).astype(np.uint32)
                    words = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24)
                    r0 = int(re.match(r"v\[?(\d+)", dst).group(1))
                    self.lds_queue.append((r0, [words[..., k].copy() for k in range(n // 4)], addr_tok == "v%d" % G.V_KADDR, (int(addr.min()), int(addr.max()) + n)))
                    continue
            elif op in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
                addr_tok, data_tok = a[0], a[1].split()[0]
                off = int(re.search(r"offset:(\d+)", rest).group(1)) if "offset:" in rest else 0
                addr = self.src32(addr_tok) + np.uint32(off)
                n = {"ds_write_b32": 4, "ds_write_b64": 8, "ds_write_b128": 16}[op]
                assert (addr % n == 0).all(), "misaligned LDS write"
                r0 = int(re.match(r"v\[?(\d+)", data_tok).group(1))
                words = np.stack([self.v[r0 + k] for k in range(n // 4)], axis=-1)
                by = np.stack([(words >> np.uint32(8 * j)) & np.uint32(0xFF) for j in range(4)], axis=-1).astype(np.uint8)
                self.lds_write(addr, by.reshape(W, L, n))
                self.lds_queue.append(None)
            elif op == "ds_write2_b64":
                addr_tok = a[0]
                o0, o1 = int(re.search(r"offset0:(\d+)", rest).group(1)), int(re.search(r"offset1:(\d+)", rest).group(1))
                datas = [int(re.match(r"v\[(\d+)", a[1]).group(1)), int(re.match(r"v\[(\d+)", a[2].split()[0]).group(1))]
                for off, r0 in zip((o0, o1), datas):
                    addr = self.src32(addr_tok) + np.uint32(8 * off)
                    assert (addr % 8 == 0).all(), "misaligned LDS write"
                    words = np.stack([self.v[r0], self.v[r0 + 1]], axis=-1)
                    by = np.stack([(words >> np.uint32(8 * j)) & np.uint32(0xFF) for j in range(4)], axis=-1).astype(np.uint8)
                    self.lds_write(addr, by.reshape(W, L, 8))
                self.lds_queue.append(None)
            elif op == "ds_write_addtid_b32":
                off = int(re.search(r"offset:(\d+)", rest).group(1)) if "offset:" in rest else 0
                assert (self.m0 < 65536).all(), "the M0-relative store takes its base from M0[15:0]"
                addr = self.m0[:, None] + np.uint32(off) + np.uint32(4) * np.arange(L, dtype=np.uint32)[None, :]
                words = self.v[int(re.match(r"v(\d+)", a[0]).group(1))]
                by = np.stack([(words >> np.uint32(8 * j)) & np.uint32(0xFF) for j in range(4)], axis=-1).astype(np.uint8)
                self.lds_write(addr, by.reshape(W, L, 4))
                self.lds_queue.append(None)
            elif op == "s_xor_b32":
                self.sdst(a[0], self.ssrc(a[1]) ^ self.ssrc(a[2]))
            elif op == "global_load_dwordx4":
                # into registers: delivered at the s_waitcnt vmcnt that guarantees it, like an LDS read
                r0 = int(re.match(r"v\[(\d+)", a[0]).group(1))
                voff = self.src32(a[1]).astype(np.int64)
                m = re.search(r"s\[(\d+):(\d+)\]", a[2])
                r = int(m.group(1))
                base = self.s[r].astype(np.int64) | (self.s[r + 1].astype(np.int64) << 32)
                off = int(re.search(r"offset:(\d+)", rest).group(1)) if "offset:" in rest else 0
                src = base[:, None] + voff + off
                by = self.key[src[..., None] + np.arange(16)].reshape(W, L, 4, 4).astype(np.uint32)
                words = by[..., 0] | (by[..., 1] << 8) | (by[..., 2] << 16) | (by[..., 3] << 24)
                self.vm_queue.append((r0, [words[..., k].copy() for k in range(4)]))
            elif op == "global_load_lds_dwordx4":
                voff = self.src32(a[0]).astype(np.int64)
                m = re.match(r"s\[(\d+):(\d+)\]", a[1])
                r = int(m.group(1))
                base = self.s[r].astype(np.int64) | (self.s[r + 1].astype(np.int64) << 32)
                off = int(re.search(r"offset:(\d+)", rest).group(1)) if "offset:" in rest else 0   # moves both addresses
                src = base[:, None] + voff + off
                data = self.key[src[..., None] + np.arange(16)]
                dst = self.m0.astype(np.int64)[:, None] + off + np.arange(L) * 16
                lo_, hi_ = int(dst.min()), int(dst.max()) + 16
                for item in self.lds_queue:      # (the waves run in lock step here, so this also covers the other waves' reads)
                    assert item is None or len(item) < 4 or item[3][1] <= lo_ or item[3][0] >= hi_, "LDS-DMA into a slot with reads in flight: line %d" % pc
                self.dma_queue.append((dst[..., None] + np.arange(16), data))
            else:
                raise NotImplementedError(ln)


def fourier_key_lanesum(bsk, n_steps):
    """Fourier key in the layout of the lane-local sum: [i][k_lo][level][j][j'][lane] c64, lane = 4 k1 + x: the word that
    multiplies slot set j (polynomial (x - j) & 3) into output slot j' (column (x - j') & 3) at bin k1 + 16 (k_lo + 4 x)."""
    out = np.zeros((n_steps + 1, 4, 2, 4, 4, 64), np.complex128)
    b = bsk.reshape(-1, 2, 4, 4, 512).view(np.int32).astype(np.float64)
    tw = np.exp(1j * np.pi * np.arange(256) / 512)
    for i in range(n_steps):
        for lv in range(2):
            F = np.zeros((4, 4, 256), np.complex128)
            for r in range(4):
                for c in range(4):
                    p = b[i, lv, r, c]
                    F[r, c] = np.fft.ifft((p[:256] + 1j * p[256:]) * tw)
            for k_lo in range(4):
                for j in range(4):
                    for jp in range(4):
                        for k1 in range(16):
                            for x in range(4):
                                out[i, k_lo, lv, j, jp, 4 * k1 + x] = F[(x - j) & 3, (x - jp) & 3, k1 + 16 * (k_lo + 4 * x)]
    return out.view(np.float64).reshape(-1).view(np.uint8)


def fourier_key(bsk, n_steps):
    """Fourier key of the first n_steps mask elements in the kernel's layout [i][level][p][slot][lane] c64 (the layout
    bsk_convert_kernel of pbs_kernel.h produces), padded by a step for the ring's prefetch."""
    if G.LANESUM:
        return fourier_key_lanesum(bsk, n_steps)
    out = np.zeros((n_steps + 1, 2, 16, 4, 64), np.complex128)
    b = bsk.reshape(-1, 2, 4, 4, 512).view(np.int32).astype(np.float64)
    j = np.arange(256)
    tw = np.exp(1j * np.pi * j / 512)
    for i in range(n_steps):
        for lv in range(2):
            for r in range(4):
                for c in range(4):
                    p = b[i, lv, r, c]
                    F = np.fft.ifft((p[:256] + 1j * p[256:]) * tw)     # = sum_j z_j e^{i pi j (4k+1)/512} / 256
                    for k2 in range(16):
                        for t in range(16):
                            out[i, lv, G.kperm(k2), c ^ r, 4 * t + r] = F[t + 16 * k2]
    return out.view(np.float64).reshape(-1).view(np.uint8)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    from oracle.pyoracle import Oracle
    o = Oracle()
    lwe_sk, glwe_sk, bsk, ksk = o.keygen(0x5446484541455321)
    bits = np.array([0, 1, 1, 0, 1, 0, 0, 1], np.uint8)[:W]
    cts = o.encrypt_bits(lwe_sk, 77, 0, bits)
    cts[2, 1] = 0      # a mask word that modulus-switches to zero
    n = 805
    key = fourier_key(bsk, steps)
    M = Machine(key)
    lane = np.arange(L)
    g, t = lane & 3, lane >> 2
    a_tilde = ((((cts >> 21) + 1) >> 1) & 1023).astype(np.uint32)     # [8][806]
    for w in range(W):
        wl, al = w * G.K_WAVE_LDS, W * G.K_WAVE_LDS + w * G.K_A_BYTES    # exchange / rotation buffer, a_tilde array
        M.v[G.V_ROTBASE][w] = wl + 4 * g
        M.v[G.V_T][w] = t
        for q in range(4):
            M.v[G.V_TRWR[q]][w] = wl + 8 * (g * G.K_TR_GROUP + (t ^ (2 * q)))
            M.v[G.V_TRRD[q]][w] = wl + 8 * (g * G.K_TR_GROUP + t * 16) + 16 * (q ^ (t >> 2))
        M.v[G.V_TW][w] = G.K_TW_OFFSET + t * (64 if G.LANESUM else 16)
        if G.LANESUM:     # rotation q of an entry sits at byte 16 q
            M.v[G.V_TWF][w] = G.K_TW_OFFSET + t * 64 + 16 * ((g * t) & 3)
        M.v[G.V_LANE16][w] = lane * 16
        M.v[G.V_KBASE][w] = lane * 16
        b_tilde = int(a_tilde[w, n])
        for m in range(32):
            jj = 16 * m + t + b_tilde
            # the loop keeps the accumulator negated, in its registers and in the rotation buffer
            M.v[G.V_ACC + m][w] = np.uint32(0) - np.where(g == 3, np.where((jj >> 9) & 1, 0xE0000000, 0x20000000), 0).astype(np.uint32)
        # accumulator into the rotation buffer, a_tilde into its LDS array
        for m in range(32):
            addr = wl + 256 * m + 4 * lane        # buffer word 4 (16 m + t) + g
            val = M.v[G.V_ACC + m][w]
            for bb in range(4):
                M.lds[addr + bb] = (val >> (8 * bb)) & 0xFF
        at = a_tilde[w].astype(np.uint16)
        M.lds[al: al + 2 * at.size] = at.view(np.uint8)
        rpw = G.ROWS_PER_WAVE
        r0 = rpw * w                           # this wave's first DMA row of a chunk: row = (position >> 1) << 3 | level << 2 | column
        kw = r0 * 1024 if G.LANESUM else ((r0 >> 2) & 1) * 65536 + (r0 >> 3) * 4096 + (r0 & 3) * 1024
        M.s[G.S_KP][w] = kw
        M.s[G.S_KP + 1][w] = 0
        M.s[G.S_STEPS][w] = steps
        M.s[G.S_ACUR][w] = al + 2
        M.s[G.S_A][w] = a_tilde[w, 0]
        M.s[G.S_M0ROW][w] = G.K_RING_OFFSET + r0 * 1024
        # the first chunks of step 0 (as many as the loop keeps ahead) are already on their way when the loop starts
        for ch in range(G.RING_AHEAD):
            for row in range(r0, r0 + rpw):
                src = kw + ch * (G.K_CHUNK_BYTES if G.LANESUM else 8192) + (row - r0) * 1024
                dst = G.K_RING_OFFSET + ch * G.K_CHUNK_BYTES + row * 1024
                M.lds[dst:dst + 1024] = key[src:src + 1024]
    if G.LANESUM:
        twt = np.zeros((16, 16, 4), np.complex128)
        for r in range(16):
            for tt in range(16):
                for q in range(4):
                    twt[r, tt, q] = np.exp(1j * np.pi * tt * (1 + 4 * r) / 512) * (1j ** q)
        M.lds[G.K_TW_OFFSET:G.K_TW_OFFSET + 16384] = twt.view(np.float64).reshape(-1).view(np.uint8)
    else:
        twt = np.zeros((16, 16), np.complex128)
        for r in range(16):
            for tt in range(16):
                twt[r, tt] = np.exp(1j * np.pi * tt * (1 + 4 * r) / 512)
        M.lds[G.K_TW_OFFSET:G.K_TW_OFFSET + 4096] = twt.view(np.float64).reshape(-1).view(np.uint8)

    lines = [ln.replace("%=", "") for ln in G.Loop().build()]
    M.run(lines)
    bad = 0
    for w in range(W):
        ref = o.blind_rotate(bsk, None, cts[w], steps)       # [4][512]
        got = np.zeros((4, 512), np.uint32)
        for m in range(32):
            got[g, 16 * m + t] = np.uint32(0) - M.v[G.V_ACC + m][w]
        diff = (got != ref).sum()
        bad += int(diff)
        if diff:
            d = (got.astype(np.int64) - ref.astype(np.int64) + 2**31) % 2**32 - 2**31
            print(f"bootstrap {w}: {diff} of 2048 words differ, max |diff| {np.abs(d).max()}; per polynomial {[int((got[q] != ref[q]).sum()) for q in range(4)]}")
    print("steps", steps, "mismatching words", bad)
    print({k: v for k, v in sorted(M.count.items())})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
