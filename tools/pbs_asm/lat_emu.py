#!/usr/bin/env python3
"""CPU execution of the generated step loop of pbs_lat_asm_kernel (tools/pbs_asm/lat_gen.py) against the exact oracle.
TEST INFRASTRUCTURE, the twin of emu.py: one workgroup = one bootstrap over four waves, numpy over the lanes, LDS reads
and key loads delivered only at the s_waitcnt that guarantees them.

The key image (layout bsk_lat: [i][level][p][c][q][L] c64, (q, L) = slot and lane holding a frequency after the forward
transform, 1/256 folded in) is produced by running the generator's own forward transform on the key polynomials, which
is what bsk_convert_lat_kernel does with the compiled one.

usage: lat_emu.py [steps]   -> exit status 0 when the accumulator after `steps` external products equals
                              oracle.blind_rotate (exact engine) word for word
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import emu as E  # noqa: E402
import lat_gen as LG  # noqa: E402

E.W = 4
W, L = 4, 64


def set_double(M, reg, x):
    M.dset(reg, np.broadcast_to(x, (W, L)))


def load_constants(M):
    """per-lane address registers and twiddles, as the kernel's prologue sets them (wave w, lane L)"""
    lane = np.arange(L)
    for w in range(W):
        wl = w * LG.K_WAVE_BYTES
        M.v[LG.V_ROTBASE][w] = wl
        M.v[LG.V_L][w] = lane
        for e in range(3):
            M.v[LG.V_PA[e]][w] = wl + 16 * np.array([LG.exch_pos(e, e, int(l), 0) for l in lane])
            M.v[LG.V_PB[e]][w] = wl + 16 * np.array([LG.exch_pos(e, e + 1, int(l), 0) for l in lane])
        M.v[LG.V_SPECWR][w] = LG.K_RED_OFFSET + 16 * (w * 8 * 64 + lane)
        for k in range(3):
            M.v[LG.V_SPECRD[k]][w] = LG.K_RED_OFFSET + 16 * ((((w + k + 1) & 3) * 8) * 64 + lane)
        M.v[LG.V_KOFF][w] = 16 * lane
        M.v[LG.V_TWADDR][w] = LG.K_TW_OFFSET + 16 * lane
    tw = np.zeros((W, L, 7), np.complex128)
    n = lane
    twist = [np.exp(1j * np.pi * (n + 64 * q) / 512) for q in range(4)]
    st0 = [np.exp(2j * np.pi * ((n * (r + 1)) & 255) / 256) for r in range(3)]
    for q in range(4):
        set_double(M, LG.R_TWIST + 4 * q, twist[q].real)
        set_double(M, LG.R_TWIST + 4 * q + 2, twist[q].imag)
    for r in range(3):
        set_double(M, LG.R_TW0 + 4 * r, st0[r].real)
        set_double(M, LG.R_TW0 + 4 * r + 2, st0[r].imag)
    table = np.zeros((2, 3, L), np.complex128)
    for s in (1, 2):
        quarter, scale = 64 >> (2 * s), 1 << (2 * s)
        o = lane & (quarter - 1)
        for r in range(3):
            table[s - 1, r] = np.exp(2j * np.pi * ((o * (r + 1) * scale) & 255) / 256)
    M.lds[LG.K_TW_OFFSET:LG.K_TW_OFFSET + 6 * 1024] = table.view(np.float64).reshape(-1).view(np.uint8)


def forward_program():
    g = LG.Lat()
    g.forward()
    lines = LG.place_waits_lat(g.out) + ["s_waitcnt lgkmcnt(0)"]
    return [ln.replace("%=", "") for ln in lines]


def forward_polys(polys):
    """polys: [n][512] float64 coefficient vectors (n a multiple of 8) -> [n][4][64] complex, slot (q, L)"""
    prog = forward_program()
    out = np.zeros((len(polys), 4, L), np.complex128)
    for base in range(0, len(polys), 8):
        M = E.Machine(np.zeros(16, np.uint8), LG.K_LDS_BYTES)
        load_constants(M)
        for w in range(W):
            for lv in (0, 1):
                p = polys[base + 2 * w + lv]
                for q in range(4):
                    M.dset_wave = None
                    re, im = p[np.arange(L) + 64 * q], p[np.arange(L) + 64 * q + 256]
                    u = np.ascontiguousarray(re).view(np.uint64)
                    M.v[LG.R_Z[lv] + 4 * q][w] = (u & 0xFFFFFFFF).astype(np.uint32)
                    M.v[LG.R_Z[lv] + 4 * q + 1][w] = (u >> 32).astype(np.uint32)
                    u = np.ascontiguousarray(im).view(np.uint64)
                    M.v[LG.R_Z[lv] + 4 * q + 2][w] = (u & 0xFFFFFFFF).astype(np.uint32)
                    M.v[LG.R_Z[lv] + 4 * q + 3][w] = (u >> 32).astype(np.uint32)
        M.run(prog)
        for w in range(W):
            for lv in (0, 1):
                for q in range(4):
                    out[base + 2 * w + lv, q] = M.dget(LG.R_Z[lv] + 4 * q)[w] + 1j * M.dget(LG.R_Z[lv] + 4 * q + 2)[w]
    return out


def key_image(bsk, n_steps):
    """bsk_lat for the first n_steps steps plus one step of padding (the loop loads one step ahead)"""
    b = bsk.reshape(-1, 2, 4, 4, 512).view(np.int32).astype(np.float64)[:n_steps]
    spec = forward_polys(b.reshape(-1, 512)) / 256.0
    img = np.zeros((n_steps + 1, 2, 4, 4, 4, L), np.complex128)
    img[:n_steps] = spec.reshape(n_steps, 2, 4, 4, 4, L)
    return img.view(np.float64).reshape(-1).view(np.uint8)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    from oracle.pyoracle import Oracle
    o = Oracle()
    lwe_sk, glwe_sk, bsk, ksk = o.keygen(0x5446484541455321)
    ct = o.encrypt_bits(lwe_sk, 77, 0, np.array([1], np.uint8))[0]
    ct[1] = 0      # a mask word that modulus-switches to zero
    n = 805
    key = key_image(bsk, steps)
    tw_base = len(key)
    k = np.arange(256)
    tw_lat = np.concatenate([np.exp(1j * np.pi * k / 512), np.exp(2j * np.pi * k / 256)])
    key = np.concatenate([key, tw_lat.view(np.float64).view(np.uint8)])
    M = E.Machine(key, LG.K_LDS_BYTES)
    load_constants(M)
    lane = np.arange(L)
    a_tilde = ((((ct >> 21) + 1) >> 1) & 1023).astype(np.uint32)
    b_tilde = int(a_tilde[n])
    at = a_tilde.astype(np.uint16)
    M.lds[LG.K_A_OFFSET:LG.K_A_OFFSET + 2 * at.size] = at.view(np.uint8)
    for w in range(W):
        for m in range(8):
            jj = 64 * m + lane + b_tilde
            # the loop keeps the accumulator negated, in its registers and in the rotation buffer
            acc = np.uint32(0) - np.where(w == 3, np.where((jj >> 9) & 1, 0xE0000000, 0x20000000), 0).astype(np.uint32)
            M.v[LG.V_ACC + m][w] = acc
            addr = w * LG.K_WAVE_BYTES + 4 * lane + 256 * m
            for bb in range(4):
                M.lds[addr + bb] = (acc >> (8 * bb)) & 0xFF
        M.s[LG.S_STEPS][w] = steps
        M.s[LG.S_ACUR][w] = LG.K_A_OFFSET + 2
        M.s[LG.S_A][w] = a_tilde[0]
        M.s[LG.S_PAR][w] = 0
        M.s[LG.S_TW][w] = tw_base
        M.s[LG.S_TW + 1][w] = 0
        for lv in range(2):
            for k in range(4):
                p = (w + k) & 3
                off = ((lv * 4 + p) * 4 + w) * 4 * 64 * 16
                M.s[LG.key_sreg(lv * 4 + k)][w] = off
                M.s[LG.key_sreg(lv * 4 + k) + 1][w] = 0
    lines = [ln.replace("%=", "") for ln in LG.Lat().build()]
    M.run(lines)
    ref = o.blind_rotate(bsk, None, ct, steps)       # [4][512]
    got = np.zeros((4, 512), np.uint32)
    for w in range(W):
        for m in range(8):
            got[w, 64 * m + lane] = np.uint32(0) - M.v[LG.V_ACC + m][w]
    bad = int((got != ref).sum())
    if bad:
        d = (got.astype(np.int64) - ref.astype(np.int64) + 2**31) % 2**32 - 2**31
        print(f"{bad} of 2048 words differ, max |diff| {np.abs(d).max()}")
    print("steps", steps, "mismatching words", bad)
    print({k: v for k, v in sorted(M.count.items())})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
