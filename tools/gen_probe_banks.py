#!/usr/bin/env python3
"""Generates tools/probe_banks.hip (not tracked): does the issue cost of a 64-bit vector instruction on gfx950 depend on
WHICH registers its operands sit in?  Every kernel runs a body of 64 instructions over independent accumulators with the
operand registers placed by a pattern (register number modulo 4 / 8, scalar operands, repeated operands, distance between
dependent instructions).  The program prints cycles per instruction from the wave's own cycle counter next to the figure
derived from the wall clock, at one and two waves per SIMD.

    python3 tools/gen_probe_banks.py && hipcc -O2 --offload-arch=gfx950 -o tools/probe_banks tools/probe_banks.hip
"""
import os

ACC0, SRC0 = 32, 96       # accumulators v32-95, sources v96-111: 112 registers, so that four waves fit a SIMD


def acc(j, stride, mod):
    return ACC0 + stride * j + mod


def d(r):
    return "v[%d:%d]" % (r, r + 1)


def fma_acc(nacc, stride, dmod, amod, bmod, aoff=0, boff=8):
    out = []
    for i in range(64):
        r = acc(i % nacc, stride, dmod)
        out.append("v_fma_f64 %s, %s, %s, %s" % (d(r), d(SRC0 + aoff + amod), d(SRC0 + boff + bmod), d(r)))
    return out


def fma_rot(nacc, stride, dmod):
    """d_j = d_{j+1} * src + d_j: two accumulator-region operands, as in a butterfly"""
    out = []
    for i in range(64):
        r, r2 = acc(i % nacc, stride, dmod), acc((i + nacc // 2) % nacc, stride, dmod)
        out.append("v_fma_f64 %s, %s, %s, %s" % (d(r), d(r2), d(SRC0), d(r)))
    return out


def fma_sgpr(nacc, stride, dmod, amod):
    out = []
    for i in range(64):
        r = acc(i % nacc, stride, dmod)
        out.append("v_fma_f64 %s, %s, s[20:21], %s" % (d(r), d(SRC0 + amod), d(r)))
    return out


def add_acc(nacc, stride, dmod, amod):
    out = []
    for i in range(64):
        r = acc(i % nacc, stride, dmod)
        out.append("v_add_f64 %s, %s, %s" % (d(r), d(r), d(SRC0 + amod)))
    return out


def add_two(nacc, stride, dmod, shift):
    """d_j = d_j + d_{j+shift}"""
    out = []
    for i in range(64):
        r, r2 = acc(i % nacc, stride, dmod), acc((i + shift) % nacc, stride, dmod)
        out.append("v_add_f64 %s, %s, %s" % (d(r), d(r), d(r2)))
    return out


def mul_acc(nacc, stride, dmod, amod):
    out = []
    for i in range(64):
        r = acc(i % nacc, stride, dmod)
        out.append("v_mul_f64 %s, %s, %s" % (d(r), d(r), d(SRC0 + amod)))
    return out


def rep(fmt, srcs=1):
    """64 copies of one instruction form over 16 destination registers; {d} destination, {a} {b} sources, {D} {A} 64-bit forms"""
    out = []
    for i in range(64):
        r = ACC0 + 2 * (i % 16)
        out.append(fmt.format(d="v%d" % r, a="v%d" % SRC0, b="v%d" % (SRC0 + 2), c="v%d" % (SRC0 + 5), D=d(r), A=d(SRC0), B=d(SRC0 + 2), m=i % 32))
    return out


CASES = [
    ("fma  d%4=0 a%4=0 b%4=0 (a,b%8=0)", fma_acc(16, 4, 0, 0, 0)),
    ("fma  d%4=0 a%4=0 b%4=2", fma_acc(16, 4, 0, 0, 2)),
    ("fma  d alternating, a%4=0 b%4=2", fma_acc(16, 2, 0, 0, 2)),
    ("fma  d=d2*src+d, d alternating", fma_rot(16, 2, 0)),
    ("fma  d=a*sgpr+d, d%4=0 a%4=0", fma_sgpr(16, 4, 0, 0)),
    ("fma  d=a*sgpr+d, d%4=0 a%4=2", fma_sgpr(16, 4, 0, 2)),
    ("fma  d=a*sgpr+d, d%4=2 a%4=0", fma_sgpr(16, 4, 2, 0)),
    ("fma  d=a*sgpr+d, d%4=2 a%4=2", fma_sgpr(16, 4, 2, 2)),
    ("fma  d=a*sgpr+d, d alternating", fma_sgpr(16, 2, 0, 0)),
    ("add  d%4=0 a%4=0", add_acc(16, 4, 0, 0)),
    ("add  d%4=0 a%4=2", add_acc(16, 4, 0, 2)),
    ("add  d alternating", add_acc(16, 2, 0, 0)),
    ("mul  d%4=0 a%4=2", mul_acc(16, 4, 0, 2)),
    ("add_f64 d, d, sgpr", rep("v_add_f64 {D}, {D}, s[20:21]")),
    ("mul_f64 d, d, sgpr", rep("v_mul_f64 {D}, {D}, s[20:21]")),
    ("cvt_f64_i32", rep("v_cvt_f64_i32 {D}, {a}")),
    ("cvt_f64_u32", rep("v_cvt_f64_u32 {D}, {a}")),
    ("cvt_f32_i32", rep("v_cvt_f32_i32 {d}, {a}")),
    ("cvt_f64_f32", rep("v_cvt_f64_f32 {D}, {a}")),
    ("v_fma_f32", rep("v_fma_f32 {d}, {a}, {b}, {d}")),
    ("v_add_f32", rep("v_add_f32 {d}, {a}, {d}")),
    ("v_pk_fma_f32", rep("v_pk_fma_f32 {D}, {A}, {B}, {D}")),
    ("v_mov_b32", rep("v_mov_b32 {d}, {a}")),
    ("v_mov_b32 literal", rep("v_mov_b32 {d}, 0x12345678")),
    ("v_not_b32", rep("v_not_b32 {d}, {a}")),
    ("v_add_u32 e32", rep("v_add_u32 {d}, {a}, {d}")),
    ("v_add_u32 e64", rep("v_add_u32_e64 {d}, {a}, {d}")),
    ("v_add_u32 inline const", rep("v_add_u32 {d}, 17, {d}")),
    ("v_add_u32 literal", rep("v_add_u32 {d}, 0x3ff00, {d}")),
    ("v_add_u32 sgpr", rep("v_add_u32 {d}, s20, {d}")),
    ("v_sub_u32", rep("v_sub_u32 {d}, {a}, {d}")),
    ("v_and_b32", rep("v_and_b32 {d}, {a}, {d}")),
    ("v_and_b32 literal", rep("v_and_b32 {d}, 0xfffffc00, {d}")),
    ("v_xor_b32", rep("v_xor_b32 {d}, {a}, {d}")),
    ("v_lshlrev_b32 const", rep("v_lshlrev_b32 {d}, 3, {d}")),
    ("v_lshrrev_b32 const", rep("v_lshrrev_b32 {d}, 12, {a}")),
    ("v_ashrrev_i32 const", rep("v_ashrrev_i32 {d}, 31, {a}")),
    ("v_cndmask_b32 vcc", rep("v_cndmask_b32 {d}, {a}, {b}, vcc")),
    ("v_mul_u32_u24", rep("v_mul_u32_u24 {d}, {a}, {d}")),
    ("v_mul_i32_i24", rep("v_mul_i32_i24 {d}, {a}, {d}")),
    ("v_min_i32", rep("v_min_i32 {d}, {a}, {d}")),
    ("v_bfe_i32 1 vgpr", rep("v_bfe_i32 {d}, {a}, {m}, 1")),
    ("v_bfe_u32 1 vgpr", rep("v_bfe_u32 {d}, {a}, 10, 10")),
    ("v_add3_u32 3 vgpr", rep("v_add3_u32 {d}, {d}, {a}, {b}")),
    ("v_add3_u32 2 vgpr + sgpr", rep("v_add3_u32 {d}, {d}, {a}, s20")),
    ("v_add3_u32 2 vgpr + inline", rep("v_add3_u32 {d}, {d}, {a}, 17")),
    ("v_xad_u32", rep("v_xad_u32 {d}, {d}, {a}, {b}")),
    ("v_lshl_add_u32", rep("v_lshl_add_u32 {d}, {a}, 4, {d}")),
    ("v_add_lshl_u32", rep("v_add_lshl_u32 {d}, {a}, {d}, 4")),
    ("v_lshl_or_b32", rep("v_lshl_or_b32 {d}, {a}, 4, {d}")),
    ("v_and_or_b32", rep("v_and_or_b32 {d}, {a}, {b}, {d}")),
    ("v_or3_b32", rep("v_or3_b32 {d}, {a}, {b}, {d}")),
    ("v_mad_u32_u24", rep("v_mad_u32_u24 {d}, {a}, {b}, {d}")),
    ("v_mad_i32_i24", rep("v_mad_i32_i24 {d}, {a}, {b}, {d}")),
    ("v_alignbit_b32", rep("v_alignbit_b32 {d}, {a}, {d}, 12")),
    ("v_bfi_b32", rep("v_bfi_b32 {d}, {a}, {b}, {d}")),
    ("v_perm_b32", rep("v_perm_b32 {d}, {a}, {d}, {b}")),
    ("v_med3_i32", rep("v_med3_i32 {d}, {a}, {b}, {d}")),
    ("v_mul_lo_u32", rep("v_mul_lo_u32 {d}, {a}, {d}")),
    ("v_lshlrev_b64", rep("v_lshlrev_b64 {D}, 3, {D}")),
    ("v_mov_b32 sdwa sext word", rep("v_mov_b32_sdwa {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0")),
    ("v_add_u32 sdwa", rep("v_add_u32_sdwa {d}, {a}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")),
    ("v_pk_add_u16", rep("v_pk_add_u16 {d}, {a}, {d}")),
    ("v_pk_sub_i16", rep("v_pk_sub_i16 {d}, {a}, {d}")),
    ("v_pk_lshrrev_b16", rep("v_pk_lshrrev_b16 {d}, 3, {d}")),
    ("v_pk_ashrrev_i16", rep("v_pk_ashrrev_i16 {d}, 3, {d}")),
    ("v_pk_mad_u16", rep("v_pk_mad_u16 {d}, {a}, {b}, {d}")),
    ("v_mov_b32_dpp quad_perm", rep("v_mov_b32_dpp {d}, {d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")),
    ("v_mov_b32_dpp row_ror:4", rep("v_mov_b32_dpp {d}, {d} row_ror:4 row_mask:0xf bank_mask:0xf")),
    ("v_add_u32_dpp quad_perm", rep("v_add_u32_dpp {d}, {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")),
    ("v_mov_b64", rep("v_mov_b64 {D}, {A}")),
    ("v_readfirstlane_b32", rep("v_readfirstlane_b32 s22, {a}")),
    ("s_nop 0", rep("s_nop 0")),
    ("s_add_u32", rep("s_add_u32 s22, s22, 1")),
    ("mix fma f64 + add_u32 (1:1)", [x for pair in zip(fma_acc(16, 2, 0, 0, 2)[:32], rep("v_add_u32 {d}, {a}, {d}")[:32]) for x in pair]),
    ("mix fma f64 + bfe (1:1)", [x for pair in zip(fma_acc(16, 2, 0, 0, 2)[:32], rep("v_bfe_i32 {d}, {a}, {m}, 1")[:32]) for x in pair]),
]



def bank_cases():
    """Every placement of the operands of the loop's four fp64 forms over the register classes r % 4 = 0 and r % 4 = 2
    (64-bit operands are even-aligned).  Sources are read-only registers, a different one per instruction, and
    destinations rotate over 8 registers, so nothing depends on anything (except the in-place forms, 8 apart)."""
    src_pool = {0: [SRC0 + 4 * j for j in range(4)], 2: [SRC0 + 4 * j + 2 for j in range(4)]}      # v96-111
    dst_pool = {0: [ACC0 + 4 * j for j in range(8)], 2: [ACC0 + 4 * j + 2 for j in range(8)]}      # v32-63
    out = []

    def body(fmt, dm, mods, inplace):
        lines = []
        for i in range(64):
            dreg_ = dst_pool[dm][i % 8]
            ops = [d(src_pool[m][(i + 3 * k) % 4]) for k, m in enumerate(mods)]
            if inplace:
                ops[-1] = d(dreg_)
            lines.append(fmt % tuple([d(dreg_)] + ops))
        return lines
    for dm in (0, 2):
        for am in (0, 2):
            for bm in (0, 2):
                out.append(("add  d%d a%d b%d" % (dm, am, bm), body("v_add_f64 %s, %s, %s", dm, (am, bm), False)))
                out.append(("mul  d%d a%d b%d" % (dm, am, bm), body("v_mul_f64 %s, %s, %s", dm, (am, bm), False)))
                out.append(("fmas d%d a%d c%d (b sgpr)" % (dm, am, bm), body("v_fma_f64 %s, %s, s[20:21], %s", dm, (am, bm), False)))
                for cm in (0, 2):
                    out.append(("fma  d%d a%d b%d c%d" % (dm, am, bm, cm), body("v_fma_f64 %s, %s, %s, %s", dm, (am, bm, cm), False)))
            out.append(("add  d%d += a%d (in place)" % (dm, am), body("v_add_f64 %s, %s, %s", dm, (am, dm), True)))
            out.append(("fmas d%d += a%d * sgpr" % (dm, am), body("v_fma_f64 %s, %s, s[20:21], %s", dm, (am, dm), True)))
            for bm in (0, 2):
                out.append(("fma  d%d += a%d * b%d" % (dm, am, bm), body("v_fma_f64 %s, %s, %s, %s", dm, (am, bm, dm), True)))
    return out


import sys
MODE = sys.argv[1] if len(sys.argv) > 1 else "ops"
if MODE == "banks":
    CASES = bank_cases()
src = ['// GENERATED by tools/gen_probe_banks.py', '#include <hip/hip_runtime.h>', '#include <cstdio>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)']
clob = ", ".join('"v%d"' % r for r in list(range(ACC0, ACC0 + 64)) + list(range(SRC0, SRC0 + 16)))
for n, (label, body) in enumerate(CASES):
    src.append('__global__ void __launch_bounds__(1024) k%d(unsigned long long* out, int iters) {' % n)
    src.append('  unsigned lo = 0x9999999au + threadIdx.x, hi = 0x3fb99999u;   // about 0.1: products shrink, sums stay finite')
    src.append('  unsigned long long t0, t1;')
    src.append('  asm volatile("s_mov_b32 s20, 0x9999999a\\n\\ts_mov_b32 s21, 0x3fb99999\\n\\t"')
    for r in range(ACC0, ACC0 + 64, 2):
        src.append('    "v_mov_b32 v%d, %%3\\n\\tv_mov_b32 v%d, %%4\\n\\t"' % (r, r + 1))
    for r in range(SRC0, SRC0 + 16, 2):
        src.append('    "v_mov_b32 v%d, %%3\\n\\tv_mov_b32 v%d, %%4\\n\\t"' % (r, r + 1))
    src.append('    "s_memtime %1\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"')
    src.append('    "L_loop%=:\\n\\t"')
    for ln in body:
        src.append('    "%s\\n\\t"' % ln)
    src.append('    "s_sub_u32 %0, %0, 1\\n\\ts_cmp_lg_u32 %0, 0\\n\\ts_cbranch_scc1 L_loop%=\\n\\t"')
    src.append('    "s_memtime %%2\\n\\ts_waitcnt lgkmcnt(0)\\n\\tv_mov_b32 %%3, v%d\\n\\t"' % ACC0)
    src.append('    : "+s"(iters), "=s"(t0), "=s"(t1), "+v"(lo), "+v"(hi) : : %s, "s20", "s21", "s22", "s23", "memory", "scc", "vcc");' % clob)
    src.append('  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = lo; }')
    src.append('}')
src.append('typedef void (*kern_t)(unsigned long long*, int);')
src.append('int main() {')
src.append('  unsigned long long* dv; CK(hipMalloc(&dv, 16 * 1024)); unsigned long long hv[2048];')
src.append('  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));')
src.append('  struct { const char* name; kern_t k; } ks[] = {')
for n, (label, body) in enumerate(CASES):
    src.append('    {"%s", k%d},' % (label, n))
src.append('  };')
src.append('  const int iters = 20000;')
src.append('  for (int rep = 0; rep < %d; rep++) for (int wps = %d; wps >= %d; wps--) {' % ((3, 2, 2) if MODE == 'banks' else (2, 4, 1)) + '')
src.append('   printf("---- %d wave(s) per SIMD: cycles per instruction and SIMD (cycle counter | wall clock at 2.4 GHz)\\n", wps);')
src.append('   for (auto& e : ks) {')
src.append('    hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * wps), 0, 0, dv, 10);')
src.append('    CK(hipEventRecord(e0));')
src.append('    hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * wps), 0, 0, dv, iters);')
src.append('    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));')
src.append('    float ms; CK(hipEventElapsedTime(&ms, e0, e1));')
src.append('    CK(hipMemcpy(hv, dv, 16 * 256, hipMemcpyDeviceToHost));')
src.append('    double cyc = 0; for (int b = 0; b < 256; b++) cyc += (double)hv[2 * b]; cyc /= 256;')
src.append('    printf("%-40s %6.2f | %6.2f   (counter %.0f ticks in %.3f ms = %.1f MHz)\\n", e.name, cyc / (64.0 * iters * wps), ms * 1e-3 * 2.4e9 / (64.0 * iters * wps), cyc, ms, cyc / ms * 1e-3);')
src.append('   }')
src.append('  }')
src.append('  return 0;')
src.append('}')
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_banks2.hip" if MODE == "banks" else "probe_banks.hip"), "w").write("\n".join(src) + "\n")
