"""The generated step loops of the hand-scheduled kernels, checked without a GPU.

1. The `.inc` files the library is compiled from are exactly what their generators emit (so the CPU check below speaks about
   the code that runs on the device).
2. The generated instruction streams, executed by the CPU emulators (LDS reads and key loads delivered only at the
   `s_waitcnt` that guarantees them), reproduce the exact oracle's accumulator word for word after a few external products.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools", "pbs_asm")
CSRC = os.path.join(ROOT, "tfhe-aes_amd", "csrc")


def _run(script, args=(), env=None):
    e = dict(os.environ)
    for k in ("PBS_ASM_EXP", "PBS_ASM_OPT", "PBS_ASM_MANUAL_WAITS", "PBS_ASM_WAVES", "PBS_LAT_PER_CU", "PBS_ASM_PAD", "PBS_ASM_FPG", "PBS_LAT_FPD"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(TOOLS, script), *args], env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("script,inc,env", [
    ("gen.py", "pbs_asm_loop.inc", {}),
    ("gen.py", "pbs_asm_loop4.inc", {"PBS_ASM_WAVES": "4"}),
    ("lat_gen.py", "pbs_lat_asm_loop.inc", {}),
    ("lat_gen.py", "pbs_lat_asm_loop2.inc", {"PBS_LAT_PER_CU": "2"}),
])
def test_committed_loop_is_generator_output(tmp_path, script, inc, env):
    out = tmp_path / inc
    r = _run(script, [str(out)], env)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == open(os.path.join(CSRC, inc)).read(), f"{inc} is stale: run make -C tfhe-aes_amd"


@pytest.mark.parametrize("script,env", [
    ("emu.py", {}),
    ("emu.py", {"PBS_ASM_WAVES": "4"}),
    ("lat_emu.py", {}),
    ("lat_emu.py", {"PBS_LAT_PER_CU": "2"}),
])
def test_emulated_loop_matches_oracle(script, env):
    r = _run(script, ["3"], env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatching words 0" in r.stdout


BASE_OPT = "mac_il,tw_spread,inv_split,lanesum,ls_pf2,wait_merge,dma_off"


@pytest.mark.parametrize("opt", [
    BASE_OPT,                                              # integer digits, three-ahead ring: the loop the shipped one is measured against
    BASE_OPT + ",fp_digits,ring2,ls_pf3,tw_il,tail_il",    # the shipped choices plus the reorderings that measured no gain
])
def test_switchable_schedule_choices_stay_exact(opt):
    """the options kept for side-by-side measurement generate loops that still reproduce the exact oracle"""
    r = _run("emu.py", ["2"], {"PBS_ASM_OPT": opt})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatching words 0" in r.stdout


@pytest.mark.parametrize("mode", ["1", "2"])
def test_narrow_loop_with_floating_point_digits_stays_exact(mode):
    """the narrow-stage loops' optional floating-point digits (two / eight coefficients at a time)"""
    r = _run("lat_emu.py", ["2"], {"PBS_LAT_FPD": mode})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatching words 0" in r.stdout
