"""The C-ABI library loads and exports every symbol the public headers declare; without a GPU the
device entry points fail with an error code, never a crash, and nothing falls back to the CPU."""
import ctypes as C
import os
import re

import pytest

from tfhe_aes_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z0-9_]+)\s*\(" % prefix, text)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    names = declared("tfhe_aes_hip.h", "tfa") + declared("tfhe_aes_host.h", "tfh")
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_rust_sys_crate_declares_exactly_the_header():
    """integration/rust/tfhe-aes-hip-sys/src/lib.rs (source only: no Rust toolchain here) names the same
    functions as include/tfhe_aes_hip.h, and all of them are exported."""
    text = open(os.path.join(ROOT, "integration", "rust", "tfhe-aes-hip-sys", "src", "lib.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (tfa_[a-z0-9_]+)\(", text)))
    assert rust == declared("tfhe_aes_hip.h", "tfa")
    lib = E.load_library()
    assert all(hasattr(lib, n) for n in rust)


def test_struct_layouts_match_header():
    assert C.sizeof(E.Gate) == 24 and C.sizeof(E.Params) == 48 and C.sizeof(E.Stats) == 64
    assert E.GATE_DTYPE.itemsize == 24


def test_sizes_and_default_params():
    p = E.default_params()
    lib = E.load_library()
    assert lib.tfa_bsk_words(C.byref(p)) == 805 * 2 * 4 * 4 * 512
    assert lib.tfa_ksk_words(C.byref(p)) == 1536 * 5 * 806


def test_bad_parameters_are_refused():
    lib = E.load_library()
    p = E.default_params()
    p.polynomial_size = 1024
    h = C.c_void_p()
    assert lib.tfa_ctx_create(C.byref(p), 0, C.byref(h)) == -1 and not h
    assert lib.tfa_ctx_create(None, 0, C.byref(h)) == -1
    assert lib.tfa_stage_submit(None, None, 0) == -1 and lib.tfa_sync(None) == -1


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.TfaError):
        E.Context(E.default_params(), 0)
    from tfhe_aes_amd import host as H
    with pytest.raises(H.HostError):
        H.gen_keys(1, 0)


def test_profile_launches_refuses_null_arguments():
    """no context is needed to see the argument checks of the entry point added for the bench's per-width table"""
    import ctypes as C
    from tfhe_aes_amd import engine as E
    lib = E.load_library()
    n = C.c_uint32(0)
    assert lib.tfa_profile_launches(None, None, C.c_uint32(0), C.byref(n)) == -1      # TFA_EINVAL
