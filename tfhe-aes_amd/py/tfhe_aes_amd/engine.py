"""ctypes binding of the C ABI in include/tfhe_aes_hip.h (libtfhe_aes_hip.so).

The library is the product; this module only marshals numpy arrays across the boundary and
raises when the HIP extension is missing or a call fails (there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get("TFHE_AES_HIP_LIB") or os.path.join(_PKG, "libtfhe_aes_hip.so")

AND, NAND, OR, NOR, XOR, XNOR, MUX, NOT, COPY = range(9)
NOALLOC_MAX_GATES = 4096   # TFA_NOALLOC_MAX_GATES / TFA_NOALLOC_MAX_PBS of include/tfhe_aes_hip.h
NOALLOC_MAX_PBS = 4096


class Params(C.Structure):
    _fields_ = [
        ("lwe_dimension", C.c_uint32),
        ("glwe_dimension", C.c_uint32),
        ("polynomial_size", C.c_uint32),
        ("pbs_base_log", C.c_uint32),
        ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32),
        ("ks_level", C.c_uint32),
        ("lwe_noise_std", C.c_double),
        ("glwe_noise_std", C.c_double),
    ]


class Gate(C.Structure):
    _fields_ = [("op", C.c_uint32), ("in0", C.c_uint32), ("in1", C.c_uint32), ("in2", C.c_uint32), ("out", C.c_uint32),
                ("neg", C.c_uint32)]


GATE_DTYPE = np.dtype([("op", "<u4"), ("in0", "<u4"), ("in1", "<u4"), ("in2", "<u4"), ("out", "<u4"), ("neg", "<u4")])


class Stats(C.Structure):
    _fields_ = [("pbs_count", C.c_uint64), ("ks_count", C.c_uint64), ("gate_count", C.c_uint64), ("stage_count", C.c_uint64),
                ("pbs_kernel_ms", C.c_double), ("ks_kernel_ms", C.c_double), ("pbs_launches", C.c_uint64),
                ("ks_launches", C.c_uint64)]


class TfaError(RuntimeError):
    pass


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TfaError(f"HIP extension not built: {LIB_PATH} is missing (run `make -C tfhe-aes_amd`)")
        # PyTorch ships its own copy of the HIP runtime: when the engine's library is loaded first, it binds the system
        # copy and torch's later initialisation finds no GPU ("No HIP GPUs are available").  Importing torch first (when it
        # is installed; the engine itself does not need it) makes both use the one already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.tfa_last_error.restype = C.c_char_p
        _lib.tfa_bsk_words.restype = C.c_uint64
        _lib.tfa_ksk_words.restype = C.c_uint64
    return _lib


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def default_params():
    p = Params()
    load_library().tfa_default_params(C.byref(p))
    return p


# ---- client side (host CPU, as in the reference) ------------------------------------------------
def client_keygen(params, seed):
    lib = load_library()
    lwe_sk = np.zeros(params.lwe_dimension, np.uint32)
    glwe_sk = np.zeros(params.glwe_dimension * params.polynomial_size, np.uint32)
    bsk = np.zeros(lib.tfa_bsk_words(C.byref(params)), np.uint32)
    ksk = np.zeros(lib.tfa_ksk_words(C.byref(params)), np.uint32)
    rc = lib.tfa_client_keygen(C.byref(params), C.c_uint64(seed), _u32p(lwe_sk), _u32p(glwe_sk), _u32p(bsk), _u32p(ksk))
    if rc:
        raise TfaError(f"tfa_client_keygen failed: {rc}")
    return lwe_sk, glwe_sk, bsk, ksk


def client_encrypt(params, lwe_sk, seed, bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros((bits.size, params.lwe_dimension + 1), np.uint32)
    rc = load_library().tfa_client_encrypt(C.byref(params), _u32p(lwe_sk), C.c_uint64(seed), _u8p(bits), C.c_uint32(bits.size), _u32p(out))
    if rc:
        raise TfaError(f"tfa_client_encrypt failed: {rc}")
    return out


def client_decrypt(params, lwe_sk, lwe):
    lwe = np.ascontiguousarray(lwe, np.uint32).reshape(-1, params.lwe_dimension + 1)
    bits = np.zeros(lwe.shape[0], np.uint8)
    rc = load_library().tfa_client_decrypt(C.byref(params), _u32p(lwe_sk), _u32p(lwe), C.c_uint32(lwe.shape[0]), _u8p(bits))
    if rc:
        raise TfaError(f"tfa_client_decrypt failed: {rc}")
    return bits


# ---- device context --------------------------------------------------------------------------
class Context:
    def __init__(self, params=None, device=0):
        self.lib = load_library()
        self.params = params or default_params()
        self._ctx = C.c_void_p()
        rc = self.lib.tfa_ctx_create(C.byref(self.params), C.c_int(device), C.byref(self._ctx))
        if rc:
            msg = self.lib.tfa_last_error(self._ctx).decode() if self._ctx else "no context"
            if self._ctx:
                self.lib.tfa_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise TfaError(f"tfa_ctx_create failed ({rc}): {msg}")
        self.n = self.params.lwe_dimension
        self.big = self.params.glwe_dimension * self.params.polynomial_size

    def _chk(self, rc, what):
        if rc:
            raise TfaError(f"{what} failed ({rc}): {self.lib.tfa_last_error(self._ctx).decode()}")

    def close(self):
        if self._ctx:
            self.lib.tfa_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def keys_upload(self, bsk, ksk):
        bsk = np.ascontiguousarray(bsk, np.uint32)
        ksk = np.ascontiguousarray(ksk, np.uint32)
        assert bsk.size == self.lib.tfa_bsk_words(C.byref(self.params)) and ksk.size == self.lib.tfa_ksk_words(C.byref(self.params))
        self._chk(self.lib.tfa_keys_upload(self._ctx, _u32p(bsk), _u32p(ksk)), "tfa_keys_upload")

    def pool_reserve(self, n_slots):
        self._chk(self.lib.tfa_pool_reserve(self._ctx, C.c_uint32(n_slots)), "tfa_pool_reserve")

    def pool_write(self, slots, lwe):
        slots = np.ascontiguousarray(slots, np.uint32)
        lwe = np.ascontiguousarray(lwe, np.uint32).reshape(slots.size, self.n + 1)
        self._chk(self.lib.tfa_pool_write(self._ctx, _u32p(slots), C.c_uint32(slots.size), _u32p(lwe)), "tfa_pool_write")

    def pool_read(self, slots):
        slots = np.ascontiguousarray(slots, np.uint32)
        out = np.zeros((slots.size, self.n + 1), np.uint32)
        self._chk(self.lib.tfa_pool_read(self._ctx, _u32p(slots), C.c_uint32(slots.size), _u32p(out)), "tfa_pool_read")
        return out

    def pool_read_device(self, slots, device_ptr, negate=None):
        """Dense rows [len(slots)][n+1] u32 written to device memory at `device_ptr` (an int: e.g. tensor.data_ptr())."""
        slots = np.ascontiguousarray(slots, np.uint32)
        neg = np.ascontiguousarray(negate, np.uint8) if negate is not None else None
        self._chk(self.lib.tfa_pool_read_device(self._ctx, _u32p(slots), _u8p(neg) if neg is not None else None, C.c_uint32(slots.size),
                                                C.c_void_p(int(device_ptr))), "tfa_pool_read_device")

    def stage_submit(self, gates):
        """gates: numpy structured array of GATE_DTYPE (or anything convertible)."""
        gates = np.ascontiguousarray(gates, GATE_DTYPE)
        self._chk(self.lib.tfa_stage_submit(self._ctx, gates.ctypes.data_as(C.POINTER(Gate)), C.c_uint32(gates.size)), "tfa_stage_submit")

    def stage_submit_noalloc(self, gates):
        """tfa_stage_submit_noalloc: the same stage on the library's static device scratch (no allocation in the call)."""
        gates = np.ascontiguousarray(gates, GATE_DTYPE)
        self._chk(self.lib.tfa_stage_submit_noalloc(self._ctx, gates.ctypes.data_as(C.POINTER(Gate)), C.c_uint32(gates.size)),
                  "tfa_stage_submit_noalloc")

    def sync(self):
        self._chk(self.lib.tfa_sync(self._ctx), "tfa_sync")

    def bootstrap_batch(self, lwe_in):
        lwe_in = np.ascontiguousarray(lwe_in, np.uint32).reshape(-1, self.n + 1)
        out = np.zeros((lwe_in.shape[0], self.big + 1), np.uint32)
        self._chk(self.lib.tfa_bootstrap_batch(self._ctx, _u32p(lwe_in), C.c_uint32(lwe_in.shape[0]), _u32p(out)), "tfa_bootstrap_batch")
        return out

    def blind_rotate_batch(self, lwe_in, steps):
        lwe_in = np.ascontiguousarray(lwe_in, np.uint32).reshape(-1, self.n + 1)
        k1 = self.params.glwe_dimension + 1
        out = np.zeros((lwe_in.shape[0], k1, self.params.polynomial_size), np.uint32)
        self._chk(self.lib.tfa_blind_rotate_batch(self._ctx, _u32p(lwe_in), C.c_uint32(lwe_in.shape[0]), C.c_uint32(steps), _u32p(out)),
                  "tfa_blind_rotate_batch")
        return out

    def keyswitch_batch(self, big_in):
        big_in = np.ascontiguousarray(big_in, np.uint32).reshape(-1, self.big + 1)
        out = np.zeros((big_in.shape[0], self.n + 1), np.uint32)
        self._chk(self.lib.tfa_keyswitch_batch(self._ctx, _u32p(big_in), C.c_uint32(big_in.shape[0]), _u32p(out)), "tfa_keyswitch_batch")
        return out

    def set_profiling(self, on):
        self._chk(self.lib.tfa_set_profiling(self._ctx, C.c_int(1 if on else 0)), "tfa_set_profiling")

    def stats(self):
        s = Stats()
        self._chk(self.lib.tfa_get_stats(self._ctx, C.byref(s)), "tfa_get_stats")
        return s

    def reset_stats(self):
        self._chk(self.lib.tfa_reset_stats(self._ctx), "tfa_reset_stats")

    def profile_launches(self):
        return profile_launches(self.lib, self._ctx)


LAUNCH_DTYPE = np.dtype([("units", np.uint32), ("kind", np.uint32), ("mapping", np.uint32), ("ms", np.float32)])


def profile_launches(lib, ctx):
    """tfa_profile_launches: every timed launch since the last reset as a structured array (units, kind, mapping, ms)"""
    n = C.c_uint32(0)
    rc = lib.tfa_profile_launches(ctx, None, C.c_uint32(0), C.byref(n))
    if rc != 0:
        raise RuntimeError("tfa_profile_launches failed (%d)" % rc)
    out = np.zeros(n.value, LAUNCH_DTYPE)
    if n.value:
        rc = lib.tfa_profile_launches(ctx, out.ctypes.data_as(C.c_void_p), C.c_uint32(n.value), C.byref(n))
        if rc != 0:
            raise RuntimeError("tfa_profile_launches failed (%d)" % rc)
    return out


def make_gates(ops, in0, in1, in2, out, neg=None):
    ops = np.asarray(ops)
    g = np.zeros(ops.size, GATE_DTYPE)
    g["op"], g["in0"], g["in1"], g["out"] = ops, in0, in1, out
    g["in2"] = in2 if in2 is not None else 0
    g["neg"] = neg if neg is not None else 0
    return g
