"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

AND, NAND, OR, NOR, XOR, XNOR, MUX, NOT, COPY = range(9)


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

    @property
    def n(self):
        return self.lwe_dimension

    @property
    def k(self):
        return self.glwe_dimension

    @property
    def N(self):
        return self.polynomial_size


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    def __init__(self):
        if not os.path.exists(_LIB):
            build()
        self.lib = C.CDLL(_LIB)
        L = self.lib
        L.orc_bsk_words.restype = C.c_uint64
        L.orc_ksk_words.restype = C.c_uint64
        L.orc_bsk_fourier_doubles.restype = C.c_uint64
        L.orc_phase.restype = C.c_uint32
        L.orc_modswitch.restype = C.c_uint32
        self.params = Params()
        L.orc_default_params(C.byref(self.params))

    # -- keys ---------------------------------------------------------------------------
    def keygen(self, seed):
        p = self.params
        lwe_sk = np.zeros(p.n, np.uint32)
        glwe_sk = np.zeros(p.k * p.N, np.uint32)
        bsk = np.zeros(self.lib.orc_bsk_words(C.byref(p)), np.uint32)
        ksk = np.zeros(self.lib.orc_ksk_words(C.byref(p)), np.uint32)
        self.lib.orc_keygen(C.byref(p), C.c_uint64(seed), _u32p(lwe_sk), _u32p(glwe_sk), _u32p(bsk), _u32p(ksk))
        return lwe_sk, glwe_sk, bsk, ksk

    def bsk_to_fourier(self, bsk):
        out = np.zeros(self.lib.orc_bsk_fourier_doubles(C.byref(self.params)), np.float64)
        self.lib.orc_bsk_to_fourier(C.byref(self.params), _u32p(bsk), _f64p(out))
        return out

    # -- client side ---------------------------------------------------------------------
    def encrypt_bits(self, lwe_sk, seed, stream, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        cts = np.zeros((bits.size, self.params.n + 1), np.uint32)
        self.lib.orc_encrypt_bits(C.byref(self.params), _u32p(lwe_sk), C.c_uint64(seed), C.c_uint64(stream), _u8p(bits),
                                  C.c_uint32(bits.size), _u32p(cts))
        return cts

    def decrypt_bits(self, lwe_sk, cts):
        cts = np.ascontiguousarray(cts, np.uint32).reshape(-1, self.params.n + 1)
        bits = np.zeros(cts.shape[0], np.uint8)
        self.lib.orc_decrypt_bits(C.byref(self.params), _u32p(lwe_sk), _u32p(cts), C.c_uint32(cts.shape[0]), _u8p(bits))
        return bits

    def phase(self, sk, ct):
        sk = np.ascontiguousarray(sk, np.uint32)
        ct = np.ascontiguousarray(ct, np.uint32)
        return int(self.lib.orc_phase(_u32p(sk), C.c_uint32(sk.size), _u32p(ct)))

    # -- building blocks -----------------------------------------------------------------
    def modswitch(self, x, log2_two_n=10):
        return int(self.lib.orc_modswitch(C.c_uint32(int(x)), C.c_uint32(log2_two_n)))

    def decompose(self, x, base_log, level):
        d = (C.c_int32 * level)()
        self.lib.orc_decompose(C.c_uint32(int(x)), C.c_uint32(base_log), C.c_uint32(level), d)
        return list(d)

    def keyswitch(self, ksk, big):
        big = np.ascontiguousarray(big, np.uint32)
        out = np.zeros(self.params.n + 1, np.uint32)
        self.lib.orc_keyswitch(C.byref(self.params), _u32p(ksk), _u32p(big), _u32p(out))
        return out

    def pbs(self, bsk, bsk_f, lwe_in):
        p = self.params
        lwe_in = np.ascontiguousarray(lwe_in, np.uint32)
        big = np.zeros(p.k * p.N + 1, np.uint32)
        self.lib.orc_pbs(C.byref(p), _u32p(bsk), _f64p(bsk_f) if bsk_f is not None else None, _u32p(lwe_in), _u32p(big))
        return big

    def blind_rotate(self, bsk, bsk_f, lwe_in, steps):
        p = self.params
        lwe_in = np.ascontiguousarray(lwe_in, np.uint32)
        acc = np.zeros((p.k + 1, p.N), np.uint32)
        self.lib.orc_blind_rotate(C.byref(p), _u32p(bsk), _f64p(bsk_f) if bsk_f is not None else None, _u32p(lwe_in),
                                  C.c_uint32(steps), _u32p(acc))
        return acc

    def external_product_exact(self, ggsw, d, out):
        self.lib.orc_external_product_exact(C.byref(self.params), _u32p(ggsw), _u32p(d), _u32p(out))
        return out

    def external_product_fft(self, ggsw_f, d, out):
        self.lib.orc_external_product_fft(C.byref(self.params), _f64p(ggsw_f), _u32p(d), _u32p(out))
        return out

    def precombine(self, op, a, b, c=None):
        w = self.params.n + 1
        u1 = np.zeros(w, np.uint32)
        u2 = np.zeros(w, np.uint32)
        a = np.ascontiguousarray(a, np.uint32)
        b = np.ascontiguousarray(b, np.uint32)
        cc = np.ascontiguousarray(c, np.uint32) if c is not None else None
        self.lib.orc_gate_precombine(C.byref(self.params), C.c_int(op), _u32p(a), _u32p(b), _u32p(cc) if cc is not None else None,
                                     _u32p(u1), _u32p(u2))
        return u1, u2

    def sample_extract(self, acc):
        p = self.params
        acc = np.ascontiguousarray(acc, np.uint32)
        big = np.zeros(p.k * p.N + 1, np.uint32)
        self.lib.orc_sample_extract(C.byref(p), _u32p(acc), _u32p(big))
        return big

    # -- gates ----------------------------------------------------------------------------
    def gates(self, bsk, bsk_f, ksk, ops, in0, in1, in2=None):
        ops = np.ascontiguousarray(ops, np.uint8)
        w = self.params.n + 1
        in0 = np.ascontiguousarray(in0, np.uint32).reshape(-1, w)
        in1 = np.ascontiguousarray(in1, np.uint32).reshape(-1, w) if in1 is not None else None
        in2 = np.ascontiguousarray(in2, np.uint32).reshape(-1, w) if in2 is not None else None
        out = np.zeros((ops.size, w), np.uint32)
        self.lib.orc_gates(C.byref(self.params), _u32p(bsk), _f64p(bsk_f) if bsk_f is not None else None, _u32p(ksk), _u8p(ops),
                           _u32p(in0), _u32p(in1) if in1 is not None else None, _u32p(in2) if in2 is not None else None,
                           C.c_uint32(ops.size), _u32p(out))
        return out

    def num_threads(self):
        return int(self.lib.orc_num_threads())
