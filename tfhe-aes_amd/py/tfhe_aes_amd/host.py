"""Python view of the host mirror (include/tfhe_aes_host.h): the reference's API names
(gen_keys, FHEByte, State, Key, ECB, CBC, OFB, CTR, reduce_mux) over the C ABI, so tests read like
the reference's own.  Pure marshalling; all work happens in libtfhe_aes_hip.so.
"""
import ctypes as C

import numpy as np

from . import engine as E


class HostError(RuntimeError):
    pass


def _lib():
    lib = E.load_library()
    if not getattr(lib, "_tfh_ready", False):
        lib.tfh_last_error.restype = C.c_char_p
        lib.tfh_context.restype = C.c_void_p
        lib.tfh_reduce_mux.restype = C.c_long
        lib.tfh_sbox_circuit_dump.restype = C.c_long
        lib._tfh_ready = True
    return lib


def _chk(rc, what):
    if rc:
        raise HostError(f"{what} failed ({rc}): {_lib().tfh_last_error().decode()}")


def _b16(data):
    data = bytes(data)
    assert len(data) == 16
    return (C.c_uint8 * 16)(*data)


class Keys:
    """(ClientKey, ServerKey) pair of gen_keys()."""

    def __init__(self, seed=0, device=0):
        self.h = C.c_void_p()
        _chk(_lib().tfh_gen_keys(C.c_uint64(seed), C.c_int(device), C.byref(self.h)), "tfh_gen_keys")
        self.params = E.default_params()

    def close(self):
        if self.h:
            _lib().tfh_keys_free(self.h)
            self.h = C.c_void_p()

    def flush(self):
        _chk(_lib().tfh_flush(self.h), "tfh_flush")

    def set_xor_rebalance(self, on):
        """tfh_set_xor_rebalance: re-association of unobservable xor chains into least-depth trees before a flush (default on)"""
        _chk(_lib().tfh_set_xor_rebalance(self.h, C.c_int(1 if on else 0)), "tfh_set_xor_rebalance")

    def host_stats(self):
        out = (C.c_uint64 * 4)()
        _lib().tfh_host_stats(self.h, out)
        return dict(gates=out[0], stages=out[1], flushes=out[2], max_stage_width=out[3])

    def lwe_secret_key(self):
        sk = np.zeros(self.params.lwe_dimension, np.uint32)
        _lib().tfh_lwe_secret_key(self.h, sk.ctypes.data_as(C.POINTER(C.c_uint32)))
        return sk

    def engine_stats(self):
        s = E.Stats()
        E.load_library().tfa_get_stats(C.c_void_p(_lib().tfh_context(self.h)), C.byref(s))
        return s

    def set_profiling(self, on):
        E.load_library().tfa_set_profiling(C.c_void_p(_lib().tfh_context(self.h)), C.c_int(1 if on else 0))

    def reset_engine_stats(self):
        E.load_library().tfa_reset_stats(C.c_void_p(_lib().tfh_context(self.h)))

    def profile_launches(self):
        return E.profile_launches(E.load_library(), C.c_void_p(_lib().tfh_context(self.h)))

    # ciphertext-level API (ClientKey / ServerKey)
    def encrypt(self, message):
        h = C.c_void_p()
        _chk(_lib().tfh_encrypt(self.h, C.c_int(1 if message else 0), C.byref(h)), "tfh_encrypt")
        return Ciphertext(self, h)

    def trivial_encrypt(self, message):
        h = C.c_void_p()
        _chk(_lib().tfh_trivial(self.h, C.c_int(1 if message else 0), C.byref(h)), "tfh_trivial")
        return Ciphertext(self, h)

    def decrypt(self, ct):
        m = C.c_int()
        _chk(_lib().tfh_decrypt(self.h, ct.h, C.byref(m)), "tfh_decrypt")
        return bool(m.value)

    def _gate(self, op, a, b=None, c=None):
        h = C.c_void_p()
        _chk(_lib().tfh_gate(self.h, C.c_uint32(op), a.h, b.h if b else None, c.h if c else None, C.byref(h)), "tfh_gate")
        return Ciphertext(self, h)

    def and_(self, a, b):
        return self._gate(E.AND, a, b)

    def nand(self, a, b):
        return self._gate(E.NAND, a, b)

    def or_(self, a, b):
        return self._gate(E.OR, a, b)

    def nor(self, a, b):
        return self._gate(E.NOR, a, b)

    def xor(self, a, b):
        return self._gate(E.XOR, a, b)

    def xnor(self, a, b):
        return self._gate(E.XNOR, a, b)

    def mux(self, c, t, e):
        return self._gate(E.MUX, c, t, e)

    def not_(self, a):
        return self._gate(E.NOT, a)


def gen_keys(seed=0, device=0):
    return Keys(seed, device)


class _Handle:
    _free = None

    def __init__(self, keys, h):
        self.keys, self.h = keys, h

    def __del__(self):
        try:
            if self.h and self._free:
                getattr(_lib(), self._free)(self.h)
                self.h = None
        except Exception:
            pass


class Ciphertext(_Handle):
    _free = "tfh_ct_free"

    def is_trivial(self):
        return bool(_lib().tfh_ct_is_trivial(self.h))


class FHEByte(_Handle):
    _free = "tfh_byte_free"

    @staticmethod
    def from_u8_enc(value, keys):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_from_u8_enc(keys.h, C.c_uint8(value), C.byref(h)), "tfh_byte_from_u8_enc")
        return FHEByte(keys, h)

    @staticmethod
    def from_u8_clear(value, keys):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_from_u8_clear(keys.h, C.c_uint8(value), C.byref(h)), "tfh_byte_from_u8_clear")
        return FHEByte(keys, h)

    trivial_clear = from_u8_clear

    def decrypt_to_u8(self):
        out = C.c_uint8()
        _chk(_lib().tfh_byte_decrypt_to_u8(self.keys.h, self.h, C.byref(out)), "tfh_byte_decrypt_to_u8")
        return out.value

    def _bin(self, op, rhs):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_binary(self.keys.h, C.c_uint32(op), self.h, rhs.h, C.byref(h)), "tfh_byte_binary")
        return FHEByte(self.keys, h)

    def xor(self, rhs):
        return self._bin(E.XOR, rhs)

    def and_(self, rhs):
        return self._bin(E.AND, rhs)

    def or_(self, rhs):
        return self._bin(E.OR, rhs)

    def not_(self):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_not(self.keys.h, self.h, C.byref(h)), "tfh_byte_not")
        return FHEByte(self.keys, h)

    def sub_byte(self):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_sub_byte(self.keys.h, self.h, C.c_int(0), C.byref(h)), "tfh_byte_sub_byte")
        return FHEByte(self.keys, h)

    def inv_sub_byte(self):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_sub_byte(self.keys.h, self.h, C.c_int(1), C.byref(h)), "tfh_byte_sub_byte")
        return FHEByte(self.keys, h)

    def mul_x_gf2(self):
        h = C.c_void_p()
        _chk(_lib().tfh_byte_mul_x_gf2(self.keys.h, self.h, C.byref(h)), "tfh_byte_mul_x_gf2")
        return FHEByte(self.keys, h)


class Key(_Handle):
    _free = "tfh_key_free"

    @staticmethod
    def from_u8_enc(data, keys):
        h = C.c_void_p()
        _chk(_lib().tfh_key_from_u8_enc(keys.h, _b16(data), C.byref(h)), "tfh_key_from_u8_enc")
        return Key(keys, h)

    @staticmethod
    def from_u128_enc(value, keys):
        return Key.from_u8_enc(int(value).to_bytes(16, "big"), keys)

    @staticmethod
    def from_u8_clear(data, keys):
        h = C.c_void_p()
        _chk(_lib().tfh_key_from_u8_clear(keys.h, _b16(data), C.byref(h)), "tfh_key_from_u8_clear")
        return Key(keys, h)

    def generate_round_keys(self):
        out = (C.c_void_p * 11)()
        _chk(_lib().tfh_key_generate_round_keys(self.keys.h, self.h, out), "tfh_key_generate_round_keys")
        return [Key(self.keys, C.c_void_p(out[i])) for i in range(11)]

    def decrypt_to_u8(self):
        out = (C.c_uint8 * 16)()
        _chk(_lib().tfh_key_decrypt_to_u8(self.keys.h, self.h, out), "tfh_key_decrypt_to_u8")
        return bytes(out)


def key_expansion_clear(key):
    out = ((C.c_uint8 * 16) * 11)()
    _lib().tfh_key_expansion_clear(_b16(key), out)
    return [bytes(out[i]) for i in range(11)]


SUB_BYTES, INV_SUB_BYTES, SHIFT_ROWS, INV_SHIFT_ROWS, MIX_COLUMNS, INV_MIX_COLUMNS = range(6)


class State(_Handle):
    _free = "tfh_state_free"

    @staticmethod
    def from_u8_enc(data, keys):
        h = C.c_void_p()
        _chk(_lib().tfh_state_from_u8_enc(keys.h, _b16(data), C.byref(h)), "tfh_state_from_u8_enc")
        return State(keys, h)

    @staticmethod
    def from_u128_enc(value, keys):
        return State.from_u8_enc(int(value).to_bytes(16, "big"), keys)

    def clone(self):
        h = C.c_void_p()
        _chk(_lib().tfh_state_clone(self.h, C.byref(h)), "tfh_state_clone")
        return State(self.keys, h)

    def _apply(self, op):
        _chk(_lib().tfh_state_apply(self.keys.h, self.h, C.c_uint32(op)), "tfh_state_apply")

    def sub_bytes(self):
        self._apply(SUB_BYTES)

    def inv_sub_bytes(self):
        self._apply(INV_SUB_BYTES)

    def shift_rows(self):
        self._apply(SHIFT_ROWS)

    def inv_shift_rows(self):
        self._apply(INV_SHIFT_ROWS)

    def mix_columns(self):
        self._apply(MIX_COLUMNS)

    def inv_mix_columns(self):
        self._apply(INV_MIX_COLUMNS)

    def xor_key_enc(self, key):
        _chk(_lib().tfh_state_xor_key_enc(self.keys.h, self.h, key.h), "tfh_state_xor_key_enc")

    def xor_key_clear(self, key):
        _chk(_lib().tfh_state_xor_key_clear(self.keys.h, self.h, _b16(key)), "tfh_state_xor_key_clear")

    def xor_state(self, other):
        _chk(_lib().tfh_state_xor_state(self.keys.h, self.h, other.h), "tfh_state_xor_state")

    def decrypt_to_u8(self):
        out = (C.c_uint8 * 16)()
        _chk(_lib().tfh_state_decrypt_to_u8(self.keys.h, self.h, out), "tfh_state_decrypt_to_u8")
        return bytes(out)

    def decrypt_to_u128(self):
        return int.from_bytes(self.decrypt_to_u8(), "big")

    def read(self):
        out = np.zeros((128, self.keys.params.lwe_dimension + 1), np.uint32)
        _chk(_lib().tfh_state_read(self.keys.h, self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "tfh_state_read")
        return out

    @staticmethod
    def import_lwe(lwe, keys):
        lwe = np.ascontiguousarray(lwe, np.uint32).reshape(128, keys.params.lwe_dimension + 1)
        h = C.c_void_p()
        _chk(_lib().tfh_state_import(keys.h, lwe.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(h)), "tfh_state_import")
        return State(keys, h)


def read_states_device(states, out):
    """The 128 LWE ciphertexts of each state, dense, into `out`: a torch int32 CUDA tensor [len(states), 128, n+1] on
    the ServerKey's GPU (written by the engine on its own stream, complete on return)."""
    keys = states[0].keys
    words = keys.params.lwe_dimension + 1
    assert out.is_cuda and out.is_contiguous() and out.element_size() == 4 and out.numel() >= len(states) * 128 * words
    _chk(_lib().tfh_states_read_device(keys.h, _handles(states), C.c_uint32(len(states)), C.c_void_p(out.data_ptr())), "tfh_states_read_device")
    return out


def _handles(objs):
    return (C.c_void_p * len(objs))(*[o.h for o in objs])


class ECB:
    def __init__(self, keys):
        assert len(keys) == 11
        self.round_keys = list(keys)
        self.k = keys[0].keys

    def encrypt(self, state):
        self.encrypt_many([state])

    def decrypt(self, state):
        self.decrypt_many([state])

    def encrypt_many(self, states):
        _chk(_lib().tfh_ecb(self.k.h, _handles(self.round_keys), _handles(states), C.c_uint32(len(states)), C.c_int(0)), "tfh_ecb")

    def decrypt_many(self, states):
        _chk(_lib().tfh_ecb(self.k.h, _handles(self.round_keys), _handles(states), C.c_uint32(len(states)), C.c_int(1)), "tfh_ecb")


class _Chained:
    _fn = None

    def __init__(self, keys, iv, n):
        self.round_keys, self.iv, self.n = list(keys), iv, n
        self.k = keys[0].keys

    def _run(self, states, dec):
        fn = getattr(_lib(), self._fn)
        _chk(fn(self.k.h, _handles(self.round_keys), self.iv.h, _handles(states), C.c_uint32(self.n), C.c_int(dec)), self._fn)

    def encrypt(self, states):
        self._run(states, 0)

    def decrypt(self, states):
        self._run(states, 1)


class CBC(_Chained):
    _fn = "tfh_cbc"


class OFB(_Chained):
    _fn = "tfh_ofb"


class CTR:
    def __init__(self, keys, counters, n):
        if len(counters) != n:
            raise HostError("CTR: counters.len() != n")
        self.round_keys, self.counters, self.n = list(keys), list(counters), n
        self.k = keys[0].keys

    def _run(self, states, dec):
        _chk(_lib().tfh_ctr(self.k.h, _handles(self.round_keys), _handles(self.counters), _handles(states), C.c_uint32(self.n), C.c_int(dec)),
             "tfh_ctr")

    def encrypt(self, states):
        self._run(states, 0)

    def decrypt(self, states):
        self._run(states, 1)


def reduce_mux(truth_table):
    tt = (C.c_uint8 * max(1, len(truth_table)))(*[1 if b else 0 for b in truth_table])
    buf = C.create_string_buffer(1 << 20)
    rc = _lib().tfh_reduce_mux(tt, C.c_uint32(len(truth_table)), buf, C.c_size_t(len(buf)))
    if rc < 0:
        raise HostError(_lib().tfh_last_error().decode())
    if rc > 0:
        buf = C.create_string_buffer(rc)
        _lib().tfh_reduce_mux(tt, C.c_uint32(len(truth_table)), buf, C.c_size_t(len(buf)))
    return buf.value.decode()


def sbox_circuit_dump(inverse=False):
    buf = C.create_string_buffer(1 << 24)
    rc = _lib().tfh_sbox_circuit_dump(C.c_int(1 if inverse else 0), buf, C.c_size_t(len(buf)))
    if rc:
        raise HostError("circuit dump buffer too small")
    return buf.value.decode()
