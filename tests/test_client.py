"""Client side of the product (key generation, encryption, decryption; host CPU as in the
reference) against the oracle: same seed -> same key material, word for word."""
import numpy as np

from tfhe_aes_amd import engine as E


def test_keygen_matches_oracle(oracle, okeys):
    from conftest import MASTER_SEED
    got = E.client_keygen(E.default_params(), MASTER_SEED)
    for a, b in zip(got, okeys):
        assert a.shape == b.shape and (a == b).all()
    assert set(np.unique(got[0])) <= {0, 1} and set(np.unique(got[1])) <= {0, 1}
    assert 300 < got[0].sum() < 500


def test_encrypt_matches_oracle_and_roundtrips(oracle, okeys):
    p = E.default_params()
    bits = np.random.default_rng(8).integers(0, 2, 300).astype(np.uint8)
    cts = E.client_encrypt(p, okeys[0], 1234, bits)
    assert (cts == oracle.encrypt_bits(okeys[0], 1234, 0, bits)).all()
    assert (E.client_decrypt(p, okeys[0], cts) == bits).all()
    assert (oracle.decrypt_bits(okeys[0], cts) == bits).all()


def test_os_seeded_encryption_is_fresh(okeys):
    p = E.default_params()
    a = E.client_encrypt(p, okeys[0], 0, np.ones(2, np.uint8))
    b = E.client_encrypt(p, okeys[0], 0, np.ones(2, np.uint8))
    assert (a != b).any() and E.client_decrypt(p, okeys[0], a).all() and E.client_decrypt(p, okeys[0], b).all()


def test_empty_batches():
    p = E.default_params()
    sk = np.zeros(p.lwe_dimension, np.uint32)
    assert E.client_encrypt(p, sk, 1, np.zeros(0, np.uint8)).shape == (0, p.lwe_dimension + 1)
    assert E.client_decrypt(p, sk, np.zeros((0, p.lwe_dimension + 1), np.uint32)).size == 0
