import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))

MASTER_SEED = 0x5446484541455321


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def okeys(oracle):
    """(lwe_sk, glwe_sk, bsk, ksk) from the oracle's own key generation."""
    return oracle.keygen(MASTER_SEED)


@pytest.fixture(scope="session")
def obsk_f(oracle, okeys):
    return oracle.bsk_to_fourier(okeys[2])


@pytest.fixture(scope="session")
def gpu_ctx(okeys):
    """Engine context on cuda:0 holding the ORACLE's keys, so outputs are comparable word for word."""
    from tfhe_aes_amd import engine as E
    ctx = E.Context(E.default_params(), 0)
    ctx.keys_upload(okeys[2], okeys[3])
    yield ctx
    ctx.close()


@pytest.fixture(scope="module", params=[1, 2, 4, 8, "1c", "2c", "4c", "8c"], ids=lambda v: f"variant{v}")
def variant_ctx(request, okeys):
    """Engine context pinned to one bootstrap kernel variant (the switches are read when the context is created):
    1 / 2 = one bootstrap per workgroup with one / two workgroups per CU, 4 / 8 = one bootstrap per wave with 4 / 8 waves
    per workgroup; a trailing "c" selects the compiler-scheduled twin of the hand-scheduled kernel (TFA_PBS_ASM=0,
    TFA_LAT_ASM=0)."""
    from tfhe_aes_amd import engine as E
    compiled = str(request.param).endswith("c")
    env = {"TFA_PBS_WAVES": str(request.param).rstrip("c")}
    if compiled:
        env.update(TFA_PBS_ASM="0", TFA_LAT_ASM="0")
    old = {k: os.environ.get(k) for k in ("TFA_PBS_WAVES", "TFA_PBS_ASM", "TFA_LAT_ASM")}
    os.environ.update(env)
    try:
        ctx = E.Context(E.default_params(), 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.keys_upload(okeys[2], okeys[3])
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def hkeys():
    """gen_keys() of the host mirror (its own client-side key generation) on cuda:0."""
    from tfhe_aes_amd import host as H
    k = H.gen_keys(MASTER_SEED ^ 0xA5A5, 0)
    yield k
    k.close()
