"""Command-line driver (tfhe-aes_amd/cli/main.cpp), the counterpart of the reference binary
(/root/reference/tfhe-aes/src/main.rs:15-32): flag parsing and error behaviour on the CPU, a full run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tfhe-aes_amd", "tfhe-aes-hip")
KEY, IV = "2b7e151628aed2a6abf7158809cf4f3c", "000102030405060708090a0b0c0d0e0f"


def run(*args, timeout=60):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=timeout)


def test_cli_is_built():
    assert os.access(CLI, os.X_OK), "run `make -C tfhe-aes_amd`"


def test_cli_rejects_bad_arguments():
    r = run("-k", "00", "-i", IV)
    assert r.returncode == 1 and "Invalid key format" in r.stderr and "currently 2 characters" in r.stderr
    r = run("-k", KEY, "-i", "zz" * 16)
    assert r.returncode == 1 and "Invalid IV format" in r.stderr
    r = run("-k", KEY, "-i", IV, "-m", "GCM")
    assert r.returncode == 1 and "Invalid mode: GCM" in r.stderr
    assert "Mode: GCM" in r.stdout and "Key Expansion Offline: false" in r.stdout  # the reference echoes its arguments first
    assert run("-k", KEY).returncode == 2
    assert run("--bogus").returncode == 2


@pytest.mark.gpu
def test_cli_ctr_two_blocks_matches_clear_aes():
    r = run("-k", KEY, "-i", IV, "-n", "2", "-x", "-m", "CTR", "--seed", "11", timeout=300)
    assert r.returncode == 0, r.stderr
    assert "AES of #2 outputs computed in" in r.stdout and "AES of #2 outputs decrypted in" in r.stdout
    assert "CTR mode test passed" in r.stdout
