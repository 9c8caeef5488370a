"""The multi-GPU path on CPU: world_size-2 gloo processes shard blocks round-robin and gather the
output ciphertexts in block order (ragged and empty shards included)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tfhe_aes_amd import shard


def fake_block(b):
    return (np.arange(128 * 6, dtype=np.uint32).reshape(128, 6) * 2654435761 + b * 40503) & 0xFFFFFFFF


def worker(rank, world, n_blocks, port, ok):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.block_partition(n_blocks, world, rank)
    local = np.stack([fake_block(b) for b in mine]) if mine else np.zeros((0, 128, 6), np.uint32)
    out = shard.gather_blocks(local, n_blocks)
    good = out.shape == (n_blocks, 128, 6) and all((out[b] == fake_block(b)).all() for b in range(n_blocks))
    # the tensor form the GPUs use (one all_gather_into_tensor, reorder on the device), here on CPU tensors over gloo
    import torch
    out_t = shard.gather_blocks_device(torch.from_numpy(local.view(np.int32)), n_blocks)
    good = good and tuple(out_t.shape) == (n_blocks, 128, 6) and (out_t.numpy().view(np.uint32) == out).all()
    ok[rank] = 1 if good else 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_blocks", [4, 5, 1])
def test_gather_blocks_world2(n_blocks):
    ctx = mp.get_context("spawn")
    ok = ctx.Array("i", [0, 0])
    port = 29500 + (os.getpid() % 2000) + n_blocks
    procs = [ctx.Process(target=worker, args=(r, 2, n_blocks, port, ok)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(ok) == [1, 1]


def test_partition_covers_all_blocks():
    for n, w in [(64, 8), (5, 2), (1, 8), (0, 4)]:
        parts = [shard.block_partition(n, w, r) for r in range(w)]
        assert sorted(b for p in parts for b in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_bench_launcher_starts_ranks_and_fails_loudly_without_gpus():
    """`bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself; with no GPU each rank refuses to run (there is no
    CPU fallback) and the launcher exits non-zero.  A launcher that started a different number of ranks is refused."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") >= 1 and "rank exit codes" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
