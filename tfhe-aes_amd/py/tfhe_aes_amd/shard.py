"""Multi-GPU sharding of independent AES blocks: one process per GPU, keys replicated, block b runs
on rank b mod world_size with no data-path collective; the only exchange is the final gather of
the output ciphertexts ([blocks][128 bits][n+1] u32) over torch.distributed (RCCL on GPUs, gloo in
the CPU tests).  Independent blocks are what the reference parallelises in CTR mode
(/root/reference/modes/src/ctr.rs:26-35)."""
import numpy as np
import torch
import torch.distributed as dist


def block_partition(n_blocks, world_size, rank):
    """Global indices of the blocks owned by `rank` (round-robin)."""
    return list(range(rank, n_blocks, world_size))


def gather_blocks(local, n_blocks, device=None):
    """All-gather per-rank outputs into global block order.

    local: uint32 array [n_local, ...] holding this rank's blocks in block_partition order.
    Returns a uint32 array [n_blocks, ...] on every rank (ragged shards are padded for the collective).
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    local = np.ascontiguousarray(local, np.uint32)
    item_shape = local.shape[1:]
    per_rank = (n_blocks + world - 1) // world
    mine = block_partition(n_blocks, world, rank)
    assert local.shape[0] == len(mine), "shard size does not match the partition"
    buf = np.zeros((per_rank,) + item_shape, np.uint32)
    buf[: len(mine)] = local
    t = torch.from_numpy(buf.view(np.int32))
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = np.zeros((n_blocks,) + item_shape, np.uint32)
    for r in range(world):
        idx = block_partition(n_blocks, world, r)
        if idx:
            out[idx] = parts[r].cpu().numpy().view(np.uint32)[: len(idx)]
    return out


def gather_blocks_device(local, n_blocks):
    """The same gather for outputs that are already on the GPU (host.read_states_device): `local` is an int32 tensor
    [n_local, ...] in block_partition order, on the device of the process group's transport (CUDA under RCCL); one all_gather_into_tensor (RCCL over xGMI) of the padded shards,
    then a device-side reorder into global block order.  Returns a tensor [n_blocks, ...] on the same device on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_blocks + world - 1) // world
    mine = block_partition(n_blocks, world, rank)
    assert local.shape[0] == len(mine), "shard size does not match the partition"
    item_shape = tuple(local.shape[1:])
    buf = local
    if len(mine) < per_rank:  # ragged last round: pad the shard for the collective
        buf = torch.zeros((per_rank,) + item_shape, dtype=local.dtype, device=local.device)
        buf[: len(mine)] = local
    allbuf = torch.empty((world * per_rank,) + item_shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(allbuf, buf.contiguous())
    # block b sits at rank b % world, position b // world
    b = torch.arange(n_blocks, device=local.device)
    return allbuf[(b % world) * per_rank + b // world]
