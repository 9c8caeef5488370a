"""Benchmark of the hot path: batched TFHE boolean-gate bootstrapping on MI355X.

A step is one tfa_stage_submit of 65536 independent NAND/XOR gates (BASELINE.json configs[1]) whose
input ciphertexts already sit in the device pool.  With --gpus N every rank runs the same batch on
its own GPU with replicated keys (weak scaling, no collective in the data path); the only exchange
is a gather of output ciphertexts after the timed region.

Prints one JSON line: metric/value per the driver contract, plus
  roofline     - achieved fp64 FLOP/s of the blind-rotation kernel (algorithmic flop per launch /
                 mean kernel time from HIP events on the engine's stream) against the fp64 peak
  cpu_baseline - the CPU oracle (C restatement, fp64 FFT engine, OpenMP over gates) on a bounded
                 sample of the same workload, timed on this node's host cores (rank 0, N=1 only)
  aes          - seconds for one homomorphic AES-128 ECB block (FIPS-197 App. B, latency), and the
                 sharded case: 8 independent CTR blocks per GPU (SP 800-38A F.5.1 extended), keys
                 replicated, outputs gathered over the process group; both checked against the standard
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tfhe-aes_amd", "py"))

import numpy as np  # noqa: E402

METRIC = "gate bootstraps/sec per GPU + seconds per AES-128 ECB block (1/8 GPUs)"
GATES_PER_STEP = 65536
FLOP_PER_PBS = 805 * 206848          # 12 FFT-256 + 12 twists + 32 complex vector MACs per blind-rotation step
FP64_PEAK_TFLOPS = 78.6              # MI355X fp64 vector = matrix peak (datasheet)
SEED = 0x5446484541455321


def measured_traffic():
    """HBM bytes per 65536-bootstrap launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as gfx950 requires for 16-byte-per-lane reads, plus WRITE_SIZE; both in KiB).  Counters
    cannot be collected inside a timed run, so this is the profile's number, or None without the file."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_summary.json")) as f:
            cur = json.load(f)["counters_current_kernels"]
        c = next(v for k, v in cur.items() if k.startswith("tfa::pbs_kernel<8"))
        return int((2 * c["FETCH_SIZE@65536"] + c["WRITE_SIZE@65536"]) * 1024)
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def cpu_baseline(p, lwe_sk, bsk, ksk, budget_s=20.0):
    from oracle.pyoracle import NAND, XOR, Oracle
    o = Oracle()
    threads = o.num_threads()
    bsk_f = o.bsk_to_fourier(bsk)

    def run(count):
        bits = np.random.default_rng(3).integers(0, 2, 2 * count).astype(np.uint8)
        cts = o.encrypt_bits(lwe_sk, 9, 0, bits)
        ops = np.where(np.arange(count) % 2 == 0, NAND, XOR).astype(np.uint8)
        t = time.perf_counter()
        out = o.gates(bsk, bsk_f, ksk, ops, cts[:count], cts[count:])
        dt = time.perf_counter() - t
        exp = np.where(np.arange(count) % 2 == 0, 1 - (bits[:count] & bits[count:]), bits[:count] ^ bits[count:])
        assert (o.decrypt_bits(lwe_sk, out) == exp).all()
        return dt

    probe = 2 * threads
    rate = probe / run(probe)
    count = max(probe, int(rate * budget_s) // threads * threads)
    dt = run(count)
    return {"value": round(count / dt, 2), "unit": "gate bootstraps/s", "cores": threads, "kind": "port",
            "sample": f"{count} of the {GATES_PER_STEP} NAND/XOR gates, oracle fp64-FFT engine, OpenMP over gates, {dt:.1f} s"}


def aes_block(device):
    """One FIPS-197 Appendix B block through the host mirror (offline round keys), encrypt only timed."""
    from tfhe_aes_amd import host as H
    keys = H.gen_keys(SEED ^ 0x77, device)
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    rk = [H.Key.from_u8_enc(k, keys) for k in H.key_expansion_clear(key)]
    ecb = H.ECB(rk)
    state = H.State.from_u8_enc(bytes.fromhex("3243f6a8885a308d313198a2e0370734"), keys)
    warm = state.clone()
    warm.sub_bytes()
    keys.flush()
    before = keys.host_stats()
    t = time.perf_counter()
    ecb.encrypt(state)
    keys.flush()
    dt = time.perf_counter() - t
    after = keys.host_stats()
    ok = state.decrypt_to_u8().hex() == "3925841d02dc09fbdc118597196a0b32"
    keys.close()
    stages = after["stages"] - before["stages"]
    return {"ecb_encrypt_block_s": round(dt, 4), "correct": ok, "stages": stages, "gates": after["gates"] - before["gates"], "pbs": 127872,
            "mean_stage_width_pbs": round(127872 / max(1, stages), 1)}


def aes_ctr_sharded(device, rank, world, coll_dev, blocks_per_gpu=8):
    """blocks_per_gpu * world independent CTR blocks, block b on rank b mod world (BASELINE configs[4] shape).
    Returns (seconds for this rank's shard, gathered output ciphertexts or None)."""
    import torch.distributed as dist
    from tfhe_aes_amd import host as H
    from tfhe_aes_amd import shard
    n_blocks = blocks_per_gpu * world
    keys = H.gen_keys(SEED ^ 0x99, device)                     # same seed everywhere: replicated key material
    rk = [H.Key.from_u8_enc(k, keys) for k in H.key_expansion_clear(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"))]
    ctr0 = int("f0f1f2f3f4f5f6f7f8f9fafbfcfdfeff", 16)
    nist_pt = ["6bc1bee22e409f96e93d7e117393172a", "ae2d8a571e03ac9c9eb76fac45af8e51", "30c81c46a35ce411e5fbc1191a0a52ef",
               "f69f2445df4f9b17ad2b417be66c3710"]
    rng = np.random.default_rng(4242)
    pts = [bytes.fromhex(nist_pt[b]) if b < 4 else rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for b in range(n_blocks)]
    mine = shard.block_partition(n_blocks, world, rank)
    counters = [H.State.from_u128_enc((ctr0 + b) % (1 << 128), keys) for b in mine]
    states = [H.State.from_u8_enc(pts[b], keys) for b in mine]
    keys.flush()
    if world > 1:
        import torch
        dist.all_reduce(torch.zeros(1, device=coll_dev))  # barrier on the transport in use
    t = time.perf_counter()
    H.CTR(rk, counters, len(mine)).encrypt(states)
    keys.flush()
    dt = time.perf_counter() - t
    local = np.stack([st.read() for st in states])
    out = shard.gather_blocks(local, n_blocks, device=coll_dev) if world > 1 else local
    ok = None
    if rank == 0:
        nist_ct = ["874d6191b620e3261bef6864990db6ce", "9806f66b7970fdff8617187bb9fffdff", "5ae4df3edbd5d35e5b4f09020db03eab",
                   "1e031dda2fbe03d1792170a0f3009cee"]
        ok = all(H.State.import_lwe(out[b], keys).decrypt_to_u8().hex() == nist_ct[b] for b in range(4))
    keys.close()
    return dt, n_blocks, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aes", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tfhe_aes_amd import engine as E
    from tfhe_aes_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo" rehearses several ranks on one GPU
        if backend == "nccl":
            # one group with both transports: CUDA tensors travel over RCCL / xGMI, CPU tensors over gloo.  One probe
            # collective, so that a broken fabric shows here and not after the timed steps; the timed path has no
            # collective, so the gather and the barriers can fall back to gloo.
            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
            try:
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as exc:
                print(f"[bench] rank {rank}: RCCL unavailable ({exc!r}); collectives fall back to gloo", file=sys.stderr, flush=True)
                backend = "gloo"
            # every rank takes the same transport: one failed probe moves all of them to gloo
            ok = torch.tensor([1 if backend == "nccl" else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                backend = "gloo"
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        backend = "none"

    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    p = E.default_params()
    lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, SEED)       # keys replicated: same seed on every rank
    ctx = E.Context(p, local_rank)
    ctx.keys_upload(bsk, ksk)
    G = GATES_PER_STEP
    bits = np.random.default_rng(100 + rank).integers(0, 2, 2 * G).astype(np.uint8)
    ctx.pool_reserve(3 * G)
    ctx.pool_write(np.arange(2 * G), E.client_encrypt(p, lwe_sk, 1000 + rank, bits))   # inputs resident in HBM
    even = np.arange(G) % 2 == 0
    gates = E.make_gates(np.where(even, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))

    def dist_barrier():  # a collective on the transport in use (dist.barrier() would pick the CUDA one by itself)
        dist.all_reduce(torch.zeros(1, device=coll_dev))

    def barrier():
        if world > 1:
            dist_barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        ctx.stage_submit(gates)
    barrier()
    ctx.reset_stats()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.stage_submit(gates)
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the timed work + the one collective of the path: gather a slice of the outputs
    out = ctx.pool_read(np.arange(2 * G, 3 * G))
    got = E.client_decrypt(p, lwe_sk, out)
    wrong = int((got != np.where(even, 1 - (bits[:G] & bits[G:]), bits[:G] ^ bits[G:])).sum())
    gathered = None
    if world > 1:
        gathered = shard.gather_blocks(out[None, :128], world, device=coll_dev)
        w = torch.tensor([wrong], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(w)
        wrong = int(w.item())

    if rank == 0:
        pbs_ms = st.pbs_kernel_ms / max(1, st.pbs_launches)
        ks_ms = st.ks_kernel_ms / max(1, st.ks_launches)
        achieved = G * FLOP_PER_PBS / (pbs_ms * 1e-3) / 1e12 if pbs_ms > 0 else 0.0
        line = {
            "metric": METRIC, "value": round(world * G * args.steps / elapsed, 1), "unit": "gate bootstraps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "65536 independent tfhe::boolean NAND/XOR gate bootstraps per GPU (BASELINE configs[1])",
                       "gates_per_step_per_gpu": G, "params": "n=805,k=3,N=512,pbs=2^10x2,ks=2^3x5", "parallelism": f"replicas x{world}, keys replicated"},
            "wrong_results": wrong,
            "roofline": {"bound": "mfma", "kernel": "pbs_kernel", "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_PEAK_TFLOPS, 4), "traffic": measured_traffic(), "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, profiles/r1_pmc_summary.json)",
                         "kernel_ms": round(pbs_ms, 3),
                         "keyswitch_kernel_ms": round(ks_ms, 3)},
        }
        if gathered is not None:
            line["gathered_ciphertexts"] = int(gathered.shape[0] * gathered.shape[1])
            line["collective_backend"] = "rccl" if backend == "nccl" else backend
    ctx.close()
    if not args.no_aes:
        # every rank takes part: 8 CTR blocks per GPU, max time over ranks
        dt, n_blocks, ok = aes_ctr_sharded(local_rank, rank, world, coll_dev)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            line["aes"] = {"ctr_blocks": n_blocks, "ctr_seconds": round(dt, 3), "ctr_s_per_block": round(dt / n_blocks, 4),
                           "ctr_first_blocks_match_sp800_38a": ok, "sharding": f"block b on GPU b mod {world}, keys replicated, outputs gathered"}
    if rank == 0:
        if world == 1 and not args.no_aes:
            line["aes"].update(aes_block(local_rank))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p, lwe_sk, bsk, ksk)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
