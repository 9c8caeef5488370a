"""Benchmark of the hot path: batched TFHE boolean-gate bootstrapping on MI355X.

A step is one tfa_stage_submit of 65536 independent NAND/XOR gates (BASELINE.json configs[1]) whose
input ciphertexts already sit in the device pool.  With --gpus N every rank runs the same batch on
its own GPU with replicated keys (weak scaling, no collective in the data path); the only exchange
is a gather of output ciphertexts after the timed region.

Launching: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself
(child processes, spawned before anything touches the GPU, rendezvous on 127.0.0.1); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.  The
collectives run on RCCL; a failed RCCL probe ends the run with a non-zero exit.  BENCH_DIST_BACKEND=gloo
is the explicit switch for rehearsing several ranks on one GPU.

Prints one JSON line: metric/value per the driver contract, plus
  roofline     - achieved fp64 FLOP/s of the blind-rotation kernel (algorithmic flop per launch /
                 mean kernel time from HIP events on the engine's stream) against the fp64 peak
  cpu_baseline - the CPU oracle (C restatement, fp64 FFT engine, OpenMP over gates) on a bounded
                 sample of the same workload, timed on this node's host cores (rank 0, N=1 only)
  aes          - the AES configurations of BASELINE.json on the same engine, each checked against
                 FIPS-197 / SP 800-38A (tests/golden/aes_vectors.json): one ECB block encrypt and decrypt
                 (configs[2]), online key expansion and the 4-block CBC and OFB chains (configs[3]; N=1 only:
                 sequential chains do not shard), and 8 independent CTR blocks per GPU, block b on rank
                 b mod N, outputs all-gathered on the device (configs[4]: 64 blocks on 8 GPUs)
"""
import argparse
import json
import os
import socket
import subprocess
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
HBM_PEAK_GBPS = 8000.0               # MI355X HBM3E
PMC_SUMMARIES = ("r3_pmc_summary.json", "r2_pmc_summary.json", "r1_pmc_summary.json")   # newest first


def measured_traffic():
    """HBM bytes per 65536-bootstrap launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as gfx950 requires for 16-byte-per-lane reads, plus WRITE_SIZE; both in KiB).  Counters
    cannot be collected inside a timed run, so this is the newest profile's number (with its file name), or None."""
    for name in PMC_SUMMARIES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                cur = json.load(f)["counters_current_kernels"]
            c = next(v for k, v in cur.items() if "pbs_asm_kernel<0, 8>" in k or "pbs_asm_kernel<0>" in k or k.startswith("tfa::pbs_kernel<8"))
            return int((2 * c["FETCH_SIZE@65536"] + c["WRITE_SIZE@65536"]) * 1024), name
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    return None, None


def measured_issue_profile():
    """Ratios of the dominant kernel from the newest committed PMC passes (derived by tools/pmc_summary.py): share of wave
    cycles spent waiting on LDS instructions and issuing, instructions per wave and blind-rotation step, and the FMA share
    of the loop's fp64 instructions; `source` names the file, the commit and the kernel the counters were collected on.
    None when no summary is readable."""
    for name in PMC_SUMMARIES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                full = json.load(f)
            d = full["derived_pbs_asm_kernel"]
            src = "profiles/" + name
            if full.get("commit"):
                src += " @ " + str(full["commit"])[:12]
            if full.get("kernel"):
                src += " (" + full["kernel"] + ")"
            return {"sq_wait_inst_lds_per_wave_cycle": d["SQ_WAIT_INST_LDS/SQ_WAVE_CYCLES"], "sq_active_inst_any_per_wave_cycle": d["SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES"],
                    "valu_instructions_per_wave_step": d["valu_instructions_per_wave_step"], "lds_instructions_per_wave_step": d["lds_instructions_per_wave_step"],
                    "fp64_instructions_per_wave_step": sum(d["fp64_instructions_per_wave_step"][k] for k in ("v_fma_f64", "v_add_f64", "v_mul_f64")),
                    "fp64_fma_share": d["fp64_instructions_per_wave_step"]["fma_share"], "source": src}
        except (OSError, KeyError, ValueError):
            continue
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
    # BASELINE configs[0]: one FHEByte::sub_byte (0x01 -> 0x7c, primitive.rs:438-452) stage by stage on the same host cores,
    # every gate of a stage in one OpenMP-parallel oracle call (the reference's rayon loop, primitive.rs:253-262)
    from oracle import boolean_tree_oracle as B
    from oracle.subbyte_oracle import sub_byte_on_oracle
    x = 0x01
    cts = o.encrypt_bits(lwe_sk, 77, 0, np.array([(x >> k) & 1 for k in range(8)], np.uint8))
    t = time.perf_counter()
    out, n_gates = sub_byte_on_oracle(o, (lwe_sk, None, bsk, ksk), bsk_f, cts, B.S_BOX)
    sub_byte_s = time.perf_counter() - t
    got = o.decrypt_bits(lwe_sk, np.stack(out))
    assert sum(int(b) << (7 - i) for i, b in enumerate(got)) == 0x7C and n_gates == 405
    return {"value": round(count / dt, 2), "unit": "gate bootstraps/s", "cores": threads, "kind": "port",
            "sample": f"{count} of the {GATES_PER_STEP} NAND/XOR gates, oracle fp64-FFT engine, OpenMP over gates, {dt:.1f} s",
            "sub_byte_s": round(sub_byte_s, 3), "sub_byte_threads": threads,
            "sub_byte_sample": "one encrypted byte 0x01 -> 0x7c through the staged S-box (405 gates, 749 bootstraps, 7 stages; BASELINE configs[0])"}


def golden():
    with open(os.path.join(ROOT, "tests", "golden", "aes_vectors.json")) as f:
        return json.load(f)


class Leg:
    """Times one AES leg on the host mirror: wall seconds around the recorded gates' execution, plus the stages,
    gates and bootstraps the engine ran for it."""

    def __init__(self, keys):
        self.keys = keys

    def __enter__(self):
        self.keys.flush()
        self.h0, self.e0 = self.keys.host_stats(), self.keys.engine_stats()
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.keys.flush()
        self.seconds = time.perf_counter() - self.t0
        h1, e1 = self.keys.host_stats(), self.keys.engine_stats()
        self.stages = h1["stages"] - self.h0["stages"]
        self.gates = h1["gates"] - self.h0["gates"]
        self.pbs = int(e1.pbs_count - self.e0.pbs_count)

    def fields(self, prefix):
        return {f"{prefix}_s": round(self.seconds, 4), f"{prefix}_stages": self.stages, f"{prefix}_pbs": self.pbs,
                f"{prefix}_mean_stage_width_pbs": round(self.pbs / max(1, self.stages), 1)}


WIDTH_BINS = (128, 256, 512, 1024, 2048, 4096)


def width_table(records):
    """Bootstrap launches of a leg (tfa_profile_launches records, kind 0) by stage width: launches, bootstraps, device
    milliseconds and bootstraps per second, the per-width view of where a single block's time goes."""
    rows = []
    pbs = records[records["kind"] == 0]
    lo = 0
    for hi in WIDTH_BINS + (1 << 31,):
        sel = pbs[(pbs["units"] > lo) & (pbs["units"] <= hi)]
        if len(sel):
            ms = float(sel["ms"].sum())
            rows.append({"width_le": hi if hi < (1 << 31) else None, "launches": int(len(sel)), "bootstraps": int(sel["units"].sum()),
                         "ms": round(ms, 2), "kpbs_per_s": round(float(sel["units"].sum()) / ms, 1) if ms > 0 else None})
        lo = hi
    ks = records[records["kind"] == 1]
    return {"bootstrap_launches_by_width": rows, "bootstrap_ms": round(float(pbs["ms"].sum()), 2), "keyswitch_ms": round(float(ks["ms"].sum()), 2)}


def aes_single_gpu(device, chains=True):
    """BASELINE configs[2] and (with `chains`) configs[3] on one GPU, every result checked against the standards' vectors."""
    from tfhe_aes_amd import host as H
    g = golden()
    fb, sp = g["fips197_b"], g["sp800_38a"]
    keys = H.gen_keys(SEED ^ 0x77, device)
    out = {}
    key = bytes.fromhex(fb["key"])
    clear_rk = H.key_expansion_clear(key)
    rk = [H.Key.from_u8_enc(k, keys) for k in clear_rk]            # offline expansion: round keys encrypted by the client
    ecb = H.ECB(rk)
    warm = H.State.from_u8_enc(bytes(16), keys)
    warm.sub_bytes()
    keys.flush()
    # configs[2]: FIPS-197 Appendix B block, encrypt then decrypt (ecb.rs:16-50)
    state = H.State.from_u8_enc(bytes.fromhex(fb["plaintext"]), keys)
    with Leg(keys) as leg:
        ecb.encrypt(state)
    ok_enc = state.decrypt_to_u8().hex() == fb["ciphertext"]
    out.update(leg.fields("ecb_encrypt_block"))
    out.update({"correct": ok_enc, "stages": leg.stages, "gates": leg.gates, "pbs": leg.pbs,
                "mean_stage_width_pbs": round(leg.pbs / max(1, leg.stages), 1)})
    with Leg(keys) as leg:
        ecb.decrypt(state)
    out.update(leg.fields("ecb_decrypt_block"))
    out["ecb_decrypt_correct"] = state.decrypt_to_u8().hex() == fb["plaintext"]
    # the same block with every gate in the association the reference's code records it in (no re-association of xor chains:
    # MixColumns and AddRoundKey as five dependent xor layers instead of three): same gates, 117 stages
    keys.set_xor_rebalance(False)
    state = H.State.from_u8_enc(bytes.fromhex(fb["plaintext"]), keys)
    with Leg(keys) as leg:
        ecb.encrypt(state)
    out.update(leg.fields("ecb_encrypt_block_as_recorded"))
    out["ecb_encrypt_block_as_recorded_correct"] = state.decrypt_to_u8().hex() == fb["ciphertext"]
    keys.set_xor_rebalance(True)
    out["xor_chain_rebalance"] = "on (default) for every leg but *_as_recorded: same gates and results, fewer dependent stages"
    # the same encryption once more with HIP events around every launch (outside the timed leg above): where the block's time goes
    state = H.State.from_u8_enc(bytes.fromhex(fb["plaintext"]), keys)
    keys.flush()
    keys.reset_engine_stats()
    keys.set_profiling(True)
    ecb.encrypt(state)
    keys.flush()
    out["ecb_encrypt_block_profile"] = width_table(keys.profile_launches())
    keys.set_profiling(False)
    if not chains:
        keys.close()
        return out
    # configs[3]: online key expansion (key_schedule.rs:162-171), then 4-block CBC and OFB chains (cbc.rs:22-31, ofb.rs:22-38)
    k0 = H.Key.from_u8_enc(bytes.fromhex(sp["key"]), keys)
    with Leg(keys) as leg:
        online = k0.generate_round_keys()
    out.update(leg.fields("key_expansion_online"))
    out["key_expansion_online_correct"] = [k.decrypt_to_u8() for k in online] == H.key_expansion_clear(bytes.fromhex(sp["key"]))
    iv = H.State.from_u8_enc(bytes.fromhex(sp["iv"]), keys)
    pts = [bytes.fromhex(x) for x in sp["plaintext"]]
    states = [H.State.from_u8_enc(x, keys) for x in pts]
    with Leg(keys) as leg:
        H.CBC(online, iv, 4).encrypt(states)
    out.update(leg.fields("cbc4_encrypt"))
    out["cbc4_correct"] = [s.decrypt_to_u8().hex() for s in states] == sp["cbc"]
    states = [H.State.from_u8_enc(x, keys) for x in pts]
    with Leg(keys) as leg:
        H.OFB(online, iv, 4).encrypt(states)
    out.update(leg.fields("ofb4"))
    out["ofb4_correct"] = [s.decrypt_to_u8().hex() for s in states] == sp["ofb"]
    keys.close()
    return out


def aes_ctr_sharded(device, rank, world, on_device, blocks_per_gpu=8):
    """blocks_per_gpu * world independent CTR blocks, block b on rank b mod world (BASELINE configs[4]: 64 blocks on
    8 GPUs).  The outputs are all-gathered: on the device (RCCL) when `on_device`, through host tensors (gloo) otherwise.
    Returns (seconds for this rank's shard, blocks, every gathered block correct against a clear AES / the standard)."""
    import torch
    import torch.distributed as dist
    from tfhe_aes_amd import host as H
    from tfhe_aes_amd import shard
    sp = golden()["sp800_38a"]
    n_blocks = blocks_per_gpu * world
    keys = H.gen_keys(SEED ^ 0x99, device)                     # same seed everywhere: replicated key material
    key = bytes.fromhex(sp["key"])
    rk = [H.Key.from_u8_enc(k, keys) for k in H.key_expansion_clear(key)]
    ctr0 = int(sp["ctr_init"], 16)
    rng = np.random.default_rng(4242)
    pts = [bytes.fromhex(sp["plaintext"][b]) if b < 4 else rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for b in range(n_blocks)]
    mine = shard.block_partition(n_blocks, world, rank)
    counters = [H.State.from_u128_enc((ctr0 + b) % (1 << 128), keys) for b in mine]
    states = [H.State.from_u8_enc(pts[b], keys) for b in mine]
    keys.flush()
    if world > 1:
        dist.barrier()
    t = time.perf_counter()
    H.CTR(rk, counters, len(mine)).encrypt(states)
    keys.flush()
    dt = time.perf_counter() - t
    words = keys.params.lwe_dimension + 1
    if on_device:
        local = torch.empty((len(mine), 128, words), dtype=torch.int32, device=torch.device("cuda", device))
        H.read_states_device(states, local)
        allb = shard.gather_blocks_device(local, n_blocks) if world > 1 else local
        torch.cuda.synchronize()
        out = allb.cpu().numpy().view(np.uint32) if rank == 0 else None
    else:
        local = np.stack([st.read() for st in states])
        out = shard.gather_blocks(local, n_blocks) if world > 1 else local
    ok = None
    if rank == 0:
        got = [H.State.import_lwe(out[b], keys).decrypt_to_u8() for b in range(n_blocks)]
        crk = H.key_expansion_clear(key)
        exp = [bytes(x ^ y for x, y in zip(aes128_encrypt_block(crk, ((ctr0 + b) % (1 << 128)).to_bytes(16, "big")), pts[b])) for b in range(n_blocks)]
        ok = got == exp and [x.hex() for x in got[:4]] == sp["ctr"]
    keys.close()
    return dt, n_blocks, ok


def _aes_sbox():
    """The AES S-box from its definition (inverse in GF(2^8) followed by the affine map), for the clear check below."""
    def mul(a, b):
        r = 0
        for _ in range(8):
            if b & 1:
                r ^= a
            a = ((a << 1) ^ 0x11B) if a & 0x80 else (a << 1)
            b >>= 1
        return r
    box = []
    for x in range(256):
        inv = next((y for y in range(1, 256) if mul(x, y) == 1), 0)
        box.append(inv ^ ((inv << 1 | inv >> 7) & 0xFF) ^ ((inv << 2 | inv >> 6) & 0xFF) ^ ((inv << 3 | inv >> 5) & 0xFF) ^ ((inv << 4 | inv >> 4) & 0xFF) ^ 0x63)
    return box


def aes128_encrypt_block(round_keys, block):
    """Clear AES-128 of one block from the 11 expanded round keys (FIPS-197), column-major state as in the standard."""
    box = _aes_sbox() if not hasattr(aes128_encrypt_block, "box") else aes128_encrypt_block.box
    aes128_encrypt_block.box = box
    xt = lambda a: ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1
    s = [b ^ k for b, k in zip(block, round_keys[0])]
    for rnd in range(1, 11):
        s = [box[b] for b in s]
        s = [s[(i + 4 * (i % 4)) % 16] for i in range(16)]                       # ShiftRows on s[4 c + r]
        if rnd < 10:
            m = []
            for c in range(4):
                a = s[4 * c:4 * c + 4]
                m += [xt(a[i]) ^ xt(a[(i + 1) % 4]) ^ a[(i + 1) % 4] ^ a[(i + 2) % 4] ^ a[(i + 3) % 4] for i in range(4)]
            s = m
        s = [b ^ k for b, k in zip(s, round_keys[rnd])]
    return bytes(s)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Starts the n ranks of this benchmark as child processes and relays rank 0's JSON line.  Runs before anything in
    this process has touched the GPU (no torch.cuda call, no HIP call): the parent only waits."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    # host threads (key generation, encryption: OpenMP) are shared out over the ranks instead of every rank taking them all
    omp = str(host_threads_per_rank(n))
    deadline = time.monotonic() + float(os.environ.get("BENCH_TIMEOUT_S", "1500"))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS=omp, BENCH_OMP_CAPPED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # wait for all of them; when one fails, the others (stuck in the rendezvous or a collective with a dead peer) are
    # ended by their exact process handles
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes, timed_out = [None] * n, False
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        timed_out = any(c is None for c in codes) and time.monotonic() > deadline
        if any(c not in (None, 0) for c in codes) or timed_out:
            if timed_out:
                print("[bench] ranks still running at BENCH_TIMEOUT_S: ending them", file=sys.stderr)
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if any(codes) or timed_out:
        print(f"[bench] rank exit codes {codes}", file=sys.stderr)
        raise SystemExit(1)


def host_threads_per_rank(n):
    """OpenMP threads each of n ranks on this node may use: the node's share (OMP_NUM_THREADS when the environment sets it,
    else the CPUs this process may run on) divided by the ranks, at least one."""
    try:
        total = int(os.environ["OMP_NUM_THREADS"])
    except (KeyError, ValueError):
        total = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, total // max(1, n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aes", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args.gpus, sys.argv[1:])
        world, rank = 1, 0
    else:
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import datetime
    import torch
    import torch.distributed as dist
    from tfhe_aes_amd import engine as E

    if world > 1 and "BENCH_OMP_CAPPED" not in os.environ and os.environ.get("LOCAL_WORLD_SIZE"):
        # started by an external launcher (torch.distributed.run): cap this rank's OpenMP team the way launch_ranks() does
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(host_threads_per_rank(int(os.environ["LOCAL_WORLD_SIZE"])))
        except OSError:
            pass
    # a rank that never reaches the rendezvous, or a collective that never completes, ends the run non-zero instead of hanging
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("BENCH_RENDEZVOUS_TIMEOUT_S", "600")))
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo": the explicit switch to rehearse several ranks on one GPU
        if backend == "nccl" and local_rank >= n_dev:
            raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({n_dev} visible); RCCL needs one GPU per rank "
                             "(set BENCH_DIST_BACKEND=gloo to rehearse on fewer GPUs)")
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            # CUDA tensors travel over RCCL / xGMI; CPU tensors (barriers, scalars) over gloo.  One probe collective so a
            # broken fabric ends the run here, with a non-zero exit, and not behind a green line.
            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=pg_timeout)
            probe = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise SystemExit(f"bench.py: RCCL probe all_reduce returned {probe.item()} on rank {rank}, expected {world}")
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    on_device = backend in ("nccl", "none")

    p = E.default_params()
    lwe_sk, glwe_sk, bsk, ksk = E.client_keygen(p, SEED)       # keys replicated: same seed on every rank
    ctx = E.Context(p, local_rank)
    ctx.keys_upload(bsk, ksk)
    G = GATES_PER_STEP
    bits = np.random.default_rng(100 + rank).integers(0, 2, 2 * G).astype(np.uint8)
    ctx.pool_reserve(3 * G)
    fresh = E.client_encrypt(p, lwe_sk, 1000 + rank, bits)
    ctx.pool_write(np.arange(16), fresh[:16])                  # first touch of the staging buffers stays out of the figure below
    t_io = time.perf_counter()
    ctx.pool_write(np.arange(2 * G), fresh)                    # inputs resident in HBM before the timed region
    h2d_ms = (time.perf_counter() - t_io) * 1e3
    even = np.arange(G) % 2 == 0
    gates = E.make_gates(np.where(even, E.NAND, E.XOR), np.arange(G), np.arange(G, 2 * G), None, np.arange(2 * G, 3 * G))

    def barrier():
        if world > 1:
            dist.barrier()
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
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the timed work + the one collective of the path: gather a slice of the outputs, device to device
    ctx.pool_read(np.arange(2 * G, 2 * G + 16))
    t_io = time.perf_counter()
    out = ctx.pool_read(np.arange(2 * G, 3 * G))
    d2h_ms = (time.perf_counter() - t_io) * 1e3
    got = E.client_decrypt(p, lwe_sk, out)
    wrong = int((got != np.where(even, 1 - (bits[:G] & bits[G:]), bits[:G] ^ bits[G:])).sum())
    gathered = None
    if world > 1:
        from tfhe_aes_amd import shard
        if on_device:
            local = torch.empty((1, 128, p.lwe_dimension + 1), dtype=torch.int32, device=torch.device("cuda", local_rank))
            ctx.pool_read_device(np.arange(2 * G, 2 * G + 128), local.data_ptr())
            allb = shard.gather_blocks_device(local, world)
            torch.cuda.synchronize()
            mine_ok = bool((allb[rank].cpu().numpy().view(np.uint32) == out[:128]).all())
            gathered = tuple(allb.shape)
        else:
            allb = shard.gather_blocks(out[None, :128], world)
            mine_ok = bool((allb[rank] == out[:128]).all())
            gathered = allb.shape
        w = torch.tensor([wrong + (0 if mine_ok else 1)], dtype=torch.int64)
        dist.all_reduce(w)
        wrong = int(w.item())

    line = None
    if rank == 0:
        pbs_ms = st.pbs_kernel_ms / max(1, st.pbs_launches)
        ks_ms = st.ks_kernel_ms / max(1, st.ks_launches)
        achieved = G * FLOP_PER_PBS / (pbs_ms * 1e-3) / 1e12 if pbs_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic()
        line = {
            "metric": METRIC, "value": round(world * G * args.steps / elapsed, 1), "unit": "gate bootstraps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "65536 independent tfhe::boolean NAND/XOR gate bootstraps per GPU (BASELINE configs[1])",
                       "gates_per_step_per_gpu": G, "params": "n=805,k=3,N=512,pbs=2^10x2,ks=2^3x5", "parallelism": f"replicas x{world}, keys replicated"},
            "wrong_results": wrong,
            # ciphertext transfer of one batch through the boundary's host-buffer calls, outside the timed region: 2 x 65536 input
            # ciphertexts up (tfa_pool_write, 3224 B each), 65536 results down (tfa_pool_read); pageable host memory, staged copies
            "h2d_ms": round(h2d_ms, 2), "d2h_ms": round(d2h_ms, 2),
            # the contract's two bounds are hbm and mfma; this kernel is bound by the fp64 arithmetic peak, which on
            # MI355X is the same number for the vector units it runs on and for the matrix cores ("bound_detail")
            "roofline": {"bound": "mfma", "bound_detail": "fp64 vector FMA units (peak equal to the fp64 matrix peak)", "kernel": "pbs_asm_kernel",
                         "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC passes, profiles/{traffic_src})",
                         "hbm_gbps": round(traffic / (pbs_ms * 1e-3) / 1e9, 1) if traffic and pbs_ms > 0 else None,
                         "hbm_frac": round(traffic / (pbs_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic and pbs_ms > 0 else None,
                         "kernel_ms": round(pbs_ms, 3), "keyswitch_kernel_ms": round(ks_ms, 3), "counters": measured_issue_profile()},
        }
        if gathered is not None:
            line["gathered_ciphertexts"] = int(gathered[0] * gathered[1])
            line["collective_backend"] = "rccl" if backend == "nccl" else backend
            line["gather"] = "device (all_gather_into_tensor on pool rows exported on the GPU)" if on_device else "host tensors"
    ctx.close()
    if not args.no_aes:
        # every rank takes part: 8 CTR blocks per GPU, max time over ranks
        dt, n_blocks, ok = aes_ctr_sharded(local_rank, rank, world, on_device)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            line["aes"] = {"ctr_blocks": n_blocks, "ctr_seconds": round(dt, 3), "ctr_s_per_block": round(dt / n_blocks, 4),
                           "ctr_all_blocks_correct": ok, "sharding": f"block b on GPU b mod {world}, keys replicated, outputs gathered"}
    if rank == 0:
        if not args.no_aes:
            # one ECB block (configs[2]) on rank 0 at every N: the "seconds per ECB block" half of the metric; the sequential
            # chains of configs[3] do not shard and run at N = 1 only
            line["aes"].update(aes_single_gpu(local_rank, chains=(world == 1)))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p, lwe_sk, bsk, ksk)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
