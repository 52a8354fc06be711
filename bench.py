#!/usr/bin/env python3
"""AES-128-CTR blocks/s under FHE on N MI355X (one process per GPU) -- the metric of BASELINE.json.

A "step" is one pass of the hot path over one batch: Server::aes_encrypt (10 rounds = 160 byte-WoPBS
= 1,280 bit circuit-bootstraps per block, /root/reference/src/server/server.rs:39-64) over the
rank's BLOCKS counter blocks, already resident in HBM.  Workload = BASELINE.json configs[2]
(128 CTR blocks on one MI355X); with N ranks every rank owns its own 128 blocks (configs[3] at N=8:
1,024 blocks), keys are broadcast once over RCCL and there is no data-path collective: "weak" scaling.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (blind rotation).  It leads with the roof that BINDS at this batch size (f64 vector
                arithmetic: algorithmic flops per launch / measured launch time vs the f64 VALU peak); the HBM view
                (algorithmic bytes per launch / launch time vs 8 TB/s) is nested under "hbm".  "traffic" (measured
                HBM-side bytes per launch) is filled only from a rocprofv3 --pmc summary under profiles/ that was
                taken from the SAME engine sources (sha256 recorded in the summary), else null.
  cpu_baseline  the CPU oracle ("port" of the reference algorithm; the reference itself is Rust + the
                un-vendored tfhe 0.11.2 crate and cannot be built here) timed on this host's cores: one thread
                (S-Box and a whole block, extrapolated from 16 S-Boxes) and all cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

IV = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF      # SP 800-38A CTR initial counter block
KEY = 0x2B7E151628AED2A6ABF7158809CF4F3C
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6                   # = half of the 157.3 TF fp32 vector peak of that guide
I8_MFMA_PEAK_TOPS = 5000.0                    # that guide: I8 MFMA = 2x BF16 per clock, BF16 ~2.5 PF dense -> ~5 POP/s dense (2 ops per MAC)
I8_MFMA_POWER_CAPPED_TOPS = 3900.0            # a bare v_mfma_i32_16x16x64_i8 loop under this socket's 1,400 W cap: 1.95e15 MAC/s (profiles/r04_ubench_energy.txt)


def ext_product_flops(p) -> float:
    """f64 flops of one external product (k+1)*L forward + (k+1) inverse 256-point FFTs + MACs"""
    k1, L = p.k + 1, p.pbs_level
    fft = 5.0 * 256 * 8 + 256 * 6          # 5 n log2 n + twist
    return (k1 * L + k1) * fft + k1 * L * k1 * 256 * 8.0


def usable_cores() -> int:
    """CPU share of this process: affinity mask, capped by the cgroup quota when there is one"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(p, keys, client, seconds_hint: float = 20.0):
    """the oracle (a C port of the reference's algorithm) timed on this host, bounded to ~10-30 s:
      (i)  ONE thread: 16 many_sbox calls' worth (one AES round of one block = 16 bytes = 128 bit-CBS) -> ms per
           S-Box, and a whole block by extrapolation (aes_encrypt = 160 byte-WoPBS; the reference's C1 run,
           main.rs:48-64, adds key expansion = 200 and the counter add = 16 WoPBS, 15 of them 9 bits wide);
      (ii) all usable cores, blocks in parallel over bytes (the reference runs blocks in parallel with rayon,
           main.rs:55-64)."""
    from oracle import oracle as orc

    O = orc.Oracle(p, keys.ksk, keys.bsk, keys.pfpksk)
    cores = usable_cores()
    luts = orc.build_lutset(orc.LUTSET_ENC_ROUND)
    x = client.encrypt_bytes(list(range(16)))
    # (i) single thread
    orc.lib().orc_set_threads(1)
    n1 = 16 if p.name == "PARAM_OPT" else 64
    t0 = time.time()
    O.wopbs_batch(x[:1], luts)                                  # one S-Box alone
    one_sbox_s = time.time() - t0
    t0 = time.time()
    for i0 in range(0, n1, 16):
        O.wopbs_batch(x, luts)
    dt1 = time.time() - t0
    sbox_1t_s = dt1 / n1
    # (ii) all cores
    orc.lib().orc_set_threads(cores)
    n_done, t0 = 0, time.time()
    while True:
        O.wopbs_batch(x, luts)
        n_done += 16
        if time.time() - t0 > seconds_hint * 0.5 or n_done >= 64:
            break
    dt = time.time() - t0
    sbox_per_s = n_done / dt
    return {
        "value": sbox_per_s / 160.0, "unit": "blocks/s", "cores": cores, "kind": "port",
        "sample": "%d many_sbox (byte-WoPBS, 3 LUTs) of the C oracle on %d cores in %.1f s; blocks/s = S-Box/s / 160" % (n_done, cores, dt),
        "ms_per_sbox_per_core": 1000.0 * dt * cores / n_done,
        "single_thread": {
            "cores": 1, "ms_one_sbox_alone": 1000.0 * one_sbox_s, "ms_per_sbox": 1000.0 * sbox_1t_s,
            "s_per_block_aes_encrypt": 160.0 * sbox_1t_s,
            "s_per_block_reference_c1": (160.0 + 200.0 + 16.0 * 9.0 / 8.0) * sbox_1t_s,
            "blocks_per_s": 1.0 / (160.0 * sbox_1t_s),
            "sample": "%d many_sbox on one thread in %.1f s; a block is extrapolated: aes_encrypt = 160 byte-WoPBS, the reference's whole "
                      "1-block run (main.rs:48-64: key expansion 200 + counter add 16 nine-bit + encrypt 160) = 378 S-Box equivalents" % (n1, dt1),
        },
        "reference_published": "84 s per block single core (README.md:186), unknown hardware",
    }


def u128_bytes(x: int):
    return [(x >> (8 * (15 - b))) & 0xFF for b in range(16)]


def wrong_blocks(client, ct, want) -> list:
    """client.rs:147-175 decrypts and asserts EVERY block (`assert_eq` at :171): indices of the blocks of `ct`
    ([n][16][8][kN+1] device tensor) whose decryption differs from want[i] (128-bit integers)"""
    got = client.decrypt_bytes(ct.cpu().numpy().view(np.uint64))            # [n][16]
    return [i for i, w in enumerate(want) if [int(v) for v in got[i]] != u128_bytes(w)]


def words_sha(t) -> str:
    import hashlib

    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=128, help="CTR blocks per GPU")
    ap.add_argument("--params", default="opt", choices=["opt", "toy"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo lets several ranks share one GPU to rehearse the N>1 path on a 1-GPU box")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="at N=1 only: initialise the nccl (= RCCL) process group with ONE rank anyway and push the key and round-key "
                         "broadcasts and the elapsed-time all-reduce through it (reported as rccl_one_rank); a builder's box has one GPU, "
                         "so this is the only RCCL code a builder can run on hardware")
    ap.add_argument("--decrypt", action="store_true", help="time Server::aes_decrypt (BASELINE configs[4] path) instead of aes_encrypt")
    ap.add_argument("--no-ctr-iteration", action="store_true",
                    help="skip the extra (untimed-step) measurement of the reference's whole CTR iteration that the default line reports at N=1")
    ap.add_argument("--ctr-add", action="store_true",
                    help="time the reference's whole CTR iteration (main.rs:59-61): Server::add_scalar(iv, i) on the GPU, then aes_encrypt")
    args = ap.parse_args()

    # host-side Client work (key generation, encryption of the synthetic inputs) is OpenMP code: give every rank
    # its share of the cores instead of letting N ranks oversubscribe the node
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(1, local_world))))

    import torch
    import torch.distributed as dist

    from tfhe_aes_amd import PARAM_OPT, PARAM_TOY, _native
    from tfhe_aes_amd.aes_clear import aes128_decrypt_block, aes128_encrypt_block
    from tfhe_aes_amd.client import Client
    from tfhe_aes_amd.dist import broadcast_keys_seeded, broadcast_tensor, shard_blocks

    p = PARAM_OPT if args.params == "opt" else PARAM_TOY
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch as `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 bench.py --gpus %d ...` (one process per GPU)" % (world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != world:           # what the communicator says, not what the environment promised
            raise SystemExit("process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), world))

    rccl1 = None
    if world == 1 and args.rccl_one_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        t0 = time.time()
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                      # the first collective creates the communicator
        torch.cuda.synchronize()
        rccl1 = {"init_and_first_all_reduce_s": round(time.time() - t0, 3), "ranks": 1}
        # marker for tests/test_gpu_bench_dist.py: past this line a communicator EXISTS, so any later RCCL error is a
        # failure of this code, not of the box
        print("RCCL_COMMUNICATOR_READY", file=sys.stderr, flush=True)

    # ---- keys: generated on rank 0, broadcast once over RCCL/xGMI --------------------------------
    client = Client(args.blocks * world, IV, KEY, params=p, seed=0xAE50001)     # secret keys: same seed on every rank
    t0 = time.time()
    keys = client.server_keys() if rank == 0 else None
    keygen_s = time.time() - t0
    # keys travel in their compressed form: (public mask seed, bodies) = 0.19 GB instead of 1.04 GB; every rank regenerates
    # the masks on its own GPU (fheaes_upload_keys_seeded)
    seeded = keys.compress() if rank == 0 else None
    t0 = time.time()
    mask_seed, dbodies = broadcast_keys_seeded(p, seeded, dev, src=0)
    torch.cuda.synchronize()
    bcast_s = time.time() - t0
    key_bytes_moved = sum(int(t.numel()) * 8 for t in dbodies) + 32
    if rccl1 is not None:                           # the same tensors through a one-rank RCCL broadcast (dist.py skips it at world 1)
        t0 = time.time()
        sums = [int(t.sum().item()) for t in dbodies]
        for t in dbodies:
            dist.broadcast(t, src=0)
        torch.cuda.synchronize()
        rccl1["key_broadcast_s"] = round(time.time() - t0, 3)
        rccl1["key_broadcast_bytes"] = key_bytes_moved - 32
        rccl1["key_broadcast_intact"] = sums == [int(t.sum().item()) for t in dbodies]
    eng = _native.Engine(p, device=dev_index)
    t0 = time.time()
    eng.upload_keys_seeded(mask_seed, *dbodies)
    expand_s = time.time() - t0
    del dbodies, seeded
    torch.cuda.empty_cache()

    # ---- round keys: expanded once on rank 0 (timed separately, as main.rs:48-51), broadcast -------
    rk = torch.empty((11, 16, 8, p.big1), dtype=torch.int64, device=dev)
    keyexp_s = None
    if rank == 0:
        ek = torch.from_numpy(client.encrypt_u128(KEY).view(np.int64)).to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        eng.aes_key_expansion(ek, rk)
        eng.synchronize()
        keyexp_s = time.time() - t0
    broadcast_tensor(rk, src=0)
    torch.cuda.synchronize()

    # ---- this rank's counter blocks (pre-incremented client side; Server::add_scalar is timed apart) --
    lo, hi = shard_blocks(args.blocks * world, world, rank)
    counters = [(IV + i) & ((1 << 128) - 1) for i in range(lo, hi)]
    n_blocks = hi - lo
    if args.ctr_add:
        iv_ct = torch.from_numpy(client.encrypt_u128(IV).view(np.int64)).to(dev)
        state = iv_ct.unsqueeze(0).repeat(n_blocks, 1, 1, 1).contiguous()        # encrypted_iv.clone() per block (main.rs:59)
        iv_states = state.clone()
    else:
        host_state = np.stack([client.encrypt_u128(c) for c in counters])
        state = torch.from_numpy(host_state.view(np.int64)).to(dev)
        del host_state
    eng.reserve(n_blocks * 128)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    crypt = eng.aes_decrypt if args.decrypt else eng.aes_encrypt
    if args.ctr_add:
        def step(rk_, state_, n_):
            state_.copy_(iv_states)
            torch.cuda.synchronize()
            eng.add_scalar(state_, n_, list(range(lo, hi)))
            crypt(rk_, state_, n_)
    else:
        step = crypt
    for _ in range(args.warmup):
        step(rk, state, n_blocks)
    eng.synchronize()
    eng.profile_enable(True)
    eng.profile_reset()
    # socket power / shader clock of this rank's GPU while the timed steps run (librocm_smi64 in-process; None where it refuses)
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        from gpu_power import Sampler
        sampler = Sampler(device=dev_index, period=0.05)
    except Exception:
        sampler = None
    barrier()
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(rk, state, n_blocks)
    eng.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    power = None
    if sampler is not None:
        sampler.__exit__()
        ps = sampler.summary()
        if ps["samples"]:
            power = {"avg_w": ps["power_w"], "max_w": ps["power_max_w"], "cap_w": ps["cap_w"], "sclk_mhz": ps["sclk_mhz"],
                     "energy_j_per_step": None if ps["energy_j"] is None else ps["energy_j"] / args.steps,
                     "note": "rank 0's GPU over the timed steps (rocm_smi socket power, shader clock, energy counter): the blind rotation runs at "
                             "the socket's power cap, so a step takes (joules per step) / (watts)"}
    prof = eng.profile_read()
    eng.profile_enable(False)
    own_elapsed = elapsed
    rank_elapsed = None
    if world > 1 or rccl1 is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rccl1 is not None:
            rccl1["elapsed_all_reduce_max_ok"] = float(tt.item()) == elapsed
        elapsed = float(tt.item())
    if world > 1:
        # every rank's own time, gathered to all (rank 0 prints them): a straggler is visible in the line itself
        mine = torch.tensor([own_elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rank_elapsed = [float(g.item()) for g in gathered]

    # ---- CPU baseline (rank 0, outside the timed region; the other ranks wait at the final barrier) -----------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(p, keys, client)

    # ---- verify EVERY block (the reference asserts every block, client.rs:171): decrypt == AES applied (warmup+steps) times ----
    verified, wrong = None, {}
    if not args.no_verify:
        want = list(counters)
        for _ in range(1 if args.ctr_add else args.warmup + args.steps):
            want = [aes128_decrypt_block(KEY, w) if args.decrypt else aes128_encrypt_block(KEY, w) for w in want]
        if os.environ.get("FHEAES_BENCH_SABOTAGE_VERIFY") == "1":       # tests/test_gpu_bench_dist.py: the failure path must be reachable
            want[0] ^= 1
        bad = wrong_blocks(client, state, want)
        if bad:
            wrong["headline_rank%d" % rank] = bad
        verified = not bad
        if world > 1:
            v = torch.tensor([1 if verified else 0], device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            verified = bool(v.item())

    # ---- the reference's whole CTR iteration (main.rs:59-61: Server::add_scalar(iv, i), then aes_encrypt), one extra step ----
    ctr_iter = None
    if rank == 0 and world == 1 and not args.ctr_add and not args.decrypt and not args.no_ctr_iteration:
        iv_ct = torch.from_numpy(client.encrypt_u128(IV).view(np.int64)).to(dev)
        st2 = iv_ct.unsqueeze(0).repeat(n_blocks, 1, 1, 1).contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.add_scalar(st2, n_blocks, list(range(lo, hi)))
        eng.aes_encrypt(rk, st2, n_blocks)
        eng.synchronize()
        dt = time.perf_counter() - t0
        bad = wrong_blocks(client, st2, [aes128_encrypt_block(KEY, (IV + lo + i) & ((1 << 128) - 1)) for i in range(n_blocks)])
        if bad:
            wrong["ctr_iteration"] = bad
        ctr_iter = {"blocks_per_s": n_blocks / dt, "ms": 1000.0 * dt, "verified_vs_aes": not bad, "blocks_checked": n_blocks,
                    "note": "one step of %d blocks: Server::add_scalar(encrypted_iv, i) on the GPU (143 bit-CBS per block, 16-step carry chain), "
                            "then Server::aes_encrypt; the timed `value` uses client-side pre-incremented counters" % n_blocks}
        del st2

    # ---- BASELINE configs[4] shard: Server::aes_decrypt on 32 blocks (256 blocks / 8 GPUs), one extra verified step ------------
    dec32 = None
    if rank == 0 and world == 1 and not args.ctr_add and not args.decrypt and not args.no_ctr_iteration and n_blocks >= 32 and not args.no_verify:
        st4 = state[:32].clone()
        eng.aes_decrypt(rk, st4, 32)                      # warm-up (workspace growth for the 4-LUT packing)
        eng.synchronize()                                 # the engine has its own stream: finish before torch overwrites st4
        sha_warm = words_sha(st4)
        st4.copy_(state[:32])
        torch.cuda.synchronize()
        eng.profile_enable(True)
        eng.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.aes_decrypt(rk, st4, 32)
        eng.synchronize()
        dt = time.perf_counter() - t0
        pr4 = eng.profile_read()
        eng.profile_enable(False)
        want4 = list(counters[:32])
        for _ in range(args.warmup + args.steps - 1):
            want4 = [aes128_encrypt_block(KEY, w) for w in want4]
        bad = wrong_blocks(client, st4, want4)
        if bad:
            wrong["configs4_decrypt"] = bad
        sha_timed = words_sha(st4)
        if sha_timed != sha_warm:                         # the same input twice: the engine is deterministic, so other words = a race
            wrong["configs4_decrypt_not_deterministic"] = [sha_warm[:16], sha_timed[:16]]
        b4 = pr4["blind_rotate"]
        l4 = max(1, b4["launches"])
        k2_ms = b4["ms"] / l4
        f4 = (b4["units"] / l4) * p.n * ext_product_flops(p) / (k2_ms * 1e-3) / 1e12 if k2_ms > 0 else 0.0
        kp4 = eng.k2_plan(int(b4["units"] // l4))
        dec32 = {"blocks_per_s": 32 / dt, "ms": 1000.0 * dt, "verified_vs_aes": not bad, "blocks_checked": 32,
                 "same_words_as_warmup_run": sha_timed == sha_warm, "words_sha256": sha_timed[:16], "k2_launches": l4, "k2_bits_per_launch": b4["units"] / l4,
                 "k2_ms_per_launch": k2_ms, "k2_form": kp4["form"], "k2_kernel": kp4["kernel"], "k2_frac_of_f64_valu_peak": f4 / F64_VALU_PEAK_TFLOPS,
                 "note": "BASELINE configs[4] per-GPU shard (256 blocks / 8 GPUs): Server::aes_decrypt on 32 resident blocks, 2,432 bit-CBS per block "
                         "(inverse S-Box + 4-LUT inverse MixColumns packing, server.rs:67-105); one step, every launch is 4,096 bits"}
        del st4

    # ---- the in-process split on ONE GPU: two contexts (keys uploaded once, cloned device to device), 64 blocks each from two host
    #      threads -- any hidden serialisation or workspace contention between contexts shows against the one-context headline ----
    two_ctx = None
    if rank == 0 and world == 1 and not args.ctr_add and not args.decrypt and not args.no_ctr_iteration and n_blocks >= 2 and not args.no_verify:
        import threading

        half = n_blocks // 2
        eng2 = _native.Engine(p, device=dev_index)
        eng2.clone_keys_from(eng)
        ci = eng2.clone_info()
        eng2.reserve(half * 128)
        halves = [state[:half].clone(), state[half:2 * half].clone()]
        engs = [eng, eng2]
        for e_, h_ in zip(engs, halves):                 # warm-up of the second context's workspace
            e_.aes_encrypt(rk, h_, half)
            e_.synchronize()
        halves = [state[:half].clone(), state[half:2 * half].clone()]
        torch.cuda.synchronize()

        def run_half(i):
            engs[i].aes_encrypt(rk, halves[i], half)
            engs[i].synchronize()

        t0 = time.perf_counter()
        ths = [threading.Thread(target=run_half, args=(i,)) for i in range(2)]
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join()
        dt = time.perf_counter() - t0
        want2 = list(counters[:2 * half])
        for _ in range(args.warmup + args.steps + 1):
            want2 = [aes128_encrypt_block(KEY, w) for w in want2]
        bad = [hidx * half + i for hidx in range(2) for i in wrong_blocks(client, halves[hidx], want2[hidx * half:(hidx + 1) * half])]
        if bad:
            wrong["two_contexts"] = bad
        two_ctx = {"blocks_per_s": 2 * half / dt, "ms": 1000.0 * dt, "verified_vs_aes": not bad, "blocks_checked": 2 * half, "blocks_per_context": half,
                   "clone": {"path": ci["path"], "bytes": ci["bytes"], "seconds": round(ci["seconds"], 4)},
                   "note": "two fheaes contexts on this one GPU (fheaes_clone_keys: one upload, device-to-device copy of the converted key images), "
                           "%d blocks of Server::aes_encrypt each from two host threads, concurrently: the in-process shape of the reference's rayon "
                           "loop (main.rs:55-64); compare with `value` (one context, %d blocks)" % (half, n_blocks)}
        eng2.close()
        del halves

    # ---- BASELINE configs[1]: one AES block = 16 S-Box WoPBS in one call (latency, not throughput) ------------
    one_block_ms = None
    if rank == 0:
        xb = state[0].clone()
        ob = torch.empty((16, 3, 8, p.big1), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.many_sbox(xb, 16, False, ob)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.many_sbox(xb, 16, False, ob)
        eng.synchronize()
        one_block_ms = 1000.0 * (time.perf_counter() - t0) / 3
        if not args.no_verify:
            from tfhe_aes_amd.aes_clear import SBOX, mul2, mul3
            xin = [int(v) for v in client.decrypt_bytes(xb.cpu().numpy().view(np.uint64))]                 # the 16 input bytes
            got = client.decrypt_bytes(ob.cpu().numpy().view(np.uint64))                                   # [16][3]
            bad = [i for i in range(16) if [int(v) for v in got[i]] != [SBOX[xin[i]], mul2(SBOX[xin[i]]), mul3(SBOX[xin[i]])]]
            if bad:
                wrong["config1_many_sbox_bytes"] = bad

    if rank == 0:
        total_blocks = args.blocks * world
        value = total_blocks * args.steps / elapsed
        ms_per_step = 1000.0 * elapsed / args.steps
        # dominant kernel = blind rotation (K2)
        br = prof["blind_rotate"]
        launches = max(1, br["launches"])
        bits_per_launch = br["units"] / launches
        avg_ms = br["ms"] / launches
        bsk_bytes = 8.0 * p.bsk_words
        io_bytes = 8.0 * ((p.n + 1) + p.big1)
        algo_bytes = bsk_bytes + bits_per_launch * io_bytes        # every key byte once per launch + per-bit I/O
        achieved_gbs = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        flops = bits_per_launch * p.n * ext_product_flops(p)
        tflops = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # measured HBM-side bytes per launch: only from a PMC summary taken from these very engine sources
        from tfhe_aes_amd import _build
        traffic, traffic_src, pm = None, None, {}
        for pmc in sorted((ROOT / "profiles").glob("*pmc_blind_rotate*.json")):
            try:
                d = json.loads(pmc.read_text())
                if (d.get("engine_src_sha256") == _build.engine_source_hash() and d.get("bits_per_launch") == bits_per_launch
                        and d.get("params") == p.name):
                    traffic, traffic_src, pm = d.get("hbm_bytes_per_launch"), pmc.name, d
            except Exception:
                pass
        stage_ms = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()}

        # ---- the second kernel of a step: K3, private functional packing key switch on the int8 matrix cores (7 % of a step) ----
        # out[m][z][o] -= sum_{i,l} digit_l(in[m][i]) KEY[z][i][l][o]: an exact integer matrix product mod 2^64 of M x Q by Q x (k+1)^2 N
        # (Q = (kN+1) x pfks_level) sliced into 15 int8 products per u64 product (kern_keyswitch.h); its HBM view: the key fragments
        # once per launch + digit planes + the output
        pf = prof["pfpks"]
        pf_l = max(1, pf["launches"])
        pf_ms = pf["ms"] / pf_l
        pf_bits = pf["units"] / pf_l
        k1_ = p.k + 1
        pf_q = p.big1 * p.pfks_level
        pf_cols = k1_ * k1_ * p.polynomial_size
        pf_mac = pf_bits * pf_q * pf_cols * 15.0
        pf_tops = 2.0 * pf_mac / (pf_ms * 1e-3) / 1e12 if pf_ms > 0 else 0.0
        pf_ksteps = -(-pf_q // 64)
        pf_bytes = (pf_ksteps * 64 * (-(-pf_cols // 16) * 16) * 8                      # balanced key bytes, fragment order (635.7 MB at PARAM_OPT)
                    + pf_bits * pf_ksteps * 64 * 2                                     # the two digit planes
                    + pf_bits * p.big1 * 8 + pf_bits * pf_cols * 8)                    # LWE words in, GGSW rows out
        roofline_k3 = {
            "kernel": "digits_kernel<12,3,2> + keyswitch_mfma_lds_kernel<2> (private functional packing key switch, K3)",
            "bound": "mfma", "achieved": pf_tops, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": pf_tops / I8_MFMA_PEAK_TOPS,
            "frac_of_power_capped_mfma_loop": pf_tops / I8_MFMA_POWER_CAPPED_TOPS,
            "avg_launch_ms": pf_ms, "bits_per_launch": pf_bits, "int8_mac_per_launch": pf_mac, "int8_products_per_u64_product": 15,
            "traffic": None,
            "hbm": {"bound": "hbm", "achieved": pf_bytes / (pf_ms * 1e-3) / 1e9 if pf_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (pf_bytes / (pf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pf_ms > 0 else 0.0, "algorithmic_bytes_per_launch": pf_bytes},
            "note": "share of a step: %.1f %%; dense int8 peak = 2 x the BF16 MFMA rate (MI355X_MICROARCH.md); the chip holds 0.78 of it in a bare "
                    "MFMA loop under the 1,400 W cap" % (100.0 * pf["ms"] / max(1e-9, sum(v["ms"] for v in prof.values()))),
        }
        for pmc in sorted((ROOT / "profiles").glob("*pmc_pfpks*.json")):
            try:
                d = json.loads(pmc.read_text())
                if d.get("engine_src_sha256") == _build.engine_source_hash() and d.get("bits_per_launch") == pf_bits and d.get("params") == p.name:
                    roofline_k3["traffic"] = d.get("hbm_bytes_per_launch")
                    roofline_k3["traffic_source"] = pmc.name
                    for k_ in ("mfma_busy_frac", "lds_bank_conflict_frac", "effective_clock_ghz", "l2_hit_rate"):
                        if k_ in d:
                            roofline_k3[k_] = d[k_]
            except Exception:
                pass
        # the kernel THIS context launches for such a batch (fheaes_k2_context_plan: after the occupancy fallbacks), not an assumption
        k2p = eng.k2_plan(int(bits_per_launch))
        line = {
            "metric": "AES-128 CTR blocks/sec (FHE)", "value": value, "unit": "blocks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64+f64", "data": "synthetic",
            "config": {
                "workload": ("reference CTR iteration (main.rs:59-61): %d blocks per GPU, Server::add_scalar (143 bit-CBS/block, 16-step carry chain) + "
                             "Server::aes_encrypt (1280 bit-CBS/block)" % args.blocks) if args.ctr_add else
                            ("configs[4] path: %d blocks per GPU, Server::aes_decrypt (2432 bit-CBS/block)" % args.blocks) if args.decrypt else
                            ("configs[2]: %d CTR blocks per GPU, Server::aes_encrypt 10 rounds (1280 bit-CBS/block), "
                             "counters pre-incremented client-side" % args.blocks),
                "params": p.name, "blocks_per_gpu": args.blocks, "total_blocks": total_blocks,
                "bit_cbs_per_step_per_gpu": args.blocks * (2432 if args.decrypt else 1280),
            },
            "ms_per_sbox": ms_per_step / (args.blocks * (304.0 if args.decrypt else 160.0)),
            "verified_vs_aes": verified,
            # every block of the headline and of every extra step is decrypted and compared (the reference asserts every block,
            # client.rs:171); anything wrong is listed by step -> block indices and the process exits 1 after printing this line
            "all_verified": None if args.no_verify else bool(verified) and not wrong, "wrong_blocks": wrong,
            "blocks_checked": None if args.no_verify else n_blocks,
            "config1_one_block_round": {"many_sbox_16_bytes_ms": one_block_ms, "verified_vs_aes": None if args.no_verify else "config1_many_sbox_bytes" not in wrong, "ms_per_sbox": None if one_block_ms is None else one_block_ms / 16.0,
                                        "note": "BASELINE configs[1]: 16 S-Box WoPBS (128 bit-CBS) in one call: latency of the 669-step rotation chain"},
            "ctr_iteration_with_add_scalar": ctr_iter,
            "configs4_decrypt_32_blocks": dec32,
            "two_contexts_64_blocks_each": two_ctx,
            "stage_ms_per_step": stage_ms,
            "roofline": {
                "kernel": "%s (blind rotation, K2)" % k2p["kernel"], "k2_form": k2p["form"],
                "k2_plan": {"units_main": k2p["units_main"], "r_main": k2p["r_main"], "units_tail": k2p["units_tail"], "r_tail": k2p["r_tail"]},
                "bound": "valu_f64",
                "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_VALU_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": avg_ms, "bits_per_launch": bits_per_launch,
                "traffic_over_algorithmic": None if traffic is None else traffic / algo_bytes,
                # from the same source-stamped counter file (null without one; tools/summarize_pmc.py says how each is derived):
                # ceiling_frac = the fraction of the roof this instruction stream reaches with a vector instruction issuing on every SIMD
                # cycle at 2.4 GHz (ISA count x per-opcode issue cost, tools/k2_dyncount.py); valu_issue_occupancy_model = SQ_INSTS_VALU
                # (measured; kept raw as sq_insts_valu) x the average issue cost per vector instruction / SIMD cycles -- a MODEL of the
                # busy share, gfx950 exposes no busy-cycle counter for the vector ALU; clock_ghz = GRBM_GUI_ACTIVE / time;
                # model_frac = ceiling x occupancy x clock / 2.4 is what those three say `frac` should be (the profiled launch's own
                # fraction is frac_of_profiled_launch: profiling boxes and bench boxes differ by a few per cent)
                "ceiling_frac": pm.get("ceiling_frac"), "valu_issue_occupancy_model": pm.get("valu_issue_occupancy_model"),
                "sq_insts_valu": pm.get("sq_insts_valu"), "clock_ghz": pm.get("effective_clock_ghz"), "model_frac": pm.get("model_frac"),
                "frac_of_profiled_launch": None if not pm.get("avg_launch_ms_profiled") else flops / (pm["avg_launch_ms_profiled"] * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS,
                "l1_fill_bytes_per_clk_per_cu": pm.get("l1_fill_bytes_per_clk_per_cu"), "l2_hit_rate": pm.get("l2_hit_rate"),
                "power": power,
                "algorithmic_flops_per_launch": flops, "flops_per_external_product": ext_product_flops(p),
                "note": "at 16,384 bits per launch the kernel is bound by f64 vector arithmetic under the socket's power cap (see power), "
                        "not by HBM: the BSK is read once per launch",
                "hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": algo_bytes,
                        "note": "one pass over the 342.5 MB Fourier BSK per launch + per-bit I/O (5,360 B in, 16,392 B out)"},
            },
            "roofline_k3": roofline_k3,
            "setup_s": {"keygen": round(keygen_s, 2), ("key_upload_h2d" if world == 1 else "key_broadcast_" + ("rccl" if args.backend == "nccl" else args.backend)): round(bcast_s, 3),
                        "key_bytes_moved": key_bytes_moved, "key_expand_and_convert_on_gpu": round(expand_s, 3),
                        "aes_key_expansion": None if keyexp_s is None else round(keyexp_s, 3)},
        }
        if world > 1:
            line["collective"] = {
                "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""), "world_size_seen": dist.get_world_size(),
                "key_broadcast_s": round(bcast_s, 3), "key_broadcast_bytes": key_bytes_moved,
                "key_broadcast_GBps": round(key_bytes_moved / max(bcast_s, 1e-9) / 1e9, 3),
                "rank_elapsed_s": {"min": min(rank_elapsed), "max": max(rank_elapsed), "per_rank": [round(x, 4) for x in rank_elapsed]},
                "data_path_collectives": 0,
                "note": "keys (seeded form) and round keys are broadcast once at start-up; the timed steps exchange nothing (CTR blocks are independent)",
            }
        if rccl1 is not None:
            rccl1["note"] = ("torch.distributed backend nccl (= RCCL) with world_size 1 on this GPU: communicator creation, the key broadcasts "
                             "and the MAX all-reduce ran through RCCL; no second GPU, so no xGMI transfer took place")
            line["rccl_one_rank"] = rccl1
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(line), flush=True)
    if world > 1 or rccl1 is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not args.no_verify and (wrong or not verified):
        # a wrong plaintext is a failed run, whatever the throughput: the line above says where, the exit code says so to a driver
        print("bench.py: verification FAILED: %s" % (json.dumps(wrong) if wrong else "another rank's blocks"), file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
