#!/usr/bin/env python3
"""AES-128-CTR blocks/s under FHE on N MI355X (one process per GPU) -- the metric of BASELINE.json.

A "step" is one pass of the hot path over one batch: Server::aes_encrypt (10 rounds = 160 byte-WoPBS
= 1,280 bit circuit-bootstraps per block, /root/reference/src/server/server.rs:39-64) over the
rank's BLOCKS counter blocks, already resident in HBM.  Workload = BASELINE.json configs[2]
(128 CTR blocks on one MI355X); with N ranks every rank owns its own 128 blocks (configs[3] at N=8:
1,024 blocks), keys are broadcast once over RCCL and there is no data-path collective: "weak" scaling.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (blind rotation): algorithmic HBM bytes per launch / measured launch time
  cpu_baseline  the CPU oracle ("port" of the reference algorithm; the reference itself is Rust + the
                un-vendored tfhe 0.11.2 crate and cannot be built here) timed on this host's cores
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
    """time the oracle's many_sbox (one AES round of one block = 16 bytes = 128 bit-CBS) on all host cores"""
    from oracle import oracle as orc

    O = orc.Oracle(p, keys.ksk, keys.bsk, keys.pfpksk)
    cores = usable_cores()
    orc.lib().orc_set_threads(cores)
    luts = orc.build_lutset(orc.LUTSET_ENC_ROUND)
    x = client.encrypt_bytes(list(range(16)))
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
        "sample": "%d many_sbox (byte-WoPBS, 3 LUTs) of the C oracle in %.1f s; blocks/s = S-Box/s / 160" % (n_done, dt),
        "ms_per_sbox_per_core": 1000.0 * dt * cores / n_done,
        "reference_published": "84 s per block single core (README.md:186), unknown hardware",
    }


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
    ap.add_argument("--decrypt", action="store_true", help="time Server::aes_decrypt (BASELINE configs[4] path) instead of aes_encrypt")
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
    from tfhe_aes_amd.dist import broadcast_keys, broadcast_tensor, shard_blocks

    p = PARAM_OPT if args.params == "opt" else PARAM_TOY
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
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

    # ---- keys: generated on rank 0, broadcast once over RCCL/xGMI --------------------------------
    client = Client(args.blocks * world, IV, KEY, params=p, seed=0xAE50001)     # secret keys: same seed on every rank
    t0 = time.time()
    keys = client.server_keys() if rank == 0 else None
    keygen_s = time.time() - t0
    t0 = time.time()
    dkeys = broadcast_keys(p, keys, dev, src=0)
    torch.cuda.synchronize()
    bcast_s = time.time() - t0
    eng = _native.Engine(p, device=dev_index)
    eng.upload_keys(*dkeys)
    del dkeys
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
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(rk, state, n_blocks)
    eng.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- verify a sample: decrypt == AES applied (warmup+steps) times to the counter -----------------
    verified = None
    if not args.no_verify:
        verified = True
        for idx in sorted({0, n_blocks // 2, n_blocks - 1}):
            got = client.decrypt_u128(state[idx].cpu().numpy().view(np.uint64))
            want = counters[idx]
            for _ in range(1 if args.ctr_add else args.warmup + args.steps):
                want = aes128_decrypt_block(KEY, want) if args.decrypt else aes128_encrypt_block(KEY, want)
            if got != want:
                verified = False
        if world > 1:
            v = torch.tensor([1 if verified else 0], device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            verified = bool(v.item())

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
        traffic = None
        pmc = ROOT / "profiles" / "pmc_blind_rotate.json"
        if pmc.exists():
            try:
                d = json.loads(pmc.read_text())
                if d.get("bits_per_launch") == bits_per_launch and d.get("params") == p.name:
                    traffic = d.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        stage_ms = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()}
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
            "config1_one_block_round": {"many_sbox_16_bytes_ms": one_block_ms, "ms_per_sbox": None if one_block_ms is None else one_block_ms / 16.0,
                                        "note": "BASELINE configs[1]: 16 S-Box WoPBS (128 bit-CBS) in one call: latency of the 669-step rotation chain"},
            "stage_ms_per_step": stage_ms,
            "roofline": {
                "kernel": "extprod_rotate_kernel (blind rotation, K2)", "bound": "hbm",
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic, "avg_launch_ms": avg_ms, "bits_per_launch": bits_per_launch,
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "one pass over the 342.5 MB BSK per launch + per-bit I/O; at this batch the kernel is f64-VALU/LDS bound, see `compute`",
                "compute": {"bound": "valu_f64", "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tflops / F64_VALU_PEAK_TFLOPS, "flops_per_external_product": ext_product_flops(p)},
            },
            "setup_s": {"keygen": round(keygen_s, 2), "key_broadcast": round(bcast_s, 3),
                        "aes_key_expansion": None if keyexp_s is None else round(keyexp_s, 3)},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(p, keys, client)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
