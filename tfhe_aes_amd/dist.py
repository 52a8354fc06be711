"""Multi-GPU plumbing: one process per GPU, keys broadcast once, CTR blocks sharded, no data-path collective.

The reference is single-process; its only parallelism is rayon over independent CTR blocks
(/root/reference/src/main.rs:55-64).  The MI355X counterpart replicates the ~1.04 GB of evaluation
keys on every GPU with one RCCL broadcast per key over xGMI at start-up and gives every rank a
contiguous range of blocks.  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo"
in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_blocks(total_blocks: int, world: int, rank: int) -> tuple[int, int]:
    """[start, end) of the blocks owned by `rank`; ranges are contiguous, disjoint and cover everything."""
    if world < 1 or not 0 <= rank < world or total_blocks < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total_blocks, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_keys(params, keys, device, src: int = 0):
    """Rank `src` passes ServerKeys, the others None.  Returns three int64 torch tensors on `device`
    holding KSK, BSK (standard domain) and PFPKSK words, identical on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    words = (params.ksk_words, params.bsk_words, params.pfpksk_words)
    if rank == src:
        if keys is None:
            raise ValueError("the source rank must hold the keys")
        tensors = [torch.from_numpy(np.ascontiguousarray(h).view(np.int64)).to(device) for h in (keys.ksk, keys.bsk, keys.pfpksk)]
        for t, w in zip(tensors, words):
            if t.numel() != w:
                raise ValueError("key size does not match the parameter set")
    else:
        tensors = [torch.empty(w, dtype=torch.int64, device=device) for w in words]
    if world > 1:
        for t in tensors:          # three large broadcasts: per-link bound on xGMI, one-off
            dist.broadcast(t, src=src)
    return tensors


def broadcast_keys_seeded(params, seeded, device, src: int = 0):
    """The same, for keys in their compressed form (client.SeededServerKeys on rank `src`, None elsewhere): the public 256-bit
    mask key and the three body arrays travel (0.19 GB instead of 1.04 GB at PARAM_OPT: one 32-byte and three tensor
    broadcasts); every rank regenerates the masks on its own GPU (Engine.upload_keys_seeded).
    Returns (mask_key as uint32[8], [ksk_body, bsk_body, pfpksk_body]) with the bodies as int64 tensors on `device`."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    k, N = params.k, params.N
    words = (params.big * params.ks_level, params.n * params.pbs_level * (k + 1) * N, (k + 1) * params.big1 * params.pfks_level * N)
    seed_t = torch.zeros(8, dtype=torch.int32, device=device)
    if rank == src:
        if seeded is None:
            raise ValueError("the source rank must hold the keys")
        seed_t.copy_(torch.from_numpy(np.asarray(seeded.mask_seed, dtype=np.uint32).reshape(8).view(np.int32).copy()))
        tensors = [torch.from_numpy(np.ascontiguousarray(h).reshape(-1).view(np.int64)).to(device)
                   for h in (seeded.ksk_body, seeded.bsk_body, seeded.pfpksk_body)]
        for t, w in zip(tensors, words):
            if t.numel() != w:
                raise ValueError("key body size does not match the parameter set")
    else:
        tensors = [torch.empty(w, dtype=torch.int64, device=device) for w in words]
    if world > 1:
        dist.broadcast(seed_t, src=src)
        for t in tensors:
            dist.broadcast(t, src=src)
    mask_key = seed_t.cpu().numpy().view(np.uint32).copy()
    return mask_key, tensors


def broadcast_tensor(t, src: int = 0):
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t
