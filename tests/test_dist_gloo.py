"""The N>1 path on CPU: world_size 2 over gloo.  Keys are broadcast once from rank 0, CTR blocks are sharded
contiguously, there is no data-path collective; each rank then evaluates its own blocks (here with the CPU
oracle standing in for the GPU) and the union of the shards is the whole CTR stream."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

from tfhe_aes_amd.dist import shard_blocks  # noqa: E402


def test_shard_blocks_cover_and_are_disjoint():
    for total in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            ranges = [shard_blocks(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_blocks(1024, 8, 3) == (384, 512)             # BASELINE configs[3]: 128 blocks per GPU
    with pytest.raises(ValueError):
        shard_blocks(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc
    from tfhe_aes_amd import PARAM_TOY, aes_clear
    from tfhe_aes_amd.client import Client, SeededServerKeys, ServerKeys
    from tfhe_aes_amd.dist import broadcast_keys, broadcast_keys_seeded, broadcast_tensor

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = PARAM_TOY
        iv, key = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF, 0x2B7E151628AED2A6ABF7158809CF4F3C
        client = Client(2, iv, key, params=p, seed=0x70F)          # same secret keys on every rank (same seed)
        keys = client.server_keys() if rank == 0 else None
        dev = torch.device("cpu")
        tk = broadcast_keys(p, keys, dev, src=0)
        got = ServerKeys(p, *[t.numpy().view(np.uint64) for t in tk])
        # the compressed form bench.py ships: (mask seed, bodies); expanding it gives the very same keys on every rank
        mseed, tb = broadcast_keys_seeded(p, keys.compress() if rank == 0 else None, dev, src=0)
        exp = SeededServerKeys(p, mseed, *[t.numpy().view(np.uint64) for t in tb]).expand()
        assert np.array_equal(exp.ksk, got.ksk) and np.array_equal(exp.bsk, got.bsk) and np.array_equal(exp.pfpksk, got.pfpksk)
        O = orc.Oracle(p, got.ksk, got.bsk, got.pfpksk)
        rk = torch.empty((11, 16, 8, p.big1), dtype=torch.int64)
        if rank == 0:
            rk.copy_(torch.from_numpy(O.aes_key_expansion(client.encrypt_u128(key)).view(np.int64)))
        broadcast_tensor(rk, src=0)
        lo, hi = shard_blocks(2, world, rank)
        res = []
        for i in range(lo, hi):
            st = O.aes_encrypt(rk.numpy().view(np.uint64), client.encrypt_u128(iv + i))
            res.append((i, client.decrypt_u128(st) == aes_clear.aes128_encrypt_block(key, iv + i)))
        digest = int(np.bitwise_xor.reduce(got.bsk[::997]))
        q.put((rank, lo, hi, res, digest))
    finally:
        dist.destroy_process_group()


def test_two_ranks_broadcast_keys_and_shard_blocks():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    out = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    out.sort()
    assert [(o[1], o[2]) for o in out] == [(0, 1), (1, 2)]
    assert out[0][4] == out[1][4]                                  # both ranks hold the same keys
    assert all(ok for o in out for _, ok in o[3])                  # every shard decrypts to AES-CTR
    assert sorted(i for o in out for i, _ in o[3]) == [0, 1]
