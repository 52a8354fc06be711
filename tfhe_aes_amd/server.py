"""``Server`` and the S-Box front end: the reference's operator API for the hot path.

Same names, argument meaning and error behaviour as
  /root/reference/src/server/server.rs        Server::{new, aes_encrypt, aes_decrypt, aes_key_expansion, add_scalar}
  /root/reference/src/server/sbox/sbox.rs     sbox, many_sbox, mul2 .. mul14
  /root/reference/src/server/sbox/many_wopbs.rs  many_wopbs_without_padding
  /root/reference/src/server/sbox/gen_lut.rs  gen_lut
but batched (a leading block axis) and over flat uint64 arrays:
  byte = [8][kN+1], state = [16][8][kN+1], round keys = [11][16][8][kN+1].
Arrays may be numpy (host; staged through HBM by the engine) or torch CUDA tensors (resident,
asynchronous on the engine's stream).  All compute happens in libfheaes.so (HIP); this module
only allocates outputs and forwards.  README.md:57-59 of the reference spells the methods
``aes_encryption`` / ``aes_decryption``; both spellings are provided.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .aes_clear import INV_SBOX, SBOX, mul2, mul3, mul9, mul11, mul13, mul14  # noqa: F401  (re-exported like sbox.rs)
from .client import ServerKeys
from .params import WopbsParameters


def gen_lut(message_mod: int, carry_mod: int, poly_size: int, nb_block: int, f) -> np.ndarray:
    """gen_lut.rs:9-42.  Returns [nb_block][max(2^nb_block, poly_size)] uint64 (gen_lut.rs:19-23), entry = output bit << 63."""
    if message_mod != 2 or carry_mod != 1:
        raise ValueError("the path uses message_modulus 2, carry_modulus 1 (client.rs:53-54)")
    if poly_size != 512 or not 1 <= nb_block <= 16:
        raise ValueError("polynomial_size must be 512 and nb_block in 1..16")
    table = np.array([int(f(x)) for x in range(1 << nb_block)], dtype=np.uint64)
    return _native.gen_lut(nb_block, table)


def _empty_like(ref, shape):
    if isinstance(ref, np.ndarray):
        return np.empty(shape, dtype=np.uint64)
    import torch

    return torch.empty(shape, dtype=torch.int64, device=ref.device)


def _to_space(arr: np.ndarray, ref):
    if isinstance(ref, np.ndarray):
        return np.ascontiguousarray(arr, dtype=np.uint64)
    import torch

    dev = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).to(ref.device)
    # the copy ran on torch's current stream; the engine reads `dev` on ITS stream: finish the copy first
    torch.cuda.current_stream(ref.device).synchronize()
    return dev


class Server:
    """``Server::new`` (server.rs:32): takes the evaluation keys by value; the engine copies them to HBM."""

    def __init__(self, keys: ServerKeys | None, device: int = 0, engine: _native.Engine | None = None, clone_from: "Server | None" = None):
        """`clone_from`: take the converted key images from another Server's context, device to device (fheaes_clone_keys),
        instead of uploading `keys` again (which may then be None)."""
        self.params: WopbsParameters = clone_from.params if clone_from is not None else keys.params
        self.engine = engine or _native.Engine(self.params, device)
        # device temporaries this wrapper created for calls that are still in flight on the engine's stream; they must
        # outlive the kernels that read them (torch's caching allocator would hand the block out again): freed in synchronize()
        self._inflight = []
        if clone_from is not None:
            self.engine.clone_keys_from(clone_from.engine)
        else:
            self.engine.upload_keys(np.ascontiguousarray(keys.ksk), np.ascontiguousarray(keys.bsk), np.ascontiguousarray(keys.pfpksk))

    # ---- S-Box front end --------------------------------------------------------
    def many_wopbs_without_padding(self, ct_in, luts):
        """many_wopbs.rs:31: ct_in [n][bits][kN+1]; luts: list of gen_lut tables [bits][W] (shared by all inputs) or an array
        [n][n_luts][bits][W] (one set per input), W = max(2^bits, 512).  Returns [n][n_luts][bits][kN+1].  The AES path uses
        bits = 8 and 9; wider inputs (up to 16 bits) go through the CMUX tree of vertical_packing first."""
        n, bits = int(ct_in.shape[0]), int(ct_in.shape[1])
        if isinstance(luts, (list, tuple)):
            lut_arr = np.stack([np.asarray(l, dtype=np.uint64) for l in luts])[None]
            per_input = False
        else:
            lut_arr = np.asarray(luts) if isinstance(luts, np.ndarray) else luts
            per_input = lut_arr.ndim == 4 and lut_arr.shape[0] == n and n > 1
            if lut_arr.ndim == 3:
                lut_arr = lut_arr[None]
        n_luts = int(lut_arr.shape[1])
        if int(lut_arr.shape[2]) != bits or int(lut_arr.shape[3]) != max(512, 1 << bits):
            raise ValueError("LUT shape does not match the input radix width")
        lut_dev = _to_space(lut_arr, ct_in) if isinstance(lut_arr, np.ndarray) else lut_arr
        out = _empty_like(ct_in, (n, n_luts, bits, self.params.big1))
        self.engine.wopbs_batch(ct_in, n, bits, lut_dev, n_luts, per_input, out)
        if not isinstance(ct_in, np.ndarray) and lut_dev is not luts:
            self._inflight.append(lut_dev)            # device call: only enqueued, the LUT copy is still being read
        return out

    def sbox(self, ct_in, inv: bool):
        """sbox.rs:46, in place over a batch of bytes [n][8][kN+1]."""
        self.engine.sbox(ct_in, int(ct_in.shape[0]), inv)
        return ct_in

    def many_sbox(self, ct_in, inv: bool):
        """sbox.rs:68: [n][8][kN+1] -> [n][L][8][kN+1]; L=3 (S, 2S, 3S) or 4 (9x, 11x, 13x, 14x)."""
        n = int(ct_in.shape[0])
        out = _empty_like(ct_in, (n, 4 if inv else 3, 8, self.params.big1))
        self.engine.many_sbox(ct_in, n, inv, out)
        return out

    # ---- Server API -------------------------------------------------------------
    def aes_key_expansion(self, key):
        """server.rs:107: key [16][8][kN+1] -> round keys [11][16][8][kN+1]."""
        rk = _empty_like(key, (11, 16, 8, self.params.big1))
        self.engine.aes_key_expansion(key, rk)
        return rk

    def aes_encrypt(self, encrypted_round_keys, state):
        """server.rs:39, in place.  state [16][8][kN+1] or a batch [B][16][8][kN+1]."""
        n_blocks = 1 if state.ndim == 3 else int(state.shape[0])
        self.engine.aes_encrypt(encrypted_round_keys, state, n_blocks)
        return state

    def aes_decrypt(self, encrypted_round_keys, state):
        """server.rs:67, in place."""
        n_blocks = 1 if state.ndim == 3 else int(state.shape[0])
        self.engine.aes_decrypt(encrypted_round_keys, state, n_blocks)
        return state

    def add_scalar(self, state, i):
        """server.rs:172, in place.  ``i`` is one integer, or one per block of a batched state."""
        n_blocks = 1 if state.ndim == 3 else int(state.shape[0])
        counters = [i] * n_blocks if isinstance(i, int) else list(i)
        if len(counters) != n_blocks:
            raise ValueError("one counter per block expected")
        self.engine.add_scalar(state, n_blocks, counters)
        return state

    # README.md:57-59 spellings
    aes_encryption = aes_encrypt
    aes_decryption = aes_decrypt

    def synchronize(self):
        self.engine.synchronize()
        self._inflight.clear()


class ServerGroup:
    """Several engine contexts behind one `Server`-shaped object: what the reference does with rayon over CTR blocks
    (main.rs:55-64, one `&Server` shared by the worker threads) done with one context per GPU -- or several on one GPU --
    and one host thread per context.  Keys are uploaded ONCE (context 0) and cloned device to device into the others
    (fheaes_clone_keys: xGMI between GPUs).  Blocks are sharded contiguously, block i -> context i * G / n (dist.shard_blocks);
    no data moves between contexts.  The Rust counterpart is `GpuServerGroup` in integration/rust_shim/src/lib.rs; the
    one-process-per-GPU path of bench.py (torch.distributed, RCCL broadcast of the seeded keys) is the other way to the same split."""

    def __init__(self, keys: ServerKeys, devices=(0,)):
        if not devices:
            raise ValueError("at least one device")
        self.params = keys.params
        self.servers = [Server(keys, device=devices[0])]
        for d in devices[1:]:
            self.servers.append(Server(None, device=d, clone_from=self.servers[0]))

    def _fan_out(self, fn, state):
        import threading

        from .dist import shard_blocks

        if state.ndim != 4:
            # Server.aes_encrypt also takes ONE state [16][8][kN+1]; here the first axis is what gets sharded, so a single state would
            # be cut into 16 "blocks" of one byte each: refuse it instead of computing nonsense
            raise ValueError("ServerGroup works on a batch [n_blocks][16][8][kN+1]; wrap a single state as state[None]")
        n, g = int(state.shape[0]), len(self.servers)
        errs = [None] * g

        def work(i):
            lo, hi = shard_blocks(n, g, i)
            try:
                if hi > lo:
                    fn(self.servers[i], state[lo:hi], lo)
                    self.servers[i].synchronize()
            except Exception as e:       # surfaced below, on the caller's thread
                errs[i] = e

        ts = [threading.Thread(target=work, args=(i,)) for i in range(g)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return state

    def aes_encrypt(self, round_keys, state):
        """[n_blocks][16][8][kN+1] in place (host arrays: every context stages its own shard)"""
        return self._fan_out(lambda s, shard, lo: s.aes_encrypt(round_keys, shard), state)

    def aes_decrypt(self, round_keys, state):
        return self._fan_out(lambda s, shard, lo: s.aes_decrypt(round_keys, shard), state)

    def add_scalar(self, state, counters):
        counters = list(counters)
        return self._fan_out(lambda s, shard, lo: s.add_scalar(shard, counters[lo:lo + int(shard.shape[0])]), state)

    def aes_key_expansion(self, key):
        return self.servers[0].aes_key_expansion(key)

    def clone_info(self):
        """per cloned context: how its keys got there ({"path": "same_device" | "peer" | "staged", "bytes", "seconds"})"""
        return [s.engine.clone_info() for s in self.servers[1:]]
