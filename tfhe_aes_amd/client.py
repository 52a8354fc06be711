"""Host-side ``Client``: key generation, per-bit encryption, decryption and verification.

Mirrors /root/reference/src/client/client.rs:59-218 (``Client::new``, ``client_encrypt``,
``client_decrypt_and_verify``, ``test_verify``) over flat ``uint64`` arrays instead of
tfhe-rs containers.  The heavy loops live in csrc/client.c (libfheaes_client.so).

Array conventions (see include/fheaes.h):
  byte  = [8][kN+1]   (block j = bit j, LSB first)
  state = [16][8][kN+1], byte index = 4*col + row, byte 0 = MSB of the u128 (client.rs:126-129)
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

from . import _build
from .params import PARAM_OPT, CParams, WopbsParameters

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.build_client()
        lib = ctypes.CDLL(str(path))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        pp = ctypes.POINTER(CParams)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        lib.fheaes_client_gen_secret_keys.argtypes = [pp, u32p, u8p, u8p]
        lib.fheaes_client_gen_ksk.argtypes = [pp, u32p, u32p, u8p, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_gen_bsk.argtypes = [pp, u32p, u32p, u8p, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_gen_pfpksk.argtypes = [pp, u32p, u32p, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_mask_word.argtypes = [u32p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        lib.fheaes_client_mask_word.restype = ctypes.c_uint64
        lib.fheaes_client_chacha20_block.argtypes = [u32p, ctypes.c_uint32, u32p, u32p]
        lib.fheaes_client_chacha20_block.restype = None
        lib.fheaes_client_encrypt_bits.argtypes = [pp, u32p, u8p, ctypes.c_double, u8p, ctypes.c_uint64, u64p]
        lib.fheaes_client_decrypt_bits.argtypes = [pp, u8p, u64p, ctypes.c_uint64, u8p, u64p]
        lib.fheaes_client_phase_small.argtypes = [pp, u8p, u64p, ctypes.c_uint64, u64p]
        lib.fheaes_client_glwe_phase.argtypes = [pp, u8p, u64p, ctypes.c_uint64, u64p]
        for f in (lib.fheaes_client_gen_secret_keys, lib.fheaes_client_gen_ksk, lib.fheaes_client_gen_bsk,
                  lib.fheaes_client_gen_pfpksk, lib.fheaes_client_encrypt_bits, lib.fheaes_client_decrypt_bits,
                  lib.fheaes_client_phase_small, lib.fheaes_client_glwe_phase):
            f.restype = None
        _lib = lib
    return _lib


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _u64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def _u32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def random_key() -> np.ndarray:
    """a fresh 256-bit ChaCha20 key from the OS"""
    return np.frombuffer(os.urandom(32), dtype=np.uint32).copy()


def test_key(seed: int, purpose: int, counter: int = 0) -> np.ndarray:
    """TEST-ONLY deterministic 256-bit key: (seed, "test", purpose, counter) -- 64 bits of entropy at most"""
    seed &= (1 << 64) - 1
    return np.array([seed & 0xFFFFFFFF, seed >> 32, 0x74736574, purpose & 0xFFFFFFFF, counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, 0, 0], dtype=np.uint32)


MASK_TAG_KSK, MASK_TAG_BSK, MASK_TAG_PFPKSK = 3, 4, 5          # csrc/client.c, csrc/kern_linear.h


def chacha20_blocks(key8: np.ndarray, counters: np.ndarray, nonce0: int, nonce1: np.ndarray, nonce2: np.ndarray) -> np.ndarray:
    """RFC 8439 block function, vectorised: one block per entry of `counters` / `nonce1` / `nonce2` (broadcast together);
    returns uint32 [..., 16].  The host-side twin of csrc/client.c::chacha20_block and kern_linear.h::fheaes_chacha20_block."""
    counters, nonce1, nonce2 = np.broadcast_arrays(np.asarray(counters, dtype=np.uint32), np.asarray(nonce1, dtype=np.uint32),
                                                   np.asarray(nonce2, dtype=np.uint32))
    shape = counters.shape
    init = [np.full(shape, c, dtype=np.uint32) for c in (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)]
    init += [np.full(shape, int(k), dtype=np.uint32) for k in np.asarray(key8, dtype=np.uint32)]
    init += [counters.copy(), np.full(shape, nonce0 & 0xFFFFFFFF, dtype=np.uint32), nonce1.copy(), nonce2.copy()]
    x = [v.copy() for v in init]

    def rotl(v, k):
        return (v << np.uint32(k)) | (v >> np.uint32(32 - k))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7)

    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return np.stack([x[i] + init[i] for i in range(16)], axis=-1)


def mask_words(mask_key: np.ndarray, tag: int, n_cts: int, words_per_ct: int) -> np.ndarray:
    """The public mask stream of csrc/client.c in numpy: [n_cts][words_per_ct] uint64 -- 64-bit word j % 8 of ChaCha20 block
    j / 8 under (mask key, nonce = (tag, ct)).  Host-side twin of the engine's expansion kernel."""
    blocks = (words_per_ct + 7) // 8
    out = np.empty((n_cts, blocks * 8), dtype=np.uint64)
    step = max(1, (1 << 21) // blocks)                                     # bound the temporaries
    for c0 in range(0, n_cts, step):
        ct = np.arange(c0, min(n_cts, c0 + step), dtype=np.uint64)
        w = chacha20_blocks(mask_key, np.arange(blocks, dtype=np.uint32)[None, :], tag, (ct & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None],
                            (ct >> np.uint64(32)).astype(np.uint32)[:, None])      # [cts][blocks][16]
        out[c0:c0 + len(ct)] = np.ascontiguousarray(w).view(np.uint64).reshape(len(ct), blocks * 8)
    return out[:, :words_per_ct]


@dataclass
class SeededServerKeys:
    """The evaluation keys as (public mask seed, bodies): every mask word is regenerated from ``mask_seed`` (on the GPU by
    ``fheaes_upload_keys_seeded``, on the host by ``expand()``), so 0.19 GB travel instead of 1.04 GB at PARAM_OPT.
    Bodies: KSK [kN][ks_level] words; BSK [n][pbs_level][k+1][N]; PFPKSK [k+1][kN+1][pfks_level][N]."""
    params: WopbsParameters
    mask_seed: np.ndarray          # the PUBLIC 256-bit mask key, uint32[8]
    ksk_body: np.ndarray
    bsk_body: np.ndarray
    pfpksk_body: np.ndarray

    @property
    def nbytes(self) -> int:
        return 32 + self.ksk_body.nbytes + self.bsk_body.nbytes + self.pfpksk_body.nbytes

    def expand(self) -> "ServerKeys":
        p = self.params
        k, N, n = p.k, p.N, p.n
        ksk = np.empty((p.big * p.ks_level, n + 1), dtype=np.uint64)
        ksk[:, :n] = mask_words(self.mask_seed, MASK_TAG_KSK, p.big * p.ks_level, n)
        ksk[:, n] = self.ksk_body.reshape(-1)
        nb = p.n * p.pbs_level * (k + 1)
        bsk = np.empty((nb, (k + 1) * N), dtype=np.uint64)
        bsk[:, :k * N] = mask_words(self.mask_seed, MASK_TAG_BSK, nb, k * N)
        bsk[:, k * N:] = self.bsk_body.reshape(nb, N)
        npf = (k + 1) * p.big1 * p.pfks_level
        pf = np.empty((npf, (k + 1) * N), dtype=np.uint64)
        pf[:, :k * N] = mask_words(self.mask_seed, MASK_TAG_PFPKSK, npf, k * N)
        pf[:, k * N:] = self.pfpksk_body.reshape(npf, N)
        return ServerKeys(p, ksk.reshape(-1), bsk.reshape(-1), pf.reshape(-1), mask_seed=self.mask_seed)

    def save(self, path) -> None:
        np.savez(path, shape=_param_shape(self.params), mask_seed=np.asarray(self.mask_seed, dtype=np.uint32),
                 ksk_body=self.ksk_body, bsk_body=self.bsk_body, pfpksk_body=self.pfpksk_body)

    @staticmethod
    def load(path, params: WopbsParameters) -> "SeededServerKeys":
        with np.load(path, allow_pickle=False) as z:
            if list(map(int, z["shape"])) != list(map(int, _param_shape(params))):
                raise ValueError("key file was generated for a different parameter set")
            out = SeededServerKeys(params, z["mask_seed"].astype(np.uint32), z["ksk_body"].astype(np.uint64), z["bsk_body"].astype(np.uint64),
                                   z["pfpksk_body"].astype(np.uint64))
        k, N = params.k, params.N
        want = (params.big * params.ks_level, params.n * params.pbs_level * (k + 1) * N, (k + 1) * params.big1 * params.pfks_level * N)
        if (out.ksk_body.size, out.bsk_body.size, out.pfpksk_body.size) != want or out.mask_seed.size != 8:
            raise ValueError("key file has the wrong array sizes")
        return out


def _param_shape(p: WopbsParameters) -> np.ndarray:
    return np.array([p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_base_log, p.pbs_level, p.ks_base_log,
                     p.ks_level, p.pfks_base_log, p.pfks_level, p.cbs_base_log, p.cbs_level], dtype=np.uint32)


@dataclass
class ServerKeys:
    """What ``client_encrypt`` hands to ``Server::new`` (client.rs:143): the evaluation keys."""

    params: WopbsParameters
    ksk: np.ndarray      # [kN][ks_level][n+1]
    bsk: np.ndarray      # [n][pbs_level][k+1][k+1][N]   standard domain
    pfpksk: np.ndarray   # [k+1][kN+1][pfks_level][(k+1)N]
    mask_seed: np.ndarray | None = None   # the public 256-bit mask key when the masks follow the stream of csrc/client.c (keys made by Client)

    def compress(self) -> "SeededServerKeys":
        """(mask_seed, bodies): drops every mask word (they are a function of the public mask seed)"""
        if self.mask_seed is None:
            raise ValueError("these keys do not carry a mask seed (not generated by Client)")
        p = self.params
        k, N, n = p.k, p.N, p.n
        ksk_b = np.ascontiguousarray(self.ksk.reshape(-1, n + 1)[:, n]).reshape(p.big, p.ks_level)
        bsk_b = np.ascontiguousarray(self.bsk.reshape(-1, (k + 1) * N)[:, k * N:]).reshape(p.n, p.pbs_level, k + 1, N)
        pf_b = np.ascontiguousarray(self.pfpksk.reshape(-1, (k + 1) * N)[:, k * N:]).reshape(k + 1, p.big1, p.pfks_level, N)
        return SeededServerKeys(p, self.mask_seed, ksk_b, bsk_b, pf_b)

    # The reference never serialises anything; these two helpers exist so that keys produced
    # elsewhere can be fed to the engine.  Plain .npz of uint64 arrays (no pickle), layouts as in include/fheaes.h.
    def save(self, path) -> None:
        extra = {} if self.mask_seed is None else {"mask_seed": np.asarray(self.mask_seed, dtype=np.uint32).reshape(8)}
        np.savez(path, shape=_param_shape(self.params), ksk=self.ksk, bsk=self.bsk, pfpksk=self.pfpksk, **extra)

    @staticmethod
    def load(path, params: WopbsParameters) -> "ServerKeys":
        with np.load(path, allow_pickle=False) as z:
            want = [params.lwe_dimension, params.glwe_dimension, params.polynomial_size, params.pbs_base_log, params.pbs_level,
                    params.ks_base_log, params.ks_level, params.pfks_base_log, params.pfks_level, params.cbs_base_log, params.cbs_level]
            if list(map(int, z["shape"])) != want:
                raise ValueError("key file was generated for a different parameter set")
            seed = None
            if "mask_seed" in z.files:          # the public mask key travels with the keys: compress() still works after a round trip
                seed = z["mask_seed"]
                if seed.shape != (8,) or seed.dtype != np.uint32:
                    raise ValueError("key file has a malformed mask_seed (expected uint32[8])")
                seed = seed.copy()
            keys = ServerKeys(params, z["ksk"].astype(np.uint64), z["bsk"].astype(np.uint64), z["pfpksk"].astype(np.uint64), seed)
        if (keys.ksk.size, keys.bsk.size, keys.pfpksk.size) != (params.ksk_words, params.bsk_words, params.pfpksk_words):
            raise ValueError("key file has the wrong array sizes")
        return keys


# ---- ciphertext interchange --------------------------------------------------------------------------------
# The reference hands whole states across in memory (client_encrypt / client_decrypt_and_verify, client.rs:123-175) and never
# serialises them.  These helpers give encrypted states and round keys an on-disk / on-wire form so that inputs produced
# elsewhere can be fed to the engine: a plain .npz of uint64 words (no pickle) in the layout of include/fheaes.h, with the
# parameter set and the kind recorded and checked on load.
CIPHERTEXT_KINDS = {
    "state": (16, 8),           # [blocks][16 bytes][8 bits][kN+1]      Server::aes_encrypt / aes_decrypt / add_scalar
    "round_keys": (11, 16, 8),  # [11][16][8][kN+1]                     Server::aes_key_expansion output
    "bytes": (8,),              # [n][8][kN+1]                          sbox / many_sbox inputs
}


def save_ciphertexts(path, params: WopbsParameters, kind: str, words: np.ndarray) -> None:
    if kind not in CIPHERTEXT_KINDS:
        raise ValueError("kind must be one of %s" % ", ".join(CIPHERTEXT_KINDS))
    tail = CIPHERTEXT_KINDS[kind] + (params.big1,)
    a = np.ascontiguousarray(words, dtype=np.uint64)
    if a.shape[-len(tail):] != tail:
        raise ValueError("a %r array must end in shape %r, got %r" % (kind, tail, a.shape))
    np.savez(path, shape=_param_shape(params), kind=np.frombuffer(kind.encode().ljust(16, b"\0"), dtype=np.uint8), words=a)


def load_ciphertexts(path, params: WopbsParameters, kind: str) -> np.ndarray:
    with np.load(path, allow_pickle=False) as z:
        if list(map(int, z["shape"])) != list(map(int, _param_shape(params))):
            raise ValueError("ciphertext file was produced for a different parameter set")
        got = bytes(z["kind"]).rstrip(b"\0").decode()
        if got != kind:
            raise ValueError("ciphertext file holds %r, expected %r" % (got, kind))
        a = z["words"].astype(np.uint64)
    tail = CIPHERTEXT_KINDS[kind] + (params.big1,)
    if a.shape[-len(tail):] != tail:
        raise ValueError("ciphertext file has the wrong array shape")
    return a


def u128_to_bytes(x: int) -> list[int]:
    """state byte i = bits [8*(15-i), 8*(16-i)) of the u128 (client.rs:126-129)."""
    return [(x >> (8 * (15 - i))) & 0xFF for i in range(16)]


def bytes_to_u128(b) -> int:
    v = 0
    for i, x in enumerate(b):
        v |= int(x) << (8 * (15 - i))
    return v


class Client:
    """``Client::new`` (client.rs:70): generates the secret and evaluation keys.

    All randomness is ChaCha20 (csrc/client.c).  ``seed=None`` (the default, the counterpart of the reference's OS-seeded
    generators, client.rs:106-107): the 256-bit secret key-generation key, the 256-bit PUBLIC mask key and a fresh 256-bit key
    for every encryption call come from ``os.urandom``.  An explicit integer ``seed`` is the TEST-ONLY deterministic mode
    (golden vectors, parity tests, synthetic bench data): all keys derive from that one number (at most 64 bits of entropy)
    and encryption call i is reproducible -- two processes with the same seed then produce the same masks and noise, which
    is exactly what fixtures need and what real use must never do."""

    def __init__(self, number_of_outputs: int = 1, iv: int = 0, key: int = 0,
                 params: WopbsParameters = PARAM_OPT, seed: int | None = None):
        self.params = params
        self.number_of_outputs = number_of_outputs
        self.iv = iv
        self.key = key
        self.deterministic = seed is not None
        self.test_seed = int(seed) if seed is not None else None
        self.seed = test_key(self.test_seed, 1) if self.deterministic else random_key()          # SECRET
        self.mask_seed = test_key(self.test_seed, 2) if self.deterministic else random_key()     # PUBLIC: travels with seeded keys
        self._enc_counter = 0
        lib = _load()
        self._c = params.c_struct()
        self.lwe_sk = np.zeros(params.n, dtype=np.uint8)
        self.glwe_sk = np.zeros(params.big, dtype=np.uint8)
        lib.fheaes_client_gen_secret_keys(ctypes.byref(self._c), _u32(self.seed), _u8(self.lwe_sk), _u8(self.glwe_sk))
        self._server_keys = None

    # -- keys -----------------------------------------------------------------
    def server_keys(self) -> ServerKeys:
        """gen_keys_radix + WopbsKey::new_wopbs_key_only_for_wopbs (client.rs:106-107)."""
        if self._server_keys is None:
            p, lib = self.params, _load()
            ksk = np.empty(p.ksk_words, dtype=np.uint64)
            bsk = np.empty(p.bsk_words, dtype=np.uint64)
            pf = np.empty(p.pfpksk_words, dtype=np.uint64)
            lib.fheaes_client_gen_ksk(ctypes.byref(self._c), _u32(self.seed), _u32(self.mask_seed), _u8(self.lwe_sk), _u8(self.glwe_sk),
                                      p.lwe_noise_std, _u64(ksk))
            lib.fheaes_client_gen_bsk(ctypes.byref(self._c), _u32(self.seed), _u32(self.mask_seed), _u8(self.lwe_sk), _u8(self.glwe_sk),
                                      p.glwe_noise_std, _u64(bsk))
            lib.fheaes_client_gen_pfpksk(ctypes.byref(self._c), _u32(self.seed), _u32(self.mask_seed), _u8(self.glwe_sk), p.pfks_noise_std,
                                         _u64(pf))
            self._server_keys = ServerKeys(p, ksk, bsk, pf, mask_seed=self.mask_seed)
        return self._server_keys

    # -- encryption -----------------------------------------------------------
    def encrypt_bits(self, bits: np.ndarray) -> np.ndarray:
        """LWE encryptions (big key, glwe noise: EncryptionKeyChoice::Big) of an array of bits; adds a last axis kN+1."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        out = np.empty(bits.shape + (self.params.big1,), dtype=np.uint64)
        self._enc_counter += 1
        enc_key = test_key(self.test_seed, 3, self._enc_counter) if self.deterministic else random_key()
        _load().fheaes_client_encrypt_bits(ctypes.byref(self._c), _u32(enc_key), _u8(self.glwe_sk),
                                           self.params.glwe_noise_std, _u8(bits), bits.size, _u64(out))
        return out

    def encrypt_bytes(self, values) -> np.ndarray:
        """cks.encrypt_without_padding per byte (client.rs:128): [len][8][kN+1]."""
        v = np.asarray(values, dtype=np.uint64).reshape(-1)
        bits = ((v[:, None] >> np.arange(8, dtype=np.uint64)[None, :]) & 1).astype(np.uint8)
        return self.encrypt_bits(bits)

    def encrypt_u128(self, x: int) -> np.ndarray:
        """one AES state / key: [16][8][kN+1]"""
        return self.encrypt_bytes(u128_to_bytes(x))

    def client_encrypt(self):
        """client.rs:123: (server keys, encrypted iv, encrypted key)."""
        return self.server_keys(), self.encrypt_u128(self.iv), self.encrypt_u128(self.key)

    # -- decryption -----------------------------------------------------------
    def decrypt_bits(self, lwe: np.ndarray, return_phase: bool = False):
        lwe = np.ascontiguousarray(lwe, dtype=np.uint64)
        assert lwe.shape[-1] == self.params.big1
        shape = lwe.shape[:-1]
        bits = np.empty(shape, dtype=np.uint8)
        phase = np.empty(shape, dtype=np.uint64)
        count = int(np.prod(shape)) if shape else 1
        _load().fheaes_client_decrypt_bits(ctypes.byref(self._c), _u8(self.glwe_sk), _u64(lwe), count, _u8(bits), _u64(phase))
        return (bits, phase) if return_phase else bits

    def decrypt_bytes(self, lwe: np.ndarray) -> np.ndarray:
        """[..., 8, kN+1] -> [...] byte values (decrypt_without_padding, client.rs:154)."""
        bits = self.decrypt_bits(lwe).astype(np.uint64)
        return (bits << np.arange(8, dtype=np.uint64)).sum(axis=-1).astype(np.uint8)

    def decrypt_u128(self, state: np.ndarray) -> int:
        return bytes_to_u128(self.decrypt_bytes(state).reshape(16))

    def phase_small(self, lwe_small: np.ndarray) -> np.ndarray:
        lwe_small = np.ascontiguousarray(lwe_small, dtype=np.uint64)
        shape = lwe_small.shape[:-1]
        out = np.empty(shape, dtype=np.uint64)
        _load().fheaes_client_phase_small(ctypes.byref(self._c), _u8(self.lwe_sk), _u64(lwe_small), out.size, _u64(out))
        return out

    def glwe_phase(self, glwe: np.ndarray) -> np.ndarray:
        """[..., (k+1)N] -> [..., N] phases B - sum A_m S_m"""
        glwe = np.ascontiguousarray(glwe, dtype=np.uint64)
        shape = glwe.shape[:-1]
        out = np.empty(shape + (self.params.N,), dtype=np.uint64)
        count = int(np.prod(shape)) if shape else 1
        _load().fheaes_client_glwe_phase(ctypes.byref(self._c), _u8(self.glwe_sk), _u64(glwe), count, _u64(out))
        return out

    # -- verification (client.rs:147-216) --------------------------------------
    def client_decrypt_and_verify(self, states) -> None:
        """decrypt every CTR output block and compare with AES-128(key, iv + index)."""
        from .aes_clear import aes128_encrypt_block

        assert len(states) == self.number_of_outputs
        for index, st in enumerate(states):
            got = self.decrypt_u128(st)
            want = aes128_encrypt_block(self.key, (self.iv + index) & ((1 << 128) - 1))
            assert got == want, "block %d: FHE %032x != AES %032x" % (index, got, want)

    def test_verify(self, state_enc, state_dec) -> None:
        from .aes_clear import aes128_encrypt_block

        got = self.decrypt_u128(state_enc)
        want = aes128_encrypt_block(self.key, self.iv)
        assert got == want, "enc: FHE %032x != AES %032x" % (got, want)
        back = self.decrypt_u128(state_dec)
        assert back == self.iv, "dec: FHE %032x != %032x" % (back, self.iv)
