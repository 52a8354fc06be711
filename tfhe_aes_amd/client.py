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
        lib.fheaes_client_gen_secret_keys.argtypes = [pp, ctypes.c_uint64, u8p, u8p]
        lib.fheaes_client_gen_ksk.argtypes = [pp, ctypes.c_uint64, ctypes.c_uint64, u8p, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_gen_bsk.argtypes = [pp, ctypes.c_uint64, ctypes.c_uint64, u8p, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_gen_pfpksk.argtypes = [pp, ctypes.c_uint64, ctypes.c_uint64, u8p, ctypes.c_double, u64p]
        lib.fheaes_client_mask_word.argtypes = [ctypes.c_uint64] * 4
        lib.fheaes_client_mask_word.restype = ctypes.c_uint64
        lib.fheaes_client_encrypt_bits.argtypes = [pp, ctypes.c_uint64, u8p, ctypes.c_double, u8p, ctypes.c_uint64, u64p]
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


MASK_TAG_KSK, MASK_TAG_BSK, MASK_TAG_PFPKSK = 3, 4, 5          # csrc/client.c, csrc/engine.hip
_M64 = (1 << 64) - 1


def _mix64(z):
    """splitmix64 finaliser on numpy uint64 arrays (wrapping arithmetic)"""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def mask_words(mask_seed: int, tag: int, n_cts: int, words_per_ct: int) -> np.ndarray:
    """The public mask stream of csrc/client.c in numpy: [n_cts][words_per_ct] uint64.  Host-side twin of the
    engine's expansion kernel (tests compare both with the masks inside the full keys)."""
    with np.errstate(over="ignore"):
        ct = np.arange(n_cts, dtype=np.uint64)
        base = _mix64(np.uint64(mask_seed & _M64) ^ np.uint64((tag * 0xD6E8FEB86659FD93) & _M64) ^ (ct * np.uint64(0xA24BAED4963EE407)))
        j = np.arange(1, words_per_ct + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        return _mix64(base[:, None] + j[None, :])


@dataclass
class SeededServerKeys:
    """The evaluation keys as (public mask seed, bodies): every mask word is regenerated from ``mask_seed`` (on the GPU by
    ``fheaes_upload_keys_seeded``, on the host by ``expand()``), so 0.19 GB travel instead of 1.04 GB at PARAM_OPT.
    Bodies: KSK [kN][ks_level] words; BSK [n][pbs_level][k+1][N]; PFPKSK [k+1][kN+1][pfks_level][N]."""
    params: WopbsParameters
    mask_seed: int
    ksk_body: np.ndarray
    bsk_body: np.ndarray
    pfpksk_body: np.ndarray

    @property
    def nbytes(self) -> int:
        return 8 + self.ksk_body.nbytes + self.bsk_body.nbytes + self.pfpksk_body.nbytes

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
        np.savez(path, shape=_param_shape(self.params), mask_seed=np.array([self.mask_seed], dtype=np.uint64),
                 ksk_body=self.ksk_body, bsk_body=self.bsk_body, pfpksk_body=self.pfpksk_body)

    @staticmethod
    def load(path, params: WopbsParameters) -> "SeededServerKeys":
        with np.load(path, allow_pickle=False) as z:
            if list(map(int, z["shape"])) != list(map(int, _param_shape(params))):
                raise ValueError("key file was generated for a different parameter set")
            out = SeededServerKeys(params, int(z["mask_seed"][0]), z["ksk_body"].astype(np.uint64), z["bsk_body"].astype(np.uint64),
                                   z["pfpksk_body"].astype(np.uint64))
        k, N = params.k, params.N
        want = (params.big * params.ks_level, params.n * params.pbs_level * (k + 1) * N, (k + 1) * params.big1 * params.pfks_level * N)
        if (out.ksk_body.size, out.bsk_body.size, out.pfpksk_body.size) != want:
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
    mask_seed: int | None = None   # set when the masks follow the public stream of csrc/client.c (keys made by Client)

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
        np.savez(path, shape=_param_shape(self.params), ksk=self.ksk, bsk=self.bsk, pfpksk=self.pfpksk)

    @staticmethod
    def load(path, params: WopbsParameters) -> "ServerKeys":
        with np.load(path, allow_pickle=False) as z:
            want = [params.lwe_dimension, params.glwe_dimension, params.polynomial_size, params.pbs_base_log, params.pbs_level,
                    params.ks_base_log, params.ks_level, params.pfks_base_log, params.pfks_level, params.cbs_base_log, params.cbs_level]
            if list(map(int, z["shape"])) != want:
                raise ValueError("key file was generated for a different parameter set")
            keys = ServerKeys(params, z["ksk"].astype(np.uint64), z["bsk"].astype(np.uint64), z["pfpksk"].astype(np.uint64))
        if (keys.ksk.size, keys.bsk.size, keys.pfpksk.size) != (params.ksk_words, params.bsk_words, params.pfpksk_words):
            raise ValueError("key file has the wrong array sizes")
        return keys


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

    ``seed=None`` (the default, the counterpart of the reference's OS-seeded generators, client.rs:106-107): the key
    seed and the seed of every encryption call are drawn from ``os.urandom``.  An explicit integer ``seed`` is the
    TEST-ONLY deterministic mode (golden vectors, parity tests, synthetic bench data): keys derive from it and
    encryption call i uses ``seed + 0x1000 * i`` -- two processes with the same seed then reuse masks and noise,
    which is exactly what reproducible fixtures need and what real use must never do.  Either way the generator
    is xoshiro256** (csrc/client.c), not a CSPRNG: this Client makes synthetic inputs for the engine, it is not a
    hardened replacement of tfhe-rs key generation."""

    def __init__(self, number_of_outputs: int = 1, iv: int = 0, key: int = 0,
                 params: WopbsParameters = PARAM_OPT, seed: int | None = None):
        self.params = params
        self.number_of_outputs = number_of_outputs
        self.iv = iv
        self.key = key
        self.deterministic = seed is not None
        self.seed = int(seed) & (2 ** 64 - 1) if seed is not None else int.from_bytes(os.urandom(8), "little")
        # PUBLIC seed of the evaluation keys' mask words (it travels with the compressed keys); independent of the
        # secret seed unless the deterministic test mode derives both from one number
        self.mask_seed = (self.seed * 0x9E3779B97F4A7C15 + 0xA5A5A5A5) & (2 ** 64 - 1) if seed is not None else int.from_bytes(os.urandom(8), "little")
        self._enc_counter = 0
        lib = _load()
        self._c = params.c_struct()
        self.lwe_sk = np.zeros(params.n, dtype=np.uint8)
        self.glwe_sk = np.zeros(params.big, dtype=np.uint8)
        lib.fheaes_client_gen_secret_keys(ctypes.byref(self._c), seed, _u8(self.lwe_sk), _u8(self.glwe_sk))
        self._server_keys = None

    # -- keys -----------------------------------------------------------------
    def server_keys(self) -> ServerKeys:
        """gen_keys_radix + WopbsKey::new_wopbs_key_only_for_wopbs (client.rs:106-107)."""
        if self._server_keys is None:
            p, lib = self.params, _load()
            ksk = np.empty(p.ksk_words, dtype=np.uint64)
            bsk = np.empty(p.bsk_words, dtype=np.uint64)
            pf = np.empty(p.pfpksk_words, dtype=np.uint64)
            lib.fheaes_client_gen_ksk(ctypes.byref(self._c), self.seed, self.mask_seed, _u8(self.lwe_sk), _u8(self.glwe_sk),
                                      p.lwe_noise_std, _u64(ksk))
            lib.fheaes_client_gen_bsk(ctypes.byref(self._c), self.seed, self.mask_seed, _u8(self.lwe_sk), _u8(self.glwe_sk),
                                      p.glwe_noise_std, _u64(bsk))
            lib.fheaes_client_gen_pfpksk(ctypes.byref(self._c), self.seed, self.mask_seed, _u8(self.glwe_sk), p.pfks_noise_std, _u64(pf))
            self._server_keys = ServerKeys(p, ksk, bsk, pf, mask_seed=self.mask_seed)
        return self._server_keys

    # -- encryption -----------------------------------------------------------
    def encrypt_bits(self, bits: np.ndarray) -> np.ndarray:
        """LWE encryptions (big key, glwe noise: EncryptionKeyChoice::Big) of an array of bits; adds a last axis kN+1."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        out = np.empty(bits.shape + (self.params.big1,), dtype=np.uint64)
        self._enc_counter += 1
        enc_seed = (self.seed + 0x1000 * self._enc_counter) & (2 ** 64 - 1) if self.deterministic else int.from_bytes(os.urandom(8), "little")
        _load().fheaes_client_encrypt_bits(ctypes.byref(self._c), enc_seed, _u8(self.glwe_sk),
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
