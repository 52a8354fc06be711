"""ctypes wrapper of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this.
See oracle/fheaes_oracle.c for what it restates and how it is pinned ("parity unpinned"
against tfhe-rs ciphertext bits; pinned by the reference's plaintext known-answer tests, an
exact schoolbook product and FIPS-197 tables).
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SO = HERE / "liboracle.so"

LUTSET_ENC_ROUND, LUTSET_SBOX, LUTSET_INV_SBOX, LUTSET_DEC_MUL, LUTSET_IDENTITY = range(5)


class OrcParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n", "k", "N", "pbs_base_log", "pbs_level", "ks_base_log", "ks_level",
        "pfks_base_log", "pfks_level", "cbs_base_log", "cbs_level")]


def _build():
    src = HERE / "fheaes_oracle.c"
    if not SO.exists() or SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        L = ctypes.CDLL(str(SO))
        u64p = ctypes.POINTER(ctypes.c_uint64)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        dp = ctypes.POINTER(ctypes.c_double)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        pp = ctypes.POINTER(OrcParams)
        vp = ctypes.c_void_p
        sig = {
            "orc_get_twiddles": (None, [dp]),
            "orc_fft_fwd_int": (None, [i64p, dp]),
            "orc_fft_fwd_torus": (None, [u64p, dp]),
            "orc_negacyclic_mul_fft": (None, [i64p, u64p, u64p]),
            "orc_negacyclic_mul_exact": (None, [i64p, u64p, u64p]),
            "orc_decompose": (None, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, i32p]),
            "orc_decompose_offset": (None, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, i32p]),
            "orc_mod_switch": (ctypes.c_int, [ctypes.c_uint64]),
            "orc_polys_to_fourier": (None, [u64p, ctypes.c_int64, dp]),
            "orc_external_product_add": (None, [pp, ctypes.c_int, ctypes.c_int, u64p, u64p, u64p]),
            "orc_keys_create": (vp, [pp, u64p, u64p, u64p]),
            "orc_keys_destroy": (None, [vp]),
            "orc_keyswitch": (None, [vp, u64p, u64p]),
            "orc_cbs_pbs": (None, [vp, u64p, ctypes.c_int, u64p]),
            "orc_pfpks": (None, [vp, ctypes.c_int, u64p, u64p]),
            "orc_cbs_pbs_batch": (None, [vp, u64p, ctypes.c_int64, ctypes.c_int, u64p]),
            "orc_keyswitch_batch": (None, [vp, u64p, ctypes.c_int64, u64p]),
            "orc_circuit_bootstrap": (None, [vp, u64p, u64p]),
            "orc_wopbs_batch": (None, [vp, u64p, ctypes.c_int, ctypes.c_int, u64p, ctypes.c_int, ctypes.c_int, u64p, u64p, u64p, u64p]),
            "orc_get_tables": (None, [u8p, u8p]),
            "orc_gen_lut": (None, [ctypes.c_int, u64p, u64p]),
            "orc_build_lutset": (ctypes.c_int, [ctypes.c_int, u64p]),
            "orc_aes_encrypt": (None, [vp, u64p, u64p]),
            "orc_aes_decrypt": (None, [vp, u64p, u64p]),
            "orc_aes_key_expansion": (None, [vp, u64p, u64p]),
            "orc_add_scalar": (None, [vp, u64p, ctypes.c_uint64, ctypes.c_uint64]),
            "orc_num_threads": (ctypes.c_int, []),
            "orc_set_threads": (None, [ctypes.c_int]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _u64(a):
    return _p(a, ctypes.c_uint64)


def orc_params(p) -> OrcParams:
    return OrcParams(p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level,
                     p.pfks_base_log, p.pfks_level, p.cbs_base_log, p.cbs_level)


# ---------------------------------------------------------------- primitives
def twiddles() -> np.ndarray:
    out = np.empty((512, 2), dtype=np.float64)
    lib().orc_get_twiddles(_p(out, ctypes.c_double))
    return out


def decompose(x: int, base_log: int, level: int) -> np.ndarray:
    out = np.empty(level, dtype=np.int32)
    lib().orc_decompose(ctypes.c_uint64(x), base_log, level, _p(out, ctypes.c_int32))
    return out


def decompose_offset(x: int, base_log: int, level: int) -> np.ndarray:
    """the decomposition of the external products (canonical form v3): closest-representable rounding, then the offset rule"""
    out = np.empty(level, dtype=np.int32)
    lib().orc_decompose_offset(ctypes.c_uint64(x), base_log, level, _p(out, ctypes.c_int32))
    return out


def mod_switch(x: int) -> int:
    return lib().orc_mod_switch(ctypes.c_uint64(x))


def negacyclic_mul_fft(small: np.ndarray, torus: np.ndarray) -> np.ndarray:
    small = np.ascontiguousarray(small, dtype=np.int64)
    torus = np.ascontiguousarray(torus, dtype=np.uint64)
    out = np.empty(512, dtype=np.uint64)
    lib().orc_negacyclic_mul_fft(_p(small, ctypes.c_int64), _u64(torus), _u64(out))
    return out


def negacyclic_mul_exact(small: np.ndarray, torus: np.ndarray) -> np.ndarray:
    small = np.ascontiguousarray(small, dtype=np.int64)
    torus = np.ascontiguousarray(torus, dtype=np.uint64)
    out = np.empty(512, dtype=np.uint64)
    lib().orc_negacyclic_mul_exact(_p(small, ctypes.c_int64), _u64(torus), _u64(out))
    return out


def polys_to_fourier(polys: np.ndarray) -> np.ndarray:
    """[..., 512] torus polynomials -> [..., 256, 2] canonical Fourier image"""
    polys = np.ascontiguousarray(polys, dtype=np.uint64)
    assert polys.shape[-1] == 512
    out = np.empty(polys.shape[:-1] + (256, 2), dtype=np.float64)
    lib().orc_polys_to_fourier(_u64(polys), polys.size // 512, _p(out, ctypes.c_double))
    return out


def external_product_add(p, level: int, base_log: int, ggsw_std: np.ndarray, d: np.ndarray, acc: np.ndarray) -> np.ndarray:
    cp = orc_params(p)
    acc = np.ascontiguousarray(acc, dtype=np.uint64).copy()
    lib().orc_external_product_add(ctypes.byref(cp), level, base_log, _u64(np.ascontiguousarray(ggsw_std)),
                                   _u64(np.ascontiguousarray(d)), _u64(acc))
    return acc


def tables():
    s = np.empty(256, dtype=np.uint8)
    i = np.empty(256, dtype=np.uint8)
    lib().orc_get_tables(_p(s, ctypes.c_uint8), _p(i, ctypes.c_uint8))
    return s, i


def gen_lut(nb_block: int, f_table) -> np.ndarray:
    f = np.ascontiguousarray(f_table, dtype=np.uint64)
    assert f.size == 1 << nb_block
    out = np.empty((nb_block, max(512, 1 << nb_block)), dtype=np.uint64)
    lib().orc_gen_lut(nb_block, _u64(f), _u64(out))
    return out


def build_lutset(which: int) -> np.ndarray:
    buf = np.zeros((4, 8, 512), dtype=np.uint64)
    n = lib().orc_build_lutset(which, _u64(buf))
    return buf[:n].copy()


# ---------------------------------------------------------------- keyed evaluation
class Oracle:
    """Evaluation of the path on the CPU with a given key set (arrays laid out as in include/fheaes.h)."""

    def __init__(self, params, ksk: np.ndarray, bsk: np.ndarray, pfpksk: np.ndarray):
        self.params = params
        self._cp = orc_params(params)
        self._ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        self._pf = np.ascontiguousarray(pfpksk, dtype=np.uint64)
        bsk = np.ascontiguousarray(bsk, dtype=np.uint64)
        self._h = lib().orc_keys_create(ctypes.byref(self._cp), _u64(self._ksk), _u64(bsk), _u64(self._pf))
        if not self._h:
            raise ValueError("oracle supports polynomial_size 512 only")

    def __del__(self):
        try:
            if self._h:
                lib().orc_keys_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def keyswitch(self, lwe_in: np.ndarray) -> np.ndarray:
        p = self.params
        x = np.ascontiguousarray(lwe_in, dtype=np.uint64).reshape(-1, p.big1)
        out = np.empty((x.shape[0], p.n + 1), dtype=np.uint64)
        lib().orc_keyswitch_batch(self._h, _u64(x), x.shape[0], _u64(out))
        return out.reshape(lwe_in.shape[:-1] + (p.n + 1,))

    def cbs_pbs(self, lwe_small: np.ndarray, level: int = 1) -> np.ndarray:
        p = self.params
        x = np.ascontiguousarray(lwe_small, dtype=np.uint64).reshape(-1, p.n + 1)
        out = np.empty((x.shape[0], p.big1), dtype=np.uint64)
        lib().orc_cbs_pbs_batch(self._h, _u64(x), x.shape[0], level, _u64(out))
        return out.reshape(lwe_small.shape[:-1] + (p.big1,))

    def pfpks(self, lwe_in: np.ndarray) -> np.ndarray:
        """[m][kN+1] -> [m][k+1][(k+1)N]"""
        p = self.params
        x = np.ascontiguousarray(lwe_in, dtype=np.uint64).reshape(-1, p.big1)
        out = np.empty((x.shape[0], p.k + 1, (p.k + 1) * p.N), dtype=np.uint64)
        for i in range(x.shape[0]):
            for r in range(p.k + 1):
                lib().orc_pfpks(self._h, r, _u64(x[i]), _u64(out[i, r]))
        return out

    def circuit_bootstrap(self, lwe_small: np.ndarray) -> np.ndarray:
        p = self.params
        x = np.ascontiguousarray(lwe_small, dtype=np.uint64).reshape(-1, p.n + 1)
        out = np.empty((x.shape[0], p.cbs_level, p.k + 1, (p.k + 1) * p.N), dtype=np.uint64)
        for i in range(x.shape[0]):
            lib().orc_circuit_bootstrap(self._h, _u64(x[i]), _u64(out[i]))
        return out

    def wopbs_batch(self, lwe_in: np.ndarray, luts: np.ndarray, lut_per_input: bool = False, debug: bool = False):
        """lwe_in [n_inputs][bits][kN+1]; luts [n_sets][n_luts][bits][W] or [n_luts][bits][W], W = max(2^bits, 512)."""
        p = self.params
        x = np.ascontiguousarray(lwe_in, dtype=np.uint64)
        n_inputs, bits = x.shape[0], x.shape[1]
        luts = np.ascontiguousarray(luts, dtype=np.uint64)
        if luts.ndim == 3:
            luts = luts[None]
        n_luts = luts.shape[1]
        assert luts.shape[2] == bits and luts.shape[0] == (n_inputs if lut_per_input else 1) and luts.shape[3] == max(512, 1 << bits)
        out = np.empty((n_inputs, n_luts, bits, p.big1), dtype=np.uint64)
        dbg = [None, None, None]
        ptrs = [None, None, None]
        if debug:
            dbg = [np.empty((n_inputs, bits, p.n + 1), dtype=np.uint64),
                   np.empty((n_inputs, bits, p.big1), dtype=np.uint64),
                   np.empty((n_inputs, bits, p.cbs_level, p.k + 1, (p.k + 1) * p.N), dtype=np.uint64)]
            ptrs = [_u64(a) for a in dbg]
        lib().orc_wopbs_batch(self._h, _u64(x), n_inputs, bits, _u64(luts), n_luts, int(bool(lut_per_input)), _u64(out), *ptrs)
        if debug:
            return out, {"small": dbg[0], "pbs": dbg[1], "ggsw": dbg[2]}
        return out

    def aes_encrypt(self, round_keys: np.ndarray, state: np.ndarray) -> np.ndarray:
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        rk = np.ascontiguousarray(round_keys, dtype=np.uint64)
        lib().orc_aes_encrypt(self._h, _u64(rk), _u64(st))
        return st

    def aes_decrypt(self, round_keys: np.ndarray, state: np.ndarray) -> np.ndarray:
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        rk = np.ascontiguousarray(round_keys, dtype=np.uint64)
        lib().orc_aes_decrypt(self._h, _u64(rk), _u64(st))
        return st

    def aes_key_expansion(self, key: np.ndarray) -> np.ndarray:
        p = self.params
        key = np.ascontiguousarray(key, dtype=np.uint64)
        out = np.empty((11, 16, 8, p.big1), dtype=np.uint64)
        lib().orc_aes_key_expansion(self._h, _u64(key), _u64(out))
        return out

    def add_scalar(self, state: np.ndarray, i: int) -> np.ndarray:
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        lib().orc_add_scalar(self._h, _u64(st), ctypes.c_uint64((i >> 64) & (2 ** 64 - 1)), ctypes.c_uint64(i & (2 ** 64 - 1)))
        return st
