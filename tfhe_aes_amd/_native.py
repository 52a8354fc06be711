"""ctypes binding of the C ABI in include/fheaes.h (libfheaes.so).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, creating an
``Engine`` raises.  ``load_library()`` alone (symbol checks) works without a GPU.
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

import numpy as np

from . import _build
from .params import CParams, WopbsParameters

HOST, DEVICE = 0, 1
STAGES = ("keyswitch", "blind_rotate", "pfpks", "ggsw_fft", "vertical_packing", "linear")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_dp = ctypes.POINTER(ctypes.c_double)
_ctx = ctypes.c_void_p
_c = ctypes

# name -> (restype, argtypes); every function declared in include/fheaes.h
SIGNATURES = {
    "fheaes_create": (_c.c_int, [_c.POINTER(CParams), _c.c_int, _c.POINTER(_ctx)]),
    "fheaes_destroy": (None, [_ctx]),
    "fheaes_last_error": (_c.c_char_p, [_ctx]),
    "fheaes_key_words": (_c.c_size_t, [_ctx, _c.c_int]),
    "fheaes_upload_keys": (_c.c_int, [_ctx, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int]),
    "fheaes_key_body_words": (_c.c_size_t, [_ctx, _c.c_int]),
    "fheaes_upload_keys_seeded": (_c.c_int, [_ctx, _c.POINTER(_c.c_uint32), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int]),
    "fheaes_clone_keys": (_c.c_int, [_ctx, _ctx]),
    "fheaes_clone_info": (_c.c_int, [_ctx, _c.POINTER(_c.c_int), _u64p, _dp]),
    "fheaes_noise_level_seen": (_c.c_int, [_ctx, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint32)]),
    "fheaes_set_stream": (_c.c_int, [_ctx, _c.c_void_p]),
    "fheaes_synchronize": (_c.c_int, [_ctx]),
    "fheaes_reserve": (_c.c_int, [_ctx, _c.c_uint64]),
    "fheaes_keyswitch_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_int]),
    "fheaes_cbs_pbs_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_uint32, _c.c_void_p, _c.c_int]),
    "fheaes_pfpks_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_int]),
    "fheaes_forward_fourier_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_int]),
    "fheaes_vertical_packing_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                                 _c.c_int, _c.c_void_p, _c.c_int]),
    "fheaes_wopbs_batch": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_uint32, _c.c_void_p, _c.c_uint32, _c.c_int,
                                      _c.c_void_p, _c.c_int]),
    "fheaes_gen_lut": (_c.c_int, [_c.c_uint32, _u64p, _u64p]),
    "fheaes_sbox": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_int]),
    "fheaes_many_sbox": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p, _c.c_int]),
    "fheaes_aes_key_expansion": (_c.c_int, [_ctx, _c.c_void_p, _c.c_void_p, _c.c_int]),
    "fheaes_aes_encrypt": (_c.c_int, [_ctx, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_int]),
    "fheaes_aes_decrypt": (_c.c_int, [_ctx, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_int]),
    "fheaes_add_scalar": (_c.c_int, [_ctx, _c.c_void_p, _c.c_uint64, _u64p, _c.c_int]),
    "fheaes_profile_enable": (_c.c_int, [_ctx, _c.c_int]),
    "fheaes_profile_reset": (_c.c_int, [_ctx]),
    "fheaes_profile_read": (_c.c_int, [_ctx, _c.c_int, _dp, _u64p, _u64p]),
    "fheaes_get_twiddles": (_c.c_int, [_dp]),
    "fheaes_read_bsk_fourier": (_c.c_int, [_ctx, _c.c_uint32, _dp]),
    "fheaes_k2_launch_plan": (_c.c_int, [_c.c_uint64, _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32),
                                       _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32)]),
    "fheaes_k2_context_plan": (_c.c_int, [_ctx, _c.c_uint64, _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32),
                                        _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32), _c.c_char_p, _c.c_size_t]),
    "fheaes_k2_set_parking": (_c.c_int, [_ctx, _c.c_int]),
    "fheaes_version": (_c.c_char_p, []),
}

_lib = None


def header_symbols() -> list[str]:
    """every function name declared in include/fheaes.h"""
    text = (_build.ROOT / "include" / "fheaes.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fheaes_[a-z0-9_]+)\s*\(", text)))


def load_library(build: bool = True):
    """dlopen libfheaes.so (building it in-tree first if needed) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build_engine() if build else _build.ENGINE_SO
    try:
        # PyTorch-ROCm ships its own libamdhip64; two HIP runtimes in one process cannot both own the
        # GPU ("No HIP GPUs are available" in whichever comes second).  Loading torch first makes
        # libfheaes.so resolve its libamdhip64.so.7 dependency to the runtime torch already mapped.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not Path(path).exists():
        raise RuntimeError("libfheaes.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class FheAesError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("fheaes error %d: %s" % (code, message))
        self.code = code


def _ptr(x):
    """numpy array (host), torch tensor (host or cuda) or int -> (void pointer value, memspace or None)"""
    if x is None:
        return None, None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return x.ctypes.data, HOST
    if hasattr(x, "data_ptr"):  # torch tensor
        assert x.is_contiguous()
        return x.data_ptr(), (DEVICE if x.is_cuda else HOST)
    raise TypeError("expected numpy array or torch tensor, got %r" % type(x))


class Engine:
    """Owns one ``fheaes_ctx``.  Thin: argument marshalling and error translation only."""

    def __init__(self, params: WopbsParameters, device: int = 0, allow_dev_build: bool = False):
        self.params = params
        self._lib = load_library()
        self._h = _ctx()
        # a library built with developer knobs (csrc/knobs.h: possibly a wrong-result ablation) says so in its version string
        version = (self._lib.fheaes_version() or b"").decode()
        if version.endswith(" dev") and not (allow_dev_build or os.environ.get("FHEAES_ALLOW_DEV_BUILD") == "1"):
            raise RuntimeError("libfheaes.so is a developer build (%r): rebuild it with tfhe_aes_amd._build.build_engine(force=True), or pass "
                               "allow_dev_build=True / FHEAES_ALLOW_DEV_BUILD=1 if that is what you mean to run" % version)
        cp = params.c_struct()
        rc = self._lib.fheaes_create(ctypes.byref(cp), device, ctypes.byref(self._h))
        if rc != 0:
            raise FheAesError(rc, (self._lib.fheaes_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fheaes_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise FheAesError(rc, (self._lib.fheaes_last_error(self._h) or b"").decode())

    def _space(self, *arrays):
        spaces = {s for _, s in map(_ptr, arrays) if s is not None}
        if len(spaces) != 1:
            raise ValueError("all arrays of one call must live in the same memory space")
        return spaces.pop()

    # -- keys / control ---------------------------------------------------------
    def key_words(self, which: int) -> int:
        return self._lib.fheaes_key_words(self._h, which)

    def upload_keys(self, ksk, bsk, pfpksk):
        sp = self._space(ksk, bsk, pfpksk)
        for which, a in enumerate((ksk, bsk, pfpksk)):
            n = a.size if isinstance(a, np.ndarray) else a.numel()
            if n != self.key_words(which):
                raise ValueError("key %d has %d words, expected %d" % (which, n, self.key_words(which)))
        self._check(self._lib.fheaes_upload_keys(self._h, _ptr(ksk)[0], _ptr(bsk)[0], _ptr(pfpksk)[0], sp))

    def upload_keys_seeded(self, mask_key, ksk_body, bsk_body, pfpksk_body):
        """keys as (public 256-bit mask key = 8 uint32, bodies): masks are regenerated on the GPU (include/fheaes.h)"""
        mk = np.ascontiguousarray(np.asarray(mask_key, dtype=np.uint32).reshape(8))
        sp = self._space(ksk_body, bsk_body, pfpksk_body)
        for which, a in enumerate((ksk_body, bsk_body, pfpksk_body)):
            n = a.size if isinstance(a, np.ndarray) else a.numel()
            if n != self._lib.fheaes_key_body_words(self._h, which):
                raise ValueError("key body %d has %d words, expected %d" % (which, n, self._lib.fheaes_key_body_words(self._h, which)))
        self._check(self._lib.fheaes_upload_keys_seeded(self._h, mk.ctypes.data_as(_c.POINTER(_c.c_uint32)), _ptr(ksk_body)[0], _ptr(bsk_body)[0],
                                                        _ptr(pfpksk_body)[0], sp))

    def clone_keys_from(self, other: "Engine"):
        """device-to-device copy of `other`'s converted key images into this context (fheaes_clone_keys): one PCIe upload serves
        several contexts -- on other GPUs (over xGMI) or on the same one (independent streams and workspaces)"""
        self._check(self._lib.fheaes_clone_keys(self._h, other._h))

    def clone_info(self) -> dict:
        """how the last clone_keys_from() into this context moved the keys: {"path": "none" | "same_device" | "peer" | "staged", "bytes", "seconds"}"""
        path, nbytes, secs = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self._lib.fheaes_clone_info(self._h, ctypes.byref(path), ctypes.byref(nbytes), ctypes.byref(secs)))
        return {"path": ("none", "same_device", "peer", "staged")[path.value], "bytes": nbytes.value, "seconds": secs.value}

    def noise_level_seen(self) -> tuple[int, int]:
        """(highest number of nominal-noise ciphertexts any linear layer of this context has summed, the parameter set's limit)"""
        seen, limit = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._lib.fheaes_noise_level_seen(self._h, ctypes.byref(seen), ctypes.byref(limit)))
        return seen.value, limit.value

    def set_stream(self, stream_handle: int | None):
        self._check(self._lib.fheaes_set_stream(self._h, stream_handle))

    def synchronize(self):
        self._check(self._lib.fheaes_synchronize(self._h))

    def reserve(self, max_bits: int):
        self._check(self._lib.fheaes_reserve(self._h, max_bits))

    # -- stages -----------------------------------------------------------------
    def keyswitch_batch(self, lwe_in, lwe_out, m: int):
        self._check(self._lib.fheaes_keyswitch_batch(self._h, _ptr(lwe_in)[0], m, _ptr(lwe_out)[0], self._space(lwe_in, lwe_out)))

    def cbs_pbs_batch(self, lwe_small, lwe_out, m: int, level: int = 1):
        self._check(self._lib.fheaes_cbs_pbs_batch(self._h, _ptr(lwe_small)[0], m, level, _ptr(lwe_out)[0], self._space(lwe_small, lwe_out)))

    def pfpks_batch(self, lwe_in, ggsw_out, m: int):
        self._check(self._lib.fheaes_pfpks_batch(self._h, _ptr(lwe_in)[0], m, _ptr(ggsw_out)[0], self._space(lwe_in, ggsw_out)))

    def forward_fourier_batch(self, polys_in, fourier_out, polys: int):
        self._check(self._lib.fheaes_forward_fourier_batch(self._h, _ptr(polys_in)[0], polys, _ptr(fourier_out)[0], self._space(polys_in, fourier_out)))

    def vertical_packing_batch(self, ggsw_fourier, n_inputs, bits, luts, n_luts, lut_per_input, lwe_out):
        self._check(self._lib.fheaes_vertical_packing_batch(self._h, _ptr(ggsw_fourier)[0], n_inputs, bits, _ptr(luts)[0], n_luts,
                                                            int(bool(lut_per_input)), _ptr(lwe_out)[0], self._space(ggsw_fourier, luts, lwe_out)))

    def wopbs_batch(self, lwe_in, n_inputs, bits, luts, n_luts, lut_per_input, lwe_out):
        self._check(self._lib.fheaes_wopbs_batch(self._h, _ptr(lwe_in)[0], n_inputs, bits, _ptr(luts)[0], n_luts, int(bool(lut_per_input)),
                                                 _ptr(lwe_out)[0], self._space(lwe_in, luts, lwe_out)))

    def sbox(self, bytes_ct, n_bytes: int, inv: bool):
        self._check(self._lib.fheaes_sbox(self._h, _ptr(bytes_ct)[0], n_bytes, int(inv), self._space(bytes_ct)))

    def many_sbox(self, bytes_ct, n_bytes: int, inv: bool, out):
        self._check(self._lib.fheaes_many_sbox(self._h, _ptr(bytes_ct)[0], n_bytes, int(inv), _ptr(out)[0], self._space(bytes_ct, out)))

    # -- Server API -------------------------------------------------------------
    def aes_key_expansion(self, key, round_keys):
        self._check(self._lib.fheaes_aes_key_expansion(self._h, _ptr(key)[0], _ptr(round_keys)[0], self._space(key, round_keys)))

    def aes_encrypt(self, round_keys, state, n_blocks: int):
        self._check(self._lib.fheaes_aes_encrypt(self._h, _ptr(round_keys)[0], _ptr(state)[0], n_blocks, self._space(round_keys, state)))

    def aes_decrypt(self, round_keys, state, n_blocks: int):
        self._check(self._lib.fheaes_aes_decrypt(self._h, _ptr(round_keys)[0], _ptr(state)[0], n_blocks, self._space(round_keys, state)))

    def add_scalar(self, state, n_blocks: int, counters):
        cnt = np.zeros((n_blocks, 2), dtype=np.uint64)
        for i, v in enumerate(counters):
            cnt[i, 0] = (int(v) >> 64) & (2 ** 64 - 1)
            cnt[i, 1] = int(v) & (2 ** 64 - 1)
        self._check(self._lib.fheaes_add_scalar(self._h, _ptr(state)[0], n_blocks, cnt.ctypes.data_as(_u64p), self._space(state)))

    # -- measurement ------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self._lib.fheaes_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._lib.fheaes_profile_reset(self._h))

    def profile_read(self) -> dict:
        out = {}
        for i, name in enumerate(STAGES):
            ms, launches, units = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
            self._check(self._lib.fheaes_profile_read(self._h, i, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(units)))
            out[name] = {"ms": ms.value, "launches": launches.value, "units": units.value}
        return out

    def k2_plan(self, bits: int) -> dict:
        """the blind-rotation kernel this context really launches for a batch of `bits` (after the occupancy fallbacks) and its cut"""
        form, um, rm, ut, rt = _c.c_int(), _c.c_uint64(), _c.c_uint32(), _c.c_uint64(), _c.c_uint32()
        name = _c.create_string_buffer(96)
        self._check(self._lib.fheaes_k2_context_plan(self._h, bits, _c.byref(form), _c.byref(um), _c.byref(rm), _c.byref(ut), _c.byref(rt), name, len(name)))
        return {"form": form.value, "kernel": name.value.decode(), "units_main": um.value, "r_main": rm.value,
                "units_tail": ut.value, "r_tail": rt.value}

    def k2_set_parking(self, claimed: bool):
        """paired blind-rotation kernel: parking slots claimed from a shared pool (default) or one private slot per workgroup"""
        self._check(self._lib.fheaes_k2_set_parking(self._h, 1 if claimed else 0))

    def read_bsk_fourier(self, i: int) -> np.ndarray:
        p = self.params
        out = np.empty((p.pbs_level, p.k + 1, p.k + 1, 256, 2), dtype=np.float64)
        self._check(self._lib.fheaes_read_bsk_fourier(self._h, i, out.ctypes.data_as(_dp)))
        return out


def get_twiddles() -> np.ndarray:
    out = np.empty((512, 2), dtype=np.float64)
    rc = load_library().fheaes_get_twiddles(out.ctypes.data_as(_dp))
    if rc != 0:
        raise FheAesError(rc, "fheaes_get_twiddles")
    return out


def gen_lut(nb_block: int, f_table) -> np.ndarray:
    f = np.ascontiguousarray(f_table, dtype=np.uint64)
    if f.size != 1 << nb_block:
        raise ValueError("f_table must have 2^nb_block entries")
    out = np.empty((nb_block, max(512, 1 << nb_block)), dtype=np.uint64)
    rc = load_library().fheaes_gen_lut(nb_block, f.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p))
    if rc != 0:
        raise FheAesError(rc, "fheaes_gen_lut")
    return out
