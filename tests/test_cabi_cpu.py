"""The C-ABI library loads without a GPU and exports exactly what include/fheaes.h declares; the product
has no CPU fallback: without a device, creating an engine fails loudly."""
import ctypes

import numpy as np
import pytest

from tfhe_aes_amd import PARAM_OPT, PARAM_TOY, _build, _native
from tfhe_aes_amd.params import CParams


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    declared = _native.header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libfheaes.so does not export %s" % name
    assert sorted(_native.SIGNATURES) == declared          # the binding covers the whole header, nothing else
    assert b"gfx950" in lib.fheaes_version()


def test_library_is_built_in_tree_for_gfx950():
    so = _build.ENGINE_SO
    assert so.exists() and so.parent == _build.PKG
    blob = so.read_bytes()
    assert b"gfx950" in blob                                # embedded code object target
    assert b"extprod_rotate_kernel" in blob and b"keyswitch_mfma" in blob and b"forward_fourier_kernel" in blob


def test_product_does_not_reference_the_oracle():
    for path in list(_build.PKG.rglob("*.py")) + list(_build.CSRC.glob("*")):
        if path.is_file() and path.suffix in (".py", ".hip", ".h", ".c"):
            text = path.read_text()
            assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, path


def test_invalid_parameters_are_errors_not_ub():
    lib = _native.load_library()
    h = ctypes.c_void_p()
    bad = PARAM_OPT.c_struct()
    bad.polynomial_size = 1024
    assert lib.fheaes_create(ctypes.byref(bad), 0, ctypes.byref(h)) == -1
    assert b"polynomial_size" in lib.fheaes_last_error(None)
    bad = PARAM_OPT.c_struct()
    bad.pbs_level = 4
    assert lib.fheaes_create(ctypes.byref(bad), 0, ctypes.byref(h)) == -1
    assert lib.fheaes_create(None, 0, ctypes.byref(h)) == -1
    assert lib.fheaes_key_words(None, 0) == 0
    assert lib.fheaes_synchronize(None) == -1


def test_no_gpu_means_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(_native.FheAesError) as e:
        _native.Engine(PARAM_TOY)
    assert e.value.code == -3                               # FHEAES_ERR_DEVICE


def test_param_struct_layout_matches_header():
    assert ctypes.sizeof(CParams) == 11 * 4
    c = PARAM_OPT.c_struct()
    assert (c.lwe_dimension, c.glwe_dimension, c.polynomial_size) == (669, 4, 512)
    assert (c.pbs_base_log, c.pbs_level, c.ks_base_log, c.ks_level) == (8, 5, 2, 6)
    assert (c.pfks_base_log, c.pfks_level, c.cbs_base_log, c.cbs_level) == (12, 3, 15, 1)
