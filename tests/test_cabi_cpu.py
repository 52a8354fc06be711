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


def test_k2_launch_plan_covers_the_batch_in_whole_generations():
    """host logic of engine.hip::launch_cbs_pbs (no GPU): how a blind-rotation batch is cut into workgroups.  Latency form (0) up to 256
    bits; up to 768 bits one unit of kern_blindrot16.h per CU (form 1: two-ciphertext units up to 2 x CUs bits, then three); beyond
    that the paired form (2, kern_blindrot_pair.h): one 512-thread workgroup per CU, units of six and of four ciphertexts that cover
    the batch in a whole number of generations, the smaller ones last"""
    import ctypes as C

    lib = _native.load_library()
    def plan(m, cus=256, k=4):
        form, um, ut = C.c_int(), C.c_uint64(), C.c_uint64()
        rm, rt = C.c_uint32(), C.c_uint32()
        assert lib.fheaes_k2_launch_plan(m, cus, k, C.byref(form), C.byref(um), C.byref(rm), C.byref(ut), C.byref(rt)) == 0
        return form.value, um.value, rm.value, ut.value, rt.value

    assert plan(1) == (0, 1, 1, 0, 0) and plan(256)[:3] == (0, 256, 1)
    assert plan(257) == (1, 0, 3, 129, 2) and plan(512) == (1, 0, 3, 256, 2)          # one 2-ciphertext unit per CU
    assert plan(513)[:4] == (1, 171, 3, 0) and plan(768)[:4] == (1, 256, 3, 0)         # one 3-ciphertext unit per CU
    assert plan(16384) == (2, 2560, 6, 256, 4)                                         # BASELINE configs[2]: 11 generations of 256
    assert plan(4096) == (2, 512, 6, 256, 4)                                           # a 32-block decrypt shard: 3 generations
    assert plan(1152) == (2, 64, 6, 192, 4)                                            # one add_scalar step of 128 blocks: 1 generation
    assert plan(800) == (2, 0, 6, 200, 4) and plan(2100) == (2, 26, 6, 486, 4) and plan(32768) == (2, 5120, 6, 512, 4)
    for m in list(range(769, 9000, 7)) + [16383, 16385, 30000, 32768]:
        form, um, rm, ut, rt = plan(m)
        assert (form, rm, rt) == (2, 6, 4) and m <= um * 6 + ut * 4 <= m + 3
        if um:
            assert (um + ut) % 256 == 0                                                # whole generations of one workgroup per CU
        else:
            assert ut == (m + 3) // 4          # 4-ciphertext units only: no empty workgroups
    for m in range(257, 769, 5):
        form, um, rm, ut, rt = plan(m)
        assert form == 1 and um * rm + ut * rt >= m and um + ut <= 256
    assert plan(530, k=1)[:4] == (1, 67, 8, 0)                                         # toy parameter set: 8 ciphertexts per unit, no tail
    for bad in ((0, 256), (16, 0)):
        f = C.c_int(); a = C.c_uint64(); b = C.c_uint32(); c = C.c_uint64(); d = C.c_uint32()
        assert lib.fheaes_k2_launch_plan(bad[0], bad[1], 4, C.byref(f), C.byref(a), C.byref(b), C.byref(c), C.byref(d)) != 0


def test_fft_constants_header_matches_the_twiddle_table():
    """csrc/fft_consts.h (generated by tools/gen_fft_consts.py) holds the lane-independent twiddles of the kernels' transform as
    literals: psi^(16 m), m < 32.  They must BE the table both sides derive from the same specification -- the engine's host
    table (fheaes_get_twiddles) and the oracle's -- bit for bit; fheaes_create checks the same at run time."""
    import re

    from oracle import oracle as orc

    text = (_build.CSRC / "fft_consts.h").read_text()
    arrays = {}
    for name, body in re.findall(r"static constexpr double (\w+)\[32\] = \{(.*?)\};", text, flags=re.S):
        arrays[name] = [float.fromhex(x.strip()) for x in body.split(",") if x.strip()]
    assert sorted(arrays) == ["FHE_PSI16_IM", "FHE_PSI16_RE"] and all(len(v) == 32 for v in arrays.values())
    for t in (_native.get_twiddles(), orc.twiddles()):
        assert np.array_equal(np.array(arrays["FHE_PSI16_RE"]).view(np.uint64), np.ascontiguousarray(t[::16, 0][:32]).view(np.uint64))
        assert np.array_equal(np.array(arrays["FHE_PSI16_IM"]).view(np.uint64), np.ascontiguousarray(t[::16, 1][:32]).view(np.uint64))


def test_product_build_takes_no_developer_knobs():
    """csrc/knobs.h: the kernels' tuning knobs and wrong-result ablation switches are an #error on a product build.  The list in
    knobs.h must name every knob the sources test, the #if chain must name every knob of the list, and the product's compile
    flags must set none of them; the product library must not call itself a developer build."""
    import re

    text = (_build.CSRC / "knobs.h").read_text()
    listed = set(re.findall(r"\bX\(([A-Z0-9_]+)\)", text))
    fenced = set(re.findall(r"defined\(([A-Z0-9_]+)\)", text))
    assert listed and listed == fenced, sorted(listed ^ fenced)
    # every `#ifndef KNOB` / `#ifdef KNOB` / `defined(KNOB)` / `#if KNOB` of the sources is a listed knob (FHEAES_* and header guards aside)
    used = set()
    for path in list(_build.CSRC.glob("*.h")) + list(_build.CSRC.glob("*.hip")):
        if path.name == "knobs.h":
            continue
        for line in path.read_text().splitlines():
            m = re.match(r"\s*#\s*(?:ifndef|ifdef)\s+([A-Z][A-Z0-9_]+)", line)
            if m:
                used.add(m.group(1))
            if re.match(r"\s*#\s*(?:if|elif)\b", line):
                used.update(re.findall(r"defined\(([A-Z][A-Z0-9_]+)\)", line))
    used = {u for u in used if not u.startswith("FHEAES_") and not u.endswith("_H")}
    assert used <= listed, "knobs tested by the sources but not fenced in knobs.h: %s" % sorted(used - listed)
    flags = " ".join(_build.engine_flags())
    for knob in listed | {"FHEAES_DEV_BUILD"}:
        assert "-D" + knob not in flags, knob
    version = _native.load_library().fheaes_version().decode()
    assert not version.endswith(" dev"), version


def test_a_stray_knob_does_not_compile():
    """one -D of a knob without -DFHEAES_DEV_BUILD must stop the build (device-side preprocessing only: fast)"""
    import subprocess

    cmd = [_build.hipcc_path(), "--offload-arch=gfx950", "-std=c++17", "-E", "--cuda-device-only", "-DBR16_ABL_NOMAC", "-I", str(_build.ROOT / "include"),
           "-I", str(_build.CSRC), "-o", "/dev/null", str(_build.CSRC / "engine.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode != 0 and "developer knob" in res.stderr
    res = subprocess.run(cmd[:5] + ["-DFHEAES_DEV_BUILD"] + cmd[5:], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-500:]
