"""Developer tool: time the blind-rotation kernel (K2) for builds with phases compiled out.
Builds variants of libfheaes.so into gpurun_out/abl/ and times fheaes_cbs_pbs_batch on M resident bits.
usage: python tools/ablate_k2.py [M]"""
import ctypes
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (HIP runtime first, see _native.load_library)

from tfhe_aes_amd import PARAM_OPT, _build, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

VARIANTS = {
    "base": [],
    "pk17_2": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=2"],
    "pk16_2": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=2"],
    "pk17_18": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=18"],
    "pk17_0": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=0"],

    "pk2_2": ["-DBR16_PARK_AUX_ST=2", "-DBR16_PARK_AUX_LD=2"],
    "pk16_16": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=16"],
    "pk17_17": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=17"],
    "pk18_18": ["-DBR16_PARK_AUX_ST=18", "-DBR16_PARK_AUX_LD=18"],
    "pk19_19": ["-DBR16_PARK_AUX_ST=19", "-DBR16_PARK_AUX_LD=19"],
    "pk2_0": ["-DBR16_PARK_AUX_ST=2", "-DBR16_PARK_AUX_LD=0"],
    "pk0_2": ["-DBR16_PARK_AUX_ST=0", "-DBR16_PARK_AUX_LD=2"],
    "pk16_0": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=0"],
    "pk0_16": ["-DBR16_PARK_AUX_ST=0", "-DBR16_PARK_AUX_LD=16"],
    "pk18_0": ["-DBR16_PARK_AUX_ST=18", "-DBR16_PARK_AUX_LD=0"],
    "pk3_3": ["-DBR16_PARK_AUX_ST=3", "-DBR16_PARK_AUX_LD=3"],
    "pk1_1": ["-DBR16_PARK_AUX_ST=1", "-DBR16_PARK_AUX_LD=1"],

    "fall": ["-DEP_FENCE_MASK=0xFFF"],
    "f0": ["-DEP_FENCE_MASK=1"],
    "f1": ["-DEP_FENCE_MASK=2"],
    "f2": ["-DEP_FENCE_MASK=4"],
    "f3": ["-DEP_FENCE_MASK=8"],
    "f4": ["-DEP_FENCE_MASK=16"],
    "f5": ["-DEP_FENCE_MASK=32"],
    "f6": ["-DEP_FENCE_MASK=64"],
    "f7": ["-DEP_FENCE_MASK=128"],
    "f8": ["-DEP_FENCE_MASK=256"],
    "f9": ["-DEP_FENCE_MASK=512"],
    "f10": ["-DEP_FENCE_MASK=1024"],
    "f11": ["-DEP_FENCE_MASK=2048"],

    "nochunk": ["-DFFT_CHUNK_BARRIERS=0"],
    "chunk2": ["-DFFT_CHUNK=2"], "chunk8": ["-DFFT_CHUNK=8"], "chunk1nb": ["-DFFT_CHUNK=1", "-DFFT_CHUNK_BARRIERS=0"], "chunk1": ["-DFFT_CHUNK=1"],
    "chunk4_e17": ["-DBR16_EARLY=17"], "chunk2_e17": ["-DFFT_CHUNK=2", "-DBR16_EARLY=17"],
    "notwist": ["-DBR16_ABL_NOTWIST"],
    "oldint": ["-DFHE_PEEL_OLD", "-DFHE_TORUS_CONV_OLD"],
    "oldpeel": ["-DFHE_PEEL_OLD"],
    "oldconv": ["-DFHE_TORUS_CONV_OLD"],
    "e17": ["-DBR16_EARLY=17"],
    "stamps": ["-DEP_STAMPS"],
    "old16": ["-DPBS_FORM16=0"],
    "form32": ["-DPBS_FORM32=1"],
    "b16_prio0": ["-DBR16_MAC_PRIO=0"],
    "e0": ["-DBR16_EARLY=0"], "e4": ["-DBR16_EARLY=4"], "e8": ["-DBR16_EARLY=8"], "e12": ["-DBR16_EARLY=12"], "e14": ["-DBR16_EARLY=14"],
    "e16": ["-DBR16_EARLY=16"], "e20": ["-DBR16_EARLY=20"], "e25": ["-DBR16_EARLY=25"],
    "e12_stamps": ["-DBR16_EARLY=12", "-DEP_STAMPS"], "e12_one_wg": ["-DBR16_EARLY=12", "-DBR16_PAD_DOUBLES=2048"],
    "e13": ["-DBR16_EARLY=13"], "e15": ["-DBR16_EARLY=15"], "e10": ["-DBR16_EARLY=10"], "e18": ["-DBR16_EARLY=18"],
    "e14_stamps": ["-DBR16_EARLY=14", "-DEP_STAMPS"], "e15_stamps": ["-DBR16_EARLY=15", "-DEP_STAMPS"],
    "e15_one_wg_stamps": ["-DBR16_EARLY=15", "-DEP_STAMPS", "-DBR16_PAD_DOUBLES=2048"], "e15_one_wg": ["-DBR16_EARLY=15", "-DBR16_PAD_DOUBLES=2048"],
    "lat512": ["-DLATENCY_BATCH_BITS=512ull"], "lat768": ["-DLATENCY_BATCH_BITS=768ull"], "lat1024": ["-DLATENCY_BATCH_BITS=1024ull"],
    "nobal": ["-DPBS_BALANCE=0"], "nor2": ["-DPBS_SMALL_R2=0"],
    "aux1": ["-DEP_KEY_AUX=1"], "aux2": ["-DEP_KEY_AUX=2"], "aux16": ["-DEP_KEY_AUX=16"], "aux17": ["-DEP_KEY_AUX=17"],
    "aux18": ["-DEP_KEY_AUX=18"], "aux3": ["-DEP_KEY_AUX=3"], "aux19": ["-DEP_KEY_AUX=19"],
    "stag8_1": ["-DBR16_STAGGER_SHIFT=8", "-DBR16_STAGGER_SLEEP=1"],
    "stag8_2": ["-DBR16_STAGGER_SHIFT=8", "-DBR16_STAGGER_SLEEP=2"],
    "stag8_4": ["-DBR16_STAGGER_SHIFT=8", "-DBR16_STAGGER_SLEEP=4"],
    "stag0_1": ["-DBR16_STAGGER_SHIFT=0", "-DBR16_STAGGER_SLEEP=1"],
    "stag3_1": ["-DBR16_STAGGER_SHIFT=3", "-DBR16_STAGGER_SLEEP=1"],
    "stag5_1": ["-DBR16_STAGGER_SHIFT=5", "-DBR16_STAGGER_SLEEP=1"],
    "stag7_1": ["-DBR16_STAGGER_SHIFT=7", "-DBR16_STAGGER_SLEEP=1"],
    "stag9_1": ["-DBR16_STAGGER_SHIFT=9", "-DBR16_STAGGER_SLEEP=1"],
    "b16_one_wg": ["-DBR16_PAD_DOUBLES=2048"],
    "b16_one_wg_stamps": ["-DBR16_PAD_DOUBLES=2048", "-DEP_STAMPS"],
    "b16_noload": ["-DBR16_ABL_NOLOAD"],
    "b16_nomac": ["-DBR16_ABL_NOMAC"],
    "b16_nomac_noload": ["-DBR16_ABL_NOMAC", "-DBR16_ABL_NOLOAD"],
    "b16_nofft": ["-DBR16_ABL_NOFFT"],
    "b16_noxpose": ["-DBR16_ABL_NOXPOSE"],
    "b16_nofft_nomac_noload": ["-DBR16_ABL_NOFFT", "-DBR16_ABL_NOMAC", "-DBR16_ABL_NOLOAD"],
    "b16_xprio0": ["-DFFT_XPOSE_PRIO=0"],
    "b16_prio1": ["-DBR16_MAC_PRIO=1"],
    "rot2": ["-DEP_ROT_CHUNK=2"],
    "rot8": ["-DEP_ROT_CHUNK=8"],
    "rot16": ["-DEP_ROT_CHUNK=16"],
    "one_wg": ["-DBR32_PAD_CPLX=2048"],
    "one_wg_stamps": ["-DBR32_PAD_CPLX=2048", "-DEP_STAMPS"],
    "b32_prio0": ["-DBR32_MAC_PRIO=0"],
    "b32_pf1": ["-DBR32_PREFETCH=1"],
    "b32_pf3": ["-DBR32_PREFETCH=3"],
    "b32_nopark": ["-DBR32_PARK=0"],
    "b32_w3": ["-DBR32_MIN_WAVES=3"],
    "r2": ["-DPBS_R=2"],
    "r2_pf3": ["-DPBS_R=2", "-DEP_PREFETCH=3"],
    "r1": ["-DPBS_R=1"],
    "sched_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "sched_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "early1": ["-DEP_EARLY_LOAD=1"],
    "early2": ["-DEP_EARLY_LOAD=2"],
    "early2_pf3": ["-DEP_EARLY_LOAD=2", "-DEP_PREFETCH=3"],
    "early3_pf3": ["-DEP_EARLY_LOAD=3", "-DEP_PREFETCH=3"],
    "late_barrier": ["-DEP_LATE_BARRIER"],
    "macprio0": ["-DEP_MAC_PRIO=0"],
    "xpose_prio0": ["-DFFT_XPOSE_PRIO=0"],
    "stage_prio": ["-DEP_STAGE_PRIO=2"],
    "macprio2": ["-DEP_MAC_PRIO=2"],
    "old_conv": ["-DEP_OLD_CONV"],
    "pf1": ["-DEP_PREFETCH=1"],
    "pf2": ["-DEP_PREFETCH=2"],
    "pf3": ["-DEP_PREFETCH=3"],
    "mac_noload": ["-DABL_MAC_NOLOAD"],
    "mac_nolds": ["-DABL_MAC_NOLDS"],
    "waves1": ["-DEP_MIN_WAVES=1"],
    "no_mac": ["-DABL_NO_MAC"],
    "no_fft": ["-DABL_NO_FFT"],
    "no_mac_no_fft": ["-DABL_NO_MAC", "-DABL_NO_FFT"],
}


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(VARIANTS)
    out = Path("gpurun_out/abl")
    out.mkdir(parents=True, exist_ok=True)
    p = PARAM_OPT
    c = Client(1, 1, 2, params=p, seed=0xAE50001)
    keys = c.server_keys()
    rng = np.random.default_rng(0)
    small = rng.integers(0, 1 << 64, (M, p.n + 1), dtype=np.uint64)
    for name in names:
        if name.startswith("so:"):                      # an already built library (A/B against an older build on the same box)
            so = Path(name[3:])
        else:
            so = out / ("libfheaes_%s.so" % name)
            cmd = [_build.hipcc_path()] + _build.engine_flags() + VARIANTS[name] + ["-o", str(so), str(_build.ENGINE_SOURCES[0])]
            subprocess.run(cmd, check=True, capture_output=True)
        lib = ctypes.CDLL(str(so))
        for fn, (res, args) in _native.SIGNATURES.items():
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        h = ctypes.c_void_p()
        cp = p.c_struct()
        assert lib.fheaes_create(ctypes.byref(cp), 0, ctypes.byref(h)) == 0
        assert lib.fheaes_upload_keys(h, keys.ksk.ctypes.data, keys.bsk.ctypes.data, keys.pfpksk.ctypes.data, 0) == 0
        d_in = torch.from_numpy(small.view(np.int64)).cuda()
        d_out = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            assert lib.fheaes_cbs_pbs_batch(h, d_in.data_ptr(), M, 1, d_out.data_ptr(), 1) == 0
            lib.fheaes_synchronize(h)
            ts.append(time.perf_counter() - t)
        print("%-16s M=%d  %.1f ms  (runs: %s)" % (name, M, 1e3 * min(ts), " ".join("%.1f" % (1e3 * x) for x in ts)), flush=True)
        lib.fheaes_destroy(h)


if __name__ == "__main__":
    main()
