"""Developer tool: time the blind-rotation kernel (K2) for builds with phases compiled out.
Builds variants of libfheaes.so into gpurun_out/abl/ and times fheaes_cbs_pbs_batch on M resident bits.
usage: python tools/ablate_k2.py [M] [name,name,... | so:<path of a prebuilt library>]"""
import ctypes
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (HIP runtime first, see _native.load_library)

sys.path.insert(0, "tools")
from gpu_power import Sampler, fmt  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _build, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

VARIANTS = {
    "base": [],
    # ---- round 5: quarter-wise last stages (stores spread over half a pass), z-form decomposition, per-lever proxies ----
    "noxstore": ["-DBRP_ABL_NOXSTORE"], "nodstore": ["-DBRP_ABL_NODSTORE"], "nostores": ["-DBRP_ABL_NOXSTORE", "-DBRP_ABL_NODSTORE"],
    "noxread": ["-DBRP_ABL_NOXREAD"], "noxpose": ["-DBRP_ABL_NOXSTORE", "-DBRP_ABL_NOXREAD"], "nolds_fwd": ["-DBRP_ABL_NOXSTORE", "-DBRP_ABL_NOXREAD", "-DBRP_ABL_NODSTORE"],
    "nobar": ["-DBRP_ABL_NOBAR"], "nobar_skew0": ["-DBRP_ABL_NOBAR", "-DBRP_ABL_SKEW=0"], "nobar_skew40": ["-DBRP_ABL_NOBAR", "-DBRP_ABL_SKEW=40"],
    "nobar_skew80": ["-DBRP_ABL_NOBAR", "-DBRP_ABL_SKEW=80"], "nobar_skew160": ["-DBRP_ABL_NOBAR", "-DBRP_ABL_SKEW=160"],
 "nopeel": ["-DBRP_ABL_NOPEEL"], "noload": ["-DBR16_ABL_NOLOAD"], "nopark": ["-DBR16_ABL_NOPARK"],
    "nostores_noload": ["-DBRP_ABL_NOXSTORE", "-DBRP_ABL_NODSTORE", "-DBR16_ABL_NOLOAD"],
    "all6": ["-DK2_PAIR_TAIL4=0"], "noskip": ["-DBRP_SKIP_IDLE_WAVES=0"], "parkwg": [],     # parkwg: the product build with fheaes_k2_set_parking(ctx, 0) (one private slot per workgroup), see RUNTIME
    "pc_1_2": ["-DBR16_PARK_AUX_ST=1"], "pc_2_2": ["-DBR16_PARK_AUX_ST=2"], "pc_3_2": ["-DBR16_PARK_AUX_ST=3"], "pc_17_2": ["-DBR16_PARK_AUX_ST=17"],
    "pc_0_0": ["-DBR16_PARK_AUX_LD=0"], "pc_0_1": ["-DBR16_PARK_AUX_LD=1"], "pc_0_3": ["-DBR16_PARK_AUX_LD=3"], "pc_0_16": ["-DBR16_PARK_AUX_LD=16"], "pc_0_18": ["-DBR16_PARK_AUX_LD=18"], "park_ld0": ["-DBR16_PARK_AUX_LD=0"],
    "park_ld0_st16": ["-DBR16_PARK_AUX_LD=0", "-DBR16_PARK_AUX_ST=16"], "fewcmul": ["-DBRP_ABL_FEWCMUL"],
    # ---- round 4: parking ----
    "nohome": ["-DBR16_W3_LDS_HOME=0"],                                   # wavefront 3 parks like the others (idle lanes still skip)
    "r3park": ["-DBR16_W3_LDS_HOME=0", "-DBR16_PARK_OWNERS_ONLY=0"],      # round-3 behaviour: every lane parks
    # ---- round 4: the paired form (kern_blindrot_pair.h, default) ----
    "nopair": ["-DK2_PAIR=0"], "pair_norh": ["-DBRP_RESIDENT_HI=0"], "pair_norh_e12": ["-DBRP_RESIDENT_HI=0", "-DBRP_EARLY=12"], "pair_norh_e15": ["-DBRP_RESIDENT_HI=0", "-DBRP_EARLY=15"],
    "pair_e6": ["-DBRP_EARLY=6"], "pair_e12": ["-DBRP_EARLY=12"], "pair_w1": ["-DBRP_W1_LATE=0"], "pair_norh_w1": ["-DBRP_RESIDENT_HI=0", "-DBRP_W1_LATE=0"],
    "nopair_stamps": ["-DK2_PAIR=0", "-DEP_STAMPS"], "pair_w2": ["-DBRP_W1_LATE=2"], "pair_w2_e6": ["-DBRP_W1_LATE=2", "-DBRP_EARLY=6"], "pair_w2_e12": ["-DBRP_W1_LATE=2", "-DBRP_EARLY=12"],
    "pair_w2_stamps": ["-DBRP_W1_LATE=2", "-DEP_STAMPS"], "pair_w2_nopark": ["-DBRP_W1_LATE=2", "-DBR16_ABL_NOPARK"], "pair_w2_noload": ["-DBRP_W1_LATE=2", "-DBR16_ABL_NOLOAD"],
    "pair_e8": ["-DBRP_EARLY=8"], "pair_e10": ["-DBRP_EARLY=10"], "pair_e7_t3": ["-DBRP_EARLY=7", "-DBRP_TAIL=3"],
    "pair_e9_t3": ["-DBRP_TAIL=3"], "pair_e10_t3": ["-DBRP_EARLY=10", "-DBRP_TAIL=3"], "pair_e9_t5": ["-DBRP_TAIL=5"], "pair_e9_t6": ["-DBRP_TAIL=6"], "pair_e8_t5": ["-DBRP_EARLY=8", "-DBRP_TAIL=5"],
    "pair_e10_t5": ["-DBRP_EARLY=10", "-DBRP_TAIL=5"], "pair_e9_t3_pk16": ["-DBRP_TAIL=3", "-DBR16_PARK_AUX_ST=16"], "pair_e9_t3_c2": ["-DBRP_TAIL=3", "-DFFT_CHUNK=2"], "pair_e6_t6": ["-DBRP_EARLY=6", "-DBRP_TAIL=6"],
    "pair_t4_c2": ["-DBRP_TAIL=4", "-DFFT_CHUNK=2"], "pair_t4_c1": ["-DBRP_TAIL=4", "-DFFT_CHUNK=1"], "pair_t4_c8": ["-DBRP_TAIL=4", "-DFFT_CHUNK=8"],
    "pair_e10_t4": ["-DBRP_EARLY=10", "-DBRP_TAIL=4"], "pair_e11_t4": ["-DBRP_EARLY=11", "-DBRP_TAIL=4"], "pair_e8_t4": ["-DBRP_EARLY=8", "-DBRP_TAIL=4"],
    "pair_t4_pk16": ["-DBRP_TAIL=4", "-DBR16_PARK_AUX_ST=16"], "pair_t4_nobar": ["-DBRP_TAIL=4", "-DFFT_CHUNK_BARRIERS=0"],
    "pair_mp0": ["-DBRP_MAC_PRIO=0"], "pair_bc4": ["-DBRP_CHUNK=4"], "pair_bc2": ["-DBRP_CHUNK=2"], "pair_mp0_bc4": ["-DBRP_MAC_PRIO=0", "-DBRP_CHUNK=4"], "pair_mp2": ["-DBRP_MAC_PRIO=2"],
    "pair_xprio1": ["-DFFT_XPOSE_PRIO=1"], "pair_xprio3": ["-DFFT_XPOSE_PRIO=3"], "pair_macprio1": ["-DBRP_MAC_PRIO=1"], "pair_macprio3": ["-DBRP_MAC_PRIO=3"],
    "pair_c2": ["-DFFT_CHUNK=2"], "pair_c1": ["-DFFT_CHUNK=1"],
    "pair_e9_t2": ["-DBRP_TAIL=2"], "pair_e9_t4": ["-DBRP_TAIL=4"], "pair_xprio0": ["-DFFT_XPOSE_PRIO=0"], "pair_pk00": ["-DBR16_PARK_AUX_LD=0"], "pair_pk16_2": ["-DBR16_PARK_AUX_ST=16"],
    "pair_chunk2": ["-DFFT_CHUNK=2"], "pair_rot8": ["-DEP_ROT_CHUNK=8"],
    "pair_nopark": ["-DBR16_ABL_NOPARK"], "pair_noload": ["-DBR16_ABL_NOLOAD"], "pair_stamps": ["-DEP_STAMPS"],
    "rh": ["-DBR16_RESIDENT_HI=1", "-DBR16_EARLY=9"], "rh15": ["-DBR16_RESIDENT_HI=1"], "rh12": ["-DBR16_RESIDENT_HI=1", "-DBR16_EARLY=12"],
    "rh_t0": ["-DBR16_RESIDENT_HI=1", "-DBR16_EARLY=9", "-DBR16_MAC_TAIL=0"], "rh_w1": ["-DBR16_RESIDENT_HI=1", "-DBR16_EARLY=9", "-DBR16_W1_LATE=0"],
    "rh_nohome": ["-DBR16_RESIDENT_HI=1", "-DBR16_EARLY=9", "-DBR16_W3_LDS_HOME=0"],
    "halfkey": ["-DBR16_ABL_HALFKEY"], "halfkey_nopark": ["-DBR16_ABL_HALFKEY", "-DBR16_ABL_NOPARK"],   # 60 % of the key bytes
    "nopark": ["-DBR16_ABL_NOPARK"], "nopark_nohome": ["-DBR16_ABL_NOPARK", "-DBR16_W3_LDS_HOME=0"],
    "nopark_samekey": ["-DBR16_ABL_NOPARK", "-DBR16_ABL_SAMEKEY"], "nopark_noload": ["-DBR16_ABL_NOPARK", "-DBR16_ABL_NOLOAD"],
    "stamps": ["-DEP_STAMPS"],                       # per-phase s_memtime stamps, printed to stderr
    # ---- ablations (wrong results, timing only) ----
    "samekey": ["-DBR16_ABL_SAMEKEY"], "b16_noload": ["-DBR16_ABL_NOLOAD"], "b16_nomac": ["-DBR16_ABL_NOMAC"], "b16_nomac_noload": ["-DBR16_ABL_NOMAC", "-DBR16_ABL_NOLOAD"],
    "b16_nofft": ["-DBR16_ABL_NOFFT"], "b16_noxpose": ["-DBR16_ABL_NOXPOSE"],
    "b16_nofft_nomac_noload": ["-DBR16_ABL_NOFFT", "-DBR16_ABL_NOMAC", "-DBR16_ABL_NOLOAD"],
    # ---- one workgroup per CU (extra LDS) ----
    "b16_one_wg": ["-DBR16_PAD_DOUBLES=2048"], "b16_one_wg_stamps": ["-DBR16_PAD_DOUBLES=2048", "-DEP_STAMPS"],
    # ---- knobs ----
    "oldconv": ["-DFHE_TORUS_CONV_OLD"],
    "nochunk": ["-DFFT_CHUNK_BARRIERS=0"], "chunk1": ["-DFFT_CHUNK=1"], "chunk2": ["-DFFT_CHUNK=2"], "chunk8": ["-DFFT_CHUNK=8"],
    "b16_prio1": ["-DBR16_MAC_PRIO=1"], "b16_xprio0": ["-DFFT_XPOSE_PRIO=0"],
    "rot2": ["-DEP_ROT_CHUNK=2"], "rot8": ["-DEP_ROT_CHUNK=8"], "rot16": ["-DEP_ROT_CHUNK=16"],
    "lat512": ["-DLATENCY_BATCH_BITS=512ull"], "lat768": ["-DLATENCY_BATCH_BITS=768ull"],
    "nobal": ["-DPBS_BALANCE=0"], "nor2": ["-DPBS_SMALL_R2=0"],
    "aux1": ["-DEP_KEY_AUX=1"], "aux2": ["-DEP_KEY_AUX=2"], "aux16": ["-DEP_KEY_AUX=16"], "aux17": ["-DEP_KEY_AUX=17"],
    "fall": ["-DEP_FENCE_MASK=0xFFF"],
    "nostore2": ["-DBR16_STORE_IN_PASS2=0"], "noxtw": ["-DBR16_XPOSE_IN_TWIDDLE=0"], "h2": ["-DBR16_HEAD=2"], "h3": ["-DBR16_HEAD=3"], "h4": ["-DBR16_HEAD=4"], "h5": ["-DBR16_HEAD=5"], "h4_e17": ["-DBR16_HEAD=4", "-DBR16_EARLY=17"], "h4_e19": ["-DBR16_HEAD=4", "-DBR16_EARLY=19"], "h6_e19": ["-DBR16_HEAD=6", "-DBR16_EARLY=19"], "nolate2": ["-DBR16_LATE_IN_PASS2=0"], "late2_e13": ["-DBR16_EARLY=13"], "late2_e10": ["-DBR16_EARLY=10"], "late2_e17": ["-DBR16_EARLY=17"], "nostage_end": ["-DBR16_STAGE_AT_END=0"], "noread2_nostage_end": ["-DBR16_READ_IN_PASS2=0", "-DBR16_STAGE_AT_END=0"], "noread2": ["-DBR16_READ_IN_PASS2=0"],
    "sched_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "sched_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "e0": ["-DBR16_EARLY=0"],
    "e8": ["-DBR16_EARLY=8"],
    "e12": ["-DBR16_EARLY=12"],
    "e13": ["-DBR16_EARLY=13"],
    "e14": ["-DBR16_EARLY=14"],
    "e16": ["-DBR16_EARLY=16"],
    "e17": ["-DBR16_EARLY=17"],
    "e18": ["-DBR16_EARLY=18"],
    "e20": ["-DBR16_EARLY=20"],
    "e25": ["-DBR16_EARLY=25"],
    "pk0_0": ["-DBR16_PARK_AUX_ST=0", "-DBR16_PARK_AUX_LD=0"],
    "pk2_2": ["-DBR16_PARK_AUX_ST=2", "-DBR16_PARK_AUX_LD=2"],
    "pk16_16": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=16"],
    "pk17_17": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=17"],
    "pk18_18": ["-DBR16_PARK_AUX_ST=18", "-DBR16_PARK_AUX_LD=18"],
    "pk19_19": ["-DBR16_PARK_AUX_ST=19", "-DBR16_PARK_AUX_LD=19"],
    "pk2_0": ["-DBR16_PARK_AUX_ST=2", "-DBR16_PARK_AUX_LD=0"],
    "pk16_0": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=0"],
    "pk0_16": ["-DBR16_PARK_AUX_ST=0", "-DBR16_PARK_AUX_LD=16"],
    "pk17_2": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=2"],
    "pk16_2": ["-DBR16_PARK_AUX_ST=16", "-DBR16_PARK_AUX_LD=2"],
    "pk17_0": ["-DBR16_PARK_AUX_ST=17", "-DBR16_PARK_AUX_LD=0"],
    "pk1_1": ["-DBR16_PARK_AUX_ST=1", "-DBR16_PARK_AUX_LD=1"],
    "pk3_3": ["-DBR16_PARK_AUX_ST=3", "-DBR16_PARK_AUX_LD=3"],
}


RUNS = 4
# variants that are a runtime setting of the context, not a build
RUNTIME = {"parkwg": lambda lib, h: lib.fheaes_k2_set_parking(h, 0)}


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
            cmd = [_build.hipcc_path()] + _build.engine_flags() + ["-DFHEAES_DEV_BUILD"] + VARIANTS[name] + ["-o", str(so), str(_build.ENGINE_SOURCES[0])]
            subprocess.run(cmd, check=True, capture_output=True)
        lib = ctypes.CDLL(str(so))
        for fn, (res, args) in _native.SIGNATURES.items():
            if not hasattr(lib, fn):                    # an older library (so:...) may lack entry points added since
                continue
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        h = ctypes.c_void_p()
        cp = p.c_struct()
        assert lib.fheaes_create(ctypes.byref(cp), 0, ctypes.byref(h)) == 0
        assert lib.fheaes_upload_keys(h, keys.ksk.ctypes.data, keys.bsk.ctypes.data, keys.pfpksk.ctypes.data, 0) == 0
        if name in RUNTIME:
            assert RUNTIME[name](lib, h) == 0
        d_in = torch.from_numpy(small.view(np.int64)).cuda()
        d_out = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ts = []
        assert lib.fheaes_cbs_pbs_batch(h, d_in.data_ptr(), M, 1, d_out.data_ptr(), 1) == 0      # warm-up (workspace, clocks)
        lib.fheaes_synchronize(h)
        with Sampler(period=0.01) as smp:                      # socket power / shader clock while the timed launches run
            for _ in range(RUNS):
                t = time.perf_counter()
                assert lib.fheaes_cbs_pbs_batch(h, d_in.data_ptr(), M, 1, d_out.data_ptr(), 1) == 0
                lib.fheaes_synchronize(h)
                ts.append(time.perf_counter() - t)
        import hashlib
        digest = hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest()[:12]      # equal digests = bit-identical outputs (real variants must match base)
        print("%-16s M=%d  %.1f ms  (runs: %s)  out %s  %s" % (name, M, 1e3 * min(ts), " ".join("%.1f" % (1e3 * x) for x in ts), digest, fmt(smp.summary())), flush=True)
        lib.fheaes_destroy(h)


if __name__ == "__main__":
    main()
