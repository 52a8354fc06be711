"""Developer tool: time the packing key switch (K3, fheaes_pfpks_batch) and the bit-extraction key switch (K1) for builds with
different -D knobs, A/B in one process, and check that every build gives the same words as the first one.
usage: python tools/ablate_k3.py [M] [name,name,...]"""
import ctypes
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401

from tfhe_aes_amd import PARAM_OPT, _build, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

# round 3 tried (all same words, all 15.2-15.6 ms per 16,384-bit launch = no change): every fragment read of a K step issued ahead of
# the matrix instructions (one LDS wait per step instead of four), reads two fragments ahead, wave priority during the products.
VARIANTS = {
    "base": [],
    "nolds": ["-DKS_LDS=0"],          # the one-wave-one-tile form (operands straight from L2)
    "k1lds": ["-DKS1_LDS=1"],         # round 6: K1 through the LDS-tiled kernel
    "ct4_k1lds": ["-DKSL_CT_TILES=4", "-DKS1_LDS=1"],
    "ct4": ["-DKSL_CT_TILES=4"],      # round 6: 256-thread workgroups of 64 ciphertexts, two per CU (no common barrier between the two)
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
    x = rng.integers(0, 1 << 64, (M, p.big1), dtype=np.uint64)
    ref = None
    for name in names:
        if name.startswith("so:"):                      # an already built library
            so = Path(name[3:])
        else:
            so = out / ("libfheaes_k3_%s.so" % name)
            cmd = [_build.hipcc_path()] + _build.engine_flags("keyswitch") + ["-DFHEAES_DEV_BUILD"] + VARIANTS[name] + ["-o", str(so), str(_build.ENGINE_SOURCES[0])]
            subprocess.run(cmd, check=True, capture_output=True)
        lib = ctypes.CDLL(str(so))
        for fn, (res, args) in _native.SIGNATURES.items():
            if not hasattr(lib, fn):
                continue
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        h = ctypes.c_void_p()
        cp = p.c_struct()
        assert lib.fheaes_create(ctypes.byref(cp), 0, ctypes.byref(h)) == 0
        assert lib.fheaes_upload_keys(h, keys.ksk.ctypes.data, keys.bsk.ctypes.data, keys.pfpksk.ctypes.data, 0) == 0
        d_in = torch.from_numpy(x.view(np.int64)).cuda()
        d_out = torch.empty((M, p.k + 1, (p.k + 1) * 512), dtype=torch.int64, device="cuda")
        d_small = torch.empty((M, p.n + 1), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ts, t1 = [], []
        for _ in range(4):
            t = time.perf_counter()
            assert lib.fheaes_pfpks_batch(h, d_in.data_ptr(), M, d_out.data_ptr(), 1) == 0
            lib.fheaes_synchronize(h)
            ts.append(time.perf_counter() - t)
            t = time.perf_counter()
            assert lib.fheaes_keyswitch_batch(h, d_in.data_ptr(), M, d_small.data_ptr(), 1) == 0
            lib.fheaes_synchronize(h)
            t1.append(time.perf_counter() - t)
        got = (d_out[:: max(1, M // 64)].cpu().numpy().copy(), d_small[:: max(1, M // 64)].cpu().numpy().copy())
        same = "reference" if ref is None else ("same words" if all(np.array_equal(a, b) for a, b in zip(got, ref)) else "DIFFERENT WORDS")
        if ref is None:
            ref = got
        print("%-10s M=%d  K3 %.2f ms  K1 %.2f ms  (K3 runs: %s)  %s" % (name, M, 1e3 * min(ts), 1e3 * min(t1), " ".join("%.2f" % (1e3 * v) for v in ts), same), flush=True)
        lib.fheaes_destroy(h)


if __name__ == "__main__":
    main()
