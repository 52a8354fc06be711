"""Developer tool: one-block-round latency for builds with different latency-kernel options."""
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

VARIANTS = {"base": [], "ahead1": ["-DBL_ROWS_AHEAD=1"],  "samekey": ["-DBL_ABL_SAMEKEY"], "nopf": ["-DBL_L2_PREFETCH=0"], "nopf_samekey": ["-DBL_L2_PREFETCH=0", "-DBL_ABL_SAMEKEY"], "pf2": ["-DBL_PREFETCH=2"], "pf3": ["-DBL_PREFETCH=3"], "pf4": ["-DBL_PREFETCH=4"], "off": ["-DLATENCY_BATCH_BITS=0ull"]}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(VARIANTS)
out = Path("gpurun_out/abl"); out.mkdir(parents=True, exist_ok=True)
p = PARAM_OPT
c = Client(1, 1, 2, params=p, seed=0xAE50001)
keys = c.server_keys()
x = c.encrypt_bytes(list(range(16)))
for name in names:
    if name.startswith("so:"):                          # an already built library
        so = Path(name[3:])
    else:
        so = out / ("libfheaes_lat_%s.so" % name)
        subprocess.run([_build.hipcc_path()] + _build.engine_flags() + ["-DFHEAES_DEV_BUILD"] + VARIANTS[name] + ["-o", str(so), str(_build.ENGINE_SOURCES[0])], check=True, capture_output=True)
    lib = ctypes.CDLL(str(so))
    for fn, (res, args) in _native.SIGNATURES.items():
        f = getattr(lib, fn); f.restype, f.argtypes = res, args
    h = ctypes.c_void_p(); cp = p.c_struct()
    assert lib.fheaes_create(ctypes.byref(cp), 0, ctypes.byref(h)) == 0
    assert lib.fheaes_upload_keys(h, keys.ksk.ctypes.data, keys.bsk.ctypes.data, keys.pfpksk.ctypes.data, 0) == 0
    d_in = torch.from_numpy(x.view(np.int64)).cuda()
    d_out = torch.empty((16, 3, 8, p.big1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        assert lib.fheaes_many_sbox(h, d_in.data_ptr(), 16, 0, d_out.data_ptr(), 1) == 0
        lib.fheaes_synchronize(h)
        ts.append(time.perf_counter() - t)
    print("%-6s one block round: %.2f ms" % (name, 1e3 * min(ts)), flush=True)
    lib.fheaes_destroy(h)
