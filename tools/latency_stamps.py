"""Developer tool: per-phase cycle stamps of the small-batch blind-rotation kernel (build with -DEP_STAMPS)."""
import ctypes
import subprocess
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401

from tfhe_aes_amd import PARAM_OPT, _build, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
extra = sys.argv[2:]
out = Path("gpurun_out/abl"); out.mkdir(parents=True, exist_ok=True)
so = out / "libfheaes_lat_stamps.so"
subprocess.run([_build.hipcc_path()] + _build.engine_flags() + ["-DFHEAES_DEV_BUILD", "-DEP_STAMPS"] + extra + ["-o", str(so), str(_build.ENGINE_SOURCES[0])], check=True, capture_output=True)
p = PARAM_OPT
c = Client(1, 1, 2, params=p, seed=0xAE50001)
keys = c.server_keys()
lib = ctypes.CDLL(str(so))
for fn, (res, args) in _native.SIGNATURES.items():
    f = getattr(lib, fn); f.restype, f.argtypes = res, args
h = ctypes.c_void_p(); cp = p.c_struct()
assert lib.fheaes_create(ctypes.byref(cp), 0, ctypes.byref(h)) == 0
assert lib.fheaes_upload_keys(h, keys.ksk.ctypes.data, keys.bsk.ctypes.data, keys.pfpksk.ctypes.data, 0) == 0
rng = np.random.default_rng(0)
small = torch.from_numpy(rng.integers(0, 1 << 64, (M, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
o = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert lib.fheaes_cbs_pbs_batch(h, small.data_ptr(), M, 1, o.data_ptr(), 1) == 0
lib.fheaes_synchronize(h)
lib.fheaes_destroy(h)
