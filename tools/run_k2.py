"""Developer tool: run the blind-rotation stage (K2) alone on M resident bits, for rocprofv3 counter passes.
usage: python3 tools/run_k2.py [M] [launches] [developer build of libfheaes.so to profile instead of the product's]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

if len(sys.argv) > 3:                      # a variant built by tools/ablate_k2.py (gpurun_out/abl/*.so)
    _alt = Path(sys.argv[3]).resolve()
    _native._build.build_engine = lambda *a, **k: _alt
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = PARAM_OPT
c = Client(1, 1, 2, params=p, seed=0xAE50001)
keys = c.server_keys()
E = _native.Engine(p, allow_dev_build=True)
E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
import os  # noqa: E402
if os.environ.get("K2_PARKING") == "private":          # one private parking slot per workgroup instead of the claimed pool
    E.k2_set_parking(False)
rng = np.random.default_rng(0)
small = torch.from_numpy(rng.integers(0, 1 << 64, (M, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
out = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(launches):
    t = time.perf_counter()
    E.cbs_pbs_batch(small, out, M)
    E.synchronize()
    print("K2 M=%d: %.1f ms" % (M, 1e3 * (time.perf_counter() - t)), flush=True)
