"""Developer tool: run the packing key switch (K3: digits_kernel + keyswitch_mfma_lds_kernel) alone on M resident bits, for rocprofv3
counter passes.  usage: python3 tools/run_k3.py [M] [launches] [developer build of libfheaes.so to profile instead of the product's]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

if len(sys.argv) > 3:
    _alt = Path(sys.argv[3]).resolve()
    _native._build.build_engine = lambda *a, **k: _alt
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = PARAM_OPT
c = Client(1, 1, 2, params=p, seed=0xAE50001)
keys = c.server_keys()
E = _native.Engine(p, allow_dev_build=True)
E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 1 << 64, (M, p.big1), dtype=np.uint64).view(np.int64)).cuda()      # any LWE words: the kernel is pure integer
out = torch.empty((M, p.k + 1, (p.k + 1) * 512), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(launches):
    t = time.perf_counter()
    E.pfpks_batch(x, out, M)
    E.synchronize()
    print("K3 M=%d: %.2f ms" % (M, 1e3 * (time.perf_counter() - t)), flush=True)
