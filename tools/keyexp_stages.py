"""Developer tool: per-stage times of Server::aes_key_expansion (40 dependent 4-byte WoPBS steps) and of one block round."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

p = PARAM_OPT
KEY = 0x2b7e151628aed2a6abf7158809cf4f3c
c = Client(1, 1, KEY, params=p, seed=0xAE50001)
keys = c.server_keys()
eng = _native.Engine(p, device=0)
eng.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
ek = torch.from_numpy(c.encrypt_u128(KEY).view(np.int64)).cuda()
rk = torch.empty((11, 16, 8, p.big1), dtype=torch.int64, device="cuda")
for rep in range(2):
    eng.profile_reset(); eng.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.aes_key_expansion(ek, rk); eng.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(); eng.profile_enable(False)
print("aes_key_expansion %.1f ms" % (dt * 1e3))
for k, v in prof.items():
    print("  %-18s %s" % (k, v))
