"""Developer tool: small-batch latencies (one block round, key expansion, counter add) at PARAM_OPT."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402

from oracle import oracle as orc  # noqa: E402
from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

p = PARAM_OPT
c = Client(1, 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF, 0x2B7E151628AED2A6ABF7158809CF4F3C, params=p, seed=0xAE50001)
keys, st, ek = c.client_encrypt()
E = _native.Engine(p)
E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
x = torch.from_numpy(c.encrypt_bytes(list(range(16))).view(np.int64)).cuda()
out = torch.empty((16, 3, 8, p.big1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); E.many_sbox(x, 16, False, out); E.synchronize(); dt = time.perf_counter() - t
print("one block round (16 many_sbox): %.2f ms" % (1e3 * dt))
O = orc.Oracle(p, keys.ksk, keys.bsk, keys.pfpksk)
want = O.wopbs_batch(x.cpu().numpy().view(np.uint64), orc.build_lutset(orc.LUTSET_ENC_ROUND))
print("  == oracle:", np.array_equal(out.cpu().numpy().view(np.uint64), want))
dk = torch.from_numpy(ek.view(np.int64)).cuda()
rk = torch.empty((11, 16, 8, p.big1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
t = time.perf_counter(); E.aes_key_expansion(dk, rk); E.synchronize(); dt = time.perf_counter() - t
print("key expansion: %.1f ms" % (1e3 * dt))
from tfhe_aes_amd.aes_clear import expand_key  # noqa: E402
print("  decrypts to the AES round keys:", np.array_equal(c.decrypt_bytes(rk.cpu().numpy().view(np.uint64)), np.array(expand_key(c.key), dtype=np.uint8)))
s8 = torch.from_numpy(np.stack([st] * 8).view(np.int64)).cuda()
torch.cuda.synchronize()
t = time.perf_counter(); E.add_scalar(s8, 8, list(range(1, 9))); E.synchronize(); dt = time.perf_counter() - t
print("add_scalar on 8 blocks: %.1f ms" % (1e3 * dt))
print("  ok:", all(c.decrypt_u128(s8[i].cpu().numpy().view(np.uint64)) == c.iv + i + 1 for i in range(8)))
