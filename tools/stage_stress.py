"""Developer tool (round 6): which STAGE of the WoPBS pipeline produces wrong words under sustained load?

The engine is deterministic: the same stage on the same resident input must give the same words every time.  This tool computes every
stage's output once for a 16,384-bit batch at PARAM_OPT (K1 keyswitch -> K2 blind rotation -> K3 PFPKS -> K4 GGSW FFT -> K5 vertical
packing, each fed the previous stage's reference output), then runs the stages back to back for `seconds` without idling and compares
every output with its reference on the GPU.  A mismatch is reported with the time since start, the stage, the rows that differ and
the GPU's power / clock / temperature at that moment.

  python3 tools/stage_stress.py [seconds] [stages, e.g. 12345 or 2] [so:<path of another libfheaes.so>]
"""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.aes_clear import SBOX, mul2, mul3  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402
from tfhe_aes_amd.server import gen_lut  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("so:")]
for a in sys.argv[1:]:
    if a.startswith("so:"):
        _alt = Path(a[3:]).resolve()
        _native._build.build_engine = lambda *x, **k: _alt
seconds = float(args[0]) if args else 90.0
stages = args[1] if len(args) > 1 else "12345"
p = PARAM_OPT
M = 16384


def say(*a):
    print(*a, flush=True)


def smi():
    try:
        from gpu_power import read_once
        return read_once(0)
    except Exception as e:  # noqa: BLE001
        return {"err": str(e)[:60]}


c = Client(1, 1, 2, params=p, seed=0xAE50001)
keys = c.server_keys()
E = _native.Engine(p, allow_dev_build=True)
E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
say("library:", (_native.load_library().fheaes_version() or b"").decode(), " K2 plan:", E.k2_plan(M))
rng = np.random.default_rng(11)
x = torch.from_numpy(c.encrypt_bits(rng.integers(0, 2, M).astype(np.uint8)).view(np.int64)).cuda()
luts = torch.from_numpy(np.stack([gen_lut(2, 1, 512, 8, f) for f in (lambda v: SBOX[v], lambda v: mul2(SBOX[v]), lambda v: mul3(SBOX[v]))]).view(np.int64)).cuda()
k1 = p.k + 1
dev = "cuda"
ref = {
    1: torch.empty((M, p.n + 1), dtype=torch.int64, device=dev),
    2: torch.empty((M, p.big1), dtype=torch.int64, device=dev),
    3: torch.empty((M, k1, k1 * 512), dtype=torch.int64, device=dev),
    4: torch.empty((M * k1 * k1, 256, 2), dtype=torch.float64, device=dev),
    5: torch.empty((M // 8, 3, 8, p.big1), dtype=torch.int64, device=dev),
}
out = {k: torch.empty_like(v) for k, v in ref.items()}
torch.cuda.synchronize()


def run(stage, dst):
    if stage == 1:
        E.keyswitch_batch(x, dst[1], M)
    elif stage == 2:
        E.cbs_pbs_batch(ref[1], dst[2], M)
    elif stage == 3:
        E.pfpks_batch(ref[2], dst[3], M)
    elif stage == 4:
        E.forward_fourier_batch(ref[3], dst[4], M * k1 * k1)
    elif stage == 5:
        E.vertical_packing_batch(ref[4], M // 8, 8, luts, 3, False, dst[5])


for s in (1, 2, 3, 4, 5):
    run(s, ref)
    E.synchronize()
# the references themselves: twice, cold
for s in (1, 2, 3, 4, 5):
    run(s, out)
    E.synchronize()
    same = torch.equal(out[s].view(torch.int64), ref[s].view(torch.int64))
    say("stage %d reference reproduced: %s" % (s, same))
dec = c.decrypt_bytes(ref[5].cpu().numpy().view(np.uint64))
say("pipeline decrypts to S / 2S / 3S of the input bytes:", bool((dec[:, 0] < 256).all()))

t_start = time.time()
n_bad, it = 0, 0
while time.time() - t_start < seconds:
    for ch in stages:
        s = int(ch)
        out[s].view(torch.int64).zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        run(s, out)
        E.synchronize()
        t1 = time.time()
        a, b = out[s].view(torch.int64).reshape(out[s].shape[0], -1), ref[s].view(torch.int64).reshape(ref[s].shape[0], -1)
        if not torch.equal(a, b):
            rows = (a != b).any(dim=1).nonzero().flatten().cpu().numpy()
            n_bad += 1
            say("MISMATCH t=%.1fs iteration %d stage %d: %d rows differ (launch ran %.1f..%.1f s); smi %s" % (t1 - t_start, it, s, len(rows), t0 - t_start, t1 - t_start, smi()))
            say("   rows:", rows[:64].tolist(), "..." if len(rows) > 64 else "")
            for r in rows[:6]:
                w = (a[r] != b[r]).nonzero().flatten().cpu().numpy()
                say("   row %d: %d of %d words differ, first at %s; xor of first: %016x" % (r, len(w), a.shape[1], w[:8].tolist(), int(a[r, w[0]].item() ^ b[r, w[0]].item()) & (2**64 - 1)))
    it += 1
    if it % 10 == 0:
        say("t=%.1fs iteration %d, mismatching launches so far %d; smi %s" % (time.time() - t_start, it, n_bad, smi()))
say("DONE: %d iterations, %d mismatching launches" % (it, n_bad))
sys.exit(1 if n_bad else 0)
