"""Developer tool (round 6): hunt for non-deterministic or wrong words in the blind rotation (K2) and in the configs[4] decrypt path.

  python3 tools/k2_stress.py k2 [rounds] [idle_s]     K2 alone on 16,384 random rows: reference = 768-bit cuts (the OTHER kernel, one parking
                                                      slab per workgroup); then `rounds` times a 16,384-bit launch, four 4,096-bit launches and
                                                      1,152-bit launches, each compared row by row; `idle_s` seconds of sleep between rounds
                                                      (clock / power state transitions)
  python3 tools/k2_stress.py dec [encrypt_steps] [decrypts]
                                                      the driver's bench shape: 128 blocks through `encrypt_steps` x aes_encrypt, then `decrypts`
                                                      x aes_decrypt of blocks 0..31 from the same input: sha256 of every output, every block
                                                      decrypted and compared, failing (block, byte, bit) and its phase error printed
An optional last argument `so:<path>` loads another build of libfheaes.so.
"""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tfhe_aes_amd import PARAM_OPT, _native  # noqa: E402
from tfhe_aes_amd.aes_clear import aes128_encrypt_block  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("so:")]
for a in sys.argv[1:]:
    if a.startswith("so:"):
        _alt = Path(a[3:]).resolve()
        _native._build.build_engine = lambda *x, **k: _alt
mode = args[0] if args else "k2"
p = PARAM_OPT
IV = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF
KEY = 0x2B7E151628AED2A6ABF7158809CF4F3C


def sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]


def say(*a):
    print(*a, flush=True)


c = Client(128, IV, KEY, params=p, seed=0xAE50001)
keys = c.server_keys()
E = _native.Engine(p, allow_dev_build=True)
E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
say("library:", (_native.load_library().fheaes_version() or b"").decode())

if mode == "k2":
    rounds = int(args[1]) if len(args) > 1 else 10
    idle_s = float(args[2]) if len(args) > 2 else 0.0
    M = 16384
    rng = np.random.default_rng(7)
    small = torch.from_numpy(rng.integers(0, 1 << 64, (M, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
    ref = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
    out = torch.empty((M, p.big1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for r0 in range(0, M, 768):
        m = min(768, M - r0)
        E.cbs_pbs_batch(small[r0:r0 + m], ref[r0:r0 + m], m)
    E.synchronize()
    say("reference (768-bit cuts, plan %s): %s" % (E.k2_plan(768), sha(ref)))

    def compare(tag, cut):
        bad = (out != ref).any(dim=1).nonzero().flatten().cpu().numpy()
        if len(bad):
            say("  MISMATCH %s: %d rows differ" % (tag, len(bad)))
            for r in bad[:24]:
                d = (out[r] != ref[r]).nonzero().flatten().cpu().numpy()
                pl = E.k2_plan(cut)
                rr = int(r) % cut
                um, rm, rt = pl["units_main"], pl["r_main"], pl["r_tail"]
                unit = rr // rm if rr < um * rm else um + (rr - um * rm) // rt
                say("    row %d (launch %d, unit %d = generation %d cu-rank %d): %d words differ, first %s" % (r, int(r) // cut, unit, unit // 256, unit % 256, len(d), d[:6]))
        return len(bad)

    total_bad = 0
    for rd in range(rounds):
        t0 = time.perf_counter()
        out.zero_()
        torch.cuda.synchronize()
        E.cbs_pbs_batch(small, out, M)
        E.synchronize()
        nb = compare("round %d, 16,384-bit launch" % rd, M)
        for cut in (4096, 1152, 1024):
            out.zero_()
            torch.cuda.synchronize()
            for r0 in range(0, M, cut):
                m = min(cut, M - r0)
                E.cbs_pbs_batch(small[r0:r0 + m], out[r0:r0 + m], m)
            E.synchronize()
            nb += compare("round %d, %d-bit launches" % (rd, cut), cut)
        total_bad += nb
        say("round %d: %d mismatching rows, %.1f s" % (rd, nb, time.perf_counter() - t0))
        if idle_s:
            time.sleep(idle_s)
    say("TOTAL mismatching rows:", total_bad)
    sys.exit(1 if total_bad else 0)

if mode == "dec":
    steps = int(args[1]) if len(args) > 1 else 25
    decrypts = int(args[2]) if len(args) > 2 else 4
    dev = torch.device("cuda", 0)
    n_blocks = 128
    counters = [(IV + i) & ((1 << 128) - 1) for i in range(n_blocks)]
    rk = torch.empty((11, 16, 8, p.big1), dtype=torch.int64, device=dev)
    ek = torch.from_numpy(c.encrypt_u128(KEY).view(np.int64)).to(dev)
    torch.cuda.synchronize()
    E.aes_key_expansion(ek, rk)
    E.synchronize()
    state = torch.from_numpy(np.stack([c.encrypt_u128(v) for v in counters]).view(np.int64)).to(dev)
    E.reserve(n_blocks * 128)
    torch.cuda.synchronize()
    want = list(counters)

    def check(t, want_list, tag):
        bad = []
        host = t.cpu().numpy().view(np.uint64)
        got = c.decrypt_bytes(host)                     # [n][16]
        for i, w in enumerate(want_list):
            wb = [(w >> (8 * (15 - b))) & 0xFF for b in range(16)]
            if [int(x) for x in got[i]] != wb:
                bad.append(i)
        if bad:
            say("  WRONG %s: blocks %s" % (tag, bad))
        return bad

    for s in range(steps):
        E.aes_encrypt(rk, state, n_blocks)
        E.synchronize()
        want = [aes128_encrypt_block(KEY, w) for w in want]
        if s % 5 == 4 or s == steps - 1:
            bad = check(state, want, "after encrypt step %d" % (s + 1))
            say("encrypt step %d: sha %s, %d wrong blocks" % (s + 1, sha(state), len(bad)))
    from tfhe_aes_amd.aes_clear import aes128_decrypt_block  # noqa: E402
    want_dec = [aes128_decrypt_block(KEY, w) for w in want[:32]]
    shas = []
    nbad = 0
    for d in range(decrypts):
        st4 = state[:32].clone()
        torch.cuda.synchronize()
        E.aes_decrypt(rk, st4, 32)
        E.synchronize()
        h = sha(st4)
        shas.append(h)
        bad = check(st4, want_dec, "decrypt %d" % d)
        nbad += len(bad)
        say("decrypt %d: sha %s, %d wrong blocks" % (d, h, len(bad)))
    say("decrypt outputs identical:", len(set(shas)) == 1)
    sys.exit(1 if nbad or len(set(shas)) != 1 else 0)
