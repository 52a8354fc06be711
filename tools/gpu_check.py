"""Quick stage-by-stage GPU-vs-oracle check (developer tool; the real tests live in tests/)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402
from tfhe_aes_amd import PARAM_OPT, PARAM_TOY  # noqa: E402
from tfhe_aes_amd import _native  # noqa: E402
from tfhe_aes_amd.client import Client  # noqa: E402


def check(name, a, b):
    same = np.array_equal(a, b)
    msg = "OK" if same else "MISMATCH (%d of %d words differ)" % (int((a != b).sum()), a.size)
    print("  %-28s %s" % (name, msg), flush=True)
    return same


def run(p, m_bits, full_aes):
    print("== %s" % p.name, flush=True)
    t = time.time()
    c = Client(1, 0x6bc1bee22e409f96e93d7e117393172a, 0x2b7e151628aed2a6abf7158809cf4f3c, params=p, seed=0xAE50001)
    keys, st, ek = c.client_encrypt()
    print("  keygen %.1fs" % (time.time() - t), flush=True)
    O = orc.Oracle(p, keys.ksk, keys.bsk, keys.pfpksk)
    E = _native.Engine(p)
    t = time.time()
    E.upload_keys(keys.ksk, keys.bsk, keys.pfpksk)
    print("  upload %.2fs" % (time.time() - t), flush=True)
    ok = True
    ok &= check("twiddles", _native.get_twiddles(), orc.twiddles())
    bf = E.read_bsk_fourier(1)
    ref = orc.polys_to_fourier(keys.bsk.reshape(p.n, p.pbs_level, p.k + 1, p.k + 1, 512)[1])
    ok &= check("bsk fourier[1]", bf.view(np.uint64), ref.view(np.uint64))
    rng = np.random.default_rng(7)
    vals = rng.integers(0, 256, size=(m_bits + 7) // 8)
    x = c.encrypt_bytes(vals).reshape(-1, p.big1)[:m_bits]
    m = x.shape[0]
    # K1
    small = np.empty((m, p.n + 1), dtype=np.uint64)
    E.keyswitch_batch(x, small, m)
    small_ref = O.keyswitch(x)
    ok &= check("K1 keyswitch", small, small_ref)
    # K2
    pbs = np.empty((m, p.big1), dtype=np.uint64)
    t = time.time(); E.cbs_pbs_batch(small_ref, pbs, m); dt = time.time() - t
    pbs_ref = O.cbs_pbs(small_ref)
    ok &= check("K2 cbs pbs (%.3fs)" % dt, pbs, pbs_ref)
    # K3
    g = np.empty((m, p.k + 1, (p.k + 1) * 512), dtype=np.uint64)
    E.pfpks_batch(pbs_ref, g, m)
    g_ref = O.pfpks(pbs_ref)
    ok &= check("K3 pfpks", g, g_ref)
    # K4
    gf = np.empty((m, (p.k + 1) ** 2, 256, 2), dtype=np.float64)
    E.forward_fourier_batch(g_ref, gf, m * (p.k + 1) ** 2)
    gf_ref = orc.polys_to_fourier(g_ref.reshape(m, (p.k + 1) ** 2, 512))
    ok &= check("K4 ggsw fourier", gf.view(np.uint64), gf_ref.view(np.uint64))
    # whole wopbs on bytes
    nb = m // 8
    if nb:
        luts = orc.build_lutset(orc.LUTSET_ENC_ROUND)
        xin = x[: nb * 8].reshape(nb, 8, p.big1)
        out = np.empty((nb, 3, 8, p.big1), dtype=np.uint64)
        t = time.time(); E.wopbs_batch(xin, nb, 8, luts, 3, False, out); dt = time.time() - t
        out_ref = O.wopbs_batch(xin, luts)
        ok &= check("wopbs 3 luts (%.3fs)" % dt, out, out_ref)
    if full_aes:
        rk = np.empty((11, 16, 8, p.big1), dtype=np.uint64)
        t = time.time(); E.aes_key_expansion(ek, rk); dt = time.time() - t
        rk_ref = O.aes_key_expansion(ek)
        ok &= check("key expansion (%.2fs)" % dt, rk, rk_ref)
        s = st.copy()
        t = time.time(); E.aes_encrypt(rk_ref, s, 1); dt = time.time() - t
        s_ref = O.aes_encrypt(rk_ref, st)
        ok &= check("aes_encrypt (%.2fs)" % dt, s, s_ref)
        print("  decrypts to %032x" % c.decrypt_u128(s))
        d = s_ref.copy()
        E.aes_decrypt(rk_ref, d, 1)
        ok &= check("aes_decrypt", d, O.aes_decrypt(rk_ref, s_ref))
        a = st.copy()
        E.add_scalar(a, 1, [0x1ff])
        ok &= check("add_scalar", a, O.add_scalar(st, 0x1ff))
    E.profile_enable(True)
    E.close()
    return ok


if __name__ == "__main__":
    ok = run(PARAM_TOY, 40, True)
    if len(sys.argv) > 1 and sys.argv[1] == "opt":
        ok &= run(PARAM_OPT, 19, False)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
