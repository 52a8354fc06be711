"""Regenerates tests/golden/golden.json.  Run from the repo root in the build container:

    python tests/golden/make_golden.py

Contents
  aes_kat     : the reference's known-answer inputs (src/main.rs:78-95 = NIST SP 800-38A F.1.1, plus FIPS-197 C.1)
                with the AES-128 answers (computed by tfhe_aes_amd.aes_clear, cross-checked with OpenSSL in SURVEY.md 4)
  sbox        : the two 256-byte tables as hex; when /root/reference is present they are compared with the
                numbers in src/tables/table.rs (data check only, nothing is copied from it)
  oracle_toy  : sha256 of the oracle's outputs on seeded toy-parameter inputs (pins oracle + Client determinism
                across machines; the GPU tests compare the HIP engine against the same hashes)
  oracle_opt  : the same at the reference's real parameter set PARAM_OPT (client.rs:31-57, k = 4): keys, K1 on 35 bits, K2 on 7,
                K3 on 3, K4, one many_sbox byte with its first words, and the exact-arithmetic check of the external product at
                k = 4 (the f64 product's error against an exact product mod 2^64, and the hash of the f64 result) -- pins the
                oracle's k = 4 words across machines and across any later change of the canonical arithmetic
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from conftest import Kit, sha  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tfhe_aes_amd import PARAM_TOY  # noqa: E402
from tfhe_aes_amd.aes_clear import INV_SBOX, SBOX, aes128_encrypt_block  # noqa: E402

KATS = [
    ("2b7e151628aed2a6abf7158809cf4f3c", "6bc1bee22e409f96e93d7e117393172a"),
    ("2b7e151628aed2a6abf7158809cf4f3c", "ae2d8a571e03ac9c9eb76fac45af8e51"),
    ("2b7e151628aed2a6abf7158809cf4f3c", "30c81c46a35ce411e5fbc1191a0a52ef"),
    ("2b7e151628aed2a6abf7158809cf4f3c", "f69f2445df4f9b17ad2b417be66c3710"),
    ("000102030405060708090a0b0c0d0e0f", "00112233445566778899aabbccddeeff"),
]
EXPECTED = ["3ad77bb40d7a3660a89ecaf32466ef97", "f5d3d58503b9699de785895a96fdbaaf", "43b1cd7f598ece23881b00e3ed030688",
            "7b0c785e27e8ad3f8223207104725dd4", "69c4e0d86a7b0430d8cdb78070b4c55a"]


def opt_section(kit=None):
    """seeded intermediates at PARAM_OPT; `kit` = tests/conftest.py's `opt` fixture (seed 0xAE50001) when called from a test"""
    from tfhe_aes_amd import PARAM_OPT

    seed = 0xAE50001
    from tfhe_aes_amd.client import Client

    kit = kit or Kit(PARAM_OPT, seed=seed)
    O, p = kit.oracle, kit.params
    # a FRESH client (same seed, hence the same secret keys): the seeded encryption stream depends on how many encryptions a client has
    # made, and the session-wide `opt` kit of the tests has made some
    c = Client(1, kit.client.iv, kit.client.key, params=p, seed=seed)
    g = {"seed": seed, "params": p.name}
    g["keys"] = {"ksk": sha(kit.keys.ksk), "bsk": sha(kit.keys.bsk), "pfpksk": sha(kit.keys.pfpksk)}
    rng = np.random.default_rng(0x0B7)
    bits = rng.integers(0, 2, 35).astype(np.uint8)
    x = c.encrypt_bits(bits)
    g["input_bits"] = "".join(str(int(b)) for b in bits)
    g["input"] = sha(x)
    small = O.keyswitch(x)                                      # K1, 35 bits
    g["keyswitch_35"] = sha(small)
    pbs = O.cbs_pbs(small[:7])                                  # K2, 7 bits
    g["cbs_pbs_7"] = sha(pbs)
    g["cbs_pbs_first_words"] = [int(v) for v in pbs.reshape(-1)[:4]]
    gg = O.pfpks(pbs[:3])                                       # K3, 3 bits
    g["pfpks_3"] = sha(gg)
    g["ggsw_fourier_3"] = sha(orc.polys_to_fourier(gg.reshape(-1, 512)))    # K4
    xb = c.encrypt_bytes([0x53])
    g["byte_input"] = sha(xb)
    y = O.wopbs_batch(xb, orc.build_lutset(orc.LUTSET_ENC_ROUND))
    g["many_sbox_0x53"] = sha(y)
    g["many_sbox_first_words"] = [int(v) for v in y.reshape(-1)[:4]]
    # the external product at k = 4 against exact arithmetic (tests/test_oracle_primitives.py does this at k = 1)
    k1, level, b = p.k + 1, p.pbs_level, p.pbs_base_log
    rng = np.random.default_rng(0xE4)
    ggsw = rng.integers(0, 1 << 64, (level, k1, k1, 512), dtype=np.uint64)
    d = rng.integers(0, 1 << 64, (k1, 512), dtype=np.uint64)
    acc0 = rng.integers(0, 1 << 64, (k1, 512), dtype=np.uint64)
    got = orc.external_product_add(p, level, b, ggsw, d, acc0)
    want = acc0.copy()
    for r in range(k1):
        digs = np.array([orc.decompose_offset(int(v), b, level) for v in d[r]], dtype=np.int64)
        for l in range(level):
            for cc in range(k1):
                want[cc] += orc.negacyclic_mul_exact(digs[:, l], ggsw[l, r, cc])
    err = (got - want).astype(np.int64)
    g["external_product_k4"] = {"f64_result": sha(got), "exact_result": sha(want), "max_abs_error": int(np.abs(err).max())}
    return g


def main():
    out = {"aes_kat": [], "sbox": bytes(SBOX).hex(), "inv_sbox": bytes(INV_SBOX).hex()}
    for (k, pt), want in zip(KATS, EXPECTED):
        ct = "%032x" % aes128_encrypt_block(int(k, 16), int(pt, 16))
        assert ct == want, (ct, want)
        out["aes_kat"].append({"key": k, "plaintext": pt, "ciphertext": ct})
    ref = Path("/root/reference/src/tables/table.rs")
    if ref.exists():
        nums = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", ref.read_text())]
        assert nums[:256] == list(SBOX) and nums[256:512] == list(INV_SBOX), "tables differ from the reference's"
        out["sbox_checked_against_reference"] = True
    kit = Kit(PARAM_TOY, seed=0x70F)
    c, O, p = kit.client, kit.oracle, kit.params
    g = {"seed": 0x70F, "params": p.name}
    g["keys"] = {"ksk": sha(kit.keys.ksk), "bsk": sha(kit.keys.bsk), "pfpksk": sha(kit.keys.pfpksk)}
    x = c.encrypt_bytes([0x00, 0x53, 0xFF, 0xA7, 0x10]).reshape(-1, p.big1)
    g["input"] = sha(x)
    small = O.keyswitch(x)
    g["keyswitch"] = sha(small)
    pbs = O.cbs_pbs(small)
    g["cbs_pbs"] = sha(pbs)
    gg = O.pfpks(pbs)
    g["pfpks"] = sha(gg)
    g["ggsw_fourier"] = sha(orc.polys_to_fourier(gg.reshape(-1, 512)))
    y = O.wopbs_batch(x.reshape(5, 8, p.big1), orc.build_lutset(orc.LUTSET_ENC_ROUND))
    g["many_sbox"] = sha(y)
    g["many_sbox_first_words"] = [int(v) for v in y.reshape(-1)[:4]]
    st, ek = c.encrypt_u128(c.iv), c.encrypt_u128(c.key)
    g["state_in"], g["key_in"] = sha(st), sha(ek)
    rk = O.aes_key_expansion(ek)
    g["round_keys"] = sha(rk)
    enc = O.aes_encrypt(rk, st)
    g["aes_encrypt"] = sha(enc)
    g["aes_decrypt"] = sha(O.aes_decrypt(rk, enc))
    g["add_scalar_0x1ff"] = sha(O.add_scalar(st, 0x1FF))
    out["oracle_toy"] = g
    out["oracle_opt"] = opt_section()
    (Path(__file__).parent / "golden.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote golden.json")


if __name__ == "__main__":
    main()
