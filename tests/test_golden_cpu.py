"""The oracle (and the Client's seeded key generation) reproduce the committed golden hashes."""
import numpy as np

from conftest import Kit, sha
from oracle import oracle as orc
from tfhe_aes_amd import PARAM_TOY


def test_oracle_reproduces_golden(golden):
    g = golden["oracle_toy"]
    kit = Kit(PARAM_TOY, seed=g["seed"])
    c, O, p = kit.client, kit.oracle, kit.params
    assert {"ksk": sha(kit.keys.ksk), "bsk": sha(kit.keys.bsk), "pfpksk": sha(kit.keys.pfpksk)} == g["keys"]
    x = c.encrypt_bytes([0x00, 0x53, 0xFF, 0xA7, 0x10]).reshape(-1, p.big1)
    assert sha(x) == g["input"]
    small = O.keyswitch(x)
    assert sha(small) == g["keyswitch"]
    pbs = O.cbs_pbs(small)
    assert sha(pbs) == g["cbs_pbs"]
    gg = O.pfpks(pbs)
    assert sha(gg) == g["pfpks"]
    assert sha(orc.polys_to_fourier(gg.reshape(-1, 512))) == g["ggsw_fourier"]
    y = O.wopbs_batch(x.reshape(5, 8, p.big1), orc.build_lutset(orc.LUTSET_ENC_ROUND))
    assert sha(y) == g["many_sbox"]
    assert [int(v) for v in y.reshape(-1)[:4]] == g["many_sbox_first_words"]
    st, ek = c.encrypt_u128(c.iv), c.encrypt_u128(c.key)
    assert sha(st) == g["state_in"] and sha(ek) == g["key_in"]
    rk = O.aes_key_expansion(ek)
    assert sha(rk) == g["round_keys"]
    enc = O.aes_encrypt(rk, st)
    assert sha(enc) == g["aes_encrypt"]
    assert sha(O.aes_decrypt(rk, enc)) == g["aes_decrypt"]
    assert sha(O.add_scalar(st, 0x1FF)) == g["add_scalar_0x1ff"]


def test_oracle_reproduces_golden_at_param_opt(golden, opt):
    """the PARAM_OPT section (client.rs:31-57, k = 4): seeded keys, K1 on 35 bits, K2 on 7, K3 on 3, K4, one many_sbox byte and the
    external product against exact arithmetic -- the oracle's k = 4 words are the same on every machine and after every change"""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("make_golden", Path(__file__).parent / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = golden["oracle_opt"]
    got = mg.opt_section(opt)
    assert got == g
    # the f64 product is an APPROXIMATE product of the same regime as tfhe-rs' fft64: error far below the gadget's payloads, not zero
    assert 0 < g["external_product_k4"]["max_abs_error"] < 2 ** 30
    assert g["external_product_k4"]["f64_result"] != g["external_product_k4"]["exact_result"]
