"""The oracle at the reference's real parameter set (client.rs:31-57): shapes of SURVEY.md 8 and one S-Box."""
import numpy as np

from oracle import oracle as orc
from tfhe_aes_amd import PARAM_OPT, aes_clear


def test_shapes_and_byte_counts():
    p = PARAM_OPT
    assert (p.n, p.k, p.N, p.big) == (669, 4, 512, 2048)
    assert 8 * p.ksk_words == 65_863_680
    assert 8 * p.bsk_words == 342_528_000
    assert 8 * p.pfpksk_words == 629_452_800
    assert p.key_bytes_per_bit == 1_037_844_480
    assert 8 * p.big1 == 16_392 and 16 * 8 * 8 * p.big1 == 2_098_176


def test_many_sbox_at_param_opt(opt):
    c, O = opt.client, opt.oracle
    vals = [0x53, 0xC7]
    y = O.wopbs_batch(c.encrypt_bytes(vals), orc.build_lutset(orc.LUTSET_ENC_ROUND))
    dec = c.decrypt_bytes(y)
    for i, v in enumerate(vals):
        s = aes_clear.SBOX[v]
        assert list(dec[i]) == [s, aes_clear.mul2(s), aes_clear.mul3(s)]
    bits, ph = c.decrypt_bits(y, return_phase=True)
    err = (ph - (bits.astype(np.uint64) << np.uint64(63))).astype(np.int64)
    # README.md:175-180: the schedule adds at most 5 ciphertexts between bootstraps
    assert np.abs(err).max() * 5 < 1 << 62
