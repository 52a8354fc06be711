"""Seeded evaluation keys (first slice): keys travel as (public 256-bit mask key, bodies) and every mask word
is regenerated from a ChaCha20 stream -- on the host by SeededServerKeys.expand(), on the GPU by fheaes_upload_keys_seeded."""
import ctypes

import numpy as np
import pytest

from tfhe_aes_amd import client as cl
from tfhe_aes_amd.client import SeededServerKeys, ServerKeys


def test_chacha20_block_known_answer():
    """RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00 -- the C generator
    (csrc/client.c) and its numpy twin (client.chacha20_blocks) both give the published block"""
    want = np.array([0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
                     0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2], dtype=np.uint32)
    key = np.frombuffer(bytes(range(32)), dtype=np.uint32).copy()
    nonce = np.array([0x09000000, 0x4a000000, 0], dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    cl._load().fheaes_client_chacha20_block(cl._u32(key), 1, cl._u32(nonce), cl._u32(out))
    assert np.array_equal(out, want)
    w = cl.chacha20_blocks(key, np.array([1], dtype=np.uint32), 0x09000000, np.array([0x4a000000], dtype=np.uint32), np.array([0], dtype=np.uint32))
    assert np.array_equal(w[0], want)


def test_mask_stream_numpy_matches_c():
    lib = cl._load()
    key = cl.test_key(0x1234567890ABCDEF, 2)
    w = cl.mask_words(key, cl.MASK_TAG_BSK, 7, 40)
    for q, j in ((0, 0), (3, 17), (6, 39)):
        assert int(w[q, j]) == lib.fheaes_client_mask_word(cl._u32(key), cl.MASK_TAG_BSK, q, j)


def test_default_client_draws_fresh_keys():
    """Client() without a seed: secret, mask and per-encryption keys come from os.urandom; an explicit seed is the
    reproducible test mode"""
    from tfhe_aes_amd import PARAM_TOY

    a, b = cl.Client(params=PARAM_TOY), cl.Client(params=PARAM_TOY)
    assert not a.deterministic and not np.array_equal(a.seed, b.seed) and not np.array_equal(a.mask_seed, b.mask_seed)
    assert not np.array_equal(a.seed, a.mask_seed)
    x1, x2 = a.encrypt_bits(np.array([1, 0], dtype=np.uint8)), a.encrypt_bits(np.array([1, 0], dtype=np.uint8))
    assert not np.array_equal(x1, x2) and list(a.decrypt_bits(x1)) == [1, 0] and list(a.decrypt_bits(x2)) == [1, 0]
    c, d = cl.Client(params=PARAM_TOY, seed=7), cl.Client(params=PARAM_TOY, seed=7)
    assert c.deterministic and np.array_equal(c.glwe_sk, d.glwe_sk)
    assert np.array_equal(c.encrypt_bits(np.array([1], dtype=np.uint8)), d.encrypt_bits(np.array([1], dtype=np.uint8)))


def test_compress_expand_roundtrip_and_sizes(toy, tmp_path):
    p, keys = toy.params, toy.keys
    sk = keys.compress()
    assert sk.ksk_body.shape == (p.big, p.ks_level) and sk.bsk_body.shape == (p.n, p.pbs_level, p.k + 1, p.N)
    assert sk.pfpksk_body.shape == (p.k + 1, p.big1, p.pfks_level, p.N)
    full = sk.expand()
    assert np.array_equal(full.ksk, keys.ksk) and np.array_equal(full.bsk, keys.bsk) and np.array_equal(full.pfpksk, keys.pfpksk)
    # file round trip (npz of uint64 arrays, no pickle)
    sk.save(tmp_path / "k.npz")
    back = SeededServerKeys.load(tmp_path / "k.npz", p)
    assert np.array_equal(back.mask_seed, sk.mask_seed) and np.array_equal(back.bsk_body, sk.bsk_body)
    with pytest.raises(ValueError):
        ServerKeys(p, keys.ksk, keys.bsk, keys.pfpksk).compress()         # foreign keys carry no mask seed
    # FULL keys through a file keep their public mask seed: save -> load -> compress -> expand is the identity
    keys.save(tmp_path / "full.npz")
    again = ServerKeys.load(tmp_path / "full.npz", p)
    assert again.mask_seed is not None and again.mask_seed.dtype == np.uint32 and np.array_equal(again.mask_seed, keys.mask_seed)
    full2 = again.compress().expand()
    assert np.array_equal(full2.ksk, keys.ksk) and np.array_equal(full2.bsk, keys.bsk) and np.array_equal(full2.pfpksk, keys.pfpksk)
    ServerKeys(p, keys.ksk, keys.bsk, keys.pfpksk).save(tmp_path / "foreign.npz")      # and foreign keys stay seedless
    assert ServerKeys.load(tmp_path / "foreign.npz", p).mask_seed is None


def test_compression_ratio_param_opt():
    from tfhe_aes_amd import PARAM_OPT as p

    full = 8 * (p.ksk_words + p.bsk_words + p.pfpksk_words)
    body = 8 * (p.big * p.ks_level + p.n * p.pbs_level * (p.k + 1) * p.N + (p.k + 1) * p.big1 * p.pfks_level * p.N)
    assert full == 1_037_844_480 and body == 194_494_464                   # 5.3x fewer bytes over PCIe / xGMI


def test_bsk_rows_keep_the_message_out_of_the_masks(toy):
    """GGSW row (l, r) carries s_i * g_l on component r through its BODY (-g_l S_r(X), or +g_l for r = k): the phases are
    those of a GGSW of s_i and no mask word depends on a secret"""
    p, c = toy.params, toy.client
    bsk = toy.keys.bsk.reshape(p.n, p.pbs_level, p.k + 1, (p.k + 1) * p.N)
    S = c.glwe_sk.reshape(p.k, p.N).astype(np.int64)
    for i in (0, p.n - 1):
        ph = c.glwe_phase(bsk[i])                                          # [L][k+1][N]
        for l in range(p.pbs_level):
            g = 1 << (64 - p.pbs_base_log * (l + 1))
            for r in range(p.k + 1):
                want = np.zeros(p.N, dtype=np.int64)
                if c.lwe_sk[i]:
                    if r == p.k:
                        want[0] = g
                    else:
                        want = -g * S[r]
                err = (ph[l, r].astype(np.int64) - want.astype(np.int64))
                assert np.abs(err).max() < 1 << 30


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["toy", "opt"])
def test_seeded_upload_equals_full_upload(which, request):
    """device keys built from (mask_seed, bodies) are the device keys of the uncompressed upload: K1, K2's Fourier BSK and
    K3 give array_equal outputs"""
    from tfhe_aes_amd import _native

    kit = request.getfixturevalue(which)
    p = kit.params
    sk = kit.keys.compress()
    E_full = kit.engine()
    E = _native.Engine(p, device=0)
    E.upload_keys_seeded(sk.mask_seed, sk.ksk_body, sk.bsk_body, sk.pfpksk_body)
    for i in (0, p.n // 2, p.n - 1):
        assert np.array_equal(E.read_bsk_fourier(i).view(np.uint64), E_full.read_bsk_fourier(i).view(np.uint64))
    rng = np.random.default_rng(11)
    m = 19
    x = rng.integers(0, 1 << 64, (m, p.big1), dtype=np.uint64)
    a, b = np.zeros((m, p.n + 1), dtype=np.uint64), np.zeros((m, p.n + 1), dtype=np.uint64)
    E.keyswitch_batch(x, a, m)
    E_full.keyswitch_batch(x, b, m)
    assert np.array_equal(a, b) and np.array_equal(a, kit.oracle.keyswitch(x))
    g1 = np.zeros((m, p.k + 1, (p.k + 1) * 512), dtype=np.uint64)
    g2 = np.zeros_like(g1)
    E.pfpks_batch(x, g1, m)
    E_full.pfpks_batch(x, g2, m)
    assert np.array_equal(g1, g2)
    bits = rng.integers(0, 2, 5).astype(np.uint8)
    ct = kit.client.encrypt_bits(bits)
    small = kit.oracle.keyswitch(ct)
    o1, o2 = np.zeros((5, p.big1), dtype=np.uint64), np.zeros((5, p.big1), dtype=np.uint64)
    E.cbs_pbs_batch(small, o1, 5)
    E_full.cbs_pbs_batch(small, o2, 5)
    assert np.array_equal(o1, o2)
    E.close()
