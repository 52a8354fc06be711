"""Pins the oracle on the reference's own tests: plaintext-level known answers
(/root/reference/src/main.rs:78-95 + FIPS-197 C.1), the random round trips of main.rs:120-141,
Client::test_verify semantics (client.rs:178-216), the CTR counter add (server.rs:172-274)."""
import numpy as np
import pytest

from conftest import KAT_KEY
from oracle import oracle as orc
from tfhe_aes_amd import aes_clear
from tfhe_aes_amd.client import bytes_to_u128, u128_to_bytes


def test_byte_order_of_the_state():
    # byte 0 is the MSB of the u128 (client.rs:126-129)
    x = 0x000102030405060708090A0B0C0D0E0F
    assert u128_to_bytes(x) == list(range(16))
    assert bytes_to_u128(range(16)) == x


def test_client_roundtrip_and_bit_order(toy):
    c = toy.client
    ct = c.encrypt_bytes([0x01, 0x80, 0xA5])
    assert ct.shape == (3, 8, toy.params.big1)
    bits = c.decrypt_bits(ct)
    assert list(bits[0]) == [1, 0, 0, 0, 0, 0, 0, 0]          # block j = bit j, LSB first
    assert list(bits[1]) == [0, 0, 0, 0, 0, 0, 0, 1]
    assert list(c.decrypt_bytes(ct)) == [0x01, 0x80, 0xA5]
    # XOR of bits == wrapping add of ciphertexts (server.rs:278-282)
    assert int(c.decrypt_bytes((ct[0] + ct[2])[None])[0]) == 0x01 ^ 0xA5


def test_sbox_and_many_sbox_plaintexts(toy):
    c, O = toy.client, toy.oracle
    vals = [0x00, 0x53, 0xFF, 0x9C]
    x = c.encrypt_bytes(vals)
    enc = c.decrypt_bytes(O.wopbs_batch(x, orc.build_lutset(orc.LUTSET_ENC_ROUND)))
    dec = c.decrypt_bytes(O.wopbs_batch(x, orc.build_lutset(orc.LUTSET_DEC_MUL)))
    inv = c.decrypt_bytes(O.wopbs_batch(x, orc.build_lutset(orc.LUTSET_INV_SBOX)))
    for i, v in enumerate(vals):
        s = aes_clear.SBOX[v]
        assert list(enc[i]) == [s, aes_clear.mul2(s), aes_clear.mul3(s)]
        assert list(dec[i]) == [aes_clear.mul9(v), aes_clear.mul11(v), aes_clear.mul13(v), aes_clear.mul14(v)]
        assert int(inv[i, 0]) == aes_clear.INV_SBOX[v]


def test_circuit_bootstrap_produces_a_ggsw_of_the_bit(toy):
    """row k of the GGSW carries bit*2^49 in its body's constant coefficient, rows r<k carry -bit*S_r*2^49"""
    c, O, p = toy.client, toy.oracle, toy.params
    x = c.encrypt_bits(np.array([0, 1], dtype=np.uint8))
    small = O.keyswitch(x)
    ph = c.phase_small(small).astype(np.int64)
    assert abs(int(ph[0])) < 1 << 58 and abs(int(ph[1]) - (-(1 << 63))) < 1 << 58
    g = O.circuit_bootstrap(small)                       # [2][1][k+1][(k+1)N]
    phases = c.glwe_phase(g).astype(np.int64)            # [2][1][k+1][N]
    delta = 1 << (64 - p.cbs_base_log)
    assert np.abs(phases[0]).max() < delta // 16         # bit 0: everything ~ 0
    body_row = phases[1, 0, p.k]
    assert abs(int(body_row[0]) - delta) < delta // 16 and np.abs(body_row[1:]).max() < delta // 16
    s0 = c.glwe_sk[:512].astype(np.int64)
    assert np.abs(phases[1, 0, 0] + s0 * delta).max() < delta // 16


@pytest.mark.parametrize("idx", range(5))
def test_known_answer_vectors(toy, golden, idx):
    """key expansion -> encrypt -> decrypt, checked both ways like Client::test_verify (client.rs:178-216)"""
    v = golden["aes_kat"][idx]
    key, pt, want = int(v["key"], 16), int(v["plaintext"], 16), int(v["ciphertext"], 16)
    c, O = toy.client, toy.oracle
    rk = O.aes_key_expansion(c.encrypt_u128(key))
    assert np.array_equal(c.decrypt_bytes(rk), np.array(aes_clear.expand_key(key), dtype=np.uint8))
    enc = O.aes_encrypt(rk, c.encrypt_u128(pt))
    assert c.decrypt_u128(enc) == want
    assert c.decrypt_u128(O.aes_decrypt(rk, enc)) == pt


def test_random_roundtrips(toy):
    rng = np.random.default_rng(2025)            # main.rs:120-141 draws from thread_rng; seeded here
    c, O = toy.client, toy.oracle
    for _ in range(2):
        key = int.from_bytes(rng.bytes(16), "big")
        pt = int.from_bytes(rng.bytes(16), "big")
        rk = O.aes_key_expansion(c.encrypt_u128(key))
        enc = O.aes_encrypt(rk, c.encrypt_u128(pt))
        assert c.decrypt_u128(enc) == aes_clear.aes128_encrypt_block(key, pt)
        assert c.decrypt_u128(O.aes_decrypt(rk, enc)) == pt


@pytest.mark.parametrize("iv,i", [
    (0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF, 1),
    (0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF, 0x1FF),                 # i >= 256: the reference's server.rs:182 carry is wrong here
    (0x000000000000000000000000FFFFFFFF, 1),                       # carry ripples through four bytes
    (0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF, 2),                       # wraps mod 2^128
    (0x0123456789ABCDEF0123456789ABCDEF, 0x00FF00FF00FF00FF00FF),  # multi-byte addend
])
def test_add_scalar(toy, iv, i):
    c, O = toy.client, toy.oracle
    out = O.add_scalar(c.encrypt_u128(iv), i)
    assert c.decrypt_u128(out) == (iv + i) % (1 << 128)


def test_noise_budget_of_the_schedule(toy):
    """outputs of one WoPBS are fresh: their phase error is far below 2^62 even after the 5 additions of a round"""
    c, O = toy.client, toy.oracle
    y = O.wopbs_batch(c.encrypt_bytes([0x3C]), orc.build_lutset(orc.LUTSET_SBOX))
    bits, ph = c.decrypt_bits(y, return_phase=True)
    err = (ph - (bits.astype(np.uint64) << np.uint64(63))).astype(np.int64)
    assert np.abs(err).max() * 5 < 1 << 61


def test_server_keys_roundtrip_through_a_file(toy, tmp_path):
    from tfhe_aes_amd import PARAM_OPT
    from tfhe_aes_amd.client import ServerKeys

    f = tmp_path / "keys.npz"
    toy.keys.save(f)
    back = ServerKeys.load(f, toy.params)
    assert np.array_equal(back.ksk, toy.keys.ksk) and np.array_equal(back.bsk, toy.keys.bsk) and np.array_equal(back.pfpksk, toy.keys.pfpksk)
    with pytest.raises(ValueError):
        ServerKeys.load(f, PARAM_OPT)


def test_ciphertext_files_round_trip(toy, tmp_path):
    """on-disk form of encrypted states / round keys (npz of uint64 words, parameter set and kind checked on load)"""
    from tfhe_aes_amd import PARAM_OPT
    from tfhe_aes_amd.client import load_ciphertexts, save_ciphertexts

    c, p = toy.client, toy.params
    st = np.stack([c.encrypt_u128(0x00112233445566778899AABBCCDDEEFF + i) for i in range(2)])
    save_ciphertexts(tmp_path / "st.npz", p, "state", st)
    back = load_ciphertexts(tmp_path / "st.npz", p, "state")
    assert np.array_equal(back, st) and c.decrypt_u128(back[1]) == 0x00112233445566778899AABBCCDDEEFF + 1
    with pytest.raises(ValueError):
        load_ciphertexts(tmp_path / "st.npz", p, "round_keys")
    with pytest.raises(ValueError):
        load_ciphertexts(tmp_path / "st.npz", PARAM_OPT, "state")
    with pytest.raises(ValueError):
        save_ciphertexts(tmp_path / "bad.npz", p, "state", st[:, :8])
