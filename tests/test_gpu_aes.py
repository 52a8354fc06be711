"""The plugin / Server API on the MI355X (through the C ABI and the Python mirror of the reference's
interface) against the oracle, the golden hashes and AES known answers."""
import numpy as np
import pytest

from conftest import Kit, sha
from oracle import oracle as orc
from tfhe_aes_amd import PARAM_TOY, aes_clear
from tfhe_aes_amd.server import Server, gen_lut

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy_server(toy):
    return Server(toy.keys, device=0, engine=toy.engine())


@pytest.mark.parametrize("n_luts_set", [orc.LUTSET_SBOX, orc.LUTSET_ENC_ROUND, orc.LUTSET_DEC_MUL])
def test_many_wopbs_8bit(toy, toy_server, n_luts_set):
    vals = [0x00, 0x53, 0xFF, 0xA7, 0x10]
    x = toy.client.encrypt_bytes(vals)
    luts = orc.build_lutset(n_luts_set)
    got = toy_server.many_wopbs_without_padding(x, list(luts))
    assert np.array_equal(got, toy.oracle.wopbs_batch(x, luts))


def test_many_wopbs_9bit_with_per_input_luts(toy, toy_server):
    """the shape add_scalar uses: 9 blocks (8 bits + carry), two 9->9 LUTs that differ per input (server.rs:216-252)"""
    c, p = toy.client, toy.params
    rng = np.random.default_rng(9)
    n = 3
    bits = rng.integers(0, 2, (n, 9)).astype(np.uint8)
    x = c.encrypt_bits(bits)
    luts = np.stack([np.stack([gen_lut(2, 1, 512, 9, lambda v, a=a: ((v & 0xFF) + (v >> 8) + a) % 256),
                               gen_lut(2, 1, 512, 9, lambda v, a=a: 1 if (v & 0xFF) + (v >> 8) + a > 255 else 0)])
                     for a in (3, 200, 255)])
    got = toy_server.many_wopbs_without_padding(x, luts)
    want = toy.oracle.wopbs_batch(x, luts, lut_per_input=True)
    assert np.array_equal(got, want)
    dec = c.decrypt_bits(got)
    for i, a in enumerate((3, 200, 255)):
        v = int(sum(int(bits[i, j]) << j for j in range(9)))
        s = (v & 0xFF) + (v >> 8) + a
        assert int(sum(int(dec[i, 0, j]) << j for j in range(8))) == s % 256 and int(dec[i, 1, 0]) == (1 if s > 255 else 0)


@pytest.mark.parametrize("nb,n_luts", [(10, 2), (12, 1)])
def test_many_wopbs_wider_than_log_n_uses_the_cmux_tree(toy, toy_server, nb, n_luts):
    """inputs wider than log2 N = 9 bits: gen_lut makes 2^(nb-9) polynomials per output bit (gen_lut.rs:19-39) and
    vertical_packing (many_wopbs.rs:277) selects one through a CMUX tree over the high bits before the blind rotation
    (SURVEY.md A.8).  Not used by AES; completes many_wopbs_without_padding.  Bit-exact against the oracle + decrypts to f(x)."""
    c = toy.client
    fs = [lambda v: (v * 37 + 5) % (1 << nb), lambda v: (v ^ (v >> 3) ^ 0x155) % (1 << nb)][:n_luts]
    luts = [gen_lut(2, 1, 512, nb, f) for f in fs]
    assert luts[0].shape == (nb, 1 << nb)
    vals = [0, 1, (1 << nb) - 1, 0x2A5 % (1 << nb), 513, 1 << (nb - 1)]
    bits = np.array([[(v >> j) & 1 for j in range(nb)] for v in vals], dtype=np.uint8)
    x = c.encrypt_bits(bits)
    got = toy_server.many_wopbs_without_padding(x, luts)
    want = toy.oracle.wopbs_batch(x, np.stack(luts))
    assert np.array_equal(got, want)
    dec = c.decrypt_bits(got)
    for i, v in enumerate(vals):
        for li, f in enumerate(fs):
            assert int(sum(int(dec[i, li, j]) << j for j in range(nb))) == f(v), (v, li)


def test_sbox_and_many_sbox(toy, toy_server):
    c = toy.client
    vals = [0x3C, 0x00, 0x80]
    x = c.encrypt_bytes(vals)
    ms = toy_server.many_sbox(x, inv=False)
    assert np.array_equal(ms, toy.oracle.wopbs_batch(x, orc.build_lutset(orc.LUTSET_ENC_ROUND)))
    mi = toy_server.many_sbox(x, inv=True)
    assert np.array_equal(mi, toy.oracle.wopbs_batch(x, orc.build_lutset(orc.LUTSET_DEC_MUL)))
    y = x.copy()
    toy_server.sbox(y, inv=False)
    assert np.array_equal(y, ms[:, 0])                      # sbox == first LUT of many_sbox (sbox.rs:52 vs :79)
    assert list(c.decrypt_bytes(y)) == [aes_clear.SBOX[v] for v in vals]
    toy_server.sbox(y, inv=True)
    assert list(c.decrypt_bytes(y)) == vals


def test_aes_pipeline_bit_exact_and_golden(golden):
    """a fresh kit replays the golden sequence: HIP output hashes == the oracle's committed hashes"""
    g = golden["oracle_toy"]
    kit = Kit(PARAM_TOY, seed=g["seed"])
    c, p = kit.client, kit.params
    srv = Server(kit.keys, device=0)
    x = c.encrypt_bytes([0x00, 0x53, 0xFF, 0xA7, 0x10])
    assert sha(x.reshape(-1, p.big1)) == g["input"]
    assert sha(srv.many_sbox(x, inv=False)) == g["many_sbox"]
    st, ek = c.encrypt_u128(c.iv), c.encrypt_u128(c.key)
    rk = srv.aes_key_expansion(ek)
    assert sha(rk) == g["round_keys"]
    enc = srv.aes_encrypt(rk, st.copy())
    assert sha(enc) == g["aes_encrypt"]
    assert sha(srv.aes_decrypt(rk, enc.copy())) == g["aes_decrypt"]
    assert sha(srv.add_scalar(st.copy(), 0x1FF)) == g["add_scalar_0x1ff"]
    c.test_verify(enc, srv.aes_decryption(rk, enc.copy()))  # Client::test_verify (client.rs:178-216)


@pytest.fixture(scope="module")
def opt_server(opt):
    return Server(opt.keys, device=0, engine=opt.engine())


@pytest.mark.parametrize("which", ["toy", "opt"])
@pytest.mark.parametrize("idx", range(5))
def test_known_answer_vectors_on_gpu(request, golden, which, idx):
    """The reference's own known answers (src/main.rs:78-95: the four SP 800-38A F.1.1 pairs) + FIPS-197 C.1, on the HIP path at the
    toy set AND at the reference's parameter set (PARAM_OPT, k = 4: the latency / paired kernels): key expansion -> encrypt ->
    decrypt, checked both ways as Client::test_verify does (client.rs:178-216)."""
    kit, srv = request.getfixturevalue(which), request.getfixturevalue(which + "_server")
    v = golden["aes_kat"][idx]
    key, pt, want = int(v["key"], 16), int(v["plaintext"], 16), int(v["ciphertext"], 16)
    c = kit.client
    rk = srv.aes_key_expansion(c.encrypt_u128(key))
    assert np.array_equal(c.decrypt_bytes(rk), np.array(aes_clear.expand_key(key), dtype=np.uint8))
    enc = srv.aes_encrypt(rk, c.encrypt_u128(pt))
    assert c.decrypt_u128(enc) == want
    assert c.decrypt_u128(srv.aes_decrypt(rk, enc.copy())) == pt


def test_batched_blocks_equal_per_block_oracle(toy, toy_server):
    """CTR batch: 3 blocks in one call == the oracle block by block (blocks are independent, main.rs:55-64)"""
    c, O = toy.client, toy.oracle
    iv = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF
    rk = O.aes_key_expansion(c.encrypt_u128(c.key))
    states = np.stack([c.encrypt_u128(iv)] * 3)
    toy_server.add_scalar(states, [0, 1, 0x101])
    for b, i in enumerate((0, 1, 0x101)):
        assert c.decrypt_u128(states[b]) == iv + i
    want = np.stack([O.aes_encrypt(rk, states[b]) for b in range(3)])
    got = toy_server.aes_encrypt(rk, states.copy())
    assert np.array_equal(got, want)
    for b, i in enumerate((0, 1, 0x101)):
        assert c.decrypt_u128(got[b]) == aes_clear.aes128_encrypt_block(c.key, iv + i)


def test_device_resident_tensors_match_host_path(toy, toy_server):
    import torch

    c, p = toy.client, toy.params
    st = c.encrypt_u128(0x00112233445566778899AABBCCDDEEFF)
    rk = toy.oracle.aes_key_expansion(c.encrypt_u128(c.key))
    host = toy_server.aes_encrypt(rk, st.copy())
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_st = torch.from_numpy(st.view(np.int64)).cuda()
    torch.cuda.synchronize()
    toy_server.aes_encrypt(d_rk, d_st)
    toy_server.synchronize()
    assert np.array_equal(d_st.cpu().numpy().view(np.uint64), host)
    with pytest.raises(ValueError):
        toy_server.aes_encrypt(rk, d_st)                   # mixed memory spaces are refused


def test_device_tensors_with_host_luts(toy, toy_server):
    """many_wopbs_without_padding on CUDA tensors with a host LUT list: the wrapper's device copy of the LUTs must be
    complete before the engine's stream reads it and must stay alive until the engine is idle"""
    import torch

    c = toy.client
    x = c.encrypt_bytes([0x11, 0xC4, 0x7E])
    luts = list(orc.build_lutset(orc.LUTSET_DEC_MUL))
    want = toy.oracle.wopbs_batch(x, np.stack(luts))
    d_x = torch.from_numpy(x.view(np.int64)).cuda()
    outs = []
    for _ in range(3):                                       # several calls in flight, temporaries churn in between
        outs.append(toy_server.many_wopbs_without_padding(d_x, luts))
        junk = torch.full((luts[0].size * 4,), -1, dtype=torch.int64, device="cuda")   # would reuse a freed LUT block
        del junk
    toy_server.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint64), want)


def test_add_scalar_then_encrypt_enqueued_without_sync(toy, toy_server):
    """FHEAES_DEVICE calls only enqueue (fheaes.h): add_scalar + aes_encrypt on device tensors with ONE final
    synchronize equal the host-path results"""
    import torch

    c = toy.client
    iv = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF
    rk = toy.oracle.aes_key_expansion(c.encrypt_u128(c.key))
    st = np.stack([c.encrypt_u128(iv)] * 2)
    host = toy_server.aes_encrypt(rk, toy_server.add_scalar(st.copy(), [5, 0x1FF]))
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_a = torch.from_numpy(st.view(np.int64)).cuda()
    d_b = torch.from_numpy(st.view(np.int64)).cuda()
    torch.cuda.synchronize()
    toy_server.add_scalar(d_a, [5, 0x1FF])
    toy_server.add_scalar(d_b, [0x1FF, 5])                   # second call reuses the pinned counter staging
    toy_server.aes_encrypt(d_rk, d_a)
    toy_server.aes_encrypt(d_rk, d_b)
    toy_server.synchronize()
    a, b = d_a.cpu().numpy().view(np.uint64), d_b.cpu().numpy().view(np.uint64)
    assert np.array_equal(a, host)
    assert np.array_equal(b[0], host[1]) and np.array_equal(b[1], host[0])


def test_concurrent_callers_are_serialised(toy, toy_server):
    """the reference shares one &Server between rayon threads (main.rs:55-61): concurrent calls on one context must not corrupt
    the workspace (the engine serialises them)"""
    import threading

    c = toy.client
    xs = [c.encrypt_bytes([17 * i + 3, 0xF0 ^ i]) for i in range(4)]
    want = [toy.oracle.wopbs_batch(x, orc.build_lutset(orc.LUTSET_ENC_ROUND)) for x in xs]
    got = [None] * 4

    def work(i):
        for _ in range(3):
            got[i] = toy_server.many_sbox(xs[i], inv=False)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(4):
        assert np.array_equal(got[i], want[i])


def test_many_sbox_param_opt(opt):
    """one AES round's worth of S-Boxes (16 bytes = 128 bit-CBS) at the reference's parameter set"""
    srv = Server(opt.keys, device=0, engine=opt.engine())
    vals = list(range(0x50, 0x60))
    x = opt.client.encrypt_bytes(vals)
    got = srv.many_sbox(x, inv=False)
    assert np.array_equal(got, opt.oracle.wopbs_batch(x, orc.build_lutset(orc.LUTSET_ENC_ROUND)))
    dec = opt.client.decrypt_bytes(got)
    for i, v in enumerate(vals):
        s = aes_clear.SBOX[v]
        assert list(dec[i]) == [s, aes_clear.mul2(s), aes_clear.mul3(s)]


def test_full_block_bit_exact_param_opt(opt):
    """One whole block at the reference's parameter set, every Server entry point word for word against the oracle:
    aes_key_expansion (server.rs:107-167), aes_encrypt (:39-64), aes_decrypt (:67-105), add_scalar(0x1FF) (:172-274).
    This is what pins the k=4 strides of the gather/add kernels, the 9-bit pack/unpack kernels and the 4-LUT packing;
    the oracle side costs ~40 s on the GPU box's host cores."""
    c, O = opt.client, opt.oracle
    srv = Server(opt.keys, device=0, engine=opt.engine())
    key, pt = c.key, 0x3243F6A8885A308D313198A2E0370734
    ek, st = c.encrypt_u128(key), c.encrypt_u128(pt)
    rk = srv.aes_key_expansion(ek)
    rk_want = O.aes_key_expansion(ek)
    assert np.array_equal(rk, rk_want)
    enc = srv.aes_encrypt(rk, st.copy())
    assert np.array_equal(enc, O.aes_encrypt(rk_want, st.copy()))
    assert c.decrypt_u128(enc) == aes_clear.aes128_encrypt_block(key, pt)
    dec = srv.aes_decrypt(rk, enc.copy())
    assert np.array_equal(dec, O.aes_decrypt(rk_want, enc.copy()))
    assert c.decrypt_u128(dec) == pt
    ctr = srv.add_scalar(st.copy(), 0x1FF)
    assert np.array_equal(ctr, O.add_scalar(st.copy(), 0x1FF))
    assert c.decrypt_u128(ctr) == (pt + 0x1FF) % (1 << 128)


def test_noise_guard_counts_what_the_schedule_sums(toy, toy_server):
    """the engine's counterpart of tfhe-rs' noise-asserts (Cargo.toml:7, MaxNoiseLevel::new(5) at client.rs:92), as a static assertion on
    the engine's own schedule (not runtime noise tracking): the linear layers declare how many WoPBS outputs they sum per word --
    MixColumns (4 terms) + AddRoundKey = 5, the limit"""
    c = toy.client
    rk = toy_server.aes_key_expansion(c.encrypt_u128(c.key))
    seen, limit = toy_server.engine.noise_level_seen()
    assert limit == 5 and 2 <= seen <= 5                      # key expansion sums two words at a time (server.rs:140-148)
    toy_server.aes_encrypt(rk, c.encrypt_u128(1))
    assert toy_server.engine.noise_level_seen() == (5, 5)
    toy_server.aes_decrypt(rk, c.encrypt_u128(1))
    assert toy_server.engine.noise_level_seen() == (5, 5)
