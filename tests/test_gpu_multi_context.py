"""The multi-GPU split behind the C ABI (include/fheaes.h: fheaes_clone_keys): several contexts, ONE key upload, device-to-device
clones of the converted key images, blocks sharded over the contexts with one host thread each -- what the reference does with
rayon over `&Server` (main.rs:55-64, server.rs:32-35).  A test box has one GPU, so both contexts sit on device 0 (the copy is then
an HBM copy instead of hipMemcpyPeerAsync over xGMI; everything else is the same code)."""
import threading

import numpy as np
import pytest

from tfhe_aes_amd import _native
from tfhe_aes_amd.server import Server, ServerGroup

pytestmark = pytest.mark.gpu


def test_cloned_context_holds_the_same_key_images_and_computes_the_same_words(toy):
    p = toy.params
    a = toy.engine()
    b = _native.Engine(p, device=0)
    with pytest.raises(_native.FheAesError):                    # nothing to evaluate with before the clone
        b.keyswitch_batch(np.zeros((1, p.big1), dtype=np.uint64), np.zeros((1, p.n + 1), dtype=np.uint64), 1)
    b.clone_keys_from(a)
    for i in (0, p.n // 2, p.n - 1):
        assert np.array_equal(a.read_bsk_fourier(i).view(np.uint64), b.read_bsk_fourier(i).view(np.uint64))
    rng = np.random.default_rng(77)
    x = toy.client.encrypt_bits(rng.integers(0, 2, 70).astype(np.uint8))
    oa, ob = np.zeros((70, p.n + 1), dtype=np.uint64), np.zeros((70, p.n + 1), dtype=np.uint64)
    a.keyswitch_batch(x, oa, 70)
    b.keyswitch_batch(x, ob, 70)
    assert np.array_equal(oa, ob) and np.array_equal(oa, toy.oracle.keyswitch(x))                    # K1 (KSK image)
    y = rng.integers(0, 1 << 64, (9, p.big1), dtype=np.uint64)
    ga, gb = (np.zeros((9, p.k + 1, (p.k + 1) * 512), dtype=np.uint64) for _ in range(2))
    a.pfpks_batch(y, ga, 9)
    b.pfpks_batch(y, gb, 9)
    assert np.array_equal(ga, gb) and np.array_equal(ga, toy.oracle.pfpks(y))                        # K3 (PFPKSK image)
    info = b.clone_info()
    assert info["path"] == "same_device" and info["bytes"] > 0 and info["seconds"] > 0     # both contexts on device 0: an HBM copy
    assert a.clone_info()["path"] == "none"
    # cloning from a context without keys is an error, not a crash
    c = _native.Engine(p, device=0)
    with pytest.raises(_native.FheAesError) as e:
        b.clone_keys_from(c)
    assert e.value.code == -2                                   # FHEAES_ERR_NOKEYS
    b.close()
    c.close()


def test_clone_between_parameter_sets_is_refused(toy):
    """the memcmp branch of fheaes_clone_keys: a context created for PARAM_OPT cannot take the toy keys"""
    from tfhe_aes_amd import PARAM_OPT

    other = _native.Engine(PARAM_OPT, device=0)
    with pytest.raises(_native.FheAesError) as e:
        other.clone_keys_from(toy.engine())
    assert e.value.code == -1 and "parameter sets differ" in str(e.value)          # FHEAES_ERR_INVALID
    other.close()


def test_last_error_is_not_inherited_by_a_new_context(toy):
    """the per-thread last-error cache is keyed by a context id, not by its address: a context created after a failed one was
    destroyed (possibly at the same address) starts with an empty message"""
    lib = _native.load_library()
    p = toy.params
    for _ in range(4):
        bad = _native.Engine(p, device=0)
        with pytest.raises(_native.FheAesError):
            bad.keyswitch_batch(np.zeros((1, p.big1), dtype=np.uint64), np.zeros((1, p.n + 1), dtype=np.uint64), 1)   # no keys
        assert b"keys" in lib.fheaes_last_error(bad._h)
        bad.close()
        fresh = _native.Engine(p, device=0)
        assert lib.fheaes_last_error(fresh._h) == b""
        fresh.close()


def test_two_contexts_two_threads_equal_one_context(toy):
    """a 4-block CTR batch (Server::add_scalar + aes_encrypt, main.rs:59-61): two halves on two contexts from two threads give
    the words of the whole batch on one context"""
    c = toy.client
    iv = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF
    one = Server(toy.keys, device=0, engine=toy.engine())
    rk = one.aes_key_expansion(c.encrypt_u128(c.key))
    st = np.stack([c.encrypt_u128(iv)] * 4)
    want = one.aes_encrypt(rk, one.add_scalar(st.copy(), [0, 1, 2, 0x1FF]))
    grp = ServerGroup(toy.keys, devices=(0, 0))
    assert len(grp.servers) == 2 and grp.servers[0].engine is not grp.servers[1].engine
    got = grp.aes_encrypt(rk, grp.add_scalar(st.copy(), [0, 1, 2, 0x1FF]))
    assert np.array_equal(got, want)
    from tfhe_aes_amd.aes_clear import aes128_encrypt_block
    for i, ctr in enumerate([0, 1, 2, 0x1FF]):
        assert c.decrypt_u128(got[i]) == aes128_encrypt_block(c.key, (iv + ctr) & ((1 << 128) - 1))
    # and the two contexts really are independent: concurrent, different inputs, repeated
    xs = [c.encrypt_bytes([0x11 * (i + 1), 0xF0 ^ i]) for i in range(2)]
    ref = [one.many_sbox(x, inv=False) for x in xs]
    outs = [None, None]

    def work(i):
        for _ in range(3):
            outs[i] = grp.servers[i].many_sbox(xs[i], inv=False)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.array_equal(outs[0], ref[0]) and np.array_equal(outs[1], ref[1])


def test_two_contexts_at_param_opt_equal_one_context(opt):
    """the in-process split at the reference's REAL parameter set on the one GPU there is: keys uploaded once (1.04 GB), cloned into a
    second context (fheaes_clone_keys), 2 x 16 CTR blocks (Server::add_scalar + aes_encrypt, main.rs:59-61) from two host threads
    == one context on all 32 blocks, word for word; the clone's path, size and time are what fheaes_clone_info reports"""
    import time

    from tfhe_aes_amd.aes_clear import aes128_encrypt_block

    c, p = opt.client, opt.params
    iv = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF
    n = 32
    one = Server(opt.keys, device=0, engine=opt.engine())
    rk = one.aes_key_expansion(c.encrypt_u128(c.key))
    st = np.stack([c.encrypt_u128(iv)] * n)
    ctrs = [0, 1, 0xFF, 0x100, 0x1FF] + list(range(5, n))
    t0 = time.perf_counter()
    want = one.aes_encrypt(rk, one.add_scalar(st.copy(), ctrs))
    one.synchronize()
    t_one = time.perf_counter() - t0
    grp = ServerGroup(opt.keys, devices=(0, 0))
    info = grp.clone_info()[0]
    assert info["path"] == "same_device"
    assert info["bytes"] == 66_060_288 + 635_699_200 + 342_528_000               # KSK + PFPKSK fragment planes (padded) + Fourier BSK
    t0 = time.perf_counter()
    got = grp.aes_encrypt(rk, grp.add_scalar(st.copy(), ctrs))
    t_two = time.perf_counter() - t0
    assert np.array_equal(got, want)
    for i in (0, 3, 4, n - 1):
        assert c.decrypt_u128(got[i]) == aes128_encrypt_block(c.key, (iv + ctrs[i]) & ((1 << 128) - 1))
    print("PARAM_OPT two contexts on one GPU: clone %.3f s for %.2f GB (%s); 32 blocks: one context %.2f s, two contexts x 16 blocks %.2f s"
          % (info["seconds"], info["bytes"] / 1e9, info["path"], t_one, t_two))
    with pytest.raises(ValueError):
        grp.aes_encrypt(rk, st[0].copy())                                         # a single state must be wrapped, not cut into bytes
    for s in grp.servers:
        if s.engine is not opt.engine():
            s.engine.close()
