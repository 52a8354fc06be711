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
    # cloning into a context with other parameters, or from one without keys, is an error, not a crash
    c = _native.Engine(p, device=0)
    with pytest.raises(_native.FheAesError):
        b.clone_keys_from(c)
    b.close()
    c.close()


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
