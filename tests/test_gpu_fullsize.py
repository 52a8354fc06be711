"""BASELINE.json sizes on the MI355X, checked through size-independent properties (the oracle would need
hours): every block decrypts to AES-CTR, decrypt(encrypt(x)) == x, and launches are deterministic."""
import numpy as np
import pytest

from conftest import sha
from tfhe_aes_amd import aes_clear
from tfhe_aes_amd.server import Server

pytestmark = pytest.mark.gpu

IV = 0xF0F1F2F3F4F5F6F7F8F9FAFBFCFDFEFF


@pytest.fixture(scope="module")
def opt_server(opt):
    return Server(opt.keys, device=0, engine=opt.engine())


def test_128_ctr_blocks_param_opt(opt, opt_server):
    """BASELINE configs[2]: 128 CTR blocks, full 10 rounds, on one MI355X"""
    import torch

    c, p = opt.client, opt.params
    key = c.key
    rk = opt_server.aes_key_expansion(c.encrypt_u128(key))
    assert np.array_equal(c.decrypt_bytes(rk), np.array(aes_clear.expand_key(key), dtype=np.uint8))
    n = 128
    states = np.stack([c.encrypt_u128(IV + i) for i in range(n)])
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_st = torch.from_numpy(states.view(np.int64)).cuda()
    torch.cuda.synchronize()
    opt_server.aes_encrypt(d_rk, d_st)
    opt_server.synchronize()
    out = d_st.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert c.decrypt_u128(out[i]) == aes_clear.aes128_encrypt_block(key, IV + i), "block %d" % i
    # noise budget over all 16,384 output bits: outputs carry one fresh WoPBS result + one round key (2 addends);
    # the decision threshold is 2^62 (README.md:175-180 of the reference: p_fail 2^-64 at 5 addends)
    bits, ph = c.decrypt_bits(out, return_phase=True)
    err = (ph - (bits.astype(np.uint64) << np.uint64(63))).astype(np.int64)
    assert np.abs(err).max() < 1 << 59, "max |noise| = 2^%.1f" % np.log2(float(np.abs(err).max()))
    assert np.abs(err).std() < 1 << 56
    # determinism at full size: a second launch on the same inputs gives the same words
    d_st2 = torch.from_numpy(states.view(np.int64)).cuda()
    torch.cuda.synchronize()
    opt_server.aes_encrypt(d_rk, d_st2)
    opt_server.synchronize()
    assert sha(d_st2.cpu().numpy()) == sha(out)
    # round trip on a few blocks (decrypt costs 1.9x, BASELINE configs[4] path)
    back = opt_server.aes_decrypt(rk, out[:4].copy())
    for i in range(4):
        assert c.decrypt_u128(back[i]) == IV + i


def test_32_block_decrypt_param_opt(opt, opt_server):
    """BASELINE configs[4] shard size: 32 blocks per GPU through aes_decrypt (inverse S-Box + 4-LUT inverse MixColumns,
    4,096 bits per launch): decrypt(encrypt(x)) == x for every block, and the ciphertexts fed in are AES ciphertexts"""
    import torch

    c = opt.client
    key = c.key
    rk = opt_server.aes_key_expansion(c.encrypt_u128(key))
    n = 32
    pts = [(IV + 0x9E3779B97F4A7C15 * i) & ((1 << 128) - 1) for i in range(n)]
    states = np.stack([c.encrypt_u128(v) for v in pts])
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_st = torch.from_numpy(states.view(np.int64)).cuda()
    torch.cuda.synchronize()
    opt_server.aes_encrypt(d_rk, d_st)
    opt_server.synchronize()
    mid = d_st.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert c.decrypt_u128(mid[i]) == aes_clear.aes128_encrypt_block(key, pts[i]), "block %d" % i
    opt_server.aes_decrypt(d_rk, d_st)
    opt_server.synchronize()
    out = d_st.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert c.decrypt_u128(out[i]) == pts[i], "block %d" % i
    bits, ph = c.decrypt_bits(out, return_phase=True)
    err = (ph - (bits.astype(np.uint64) << np.uint64(63))).astype(np.int64)
    assert np.abs(err).max() < 1 << 59


def test_ctr_counter_add_param_opt(opt, opt_server):
    c = opt.client
    st = np.stack([c.encrypt_u128(IV)] * 2)
    opt_server.add_scalar(st, [0x1FF, 0xFFFFFFFF])
    assert c.decrypt_u128(st[0]) == (IV + 0x1FF) % (1 << 128)
    assert c.decrypt_u128(st[1]) == (IV + 0xFFFFFFFF) % (1 << 128)


def test_1024_ctr_blocks_on_one_gpu(opt, opt_server):
    """BASELINE configs[3] total size (1,024 CTR blocks) on ONE GPU: 131,072 bits per round, processed in four
    workspace chunks; every block must decrypt to AES-CTR (the 8-GPU run shards the same stream 128 per GPU)."""
    import torch

    c, p = opt.client, opt.params
    key = c.key
    rk = opt_server.aes_key_expansion(c.encrypt_u128(key))
    n = 1024
    bits = np.zeros((n, 16, 8), dtype=np.uint8)
    for i in range(n):
        v = (IV + i) & ((1 << 128) - 1)
        for byte in range(16):
            bv = (v >> (8 * (15 - byte))) & 0xFF
            bits[i, byte] = [(bv >> j) & 1 for j in range(8)]
    states = c.encrypt_bits(bits)                                   # one call: [1024][16][8][kN+1]
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_st = torch.from_numpy(states.view(np.int64)).cuda()
    del states
    torch.cuda.synchronize()
    opt_server.aes_encrypt(d_rk, d_st)
    opt_server.synchronize()
    out = d_st.cpu().numpy().view(np.uint64)
    got = c.decrypt_bytes(out)                                      # [1024][16]
    for i in range(n):
        want = aes_clear.aes128_encrypt_block(key, (IV + i) & ((1 << 128) - 1))
        assert [int(x) for x in got[i]] == [(want >> (8 * (15 - b))) & 0xFF for b in range(16)], "block %d" % i


def test_k2_launch_forms_agree_at_full_size(opt):
    """BASELINE configs[2] launches 16,384 bits of blind rotation at a time: 2,560 six-ciphertext + 256 four-ciphertext workgroups of the
    paired kernel (kern_blindrot_pair.h; fheaes_k2_launch_plan form 2).  The oracle would need minutes for that batch; the size-independent
    property is that neither the cut of a batch into launches and workgroups NOR THE KERNEL that runs it changes a single word: the same
    rows in 1,024-bit launches (256 four-ciphertext paired units), in a 4,096-bit launch (512 six + 256 four), in 768-bit launches -- which
    take the OTHER throughput kernel (kern_blindrot16.h: three ciphertexts per 256-thread workgroup, written a round earlier, compared
    with the oracle up to 520 bits in test_gpu_stages.py) -- and, for the first 256 rows, in the latency form (one ciphertext per
    512-thread workgroup) give identical outputs."""
    import ctypes as C

    import torch

    from tfhe_aes_amd import _native

    lib = _native.load_library()
    def form(m):
        f, um, ut = C.c_int(), C.c_uint64(), C.c_uint64()
        rm, rt = C.c_uint32(), C.c_uint32()
        assert lib.fheaes_k2_launch_plan(m, 256, 4, C.byref(f), C.byref(um), C.byref(rm), C.byref(ut), C.byref(rt)) == 0
        return f.value
    assert (form(16384), form(4096), form(1024), form(768), form(256)) == (2, 2, 2, 1, 0)     # the device-independent plan of each cut below

    p, E = opt.params, opt.engine()
    # ... and what THIS context really launches on the MI355X (fheaes_k2_context_plan: after the occupancy queries): the paired kernel
    # must be placeable here, or every number published under its name would be another kernel's
    got = [E.k2_plan(m_) for m_ in (16384, 4096, 1024, 768, 256)]
    assert [g["form"] for g in got] == [2, 2, 2, 1, 0]
    assert got[0]["kernel"].startswith("blind_rotate_pair_kernel") and (got[0]["units_main"], got[0]["r_main"], got[0]["units_tail"], got[0]["r_tail"]) == (2560, 6, 256, 4)
    assert got[3]["kernel"].startswith("blind_rotate16_kernel<5,5,8,3,2,true") and got[4]["kernel"].startswith("blind_rotate_latency_kernel<5")
    rng = np.random.default_rng(16384)
    m = 16384
    small = torch.from_numpy(rng.integers(0, 1 << 64, (m, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
    full = torch.empty((m, p.big1), dtype=torch.int64, device="cuda")
    E.cbs_pbs_batch(small, full, m)
    E.synchronize()
    part = torch.empty_like(full)
    for lo in range(0, m, 1024):
        E.cbs_pbs_batch(small[lo:lo + 1024], part[lo:lo + 1024], 1024)
    E.synchronize()
    assert torch.equal(full, part)
    part.zero_()
    for lo in range(0, m, 768):                                     # the whole batch again through the 16-form kernel
        n = min(768, m - lo)
        E.cbs_pbs_batch(small[lo:lo + n], part[lo:lo + n], n)
    E.synchronize()
    assert torch.equal(full, part)
    mid = torch.empty((4096, p.big1), dtype=torch.int64, device="cuda")
    E.cbs_pbs_batch(small[4096:8192], mid, 4096)
    lat = torch.empty((256, p.big1), dtype=torch.int64, device="cuda")
    E.cbs_pbs_batch(small[:256], lat, 256)
    E.synchronize()
    assert torch.equal(full[4096:8192], mid) and torch.equal(full[:256], lat)
    assert sha(full.cpu().numpy()) != sha(np.zeros_like(full.cpu().numpy()))


def test_six_generation_launch_against_the_oracle_word_for_word(opt):
    """A launch of the bench's kind -- 8,192 bits: six generations of the paired kernel on 256 CUs (1,024 six- and 512 four-ciphertext
    units; the bench's 16,384-bit launch is eleven) -- compared with the CPU oracle on EVERY word, not only with the other kernel form
    (test_k2_launch_forms_agree_at_full_size) or at two generations (test_gpu_stages.py: 2,100 bits).  8,192 x 669 external products on
    the host: about 80 s on the GPU box's 16 cores (the oracle's batch entry point is OpenMP-parallel)."""
    p, E = opt.params, opt.engine()
    rng = np.random.default_rng(0xB16)
    m = 8192
    small = rng.integers(0, 1 << 64, (m, p.n + 1), dtype=np.uint64)
    plan = E.k2_plan(m)
    assert plan["form"] == 2 and (plan["units_main"], plan["units_tail"]) == (1024, 512)
    out = np.zeros((m, p.big1), dtype=np.uint64)
    E.cbs_pbs_batch(small, out, m)
    want = opt.oracle.cbs_pbs(small)
    assert out.shape == want.shape and np.array_equal(out, want)


def test_k2_parking_modes_agree(opt):
    """The paired kernel parks half of every accumulator in memory.  Round 6: a slot of the parking slab is CLAIMED from a shared pool
    (one compare-and-swap per workgroup; kern_blindrot_pair.h) or, with fheaes_k2_set_parking(ctx, 0), private to the workgroup.  Where
    the words are parked must not change a single one of them: 16,384-bit and 4,096-bit launches under both settings, and the context
    plan names the setting that ran."""
    import torch

    p, E = opt.params, opt.engine()
    rng = np.random.default_rng(0x9A6B)
    m = 16384
    small = torch.from_numpy(rng.integers(0, 1 << 64, (m, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
    outs = {}
    try:
        for claimed in (True, False):
            E.k2_set_parking(claimed)
            assert E.k2_plan(m)["kernel"].endswith("parking=claimed" if claimed else "parking=private")
            full = torch.empty((m, p.big1), dtype=torch.int64, device="cuda")
            E.cbs_pbs_batch(small, full, m)
            cut = torch.empty_like(full)
            for lo in range(0, m, 4096):
                E.cbs_pbs_batch(small[lo:lo + 4096], cut[lo:lo + 4096], 4096)
            E.synchronize()
            assert torch.equal(full, cut)
            outs[claimed] = full
    finally:
        E.k2_set_parking(True)
    assert torch.equal(outs[True], outs[False])
    assert int((outs[True] != 0).sum().item()) > m * p.big1 // 2


def test_k2_sustained_launches_stay_deterministic(opt):
    """The regression test of round 5's defect.  A queue that runs for tens of seconds is preempted now and then
    (compute wave save / restore: on the MI355X boxes of this pool about every 30 s of sustained load), and its workgroups resume on
    OTHER compute units.  Round 5 indexed the parking slab by the compute unit a workgroup STARTED on: after such a preemption two live
    workgroups shared a slot and 200-260 rows of one launch came out wrong -- never in a short test, always in the driver's 80-second
    bench.  Here: 16,384-bit launches back to back for 75 s, every launch compared with the first on the GPU, plus interleaved 4,096-bit
    launches (the configs[4] shard's size)."""
    import time

    import torch

    p, E = opt.params, opt.engine()
    rng = np.random.default_rng(0x50AC)
    m = 16384
    small = torch.from_numpy(rng.integers(0, 1 << 64, (m, p.n + 1), dtype=np.uint64).view(np.int64)).cuda()
    ref = torch.empty((m, p.big1), dtype=torch.int64, device="cuda")
    out = torch.empty_like(ref)
    E.cbs_pbs_batch(small, ref, m)
    E.synchronize()
    assert E.k2_plan(m)["kernel"].startswith("blind_rotate_pair_kernel")
    t0, launches, bad = time.time(), 0, []
    while time.time() - t0 < 75.0:
        E.cbs_pbs_batch(small, out, m)
        E.synchronize()
        launches += 1
        if not torch.equal(out, ref):
            bad.append((launches, round(time.time() - t0, 1), int((out != ref).any(dim=1).sum().item())))
        if launches % 8 == 0:
            E.cbs_pbs_batch(small[4096:8192], out[4096:8192], 4096)
            E.synchronize()
            if not torch.equal(out[4096:8192], ref[4096:8192]):
                bad.append((launches, round(time.time() - t0, 1), -int((out[4096:8192] != ref[4096:8192]).any(dim=1).sum().item())))
    assert launches > 200
    assert not bad, "launches that differ from the first (launch, seconds, rows): %s" % bad


def test_chained_encrypt_steps_then_decrypt_shard_every_block(opt, opt_server):
    """The driver's bench shape under pytest: 128 blocks through six chained aes_encrypt steps (16,384-bit launches), every block checked
    after the last; then the configs[4] shard -- aes_decrypt of blocks 0..31 (4,096-bit launches) -- twice from the same input: every
    block checked, and the two runs give the same words (server.rs:39-105; the reference asserts every block, client.rs:171)."""
    import torch

    c = opt.client
    key = c.key
    rk = opt_server.aes_key_expansion(c.encrypt_u128(key))
    n, steps = 128, 6
    pts = [(IV + i) & ((1 << 128) - 1) for i in range(n)]
    d_rk = torch.from_numpy(rk.view(np.int64)).cuda()
    d_st = torch.from_numpy(np.stack([c.encrypt_u128(v) for v in pts]).view(np.int64)).cuda()
    torch.cuda.synchronize()
    want = list(pts)
    for _ in range(steps):
        opt_server.aes_encrypt(d_rk, d_st)
        want = [aes_clear.aes128_encrypt_block(key, w) for w in want]
    opt_server.synchronize()
    got = c.decrypt_bytes(d_st.cpu().numpy().view(np.uint64))
    wrong = [i for i in range(n) if [int(v) for v in got[i]] != [(want[i] >> (8 * (15 - b))) & 0xFF for b in range(16)]]
    assert not wrong, "blocks wrong after %d chained encrypt steps: %s" % (steps, wrong)
    shas = []
    for _ in range(2):
        d4 = d_st[:32].clone()
        torch.cuda.synchronize()
        opt_server.aes_decrypt(d_rk, d4)
        opt_server.synchronize()
        host = d4.cpu().numpy().view(np.uint64)
        shas.append(sha(host))
        got = c.decrypt_bytes(host)
        back = [aes_clear.aes128_decrypt_block(key, w) for w in want[:32]]
        wrong = [i for i in range(32) if [int(v) for v in got[i]] != [(back[i] >> (8 * (15 - b))) & 0xFF for b in range(16)]]
        assert not wrong, "blocks wrong after aes_decrypt: %s" % wrong
    assert shas[0] == shas[1]


def test_random_key_round_trips_param_opt(opt, opt_server):
    """main.rs:120-141 at the reference's parameter set on the GPU: ten random (AES key, plaintext) pairs, one block each, key expansion ->
    encrypt -> decrypt, both directions checked against FIPS-197 arithmetic (`test_verify`, client.rs:147-175) -- eight of them under
    the session's evaluation keys, and, as the reference draws a NEW client per pair, two under freshly generated FHE keys (a fresh
    `Client`, a fresh `Server` with its own engine context and upload)."""
    from tfhe_aes_amd.client import Client

    rng = np.random.default_rng(0x120141)

    def round_trip(c, srv, key, pt):
        rk = srv.aes_key_expansion(c.encrypt_u128(key))
        assert np.array_equal(c.decrypt_bytes(rk), np.array(aes_clear.expand_key(key), dtype=np.uint8))
        st = c.encrypt_u128(pt)[None]
        enc = srv.aes_encrypt(rk, st.copy())
        assert c.decrypt_u128(enc[0]) == aes_clear.aes128_encrypt_block(key, pt)
        dec = srv.aes_decrypt(rk, enc.copy())
        assert c.decrypt_u128(dec[0]) == pt

    for _ in range(8):
        round_trip(opt.client, opt_server, int.from_bytes(rng.bytes(16), "big"), int.from_bytes(rng.bytes(16), "big"))
    for _ in range(2):
        key, pt = int.from_bytes(rng.bytes(16), "big"), int.from_bytes(rng.bytes(16), "big")
        c2 = Client(1, pt, key, params=opt.params, seed=int.from_bytes(rng.bytes(8), "big"))
        srv2 = Server(c2.server_keys(), device=0)            # Server::new(public_key, sks, wopbs_key): its own context, its own upload
        try:
            round_trip(c2, srv2, key, pt)
        finally:
            srv2.engine.close()
