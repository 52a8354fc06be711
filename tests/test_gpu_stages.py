"""Parity tests proper: every kernel of the hot path, called through the C ABI on the MI355X, against the CPU
oracle on the same seeded inputs -- bit-exact (integer stages AND the f64 FFT stages, whose arithmetic is
canonical: see oracle/fheaes_oracle.c header)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tfhe_aes_amd import _native

pytestmark = pytest.mark.gpu


def _inputs(kit, m, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, m).astype(np.uint8)
    return kit.client.encrypt_bits(bits), bits


@pytest.mark.parametrize("which", ["toy", "opt"])
def test_twiddles_and_bsk_fourier(which, request):
    kit = request.getfixturevalue(which)
    p, E = kit.params, kit.engine()
    assert np.array_equal(_native.get_twiddles().view(np.uint64), orc.twiddles().view(np.uint64))
    bsk = kit.keys.bsk.reshape(p.n, p.pbs_level, p.k + 1, p.k + 1, 512)
    for i in (0, p.n // 2, p.n - 1):
        assert np.array_equal(E.read_bsk_fourier(i).view(np.uint64), orc.polys_to_fourier(bsk[i]).view(np.uint64))


@pytest.mark.parametrize("which,m", [("toy", 1), ("toy", 33), ("toy", 100), ("opt", 1), ("opt", 35), ("opt", 200)])
def test_k1_keyswitch(which, m, request):
    # opt 200: four 64-ciphertext tiles (grid.y = 4), ragged last one, every word against the oracle
    kit = request.getfixturevalue(which)
    p, E = kit.params, kit.engine()
    x, _ = _inputs(kit, m, 10 + m)
    out = np.zeros((m, p.n + 1), dtype=np.uint64)
    E.keyswitch_batch(x, out, m)
    assert np.array_equal(out, kit.oracle.keyswitch(x))


@pytest.mark.parametrize("which,m", [("toy", 1), ("toy", 8), ("toy", 21), ("toy", 300), ("toy", 530),
                                     ("opt", 1), ("opt", 7), ("opt", 300), ("opt", 520), ("opt", 800), ("opt", 1001), ("opt", 2100)])
def test_k2_blind_rotation(which, m, request):
    # every launch form of engine.hip::launch_cbs_pbs: m <= 256 the one-ciphertext-per-512-thread-workgroup latency kernel
    # (1, 7/8/21 bits: every sharing degree of its L2 walk); up to 768 bits kern_blindrot16.h with 3 (k=4) or 8 (k=1) ciphertexts per
    # 256-thread workgroup, at most one per CU (opt 300 in two-ciphertext units, 520 in three-ciphertext ones, ragged last one; the toy
    # set always); beyond that, at k=4, the paired kernel kern_blindrot_pair.h: opt 800 = 200 four-ciphertext 512-thread units, opt
    # 1001 = 251 of them with a ragged last one (three of its four slots empty), opt 2100 = two full generations of 26 six- and 486 four-ciphertext units (both unit bodies, the home wavefronts, the mirrored groups)
    kit = request.getfixturevalue(which)
    p, E = kit.params, kit.engine()
    x, bits = _inputs(kit, m, 20 + m)
    small = kit.oracle.keyswitch(x)
    out = np.zeros((m, p.big1), dtype=np.uint64)
    E.cbs_pbs_batch(small, out, m)
    assert np.array_equal(out, kit.oracle.cbs_pbs(small))
    # and it means what it should: LWE of bit * 2^(64-15)
    _, ph = kit.client.decrypt_bits(out, return_phase=True)
    delta = 1 << (64 - p.cbs_base_log)
    assert np.abs(ph.astype(np.int64) - bits.astype(np.int64) * delta).max() < delta // 8


@pytest.mark.parametrize("which,m", [("toy", 1), ("toy", 40), ("toy", 300), ("opt", 3), ("opt", 33), ("opt", 300)])
def test_k3_pfpks(which, m, request):
    # the LDS-tiled kernel owns 128 ciphertexts per workgroup: 300 = three tiles (grid.y = 3), ragged last one, with the k = 4
    # column tiling (2,560 columns, out_z_stride) -- every word against the oracle (~80 M multiply-adds per bit on the CPU)
    kit = request.getfixturevalue(which)
    p, E = kit.params, kit.engine()
    rng = np.random.default_rng(30 + m)
    x = rng.integers(0, 1 << 64, (m, p.big1), dtype=np.uint64)         # any LWE words: the kernel is pure integer
    out = np.zeros((m, p.k + 1, (p.k + 1) * 512), dtype=np.uint64)
    E.pfpks_batch(x, out, m)
    assert np.array_equal(out, kit.oracle.pfpks(x))


@pytest.mark.parametrize("which", ["toy", "opt"])
def test_k4_forward_fourier(which, request):
    # the kernel does not depend on the parameter set at N = 512 (one polynomial per 16-lane group); both engines anyway, every word
    # against the oracle, up to 300 polynomials (19 workgroups, ragged last one)
    E = request.getfixturevalue(which).engine()
    rng = np.random.default_rng(4)
    for polys in (1, 15, 16, 17, 130, 300):
        x = rng.integers(0, 1 << 64, (polys, 512), dtype=np.uint64)
        out = np.zeros((polys, 256, 2), dtype=np.float64)
        E.forward_fourier_batch(x, out, polys)
        assert np.array_equal(out.view(np.uint64), orc.polys_to_fourier(x).view(np.uint64))


@pytest.mark.parametrize("which", ["toy", "opt"])
def test_k5_vertical_packing(which, request):
    kit = request.getfixturevalue(which)
    p, E, c, O = kit.params, kit.engine(), kit.client, kit.oracle
    vals = [0x53, 0xE1]
    x = c.encrypt_bytes(vals)
    luts = orc.build_lutset(orc.LUTSET_DEC_MUL)                        # 4 LUTs: 32 instances per byte, ragged vs R
    want, dbg = O.wopbs_batch(x, luts, debug=True)
    ggsw_f = orc.polys_to_fourier(dbg["ggsw"].reshape(2, 8, -1, 512))   # [inputs][bits][(k+1)^2][256][2]
    out = np.zeros_like(want)
    E.vertical_packing_batch(np.ascontiguousarray(ggsw_f), 2, 8, luts, 4, False, out)
    assert np.array_equal(out, want)


def test_k5_vertical_packing_9bit_per_input_luts_param_opt(opt):
    """the add_scalar shape at the reference's parameter set: 9 input bits, 2 LUTs that differ per input (server.rs:216-252);
    exercises the 9-iteration instance of the vertical-packing kernel and the per-input LUT indexing"""
    p, E, c, O = opt.params, opt.engine(), opt.client, opt.oracle
    rng = np.random.default_rng(95)
    n = 2
    bits = rng.integers(0, 2, (n, 9)).astype(np.uint8)
    x = c.encrypt_bits(bits)
    from tfhe_aes_amd.server import gen_lut
    adds = (0x7F, 0xF3)
    luts = np.stack([np.stack([gen_lut(2, 1, 512, 9, lambda v, a=a: ((v & 0xFF) + (v >> 8) + a) % 256),
                               gen_lut(2, 1, 512, 9, lambda v, a=a: 1 if (v & 0xFF) + (v >> 8) + a > 255 else 0)]) for a in adds])
    want, dbg = O.wopbs_batch(x, luts, lut_per_input=True, debug=True)
    out = np.zeros_like(want)
    E.wopbs_batch(x, n, 9, luts, 2, True, out)                         # whole pipeline, bit-exact
    assert np.array_equal(out, want)
    ggsw_f = orc.polys_to_fourier(dbg["ggsw"].reshape(n, 9, -1, 512))
    out2 = np.zeros_like(want)
    E.vertical_packing_batch(np.ascontiguousarray(ggsw_f), n, 9, luts, 2, True, out2)   # K5 alone on the oracle's GGSWs
    assert np.array_equal(out2, want)
    dec = c.decrypt_bits(out)
    for i, a in enumerate(adds):
        v = int(sum(int(bits[i, j]) << j for j in range(9)))
        s_ = (v & 0xFF) + (v >> 8) + a
        assert int(sum(int(dec[i, 0, j]) << j for j in range(8))) == s_ % 256 and int(dec[i, 1, 0]) == (1 if s_ > 255 else 0)


def test_zero_sized_batches_are_noops(toy):
    p, E = toy.params, toy.engine()
    x = np.zeros((1, p.big1), dtype=np.uint64)
    out = np.full((1, p.n + 1), 7, dtype=np.uint64)
    E.keyswitch_batch(x, out, 0)
    assert (out == 7).all()


def test_errors_are_status_codes(toy):
    p = toy.params
    E = _native.Engine(p)                                              # no keys uploaded
    x = np.zeros((1, p.big1), dtype=np.uint64)
    out = np.zeros((1, p.n + 1), dtype=np.uint64)
    with pytest.raises(_native.FheAesError) as e:
        E.keyswitch_batch(x, out, 1)
    assert e.value.code == -2 and "keys" in str(e.value)
    with pytest.raises(ValueError):
        E.upload_keys(toy.keys.ksk[:-1].copy(), toy.keys.bsk, toy.keys.pfpksk)
    E2 = toy.engine()
    luts = orc.build_lutset(orc.LUTSET_SBOX)
    for bits in (0, 17):                                                 # 1..16 are valid (9 < bits: CMUX tree first)
        with pytest.raises(_native.FheAesError) as e:
            E2.wopbs_batch(np.zeros((1, max(bits, 1), p.big1), dtype=np.uint64), 1, bits, luts, 1, False,
                           np.zeros((1, 1, max(bits, 1), p.big1), dtype=np.uint64))
        assert e.value.code == -1
    E.close()


def test_large_batches_are_chunked_consistently(toy):
    """more bits than one workspace chunk (32,768): the chunked call equals the same inputs processed in two calls"""
    p, E, c = toy.params, toy.engine(), toy.client
    n = 4200                                                           # 33,600 bits > MAX_CHUNK_BITS
    rng = np.random.default_rng(5)
    x = c.encrypt_bytes(rng.integers(0, 256, n))
    luts = orc.build_lutset(orc.LUTSET_SBOX)
    whole = np.zeros((n, 1, 8, p.big1), dtype=np.uint64)
    E.wopbs_batch(x, n, 8, luts, 1, False, whole)
    half = n // 2
    parts = np.zeros_like(whole)
    E.wopbs_batch(x[:half], half, 8, luts, 1, False, parts[:half])
    E.wopbs_batch(np.ascontiguousarray(x[half:]), n - half, 8, luts, 1, False, parts[half:])
    assert np.array_equal(whole, parts)
    # spot-check against the oracle and the S-Box
    idx = [0, half - 1, half, n - 1]
    assert np.array_equal(whole[idx], toy.oracle.wopbs_batch(x[idx], luts))


def test_param_opt_golden_hashes_on_gpu(opt, golden):
    """the committed PARAM_OPT hashes (tests/golden/make_golden.py: oracle outputs on seeded inputs, generated in the build container)
    against the HIP kernels on this box: K1 on 35 bits, K2 on 7, K3 on 3, K4, one many_sbox byte -- the oracle need not run here"""
    from conftest import sha
    from tfhe_aes_amd.client import Client

    g = golden["oracle_opt"]
    p, E = opt.params, opt.engine()
    c = Client(1, opt.client.iv, opt.client.key, params=p, seed=g["seed"])           # fresh: same secret keys, encryption stream from 0
    bits = np.array([int(ch) for ch in g["input_bits"]], dtype=np.uint8)
    x = c.encrypt_bits(bits)
    assert sha(x) == g["input"]
    small = np.zeros((35, p.n + 1), dtype=np.uint64)
    E.keyswitch_batch(x, small, 35)
    assert sha(small) == g["keyswitch_35"]
    pbs = np.zeros((7, p.big1), dtype=np.uint64)
    E.cbs_pbs_batch(np.ascontiguousarray(small[:7]), pbs, 7)
    assert sha(pbs) == g["cbs_pbs_7"] and [int(v) for v in pbs.reshape(-1)[:4]] == g["cbs_pbs_first_words"]
    gg = np.zeros((3, p.k + 1, (p.k + 1) * 512), dtype=np.uint64)
    E.pfpks_batch(np.ascontiguousarray(pbs[:3]), gg, 3)
    assert sha(gg) == g["pfpks_3"]
    polys = 3 * (p.k + 1) * (p.k + 1)
    ff = np.zeros((polys, 256, 2), dtype=np.float64)
    E.forward_fourier_batch(np.ascontiguousarray(gg.reshape(polys, 512)), ff, polys)
    assert sha(ff) == g["ggsw_fourier_3"]
    xb = c.encrypt_bytes([0x53])
    assert sha(xb) == g["byte_input"]
    y = np.zeros((1, 3, 8, p.big1), dtype=np.uint64)
    E.many_sbox(xb, 1, False, y)
    assert sha(y) == g["many_sbox_0x53"] and [int(v) for v in y.reshape(-1)[:4]] == g["many_sbox_first_words"]
