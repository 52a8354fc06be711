"""Pins the oracle's primitives: gadget decomposition, modulus switch (A.4), the canonical FFT
against an exact schoolbook product, the external product against exact integer arithmetic."""
import numpy as np
import pytest

from oracle import oracle as orc
from tfhe_aes_amd import PARAM_TOY, _native

M64 = (1 << 64) - 1


def _decompose_restated(x, b, level):
    """SURVEY.md Appendix A.3, digit list index 0 = level 1"""
    r = 64 - b * level
    st = ((x >> r) + ((x >> (r - 1)) & 1)) & ((1 << (b * level)) - 1)
    out = [0] * level
    for l in range(level - 1, -1, -1):
        d = st & ((1 << b) - 1)
        st >>= b
        carry = ((((d - 1) & M64) | st) & d) >> (b - 1)
        st += carry
        out[l] = d - (carry << b)
    return out


@pytest.mark.parametrize("b,level", [(8, 5), (2, 6), (12, 3), (15, 1)])
def test_decomposition(b, level):
    rng = np.random.default_rng(b * 100 + level)
    xs = [int(v) for v in rng.integers(0, 1 << 64, 300, dtype=np.uint64)]
    r = 64 - b * level
    xs += [0, 1, M64, 1 << 63, (1 << 63) - 1, (1 << r) - 1, 1 << (r - 1), (1 << (r - 1)) - 1,
           ((1 << (b - 1)) << r), (((1 << (b - 1)) | (1 << (2 * b - 1 if level > 1 else 0))) << r) & M64, M64 - (1 << (r - 1)) + 1]
    for x in xs:
        d = [int(v) for v in orc.decompose(x, b, level)]
        assert d == _decompose_restated(x, b, level)
        assert all(-(1 << (b - 1)) <= v <= (1 << (b - 1)) for v in d)
        closest = (((x >> r) + ((x >> (r - 1)) & 1)) << r) & M64
        recomposed = sum(v << (64 - b * (l + 1)) for l, v in enumerate(d)) & M64
        assert recomposed == closest


def _decompose_offset_restated(x, b, level):
    """canonical form v3 (external products): closest representable, then the offset rule of the original TFHE library:
    digit_l = ((x' + sum_l (B/2) 2^(64 - b(l+1))) >> (64 - b(l+1))) mod B - B/2"""
    r = 64 - b * level
    z = (x + ((1 << (r - 1)) if r > 0 else 0)) & M64
    for l in range(level):
        z = (z + ((1 << (b - 1)) << (64 - b * (l + 1)))) & M64
    return [((z >> (64 - b * (l + 1))) & ((1 << b) - 1)) - (1 << (b - 1)) for l in range(level)]


@pytest.mark.parametrize("b,level", [(8, 5), (15, 1), (8, 2), (4, 8)])
def test_offset_decomposition_of_the_external_products(b, level):
    """the rule the external products use since round 5: same recomposed value as the tfhe-rs rule (the closest representable),
    digits in [-B/2, B/2), identical digits wherever the tfhe-rs rule produces no digit of magnitude B/2"""
    rng = np.random.default_rng(b * 1000 + level)
    r = 64 - b * level
    xs = [int(v) for v in rng.integers(0, 1 << 64, 400, dtype=np.uint64)]
    half = 1 << (b - 1)
    # ties at every level, with and without the next digit's top bit, carry ripples through digits of B - 1, the wrap at the top
    for l in range(level):
        sh = 64 - b * (l + 1)
        xs += [(half << sh) & M64, ((half << sh) | (half << (sh + b))) & M64 if l else (half << sh) & M64, (((1 << b) - 1) << sh | (half << (sh - b) if sh >= b + r else 0)) & M64]
    xs += [0, 1, M64, 1 << 63, (1 << 63) - 1, (1 << r) - 1 if r else 0, M64 - ((1 << (r - 1)) if r else 0) + 1 & M64]
    same = 0
    for x in xs:
        d = [int(v) for v in orc.decompose_offset(x, b, level)]
        assert d == _decompose_offset_restated(x, b, level)
        assert all(-half <= v < half for v in d)
        closest = ((((x >> r) + ((x >> (r - 1)) & 1)) << r) & M64) if r else x
        assert sum(v << (64 - b * (l + 1)) for l, v in enumerate(d)) & M64 == closest
        t = _decompose_restated(x, b, level) if r else None
        if t is not None and all(abs(v) != half for v in t):
            assert d == t
            same += 1
    assert same > 150


def test_mod_switch():
    assert orc.mod_switch(0) == 0
    assert orc.mod_switch(1 << 54) == 1
    assert orc.mod_switch((1 << 53) - 1) == 0 and orc.mod_switch(1 << 53) == 1
    assert orc.mod_switch(M64) == 0            # rounds up to 1024 == 0
    assert orc.mod_switch(1 << 63) == 512
    assert orc.mod_switch((1 << 62) + (1 << 63)) == 768


def test_twiddles_accurate_and_shared_with_the_engine():
    t = orc.twiddles()
    j = np.arange(512)
    assert np.abs(t[:, 0] - np.cos(np.pi * j / 512)).max() < 5e-16
    assert np.abs(t[:, 1] - np.sin(np.pi * j / 512)).max() < 5e-16
    assert t[0, 0] == 1.0 and t[0, 1] == 0.0 and t[256, 0] == 0.0 and t[256, 1] == 1.0 and t[128, 0] == t[128, 1]
    # the engine builds its own table from the same specification: must agree bit for bit
    assert np.array_equal(_native.get_twiddles().view(np.uint64), t.view(np.uint64))


def test_fft_product_matches_exact_schoolbook():
    rng = np.random.default_rng(3)
    for base_half in (128, 2048, 16384):          # digit ranges of PBS, PFKS, CBS gadgets
        small = rng.integers(-base_half, base_half + 1, 512)
        tor = rng.integers(0, 1 << 64, 512, dtype=np.uint64)
        err = (orc.negacyclic_mul_fft(small, tor) - orc.negacyclic_mul_exact(small, tor)).astype(np.int64)
        # f64 rounding: |product| <= base_half * 2^63 * 512, relative 2^-53 per op, a few log-steps
        assert np.abs(err).max() < base_half * 512 * (1 << 63) * 2.0 ** -50
    # small * small is exact
    a = rng.integers(-128, 129, 512)
    bb = rng.integers(0, 1 << 20, 512, dtype=np.uint64)
    assert np.array_equal(orc.negacyclic_mul_fft(a, bb), orc.negacyclic_mul_exact(a, bb))
    # monomials: X^511 * X = -1
    x1 = np.zeros(512, dtype=np.int64); x1[1] = 1
    t = np.zeros(512, dtype=np.uint64); t[511] = 5
    assert orc.negacyclic_mul_fft(x1, t)[0] == np.uint64(M64 - 4)


def test_external_product_against_exact_arithmetic():
    """acc += GGSW (x) d with the FFT path  vs  sum of exact products of the decomposed digits"""
    p = PARAM_TOY
    k1, level, b = p.k + 1, p.pbs_level, p.pbs_base_log
    rng = np.random.default_rng(11)
    ggsw = rng.integers(0, 1 << 64, (level, k1, k1, 512), dtype=np.uint64)
    d = rng.integers(0, 1 << 64, (k1, 512), dtype=np.uint64)
    acc0 = rng.integers(0, 1 << 64, (k1, 512), dtype=np.uint64)
    got = orc.external_product_add(p, level, b, ggsw, d, acc0)
    want = acc0.copy()
    for r in range(k1):
        digs = np.array([orc.decompose_offset(int(x), b, level) for x in d[r]], dtype=np.int64)     # [512][level] (the external products' rule)
        for l in range(level):
            for c in range(k1):
                want[c] += orc.negacyclic_mul_exact(digs[:, l], ggsw[l, r, c])
    err = (got - want).astype(np.int64)
    assert np.abs(err).max() < 2.0 ** 30       # f64 FFT noise, far below the gadget's 2^24-granular payloads * noise budget
    assert np.abs(err).max() > 0               # and it IS an approximate product (same regime as tfhe-rs' fft64)
