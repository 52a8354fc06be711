"""Clear-text AES-128 and the S-Box LUT functions of the path.

Stands in for (a) the RustCrypto ``aes`` crate the reference verifies against
(/root/reference/src/client/client.rs:166-171) and (b) the reference's tables and GF(2^8)
helpers (src/tables/table.rs, src/server/sbox/sbox.rs:20-42).  Tables are derived from the
field definition (inverse in GF(2^8) mod x^8+x^4+x^3+x+1, then the FIPS-197 affine map), not
copied; tests pin them against FIPS-197 values.
"""
from __future__ import annotations


def gf_mul(a: int, b: int) -> int:
    r = 0
    for _ in range(8):
        if b & 1:
            r ^= a
        hi = a & 0x80
        a = (a << 1) & 0xFF
        if hi:
            a ^= 0x1B
        b >>= 1
    return r


def _make_tables():
    sbox = [0] * 256
    inv = [0] * 256
    for x in range(256):
        y = 0
        if x:
            for c in range(1, 256):
                if gf_mul(x, c) == 1:
                    y = c
                    break
        s = v = y
        for _ in range(4):
            v = ((v << 1) | (v >> 7)) & 0xFF
            s ^= v
        s ^= 0x63
        sbox[x] = s
        inv[s] = x
    return tuple(sbox), tuple(inv)


SBOX, INV_SBOX = _make_tables()
RCON = (0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36)


def mul2(x): return gf_mul(x, 2)
def mul3(x): return gf_mul(x, 3)
def mul9(x): return gf_mul(x, 9)
def mul11(x): return gf_mul(x, 11)
def mul13(x): return gf_mul(x, 13)
def mul14(x): return gf_mul(x, 14)


def expand_key(key: int):
    kb = [(key >> (8 * (15 - i))) & 0xFF for i in range(16)]
    w = [kb[4 * i:4 * i + 4] for i in range(4)]
    for i in range(4, 44):
        t = list(w[i - 1])
        if i % 4 == 0:
            t = t[1:] + t[:1]
            t = [SBOX[b] for b in t]
            t[0] ^= RCON[i // 4 - 1]
        w.append([a ^ b for a, b in zip(w[i - 4], t)])
    return [sum(w[4 * r:4 * r + 4], []) for r in range(11)]


def _shift_rows(s):
    return [s[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4)]


def _inv_shift_rows(s):
    return [s[4 * ((c - r) % 4) + r] for c in range(4) for r in range(4)]


def _mix(s, m):
    out = []
    for c in range(4):
        col = s[4 * c:4 * c + 4]
        for r in range(4):
            v = 0
            for j in range(4):
                v ^= gf_mul(col[j], m[(j - r) % 4])
            out.append(v)
    return out


def aes128_encrypt_block(key: int, block: int) -> int:
    rk = expand_key(key)
    s = [(block >> (8 * (15 - i))) & 0xFF for i in range(16)]
    s = [a ^ b for a, b in zip(s, rk[0])]
    for rnd in range(1, 10):
        s = _mix(_shift_rows([SBOX[b] for b in s]), (2, 3, 1, 1))
        s = [a ^ b for a, b in zip(s, rk[rnd])]
    s = _shift_rows([SBOX[b] for b in s])
    s = [a ^ b for a, b in zip(s, rk[10])]
    return sum(b << (8 * (15 - i)) for i, b in enumerate(s))


def aes128_decrypt_block(key: int, block: int) -> int:
    rk = expand_key(key)
    s = [(block >> (8 * (15 - i))) & 0xFF for i in range(16)]
    s = [a ^ b for a, b in zip(s, rk[10])]
    for rnd in range(9, 0, -1):
        s = [INV_SBOX[b] for b in _inv_shift_rows(s)]
        s = [a ^ b for a, b in zip(s, rk[rnd])]
        s = _mix(s, (14, 11, 13, 9))
    s = [INV_SBOX[b] for b in _inv_shift_rows(s)]
    s = [a ^ b for a, b in zip(s, rk[0])]
    return sum(b << (8 * (15 - i)) for i, b in enumerate(s))
