// kern_linear.h -- K6: the linear AES layers on LWE vectors (XOR == uint64 wrapping add of the
// 1-bit-at-the-MSB encodings) and the small data-movement kernels of the schedule.
//   mix_columns (+ShiftRows)  src/server/encrypt/mix_columns.rs:4-78
//   shift_rows                src/server/encrypt/shift_rows.rs:5-21
//   inv_mix_columns           src/server/decrypt/inv_mix_columns.rs:4-58
//   inv_shift_rows            src/server/decrypt/inv_shift_rows.rs:5-21
//   add_round_key             src/server/server.rs:278-282
// All of them are one "gather-add": out[blk][byte] = sum_t src[blk][tab.src[byte][t]][tab.lut[byte][t]] (+ rk[byte]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GatherTable {
    int32_t terms;          // 1..4
    int8_t src[16][4];      // source byte index inside the block
    int8_t lut[16][4];      // which LUT output of that byte
};

// src: [n_blocks][16][n_luts][byte_words]; rk: [16][byte_words] or null; out: [n_blocks][16][byte_words]
__global__ __launch_bounds__(256) void gather_add_kernel(const uint64_t *src, uint32_t n_luts, const uint64_t *rk, uint64_t *out,
                                                         uint64_t n_blocks, uint32_t byte_words, const GatherTable tab)
{
    const uint64_t blk = blockIdx.z;
    const uint32_t byte = blockIdx.y;
    const uint64_t *sb = src + blk * 16 * (uint64_t)n_luts * byte_words;
    uint64_t *ob = out + (blk * 16 + byte) * (uint64_t)byte_words;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < byte_words; w += gridDim.x * blockDim.x) {
        uint64_t v = rk ? rk[(uint64_t)byte * byte_words + w] : 0;
        for (int t = 0; t < tab.terms; ++t)
            v += sb[((uint64_t)tab.src[byte][t] * n_luts + tab.lut[byte][t]) * byte_words + w];
        ob[w] = v;
    }
}

// dst[blk][i] += src[i]  (add_round_key with one key set for all blocks)
__global__ __launch_bounds__(256) void add_bcast_kernel(uint64_t *dst, const uint64_t *src, uint64_t words_per_block, uint64_t n_blocks)
{
    uint64_t total = words_per_block * n_blocks;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] += src[i % words_per_block];
}

// dst[i] = a[i] + b[i]
__global__ __launch_bounds__(256) void add2_kernel(uint64_t *dst, const uint64_t *a, const uint64_t *b, uint64_t words)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = a[i] + b[i];
}

// trivial (noise-free) byte constant added to an encrypted byte: body of bit j += ((value >> j) & 1) << 63
__global__ void add_const_byte_kernel(uint64_t *byte_ct, uint32_t lwe_words, uint32_t value)
{
    int j = threadIdx.x;
    if (j < 8) byte_ct[(uint64_t)j * lwe_words + lwe_words - 1] += (uint64_t)((value >> j) & 1u) << 63;
}

// Builds the radix inputs of add_scalar (server.rs:216-222): in[blk][0..8) = state[blk][byte] and, for
// bits == 9, in[blk][8] = carry[blk]
__global__ __launch_bounds__(256) void pack9_kernel(const uint64_t *state, const uint64_t *carry, uint64_t *in9,
                                                    uint32_t byte, uint32_t lwe_words, uint64_t n_blocks, uint32_t bits)
{
    const uint64_t blk = blockIdx.y;
    const uint64_t *sb = state + (blk * 16 + byte) * 8ull * lwe_words;
    const uint64_t *cb = carry + blk * lwe_words;
    uint64_t *o = in9 + blk * (uint64_t)bits * lwe_words;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < bits * lwe_words; w += gridDim.x * blockDim.x)
        o[w] = (w < 8 * lwe_words) ? sb[w] : cb[w - 8 * lwe_words];
}

// Scatter the results of one add_scalar step: res [n_blocks][2][bits][lwe]: LUT 0 blocks 0..7 -> state byte,
// LUT 1 block 0 -> carry
__global__ __launch_bounds__(256) void unpack_sum_carry_kernel(const uint64_t *res, uint32_t bits, uint64_t *state, uint64_t *carry,
                                                               uint32_t byte, uint32_t lwe_words, uint64_t n_blocks)
{
    const uint64_t blk = blockIdx.y;
    const uint64_t *rb = res + blk * 2ull * bits * lwe_words;
    uint64_t *sb = state + (blk * 16 + byte) * 8ull * lwe_words;
    uint64_t *cb = carry + blk * lwe_words;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < 9 * lwe_words; w += gridDim.x * blockDim.x) {
        if (w < 8 * lwe_words) sb[w] = rb[w];
        else cb[w - 8 * lwe_words] = rb[(uint64_t)bits * lwe_words + (w - 8 * lwe_words)];
    }
}

// LUTs of add_scalar (server.rs:181-197, :225-248) for every block: luts[blk][2][bits][512];
// addend[blk] = the counter byte added at this position.  bits = 8: f=(x+c)%256, g = x+c>255;
// bits = 9: x = byte | carry<<8.
__global__ __launch_bounds__(256) void counter_lut_kernel(uint64_t *luts, const uint8_t *addend, uint32_t bits, uint64_t n_blocks)
{
    const uint64_t blk = blockIdx.y;
    const uint32_t c = addend[blk];
    uint64_t *L = luts + blk * 2ull * bits * 512;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * bits * 512; e += gridDim.x * blockDim.x) {
        uint32_t which = e / (bits * 512), rem = e % (bits * 512);
        uint32_t bit = rem / 512, idx = rem % 512;
        uint32_t x = idx & ((1u << bits) - 1);
        uint32_t s = (x & 0xFF) + ((bits == 9) ? ((x >> 8) & 1) : 0) + c;
        uint32_t val = which == 0 ? (s & 0xFF) : (s > 255 ? 1u : 0u);
        L[e] = (uint64_t)((val >> bit) & 1u) << 63;
    }
}

// ---- seeded evaluation keys (fheaes_upload_keys_seeded) -------------------------------------------------------------
// Mask word j of key ciphertext q of key `tag` = 64-bit word j % 8 of ChaCha20 block j / 8 under (public mask key,
// nonce = (tag, q)): the counter-based stream of csrc/client.c (RFC 8439 block function; kept identical here).
struct MaskKey {
    uint32_t k[8];
};

#define FHEAES_QR(a, b, c, d)                                                   \
    a += b; d ^= a; d = (d << 16) | (d >> 16); c += d; b ^= c; b = (b << 12) | (b >> 20); \
    a += b; d ^= a; d = (d << 8) | (d >> 24);  c += d; b ^= c; b = (b << 7) | (b >> 25)

__device__ __forceinline__ void fheaes_chacha20_block(const MaskKey &key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t (&out)[16])
{
    const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                            key.k[4], key.k[5], key.k[6], key.k[7], counter, n0, n1, n2};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll 1
    for (int r = 0; r < 10; ++r) {
        FHEAES_QR(x[0], x[4], x[8], x[12]); FHEAES_QR(x[1], x[5], x[9], x[13]);
        FHEAES_QR(x[2], x[6], x[10], x[14]); FHEAES_QR(x[3], x[7], x[11], x[15]);
        FHEAES_QR(x[0], x[5], x[10], x[15]); FHEAES_QR(x[1], x[6], x[11], x[12]);
        FHEAES_QR(x[2], x[7], x[8], x[13]); FHEAES_QR(x[3], x[4], x[9], x[14]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

// out [n_cts][mask_words + body_words] <- regenerated masks | bodies [n_cts][body_words].  One thread per (ciphertext,
// 8-word block of its mask); the bodies are copied by the threads of the last block(s).
__global__ __launch_bounds__(256) void expand_masks_kernel(uint64_t *out, const uint64_t *bodies, uint64_t n_cts, uint32_t mask_words,
                                                           uint32_t body_words, MaskKey key, uint32_t tag)
{
    const uint32_t ct_words = mask_words + body_words;
    const uint32_t mask_blocks = (mask_words + 7) / 8, body_blocks = (body_words + 7) / 8;
    const uint32_t blocks = mask_blocks + body_blocks;
    const uint64_t total = n_cts * blocks;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = i / blocks;
        const uint32_t bi = (uint32_t)(i - q * blocks);
        uint64_t *o = out + q * ct_words;
        if (bi < mask_blocks) {
            uint32_t w[16];
            fheaes_chacha20_block(key, bi, tag, (uint32_t)q, (uint32_t)(q >> 32), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t j = bi * 8 + e;
                if (j < mask_words) o[j] = (uint64_t)w[2 * e] | ((uint64_t)w[2 * e + 1] << 32);
            }
        } else {
            const uint32_t b0 = (bi - mask_blocks) * 8;
            for (uint32_t e = 0; e < 8 && b0 + e < body_words; ++e) o[mask_words + b0 + e] = bodies[q * body_words + b0 + e];
        }
    }
}
