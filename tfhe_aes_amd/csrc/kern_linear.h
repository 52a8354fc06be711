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
// Mask word j of key ciphertext q of key `tag` = the j-th output of the splitmix64 sequence started at
// mix(mask_seed, tag, q): the public, counter-based stream of csrc/client.c (kept identical here).
__device__ __host__ __forceinline__ uint64_t fheaes_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ uint64_t fheaes_mask_word(uint64_t mask_seed, uint64_t tag, uint64_t q, uint64_t j)
{
    const uint64_t base = fheaes_mix64(mask_seed ^ (tag * 0xD6E8FEB86659FD93ULL) ^ (q * 0xA24BAED4963EE407ULL));
    return fheaes_mix64(base + (j + 1) * 0x9E3779B97F4A7C15ULL);
}

// out [n_cts][mask_words + body_words] <- regenerated masks | bodies [n_cts][body_words]   (HBM-bound: one 8-byte store per word)
__global__ __launch_bounds__(256) void expand_masks_kernel(uint64_t *out, const uint64_t *bodies, uint64_t n_cts, uint32_t mask_words,
                                                           uint32_t body_words, uint64_t mask_seed, uint64_t tag)
{
    const uint32_t ct_words = mask_words + body_words;
    const uint64_t total = n_cts * ct_words;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = i / ct_words;
        const uint32_t j = (uint32_t)(i - q * ct_words);
        out[i] = j < mask_words ? fheaes_mask_word(mask_seed, tag, q, j) : bodies[q * body_words + (j - mask_words)];
    }
}
