// kern_keyswitch.h -- the two integer key-switching kernels:
//   K1  LWE keyswitch big -> small ("bit extraction")
//   K3  private functional packing keyswitch LWE -> GLWE, k+1 keys (8 a13)
// Both are  out[m][o] = init[o] (+ body) - sum_{i,l} digit_l(in[m][i]) * KEY[i][l][o]   in uint64 wrapping
// arithmetic, i.e. a (ciphertexts x (inputs*levels)) by ((inputs*levels) x columns) product.
//
// Tiling: a 256-thread workgroup owns 256 key columns (one per thread: every key load is a
// coalesced 2 KB row segment) and TM ciphertexts; digits are decomposed once per workgroup into
// LDS and read back as wave-uniform broadcasts, so each 8-byte key element fetched feeds TM
// multiply-adds.  Digits are stored with an offset (d' = d + B/2 >= 0) so the products are plain
// unsigned v_mad_u64_u32; the offset is undone by init[o] = (B/2) * sum_{i,l} KEY[i][l][o],
// which is computed once at key upload (keysum_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_dev.h"

#define KS_THREADS 256
#define KS_KI 16          /* input elements decomposed per LDS chunk */

struct KeyswitchArgs {
    const uint64_t *in;        // [m][in_stride]
    uint64_t in_stride;
    uint32_t n_in;             // elements that are decomposed
    int32_t body_index;        // >= 0: in[m][body_index] is added to column body_col (K1); < 0: none
    uint32_t body_col;
    const uint64_t *key;       // [z][n_in][LEVELS][ncols]
    uint64_t key_z_stride;
    const uint64_t *init;      // [z][ncols]
    uint32_t ncols;
    uint64_t *out;             // [m][out_stride], column o of key z at out[m*out_stride + z*out_z_stride + o]
    uint64_t out_stride;
    uint64_t out_z_stride;
    uint64_t m;
};

template <int BASE_LOG, int LEVELS, int TM>
__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(const KeyswitchArgs A)
{
    static_assert(TM % 8 == 0, "TM must be a multiple of 8");
    __shared__ __attribute__((aligned(16))) uint16_t dig[KS_KI * LEVELS * TM];   // [i][l][m]
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.x * KS_THREADS + tid;
    const bool col_ok = col < A.ncols;
    const uint32_t colc = col_ok ? col : A.ncols - 1;
    const uint64_t m0 = (uint64_t)blockIdx.y * TM;
    const int z = blockIdx.z;
    const uint64_t *key = A.key + (uint64_t)z * A.key_z_stride + colc;

    uint64_t acc_lo[TM], acc_hi[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) { acc_lo[m] = 0; acc_hi[m] = 0; }

    for (uint32_t i0 = 0; i0 < A.n_in; i0 += KS_KI) {
        __syncthreads();
        // cooperative decomposition of TM x KS_KI inputs
        for (int e = tid; e < TM * KS_KI; e += KS_THREADS) {
            int m = e / KS_KI, ii = e % KS_KI;
            uint64_t mm = m0 + m;
            uint32_t i = i0 + ii;
            int d[LEVELS];
            if (mm < A.m && i < A.n_in) decompose_all<BASE_LOG, LEVELS>(A.in[mm * A.in_stride + i], d);
            else {
#pragma unroll
                for (int l = 0; l < LEVELS; ++l) d[l] = -(1 << (BASE_LOG - 1));   // offset digit 0: contributes nothing
            }
#pragma unroll
            for (int l = 0; l < LEVELS; ++l) dig[(ii * LEVELS + l) * TM + m] = (uint16_t)(d[l] + (1 << (BASE_LOG - 1)));
        }
        __syncthreads();
        const uint32_t ilim = (A.n_in - i0 < KS_KI) ? (A.n_in - i0) : KS_KI;
        for (uint32_t ii = 0; ii < ilim; ++ii) {
#pragma unroll
            for (int l = 0; l < LEVELS; ++l) {
                uint64_t kv = key[((uint64_t)(i0 + ii) * LEVELS + l) * A.ncols];
                uint32_t klo = (uint32_t)kv, khi = (uint32_t)(kv >> 32);
                const uint4 *dp = reinterpret_cast<const uint4 *>(dig + (ii * LEVELS + l) * TM);
#pragma unroll
                for (int q = 0; q < TM / 8; ++q) {
                    uint4 w = dp[q];
                    uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        uint32_t d0 = ww[h] & 0xFFFFu, d1 = ww[h] >> 16;
                        acc_lo[q * 8 + 2 * h] += (uint64_t)d0 * klo;
                        acc_hi[q * 8 + 2 * h] += (uint64_t)d0 * khi;
                        acc_lo[q * 8 + 2 * h + 1] += (uint64_t)d1 * klo;
                        acc_hi[q * 8 + 2 * h + 1] += (uint64_t)d1 * khi;
                    }
                }
            }
        }
    }
    if (!col_ok) return;
    const uint64_t init = A.init[(uint64_t)z * A.ncols + col];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        uint64_t mm = m0 + m;
        if (mm >= A.m) break;
        uint64_t v = init - (acc_lo[m] + (acc_hi[m] << 32));
        if (A.body_index >= 0 && col == A.body_col) v += A.in[mm * A.in_stride + (uint32_t)A.body_index];
        A.out[mm * A.out_stride + (uint64_t)z * A.out_z_stride + col] = v;
    }
}

// init[z][o] = (B/2) * sum over rows of KEY[z][row][o]
__global__ void keysum_kernel(const uint64_t *key, uint64_t key_z_stride, uint32_t rows, uint32_t ncols, uint64_t half_base, uint64_t *init)
{
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const uint64_t *k = key + (uint64_t)blockIdx.y * key_z_stride + col;
    uint64_t s = 0;
    for (uint32_t r = 0; r < rows; ++r) s += k[(uint64_t)r * ncols];
    init[(uint64_t)blockIdx.y * ncols + col] = s * half_base;
}
