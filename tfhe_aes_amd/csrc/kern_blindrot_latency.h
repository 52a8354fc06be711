// kern_blindrot_latency.h -- K2 for SMALL batches (one AES block, key expansion, the counter-add chain).
//
// The throughput kernel (kern_extprod.h) walks the L decomposition levels one after the other so that three
// ciphertexts fit a workgroup; with at most a few hundred bits in flight the GPU is mostly empty and what counts
// is the length of the dependent chain of one iteration (11 barriers, 6 transform passes).  Here ONE ciphertext
// owns a 512-thread workgroup and all L x K1 digit polynomials of an iteration are transformed at once:
//   groups 0..K1-1          own the accumulator polynomials (registers), publish them in LDS every iteration,
//                           and run the K1 inverse transforms;
//   groups 0..L*K1-1        group (l, p) rebuilds d_p = acc_p * X^t - acc_p from LDS, peels the decomposition down
//                           to its level l and transforms that digit polynomial into tile (l, p);
//   threads 0..255          thread t owns Fourier point t and runs the whole L*K1-row multiply-accumulate chain
//                           (same order as everywhere: least significant level first, rows ascending) with the
//                           GGSW rows prefetched two rows ahead.
// 4 workgroup barriers and 2 transform passes per iteration (11 and 6 in the throughput kernel).  Same arithmetic as the throughput kernel: results are
// bit-identical (tests/test_gpu_stages.py::test_k2_blind_rotation covers both).
#pragma once
#include "fft_dev.h"
#include "kern_extprod.h"

#define BL_THREADS 512
#ifndef BL_PREFETCH
#define BL_PREFETCH 2            /* GGSW rows (of K1 entries) in flight per multiply-accumulate thread */
#endif

template <int K1, int LEVELS, int BASE_LOG>
__global__ __launch_bounds__(BL_THREADS, 2) void blind_rotate_latency_kernel(const ExtProdArgs A)
{
    constexpr int ROWS = LEVELS * K1;
    static_assert(ROWS <= BL_THREADS / 16, "one lane group per digit polynomial");
    constexpr int LDS_DOUBLES = ROWS * GROUP_TILE_DOUBLES + K1 * FHE_N + 2 * 2 * FHE_H;
    __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES];
    uint64_t *accs = reinterpret_cast<uint64_t *>(lds + ROWS * GROUP_TILE_DOUBLES);           // [K1][512]
    double2 *psi = reinterpret_cast<double2 *>(lds + ROWS * GROUP_TILE_DOUBLES + K1 * FHE_N);
    double2 *tw = psi + FHE_H;

    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool transform = g < ROWS;                  // group (l_idx, p): row index g = k*K1 + p, k = 0 is the least significant level
    const int kk = transform ? g / K1 : 0;            // how many levels to peel before ours
    const int p_own = transform ? g % K1 : 0;
    const bool owner = g < K1;                        // accumulator polynomial g lives in this group's registers
    double *tile = lds + (transform ? g : 0) * GROUP_TILE_DOUBLES;
    const FftConsts fc = A.fc;

    if (tid < FHE_H) {
        psi[tid] = A.psi[tid];
        tw[tid] = A.tw[tid];
    }

    const uint64_t inst = blockIdx.x;                 // one ciphertext per workgroup, grid = count
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    uint64_t lo[16], hi[16];
    {
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            int j0 = 16 * a + b, j1 = j0 + 256;
            int e0 = ((j0 - t) & 511) + t, e1 = ((j1 - t) & 511) + t;
            uint64_t v0 = ((e0 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            uint64_t v1 = ((e1 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            lo[a] = (owner && g == K1 - 1) ? v0 : 0;
            hi[a] = (owner && g == K1 - 1) ? v1 : 0;
        }
    }
    __syncthreads();

    constexpr size_t GGSW_STRIDE = (size_t)LEVELS * K1 * K1 * FHE_H;
    // row index in chain order (k = 0 least significant level) -> GGSW storage row (level LEVELS-1-k, polynomial p)
    auto row_ptr = [&](uint32_t it, int row) -> const double2 * {
        const int k = row / K1, p = row % K1;
        return A.ggsw + (size_t)it * GGSW_STRIDE + ((size_t)(LEVELS - 1 - k) * K1 + p) * K1 * FHE_H + (tid & 255);
    };

    for (uint32_t it = 0; it < A.iters; ++it) {
        const int t = mod_switch_1024(lwe[it]);
        // ---- 1. owners publish the accumulator -------------------------------------------------------------
        if (owner) {
            uint64_t *mine = accs + (size_t)g * FHE_N;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                mine[16 * a + b] = lo[a];
                mine[256 + 16 * a + b] = hi[a];
            }
        }
        __syncthreads();
        // ---- 2. every (level, polynomial) group: rotate, subtract, peel to its level, transform ---------------
        if (transform) {
            const uint64_t *src = accs + (size_t)p_own * FHE_N;
            double xr[16], xi[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                int j0 = 16 * a + b, j1 = j0 + 256;
                int s0 = (j0 - t) & 511, s1 = (j1 - t) & 511;
                uint64_t v0 = src[s0], v1 = src[s1];
                if (((s0 + t) >> 9) & 1) v0 = (uint64_t)0 - v0;
                if (((s1 + t) >> 9) & 1) v1 = (uint64_t)0 - v1;
                v0 -= src[j0]; v1 -= src[j1];
                uint32_t s_lo, s_hi;
                int d0 = decompose_first<BASE_LOG, LEVELS>(v0, s_lo);
                int d1 = decompose_first<BASE_LOG, LEVELS>(v1, s_hi);
                for (int q = 0; q < kk; ++q) {        // wave-divergent trip count only between groups of different level
                    d0 = decompose_next<BASE_LOG>(s_lo);
                    d1 = decompose_next<BASE_LOG>(s_hi);
                }
                xr[a] = (double)d0; xi[a] = (double)d1;
            }
            nega_fwd(xr, xi, psi, tw, tile, b, fc);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (b + 16 * k2)) = v;
            }
        }
        // ---- 3. multiply-accumulate: threads 0..255, one Fourier point each; rows prefetched BL_PREFETCH ahead ----
        //         (a 512-thread split by output polynomial doubles the loads in flight on paper, but hipcc then spills
        //          ~370 registers; 2 rows ahead is the measured optimum of this form)
        double fr[K1], fi[K1];
        double2 bq[BL_PREFETCH][K1];
        if (tid < 256) {
#pragma unroll
            for (int r = 0; r < BL_PREFETCH; ++r)
#pragma unroll
                for (int c = 0; c < K1; ++c) bq[r][c] = row_ptr(it, r)[(size_t)c * FHE_H];
        }
        wg_barrier_lds_only();                         // digits visible; the key loads stay in flight
        if (tid < 256) {
#pragma unroll
            for (int c = 0; c < K1; ++c) { fr[c] = 0.0; fi[c] = 0.0; }
#pragma unroll
            for (int row = 0; row < ROWS; ++row) {
                double2 bv[K1];
#pragma unroll
                for (int c = 0; c < K1; ++c) bv[c] = bq[row % BL_PREFETCH][c];
                if (row + BL_PREFETCH < ROWS) {
#pragma unroll
                    for (int c = 0; c < K1; ++c) bq[row % BL_PREFETCH][c] = row_ptr(it, row + BL_PREFETCH)[(size_t)c * FHE_H];
                }
                const double2 d = *reinterpret_cast<const double2 *>(lds + row * GROUP_TILE_DOUBLES + 2 * tid);
#pragma unroll
                for (int c = 0; c < K1; ++c) {
                    fr[c] = __builtin_fma(d.x, bv[c].x, fr[c]);
                    fr[c] = __builtin_fma(-d.y, bv[c].y, fr[c]);
                    fi[c] = __builtin_fma(d.x, bv[c].y, fi[c]);
                    fi[c] = __builtin_fma(d.y, bv[c].x, fi[c]);
                }
            }
        }
        __syncthreads();                               // all digits consumed: tiles 0..K1-1 may take the products
        if (tid < 256) {
#pragma unroll
            for (int c = 0; c < K1; ++c) {
                double2 v; v.x = fr[c]; v.y = fi[c];
                *reinterpret_cast<double2 *>(lds + c * GROUP_TILE_DOUBLES + 2 * tid) = v;
            }
        }
        __syncthreads();
        // ---- 4. owners: inverse transform, accumulate -------------------------------------------------------------
        if (owner) {
            double xr[16], xi[16];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (b + 16 * k2));
                xr[k2] = v.x; xi[k2] = v.y;
            }
            wave_lds_sync();
            nega_inv(xr, xi, psi, tw, tile, b, fc);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                lo[a] += torus_from_double(xr[a]);
                hi[a] += torus_from_double(xi[a]);
            }
        }
        // the next iteration's first barrier (after the owners' stores to `accs`) orders everything else: tiles are
        // written again only in step 2, which follows that barrier
    }

    // ---- sample extract coefficient 0 ---------------------------------------------------------
    if (owner) {
        const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
        uint64_t *o = A.out + inst * (big + 1);
        if (g < K1 - 1) {
            uint64_t *om = o + (uint64_t)g * FHE_N;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                int j0 = 16 * a + b, j1 = j0 + 256;
                if (j0 == 0) om[0] = lo[a]; else om[FHE_N - j0] = (uint64_t)0 - lo[a];
                om[FHE_N - j1] = (uint64_t)0 - hi[a];
            }
        } else if (b == 0) {
            o[big] = lo[0] + A.post_add;
        }
    }
}
