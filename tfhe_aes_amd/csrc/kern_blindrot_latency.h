// kern_blindrot_latency.h -- K2 for SMALL batches (one AES block, key expansion, the counter-add chain).
//
// The throughput kernel (kern_blindrot16.h) walks the L decomposition levels one after the other so that three
// ciphertexts fit a workgroup; with at most a few hundred bits in flight the GPU is mostly empty and what counts
// is the length of the dependent chain of one iteration.  Here ONE ciphertext owns a 512-thread workgroup and all
// L x K1 digit polynomials of an iteration are transformed at once:
//   LDS                     the accumulator (K1 x 512 torus words) LIVES here: read rotated and unrotated by the digit
//                           groups, updated in place by the owner groups at the end of the iteration;
//   groups 0..L*K1-1        group (l, p) rebuilds d_p = acc_p * X^t - acc_p from LDS, peels the decomposition down
//                           to its level l and transforms that digit polynomial into tile (l, p);
//   all 512 threads         multiply-accumulate: thread = (Fourier point, share of the K1 output columns); the whole
//                           L*K1-row chain (same order as everywhere: least significant level first, rows ascending);
//   groups 0..K1-1          inverse transform of output polynomial c, accumulate into LDS.
// Round 2: the chain of an iteration was dominated by exposed round trips, not by arithmetic -- 25 dependent GGSW
// row fetches with two rows in flight (~600 cycles each), and one LDS table read in flight per twiddle multiply.  Now
//   * the K1 rows of the first level of the multiply-accumulate are requested at the TOP of the iteration (they do not
//     depend on data) and land during the rotation / decomposition / transform; afterwards one level (K1 rows) stays in flight;
//   * the multiply-accumulate is split by output column over all 512 threads (3 + 2 columns), halving its length and
//     the registers a row occupies; key rows come through raw buffer loads with scalar row offsets;
//   * twiddle reads are batched eight at a time, one step ahead of their use (fft_dev.h, nega_*_batched);
//   * the accumulator no longer has a register copy (64 VGPRs, and the publish step with its barrier, are gone).
// 4 workgroup barriers and 2 transform passes per iteration (11 and 6 in the throughput kernel).  Same arithmetic as
// the throughput kernel: results are bit-identical (tests/test_gpu_stages.py::test_k2_blind_rotation covers both).
#pragma once
#include <type_traits>
#include "fft_dev.h"
#include "kern_extprod.h"

#define BL_THREADS 512
#ifndef BL_L2_PREFETCH
#define BL_L2_PREFETCH 1
#endif

template <int K1, int LEVELS, int BASE_LOG>
__global__ __launch_bounds__(BL_THREADS, 2) void blind_rotate_latency_kernel(const ExtProdArgs A)
{
    constexpr int ROWS = LEVELS * K1;
    constexpr int CA = (K1 + 1) / 2, CB = K1 - CA;        // output columns of threads 0..255 / 256..511
    static_assert(ROWS <= BL_THREADS / 16, "one lane group per digit polynomial");
    constexpr int LDS_DOUBLES = 2 * 2 * FHE_H + ROWS * GROUP_TILE_DOUBLES + K1 * FHE_N;
    __shared__ __attribute__((aligned(16))) double lds_all[LDS_DOUBLES];
    double2 *psi = reinterpret_cast<double2 *>(lds_all);                                       // tables first: 16-bit offsets reach them
    double2 *tw = psi + FHE_H;
    double *lds = lds_all + 2 * 2 * FHE_H;                                                      // ROWS tiles
    uint64_t *accs = reinterpret_cast<uint64_t *>(lds + ROWS * GROUP_TILE_DOUBLES);            // [K1][512]

    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool transform = g < ROWS;                  // group (l_idx, p): row index g = k*K1 + p, k = 0 is the least significant level
    const int kk = transform ? g / K1 : 0;            // how many levels to peel before ours
    const int p_own = transform ? g % K1 : 0;
    const bool owner = g < K1;                        // output polynomial g is inverse-transformed by this group
    const bool half_b = tid >= 256;                   // multiply-accumulate role: columns CA..K1-1 (wave-uniform)
    const FftConsts fc = A.fc;

    if (tid < FHE_H) {
        psi[tid] = A.psi[tid];
        tw[tid] = A.tw[tid];
    }

    const uint64_t inst = blockIdx.x;                 // one ciphertext per workgroup, grid = count
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    // accumulator init, straight into LDS
    {
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
        for (int j = tid; j < K1 * FHE_N; j += BL_THREADS) {
            const int p = j >> 9, c = j & 511;
            const int e = ((c - t) & 511) + t;
            const uint64_t v = ((e >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            accs[j] = (p == K1 - 1) ? v : 0;
        }
    }
    __syncthreads();

    constexpr unsigned GGSW_BYTES = LEVELS * K1 * K1 * FHE_H * 16;
    const __amdgpu_buffer_rsrc_t bsk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(A.ggsw), 0, (int)(A.iters * GGSW_BYTES), 0x00020000);
    // chain level k (0 = least significant) is GGSW storage level LEVELS-1-k; byte offset of its row p, column 0
    auto row_bytes = [&](int k, int p) -> unsigned { return (unsigned)(((LEVELS - 1 - k) * K1 + p) * K1) * (FHE_H * 16); };
    uint64_t a_next = lwe[0];
#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif

    unsigned pf_sink = 0;
    for (uint32_t it = 0; it < A.iters; ++it) {
        const int t = mod_switch_1024(a_next);
        a_next = lwe[it + 1];                          // one iteration ahead (the last one reads the body: unused)
#ifdef BL_ABL_SAMEKEY
        const unsigned g_bytes = 0;                    // developer ablation (wrong results): every iteration reads GGSW 0, L2-resident
#else
        const unsigned g_bytes = it * GGSW_BYTES;
#endif
        int tq = tid;
        asm volatile("" : "+v"(tq));                   // addresses below are recomputed from this, not kept across iterations
        const int bq_ = tq & 15, mp = tq & 255;
        double *tile = lds + (transform ? (tq >> 4) : 0) * GROUP_TILE_DOUBLES;

        // ---- 0. the first rows of this iteration's multiply-accumulate: no data dependence, request them now ----------
        double2 bq[K1][CA];                            // one level of rows in flight
        const unsigned col_bytes = half_b ? (unsigned)CA * (FHE_H * 16) : 0u;      // wave-uniform
        auto load_row = [&](int k, int p, double2 (&dst)[CA]) {
#pragma unroll
            for (int c = 0; c < CA; ++c)
                if (c < CB || !half_b) dst[c] = ep_key_load(bsk_rsrc, (unsigned)mp * 16u, g_bytes + row_bytes(k, p) + col_bytes + (unsigned)c * (FHE_H * 16));
        };
#pragma unroll
        for (int p = 0; p < K1; ++p) load_row(0, p, bq[p]);
        EP_STAMP(0);

#if BL_L2_PREFETCH
        // ---- 0b. the last wave holds no digit polynomial and would only wait at the barrier: it walks the NEXT iteration's
        //      GGSW (one dword per 128-byte line) so that the 512 KB come from HBM into this XCD's L2 an iteration ahead of
        //      their use; in a small batch every workgroup is at the same iteration and the first touch of a GGSW is otherwise a
        //      DRAM round trip on the multiply-accumulate's critical path.  Workgroups are dealt to the 8 XCDs round-robin, so
        //      the up to 16 workgroups of an XCD share the walk.
        if (tid >= BL_THREADS - 64 && it + 1 < A.iters) {
            const unsigned nshare = gridDim.x >= 128 ? 16u : gridDim.x >= 64 ? 8u : gridDim.x >= 32 ? 4u : gridDim.x >= 16 ? 2u : 1u;
            const unsigned mine = (blockIdx.x >> 3) % nshare;
            const unsigned lane = (unsigned)tid & 63u;
            unsigned sink = 0;
            for (unsigned line = mine + nshare * lane; line < GGSW_BYTES / 128; line += nshare * 64)
                sink ^= __builtin_amdgcn_raw_buffer_load_b32(bsk_rsrc, line * 128u, g_bytes + GGSW_BYTES, 0);
            pf_sink ^= sink;
        }
#endif
        // ---- 1. every (level, polynomial) group: rotate, subtract, peel to its level, transform -----------------------
        if (transform) {
            const uint64_t *src = accs + (size_t)p_own * FHE_N;
            double xr[16], xi[16];
            double2 w0[8], w1[8];
            fft_tw_load8(w0, psi, bq_, 16);            // lands during the rotation
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                int j0 = 16 * a + bq_;
                int s0 = (j0 - t) & 511, s1 = s0 ^ 256;
                uint64_t v0 = src[s0], v1 = src[s1];
                if (((s0 + t) >> 9) & 1) v0 = (uint64_t)0 - v0;
                if (((s1 + t) >> 9) & 1) v1 = (uint64_t)0 - v1;
                v0 -= src[j0]; v1 -= src[j0 + 256];
                uint32_t s_lo, s_hi;
                int d0 = decompose_first<BASE_LOG, LEVELS>(v0, s_lo);
                int d1 = decompose_first<BASE_LOG, LEVELS>(v1, s_hi);
                for (int q = 0; q < kk; ++q) {        // wave-divergent trip count only between groups of different level
                    d0 = decompose_next<BASE_LOG>(s_lo);
                    d1 = decompose_next<BASE_LOG>(s_hi);
                }
                xr[a] = (double)d0; xi[a] = (double)d1;
            }
            EP_STAMP(1);
            nega_fwd_batched(xr, xi, w0, w1, psi, tw, tile, bq_, fc);
            EP_STAMP(2);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (bq_ + 16 * k2)) = v;
            }
        }
        EP_STAMP(3);
        wg_barrier_lds_only();                         // digits visible; the key loads stay in flight
        EP_STAMP(4);
        // ---- 2. multiply-accumulate: all 512 threads, one Fourier point and a share of the columns each ---------------
        double fr[CA], fi[CA];
#pragma unroll
        for (int c = 0; c < CA; ++c) { fr[c] = 0.0; fi[c] = 0.0; }
#pragma unroll 1
        for (int k = 0; k < LEVELS; ++k) {
            const double *dl = lds + (size_t)k * K1 * GROUP_TILE_DOUBLES + 2 * mp;      // digits of chain level k
            double2 dn = *reinterpret_cast<const double2 *>(dl);
#pragma unroll
            for (int p = 0; p < K1; ++p) {
                double2 bv[CA];
#pragma unroll
                for (int c = 0; c < CA; ++c) bv[c] = bq[p][c];
                if (k + 1 < LEVELS) load_row(k + 1, p, bq[p]);               // the same row of the next level, one level ahead
                const double2 d = dn;
                if (p + 1 < K1) dn = *reinterpret_cast<const double2 *>(dl + (p + 1) * GROUP_TILE_DOUBLES);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    if (c < CB || !half_b) {
                        fr[c] = __builtin_fma(d.x, bv[c].x, fr[c]);
                        fr[c] = __builtin_fma(-d.y, bv[c].y, fr[c]);
                        fi[c] = __builtin_fma(d.x, bv[c].y, fi[c]);
                        fi[c] = __builtin_fma(d.y, bv[c].x, fi[c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        EP_STAMP(5);
        wg_barrier_lds_only();                         // all digits consumed: tiles 0..K1-1 may take the products
        EP_STAMP(6);
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            if (c < CB || !half_b) {
                double2 v; v.x = fr[c]; v.y = fi[c];
                *reinterpret_cast<double2 *>(lds + ((half_b ? CA : 0) + c) * GROUP_TILE_DOUBLES + 2 * mp) = v;
            }
        }
        EP_STAMP(7);
        wg_barrier_lds_only();
        EP_STAMP(8);
        // ---- 3. owner groups: inverse transform of output polynomial g, accumulate into the LDS accumulator ------------
        if (owner) {
            double xr[16], xi[16];
            double2 w0[8], w1[8];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ + 16 * k2));
                xr[k2] = v.x; xi[k2] = v.y;
            }
            wave_lds_sync();
            nega_inv_batched(xr, xi, w0, w1, psi, tw, tile, bq_, fc);
            uint64_t *mine = accs + (size_t)(tq >> 4) * FHE_N + bq_;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                mine[16 * a] += torus_from_double(xr[a]);
                mine[256 + 16 * a] += torus_from_double(xi[a]);
            }
        }
        EP_STAMP(9);
        wg_barrier_lds_only();                         // accumulator complete before the next iteration's rotation reads it
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 8 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- sample extract coefficient 0, from the LDS accumulator -------------------------------------
    {
        const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
        uint64_t *o = A.out + inst * (big + 1);
        for (int j = tid; j < (K1 - 1) * FHE_N; j += BL_THREADS) {
            const int p = j >> 9, c = j & 511;
            const uint64_t v = accs[j];
            if (c == 0) o[(uint64_t)p * FHE_N] = v; else o[(uint64_t)p * FHE_N + FHE_N - c] = (uint64_t)0 - v;
        }
        if (tid == 0) o[big] = accs[(K1 - 1) * FHE_N] + A.post_add;
        if (pf_sink == 0x9e3779b9u && A.count == 0) o[0] = pf_sink;      // keeps the prefetch loads alive; never true
    }
}
