// kern_blindrot32.h -- K2, blind rotation of the circuit-bootstrap PBS + sample extract,
// throughput form for large batches.  Same arithmetic, bit for bit, as kern_extprod.h / the oracle.
//
// One 512-thread workgroup = 16 lane groups of 32 lanes; group g < R*K1 owns polynomial p = g % K1 of
// ciphertext r = g / K1 of the workgroup's R ciphertexts, 8 complex points (16 coefficients) per lane
// (fft32_dev.h).  Per-lane state is half that of the 16-lane form, and the accumulator itself is parked
// in HBM/L2 between the end of one iteration and the end of the next (own-lane 16-byte stores and loads,
// issued a whole inverse transform ahead of their use): the level loop holds decomposition state (16) +
// multiply-accumulate sums (<= 36) + transform working set (32) VGPRs, the kernel fits 128 VGPRs and FOUR
// waves share each SIMD (two workgroups per CU), which is what keeps the f64 vector unit busy while
// other waves sit in LDS trips, barriers or key loads.
//
// Per iteration: the accumulator goes to the group's LDS tile, comes back rotated (X^t) minus itself in
// the transform's input layout, is decomposed level by level (streaming 32-bit state); each level's digit
// polynomial is transformed (3 register steps, 2 LDS trips) into the tile in natural point order; then all
// 512 threads switch roles: thread = (Fourier point, half of the output columns) multiplies the R x K1
// transformed digits with its share of the K1 x K1 GGSW entries of that level streamed from L2, so each
// 16-byte key element fetched serves R ciphertexts.  After the last level the sums return through the
// tiles to the owning groups for the inverse transform.
#pragma once
#include "fft_dev.h"
#include "fft32_dev.h"   /* next to this file */
#include "kern_extprod.h"

#define BR32_THREADS 512
#define BR32_GROUPS 16
#ifndef BR32_PARK
#define BR32_PARK 1          /* accumulator parked in global memory during the level loop */
#endif
#ifndef BR32_PREFETCH
#define BR32_PREFETCH 2      /* GGSW rows in flight per multiply-accumulate thread */
#endif
#ifndef BR32_MAC_PRIO
#define BR32_MAC_PRIO 1
#endif
#ifndef BR32_MIN_WAVES
#define BR32_MIN_WAVES 4     /* waves per SIMD the register budget is sized for (two 512-thread workgroups per CU) */
#endif
#ifndef BR32_PAD_CPLX
#define BR32_PAD_CPLX 0      /* developer ablation: extra LDS so that only one workgroup fits a CU */
#endif
#define BR32_LDS_CPLX (BR32_GROUPS * FFT32_TILE_CPLX + 2 * FHE_H + 8 + BR32_PAD_CPLX)

__device__ __forceinline__ size_t br32_park_words(uint64_t workgroups) { return (size_t)workgroups * 8 * BR32_THREADS * 2; }

// The thread index, made opaque to the optimiser.  Every LDS / table address in the kernel is (a base that takes 1-3
// integer instructions from the lane index) + (a constant); without this the compiler computes all of them once,
// keeps them live across the whole 669-iteration loop and spills them.
__device__ __forceinline__ int br32_opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

template <int K1, int LEVELS, int BASE_LOG, int R>
__global__ __launch_bounds__(BR32_THREADS, BR32_MIN_WAVES) void blind_rotate32_kernel(const ExtProdArgs A)
{
    static_assert(R * K1 <= BR32_GROUPS, "too many polynomials for 16 lane groups");
    constexpr int CA = (K1 + 1) / 2;      // output columns of the multiply-accumulate threads 0..255; threads 256..511 take the rest
    constexpr int CB = K1 - CA;
    __shared__ __attribute__((aligned(256))) Fft32Cplx lds[BR32_LDS_CPLX];
    Fft32Cplx *psi = lds + BR32_GROUPS * FFT32_TILE_CPLX;
    Fft32Cplx *tw = psi + FHE_H;
    Fft32Cplx *w16 = tw + FHE_H;

    const int tid = threadIdx.x;
    const int g = tid >> 5, L = tid & 31;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    Fft32Consts fc; fc.c1 = A.fc.c1; fc.s1 = A.fc.s1; fc.h = A.fc.h;

    if (tid < FHE_H) {
        double2 v = A.psi[tid]; psi[tid].x = v.x; psi[tid].y = v.y;
        v = A.tw[tid]; tw[tid].x = v.x; tw[tid].y = v.y;
    }
    if (tid < 8) {
        // w16^m exactly as the canonical dft16 spells its twiddles
        double wr = 1.0, wi = 0.0;
        switch (tid) {
        case 1: wr = fc.c1; wi = fc.s1; break;
        case 2: wr = fc.h; wi = fc.h; break;
        case 3: wr = fc.s1; wi = fc.c1; break;
        case 4: wr = 0.0; wi = 1.0; break;
        case 5: wr = -fc.s1; wi = fc.c1; break;
        case 6: wr = -fc.h; wi = fc.h; break;
        case 7: wr = -fc.c1; wi = fc.s1; break;
        default: break;
        }
        w16[tid].x = wr; w16[tid].y = wi;
    }

    uint64_t inst = (uint64_t)blockIdx.x * R + r_own;
    const bool valid = inst < A.count;
    if (!valid) inst = A.count - 1;
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    // ---- accumulator: lane (kk, blk) holds coefficients 16 a + kk and 256 + 16 a + kk, a = bitrev4(8 blk + q),
    //      i.e. j0 = f32_c_out(L) + 32 bitrev3(q) ----
    uint64_t lo[8], hi[8];
    {
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j0 = f32_c_out(L) + 32 * f32_bitrev3(q), j1 = j0 + 256;
            const int e0 = ((j0 - t) & 511) + t, e1 = ((j1 - t) & 511) + t;
            const uint64_t v0 = ((e0 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            const uint64_t v1 = ((e1 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            lo[q] = (p_own == K1 - 1) ? v0 : 0;
            hi[q] = (p_own == K1 - 1) ? v1 : 0;
        }
    }
#if BR32_PARK
    ulonglong2 *park = reinterpret_cast<ulonglong2 *>(A.park) + (size_t)blockIdx.x * 8 * BR32_THREADS;   // wave-uniform base
#endif
    wg_barrier_lds_only();   // tables visible

    constexpr size_t GGSW_STRIDE = (size_t)LEVELS * K1 * K1 * FHE_H;
    const bool half_b = tid >= 256;           // ... and which share of the output columns (wave-uniform)
    uint64_t a_next = lwe[0];
#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif

    for (uint32_t it = 0; it < A.iters; ++it) {
        // the lane index is made opaque once per iteration: every address below is (a base that takes 1-3 integer
        // instructions from it) + (a constant), recomputed where it is used instead of being kept in registers
        // across the whole loop
        const int t = mod_switch_1024(a_next);
        a_next = lwe[it + 1];                                   // one iteration ahead (the last one reads the body: unused)
        const double2 *G = A.ggsw + (size_t)it * GGSW_STRIDE;   // wave-uniform
        EP_STAMP(11);

        // ---- accumulator -> tile (natural coefficient order) [-> parking slot] ----------------------------
        uint32_t st_lo[8], st_hi[8];
        double xr[8], xi[8];
        {
            const int tq = br32_opaque_tid();
            const int Lq = tq & 31;
            Fft32Cplx *tile = lds + (tq >> 5) * FFT32_TILE_CPLX;
            {
                uint64_t *st_w = reinterpret_cast<uint64_t *>(tile) + f32_c_out(Lq);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    st_w[32 * f32_bitrev3(q)] = lo[q];
                    st_w[32 * f32_bitrev3(q) + 256] = hi[q];
                }
            }
#if BR32_PARK
#pragma unroll
            for (int q = 0; q < 8; ++q) { ulonglong2 v; v.x = lo[q]; v.y = hi[q]; (park + q * BR32_THREADS)[(unsigned)tq] = v; }
#endif
            wave_lds_sync();
            // ---- d = acc * X^t - acc in the transform's input layout, first (least significant) digit ----------
            const uint64_t *stage = reinterpret_cast<const uint64_t *>(tile);
            const int ja = f32_a_base(Lq);
            const uint64_t *own = stage + ja;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j0 = ja + 64 * (e >> 1) + 8 * (e & 1);
                const int s0 = (j0 - t) & 511, s1 = s0 ^ 256;
                uint64_t v0 = stage[s0], v1 = stage[s1];
                if (((s0 + t) >> 9) & 1) v0 = (uint64_t)0 - v0;
                if (((s1 + t) >> 9) & 1) v1 = (uint64_t)0 - v1;
                v0 -= own[64 * (e >> 1) + 8 * (e & 1)]; v1 -= own[64 * (e >> 1) + 8 * (e & 1) + 256];
                xr[e] = (double)decompose_first<BASE_LOG, LEVELS>(v0, st_lo[e]);
                xi[e] = (double)decompose_first<BASE_LOG, LEVELS>(v1, st_hi[e]);
            }
            wave_lds_sync();
        }
        EP_STAMP(0);

        double fr[R][CA], fi[R][CA];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < CA; ++c) { fr[r][c] = 0.0; fi[r][c] = 0.0; }

        auto level_body = [&](const int l, const bool tiles_busy) {
            const int tq = br32_opaque_tid();
            const int Lq = tq & 31, mp = tq & 255;
            Fft32Cplx *tile = lds + (tq >> 5) * FFT32_TILE_CPLX;
            // ---- forward transform of this level's digit polynomial ------------------------------------------
            {
                const Fft32Cplx *pb = psi + f32_a_base(Lq);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const Fft32Cplx w = pb[64 * (e >> 1) + 8 * (e & 1)];
                    f32_cmul(xr[e], xi[e], w.x, w.y);
                }
            }
            f32_step_a<false>(xr, xi, w16, Lq);
            EP_STAMP(2);
            if (tiles_busy) wg_barrier_lds_only();               // every thread is done reading the previous level's digits
            EP_STAMP(3);
            f32_trip1_write(xr, xi, tile, Lq);
            wave_lds_sync();
            f32_trip1_read(xr, xi, tile, Lq);
            f32_step_b<false>(xr, xi, tw, w16, fc, Lq);
            wave_lds_sync();
            EP_STAMP(4);
            f32_trip2_write(xr, xi, tile, Lq);
            wave_lds_sync();
            f32_trip2_read(xr, xi, tile, Lq);
            f32_step_c<false>(xr, xi, fc);
            wave_lds_sync();
            {
                Fft32Cplx *ob = tile + f32_c_out(Lq);             // natural point order
#pragma unroll
                for (int q = 0; q < 8; ++q) { Fft32Cplx v; v.x = xr[q]; v.y = xi[q]; ob[32 * f32_bitrev3(q)] = v; }
            }
            // ---- multiply-accumulate role -------------------------------------------------------------------
            const double2 *Gl = G + (size_t)l * K1 * K1 * FHE_H;    // wave-uniform; row pointers stay scalar, the lane adds 16 * mp
            auto mac = [&](auto c0_tag, auto nc_tag) {
                constexpr int C0 = decltype(c0_tag)::value, NC = decltype(nc_tag)::value;
                constexpr int PF = (K1 < BR32_PREFETCH) ? K1 : BR32_PREFETCH;
                double2 bq[PF][NC];
#pragma unroll
                for (int p = 0; p < PF; ++p)
#pragma unroll
                    for (int c = 0; c < NC; ++c) bq[p][c] = (Gl + (p * K1 + C0 + c) * FHE_H)[(unsigned)mp];
                __builtin_amdgcn_sched_barrier(0);
                EP_STAMP(5);
                wg_barrier_lds_only();                           // digits of all groups visible; key loads stay in flight
                EP_STAMP(6);
#if BR32_MAC_PRIO
                __builtin_amdgcn_s_setprio(BR32_MAC_PRIO);
#endif
#pragma unroll
                for (int p = 0; p < K1; ++p) {
                    double2 bv[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) bv[c] = bq[p % PF][c];
                    if (p + PF < K1) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) bq[p % PF][c] = (Gl + ((p + PF) * K1 + C0 + c) * FHE_H)[(unsigned)mp];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const Fft32Cplx d = lds[(r * K1 + p) * FFT32_TILE_CPLX + mp];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            fr[r][c] = __builtin_fma(d.x, bv[c].x, fr[r][c]);
                            fr[r][c] = __builtin_fma(-d.y, bv[c].y, fr[r][c]);
                            fi[r][c] = __builtin_fma(d.x, bv[c].y, fi[r][c]);
                            fi[r][c] = __builtin_fma(d.y, bv[c].x, fi[r][c]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#if BR32_MAC_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                EP_STAMP(7);
            };
            if (!half_b) mac(std::integral_constant<int, 0>{}, std::integral_constant<int, CA>{});
            else mac(std::integral_constant<int, CA>{}, std::integral_constant<int, CB>{});
        };

        level_body(LEVELS - 1, false);
#pragma unroll 1
        for (int l = LEVELS - 2; l >= 0; --l) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xr[e] = (double)decompose_next<BASE_LOG>(st_lo[e]);
                xi[e] = (double)decompose_next<BASE_LOG>(st_hi[e]);
            }
            EP_STAMP(1);
            level_body(l, true);
        }

        const int tq = br32_opaque_tid();
        const int Lq = tq & 31, mp = tq & 255;
        Fft32Cplx *tile = lds + (tq >> 5) * FFT32_TILE_CPLX;
#if BR32_PARK
        // the parked accumulator comes back during the products exchange and the inverse transform
        ulonglong2 pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pk[q] = (park + q * BR32_THREADS)[(unsigned)tq];
#endif
        // ---- sums back to the owning groups, inverse transform, accumulate ----------------------------------
        wg_barrier_lds_only();       // every thread is done reading the last level's digits
        if (!half_b) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < CA; ++c) { Fft32Cplx v; v.x = fr[r][c]; v.y = fi[r][c]; lds[(r * K1 + c) * FFT32_TILE_CPLX + mp] = v; }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c) { Fft32Cplx v; v.x = fr[r][c]; v.y = fi[r][c]; lds[(r * K1 + CA + c) * FFT32_TILE_CPLX + mp] = v; }
        }
        wg_barrier_lds_only();
        {
            const Fft32Cplx *ib = tile + f32_a_base(Lq);                       // row = k2, column = k1: p = k1 + 16 k2
#pragma unroll
            for (int e = 0; e < 8; ++e) { const Fft32Cplx v = ib[64 * (e >> 1) + 8 * (e & 1)]; xr[e] = v.x; xi[e] = v.y; }
        }
        wave_lds_sync();
        EP_STAMP(8);
        f32_step_a<true>(xr, xi, w16, Lq);
        f32_trip1_write(xr, xi, tile, Lq);
        wave_lds_sync();
        f32_trip1_read(xr, xi, tile, Lq);
        f32_step_b<true>(xr, xi, tw, w16, fc, Lq);
        wave_lds_sync();
        f32_trip2_write(xr, xi, tile, Lq);
        wave_lds_sync();
        f32_trip2_read(xr, xi, tile, Lq);
        f32_step_c<true>(xr, xi, fc);
        wave_lds_sync();
        EP_STAMP(9);
        const Fft32Cplx *pc = psi + f32_c_out(Lq);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const Fft32Cplx w = pc[32 * f32_bitrev3(q)];
            f32_cmulc(xr[q], xi[q], w.x, w.y);
#if BR32_PARK
            lo[q] = pk[q].x + torus_from_double(xr[q]);
            hi[q] = pk[q].y + torus_from_double(xi[q]);
#else
            lo[q] += torus_from_double(xr[q]);
            hi[q] += torus_from_double(xi[q]);
#endif
        }
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 8 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- sample extract coefficient 0 -----------------------------------------------------
    if (owner && valid) {
        const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
        uint64_t *o = A.out + inst * (big + 1);
        if (p_own < K1 - 1) {
            uint64_t *om = o + (uint64_t)p_own * FHE_N;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j0 = f32_c_out(L) + 32 * f32_bitrev3(q), j1 = j0 + 256;
                if (j0 == 0) om[0] = lo[q]; else om[FHE_N - j0] = (uint64_t)0 - lo[q];
                om[FHE_N - j1] = (uint64_t)0 - hi[q];
            }
        } else if (L == 0) {
            o[big] = lo[0] + A.post_add;
        }
    }
}
