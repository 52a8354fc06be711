// kern_blindrot16.h -- K2, blind rotation of the circuit-bootstrap PBS + sample extract,
// throughput form for large batches (the same arithmetic and lane mapping as extprod_rotate_kernel in
// kern_extprod.h, which stays the form of K5 and of medium batches; bit-identical results).
//
// What differs from kern_extprod.h, and why (measured with per-phase s_memtime stamps, profiles/r02_*):
//  * the accumulator (64 VGPRs per lane) is dead weight between the rotation at the top of an iteration and the
//    accumulate at its end; the compiler spilled it (252 B/lane of scratch, ~430 GB of HBM traffic per launch, and
//    the reloads sat on the critical path).  Here it is PARKED explicitly: stored once per iteration with
//    coalesced 16-byte stores into a per-workgroup slab, reloaded into registers that are free by then, a whole
//    products exchange + inverse transform ahead of its use.
//  * with only two waves per SIMD nothing hides an LDS round trip but the wave's own instruction stream, and the
//    compiler (at its register limit) kept ONE twiddle read in flight: each of the 31 table reads of a transform
//    exposed ~100 cycles -- the "forward head" phase ran 3.6x longer than its arithmetic.  The 64 registers that
//    parking frees hold two 8-entry twiddle buffers; table reads are issued a whole batch (and a whole DFT16 or
//    decomposition step) ahead of their use.  The transformed digits of the next multiply-accumulate row are read
//    one row ahead the same way.
//  * a burst of 25 key loads blocks the in-order wave for as long as the L1 takes to accept them (~150 cycles each with one
//    workgroup on the CU: 28 % of a lone wave's time).  15 of the 25 GGSW entries of a level are therefore requested one or two
//    at a time BETWEEN the instructions of the transpose and of the second DFT16, into the registers the twiddle buffers have
//    just left; the other 10 right after the digit stores, into the registers of the transform working set, before the
//    exchange barrier.
//  * key rows are addressed as (scalar row pointer) + (16 * point) so no 64-bit vector address arithmetic is issued;
//    LDS addresses are (one base per lane) + constants, recomputed per phase from an opaque lane index so the
//    compiler does not keep dozens of them live across the 669-iteration loop.
#pragma once
#include "fft_dev.h"
#include "kern_extprod.h"

#ifndef BR16_MAC_PRIO
#define BR16_MAC_PRIO 0
#endif
#ifndef BR16_EARLY
#define BR16_EARLY 15      /* GGSW entries (of 25 per level; scaled to K1*K1) requested between the instructions of the transform's
                              second half.  Measured at 16,384 bits: 0 -> 267 ms, 12 -> 267, 13 -> 255, 14 -> 253, 15 -> 251.5, 16 -> 263,
                              20 -> 268, 25 -> 297 (spills past 18); some of them ahead of the tiles-free barrier, or the late ones
                              between the digit stores: no gain. */
#endif
#ifndef BR16_PARK_NT
#define BR16_PARK_NT 0     /* nontemporal parking stores/loads (measured: see DESIGN.md) */
#endif
#define BR16_PARK_WORDS_PER_WG (16 * EP_THREADS * 2 * 2)   /* 16 chunks of 32 bytes per thread: lo[a], hi[a] pairs */

__device__ __forceinline__ int br16_opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

#ifndef BR16_PAD_DOUBLES
#define BR16_PAD_DOUBLES 0     /* developer ablation: extra LDS so that only one workgroup fits a CU */
#endif
// One unit of work = R ciphertexts starting at `inst0`, one workgroup (the body of the kernel below).
template <int K1, int LEVELS, int BASE_LOG, int R>
__device__ __forceinline__ void blind_rotate16_unit(const ExtProdArgs &A, double *lds_all, const uint64_t inst0)
{
    static_assert(R * K1 <= EP_GROUPS, "too many polynomials for 16 lane groups");
    // twiddle tables first: their addresses then fit the 16-bit offset field of the LDS instructions
    double2 *psi = reinterpret_cast<double2 *>(lds_all);
    double2 *tw = psi + FHE_H;
    double *lds = lds_all + 2 * 2 * FHE_H;                        // the 16 group tiles

    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    const FftConsts fc = A.fc;

    psi[tid] = A.psi[tid];
    tw[tid] = A.tw[tid];

    uint64_t inst = inst0 + r_own;
    const bool valid = inst < A.count;
    if (!valid) inst = A.count - 1;
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    // ---- accumulator init (coefficients 16a+b and 256+16a+b in lane b) ------------------------------------------
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
            lo[a] = (p_own == K1 - 1) ? v0 : 0;
            hi[a] = (p_own == K1 - 1) ? v1 : 0;
        }
    }
    ulonglong2 *park = reinterpret_cast<ulonglong2 *>(A.park) + (size_t)blockIdx.x * 16 * EP_THREADS;   // wave-uniform
    __syncthreads();   // tables visible
#if defined(BR16_STAGGER_SLEEP) && BR16_STAGGER_SLEEP > 0
    // developer experiment: start every second generation of workgroups a fraction of a level later, so that the two
    // workgroups of a CU run their key-load phase and their transform phase against each other instead of in phase
    if ((blockIdx.x >> BR16_STAGGER_SHIFT) & 1) {
        for (int i = 0; i < BR16_STAGGER_SLEEP; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif

    constexpr unsigned GGSW_BYTES = LEVELS * K1 * K1 * FHE_H * 16;   // one GGSW of the Fourier BSK
    // the whole Fourier BSK as one raw buffer (< 2^31 bytes for every supported parameter set: checked by the launcher)
    const __amdgpu_buffer_rsrc_t bsk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(A.ggsw), 0, (int)(A.iters * GGSW_BYTES), 0x00020000);
    uint64_t a_next = lwe[0];

#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    for (uint32_t it = 0; it < A.iters; ++it) {
        const int t = mod_switch_1024(a_next);
        a_next = lwe[it + 1];                                    // one iteration ahead (the last one reads the body: unused)
        const unsigned g_bytes = it * GGSW_BYTES;                // wave-uniform byte offset of this iteration's GGSW

        // ---- accumulator -> tile and parking slab; d = acc * X^t - acc; first (least significant) digit -----------
        uint32_t st_lo[16], st_hi[16];
        double xr[16], xi[16];
        double2 w0[8], w1[8];                                     // twiddle buffers
        EP_STAMP(11);
        {
            const int tq = br16_opaque_tid();
            const int bq_ = tq & 15;
            uint64_t *stage = reinterpret_cast<uint64_t *>(lds + (tq >> 4) * GROUP_TILE_DOUBLES);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                stage[16 * a + bq_] = lo[a];
                stage[256 + 16 * a + bq_] = hi[a];
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                ulonglong2 v; v.x = lo[a]; v.y = hi[a];
#if BR16_PARK_NT
                typedef unsigned long long br16_u64x2 __attribute__((ext_vector_type(2)));
                br16_u64x2 nv; nv[0] = v.x; nv[1] = v.y;
                __builtin_nontemporal_store(nv, reinterpret_cast<br16_u64x2 *>(park + a * EP_THREADS) + (unsigned)tq);
#else
                (park + a * EP_THREADS)[(unsigned)tq] = v;
#endif
            }
            wave_lds_sync();
            fft_tw_load8(w0, psi, bq_, 16);                       // psi^(16a+b), a = 0..7: lands during the rotation
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                int j0 = 16 * a + bq_;
                int s0 = (j0 - t) & 511, s1 = s0 ^ 256;
                uint64_t v0 = stage[s0], v1 = stage[s1];
                if (((s0 + t) >> 9) & 1) v0 = (uint64_t)0 - v0;
                if (((s1 + t) >> 9) & 1) v1 = (uint64_t)0 - v1;
                v0 -= lo[a]; v1 -= hi[a];
                xr[a] = (double)decompose_first<BASE_LOG, LEVELS>(v0, st_lo[a]);
                xi[a] = (double)decompose_first<BASE_LOG, LEVELS>(v1, st_hi[a]);
                if ((a & (EP_ROT_CHUNK - 1)) == EP_ROT_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_sync();
        }
        EP_STAMP(0);

        double fr[R][K1], fi[R][K1];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) { fr[r][c] = 0.0; fi[r][c] = 0.0; }

        // One decomposition level.  On entry w0 holds (or is about to receive) psi^(16a+b), a = 0..7, and xr/xi the digits.
        auto level_body = [&](const int l, const bool tiles_busy) {
            const int tq = br16_opaque_tid();
            const int bq_ = tq & 15;
            double *tile = lds + (tq >> 4) * GROUP_TILE_DOUBLES;
            // ---- forward transform: fold + twist, DFT16, twiddle, transpose, DFT16 (fft_dev.h's nega_fwd, with the
            //      table reads batched eight at a time and issued one step ahead) ------------------------------------
            fft_tw_load8(w1, psi, 128 + bq_, 16);                  // a = 8..15
            __builtin_amdgcn_sched_barrier(0);
            fft_tw_mul<false, 8>(xr, xi, w0);
            __builtin_amdgcn_sched_barrier(0);
            fft_tw_load8(w0, tw, 16 + bq_, 16);                    // w256^(k1 b), k1 = 1..8
            __builtin_amdgcn_sched_barrier(0);
            fft_tw_mul<false, 8>(xr + 8, xi + 8, w1);
            __builtin_amdgcn_sched_barrier(0);
            fft_tw_load8(w1, tw, 128 + bq_, 16);                   // k1 = 8..15 (entry 0 unused)
            __builtin_amdgcn_sched_barrier(0);
#ifndef BR16_ABL_NOFFT
            dft16<false>(xr, xi, fc);
#endif
            __builtin_amdgcn_sched_barrier(0);
            fft_tw_mul<false, 8>(xr + 1, xi + 1, w0);
#pragma unroll
            for (int k = 1; k < 8; ++k) cmul(xr[8 + k], xi[8 + k], w1[k].x, w1[k].y);
            EP_STAMP(2);
            const unsigned gl_bytes = g_bytes + (unsigned)l * (K1 * K1 * FHE_H * 16);   // scalar; the lane adds 16 * point
            double2 bm[K1][K1];
            // GGSW entries [from, to) of this level (row-major: the multiply-accumulate consumes them in this order)
            auto key_rows = [&](const int from, const int to) {
#pragma unroll
                for (int q = 0; q < K1 * K1; ++q) {
                    if (q < from || q >= to) continue;
#ifdef BR16_ABL_NOLOAD
                    bm[q / K1][q % K1] = make_double2((double)(tq + q), (double)(tq - q));
#else
                    bm[q / K1][q % K1] = ep_key_load(bsk_rsrc, (unsigned)tq * 16u, gl_bytes + (unsigned)q * (FHE_H * 16));
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // The first BR16_EARLY entries are requested a few at a time BETWEEN the instructions of the transpose and of the
            // second DFT16, into the registers the twiddle buffers have just left: a burst of 25 loads blocks the in-order
            // wave for as long as the L1 takes to accept them (~150 cycles each with one workgroup per CU); spaced out,
            // the same acceptance time passes under the wave's own LDS and vector work.
            constexpr int NE = BR16_EARLY * K1 * K1 / 25, NHOOK = 7;
            auto early = [&](const int h) { key_rows(NE * h / NHOOK, NE * (h + 1) / NHOOK); };
            if (tiles_busy) wg_barrier_lds_only();                // every thread is done reading the previous level's digits
            EP_STAMP(3);
#if defined(BR16_ABL_NOFFT)
            group_transpose(xr, xi, tile, bq_);
            key_rows(0, NE);
#elif defined(BR16_ABL_NOXPOSE)
            dft16<false>(xr, xi, fc);
            key_rows(0, NE);
#else
            {
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(FFT_XPOSE_PRIO);
#endif
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
                    double2 v; v.x = xr[k1]; v.y = xi[k1];
                    *reinterpret_cast<double2 *>(tile + 2 * (k1 * 17 + bq_)) = v;
                    if (NE && k1 == 7) { __builtin_amdgcn_sched_barrier(0); early(0); }
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(1); }
                wave_lds_sync();
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
                    xr[c] = v.x; xi[c] = v.y;
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(2); }
                wave_lds_sync();
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                dft16<false>(xr, xi, fc, [&](const int stage) { if (NE) { __builtin_amdgcn_sched_barrier(0); early(3 + stage); } });
            }
#endif
            EP_STAMP(4);
            // store the transformed digits, then request the remaining GGSW entries of this level into the registers the
            // working set has just left
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (bq_ + 16 * k2)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
            key_rows(NE, K1 * K1);
            EP_STAMP(5);
            wg_barrier_lds_only();                                // digits of all groups visible; key loads stay in flight
            EP_STAMP(6);
            // ---- multiply-accumulate role: thread tq owns Fourier point tq; digits are read one row ahead ---------------
#if BR16_MAC_PRIO
            __builtin_amdgcn_s_setprio(BR16_MAC_PRIO);
#endif
            double2 dn[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dn[r] = *reinterpret_cast<const double2 *>(lds + (r * K1) * GROUP_TILE_DOUBLES + 2 * tq);
#pragma unroll
            for (int p = 0; p < K1; ++p) {
                double2 d[R];
#pragma unroll
                for (int r = 0; r < R; ++r) d[r] = dn[r];
                if (p + 1 < K1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dn[r] = *reinterpret_cast<const double2 *>(lds + (r * K1 + p + 1) * GROUP_TILE_DOUBLES + 2 * tq);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef BR16_ABL_NOMAC
                if (p == 0)
#endif
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int c = 0; c < K1; ++c) {
                        fr[r][c] = __builtin_fma(d[r].x, bm[p][c].x, fr[r][c]);
                        fr[r][c] = __builtin_fma(-d[r].y, bm[p][c].y, fr[r][c]);
                        fi[r][c] = __builtin_fma(d[r].x, bm[p][c].y, fi[r][c]);
                        fi[r][c] = __builtin_fma(d[r].y, bm[p][c].x, fi[r][c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#if BR16_MAC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            EP_STAMP(7);
        };

        level_body(LEVELS - 1, false);
#pragma unroll 1
        for (int l = LEVELS - 2; l >= 0; --l) {
            {
                const int tq = br16_opaque_tid();
                fft_tw_load8(w0, psi, tq & 15, 16);               // lands during the decomposition step
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                xr[a] = (double)decompose_next<BASE_LOG>(st_lo[a]);
                xi[a] = (double)decompose_next<BASE_LOG>(st_hi[a]);
            }
            EP_STAMP(1);
            level_body(l, true);
        }

        // ---- parked accumulator back (lands during the products exchange and the inverse transform) ---------------
        const int tq = br16_opaque_tid();
        const int bq_ = tq & 15;
        double *tile = lds + (tq >> 4) * GROUP_TILE_DOUBLES;
        ulonglong2 pk[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#if BR16_PARK_NT
            typedef unsigned long long br16_u64x2 __attribute__((ext_vector_type(2)));
            br16_u64x2 nv = __builtin_nontemporal_load(reinterpret_cast<const br16_u64x2 *>(park + a * EP_THREADS) + (unsigned)tq);
            pk[a].x = nv[0]; pk[a].y = nv[1];
#else
            pk[a] = (park + a * EP_THREADS)[(unsigned)tq];
#endif
        }
        // ---- products back to the owning groups, inverse transform, accumulate --------------------------------------
        wg_barrier_lds_only();       // every thread is done reading the last level's digits from the tiles
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) {
                double2 v; v.x = fr[r][c]; v.y = fi[r][c];
                *reinterpret_cast<double2 *>(lds + (r * K1 + c) * GROUP_TILE_DOUBLES + 2 * tq) = v;
            }
        wg_barrier_lds_only();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ + 16 * k2));
            xr[k2] = v.x; xi[k2] = v.y;
        }
        fft_tw_load8(w0, tw, 16 + bq_, 16);                       // k1 = 1..8
        fft_tw_load8(w1, tw, 128 + bq_, 16);                      // k1 = 8..15
        wave_lds_sync();
        EP_STAMP(8);
        // inverse transform (fft_dev.h's nega_inv, table reads batched and issued a step ahead)
        dft16<true>(xr, xi, fc);
        __builtin_amdgcn_sched_barrier(0);
        fft_tw_mul<true, 8>(xr + 1, xi + 1, w0);
#pragma unroll
        for (int k = 1; k < 8; ++k) cmulc(xr[8 + k], xi[8 + k], w1[k].x, w1[k].y);
        __builtin_amdgcn_sched_barrier(0);
        fft_tw_load8(w0, psi, bq_, 16);
        fft_tw_load8(w1, psi, 128 + bq_, 16);
        __builtin_amdgcn_sched_barrier(0);
        group_transpose(xr, xi, tile, bq_);
        dft16<true>(xr, xi, fc);
        __builtin_amdgcn_sched_barrier(0);
        fft_tw_mul<true, 8>(xr, xi, w0);
        fft_tw_mul<true, 8>(xr + 8, xi + 8, w1);
        EP_STAMP(9);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            lo[a] = pk[a].x + torus_from_double(xr[a]);
            hi[a] = pk[a].y + torus_from_double(xi[a]);
        }
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 4 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- sample extract coefficient 0 ---------------------------------------------------------------
    if (owner && valid) {
        const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
        uint64_t *o = A.out + inst * (big + 1);
        if (p_own < K1 - 1) {
            uint64_t *om = o + (uint64_t)p_own * FHE_N;
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

// The launch: workgroups 0 .. units_main-1 carry R ciphertexts each, the rest R2 (R2 = 0: none).  With more units than the chip
// has slots (two per CU) the launcher picks both counts so that the total is a whole number of generations and covers the batch
// exactly: 16,384 bits = 5,120 x 3 + 512 x 2 = 11 full generations, instead of 5,462 x 3 whose eleventh generation leaves a third of
// the CUs idle for the length of a full one.  Workgroups are dispatched in index order, so the smaller units form the last generation.
template <int K1, int LEVELS, int BASE_LOG, int R, int R2 = 0>
__global__ __launch_bounds__(EP_THREADS, 2) void blind_rotate16_kernel(const ExtProdArgs A)
{
    __shared__ __attribute__((aligned(16))) double lds_all[EP_LDS_DOUBLES + BR16_PAD_DOUBLES];
    if constexpr (R2 > 0) {
        if (blockIdx.x >= A.units_main) {       // scalar branch
            blind_rotate16_unit<K1, LEVELS, BASE_LOG, R2>(A, lds_all, (uint64_t)A.units_main * R + (uint64_t)(blockIdx.x - A.units_main) * R2);
            return;
        }
    }
    blind_rotate16_unit<K1, LEVELS, BASE_LOG, R>(A, lds_all, (uint64_t)blockIdx.x * R);
}
