// fft_dev.h -- device-side negacyclic f64 FFT (N = 512 -> 256 complex points), gadget
// decomposition and torus conversions for gfx950.
//
// Mapping: one polynomial per GROUP of 16 lanes (4 polynomials per 64-lane wavefront), 16 complex
// points per lane, so the 256-point transform is two in-register DFT16 passes with ONE transpose
// through LDS (a 16 x 17 padded tile per group: conflict-free ds_write_b128 / ds_read_b128).
//
// The arithmetic is the canonical one stated in oracle/fheaes_oracle.c (header) and DESIGN.md:
//   fold+twist by psi^j, DFT16 over the row index (radix-2 DIF), twiddle w256^(k1*b), transpose,
//   DFT16; cmul / cmulc with exactly one fma per component; no contraction elsewhere
//   (the translation unit is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FHE_N 512
#ifndef FFT_XPOSE_PRIO
#define FFT_XPOSE_PRIO 2   /* wave priority while a transpose's LDS writes/reads are being issued (measured -0.7 %) */
#endif
#define FHE_H 256
#define GROUP_TILE_BYTES 4352          /* 16 rows x 17 complex x 16 B */
#define GROUP_TILE_DOUBLES (GROUP_TILE_BYTES / 8)

struct FftConsts {           // w16^1 = (c1, s1), w16^2 = (h, h)
    double c1, s1, h;
};

__device__ __forceinline__ void cmul(double &xr, double &xi, double wr, double wi)
{
    double t = xi * wi;
    double u = xi * wr;
    double re = __builtin_fma(xr, wr, -t);
    double im = __builtin_fma(xr, wi, u);
    xr = re; xi = im;
}

__device__ __forceinline__ void cmulc(double &xr, double &xi, double wr, double wi)
{
    double t = xi * wi;
    double u = xr * wi;
    double re = __builtin_fma(xr, wr, t);
    double im = __builtin_fma(xi, wr, -u);
    xr = re; xi = im;
}

// DFT of length 16 in registers, radix-2 DIF, natural-order output.
// INV = false: kernel e^{+2 pi i a k / 16}; INV = true: conjugate kernel.
// `hook(stage)` runs after butterfly stage 0..3 (a caller interleaves independent memory instructions there).
struct FftNoHook { __device__ __forceinline__ void operator()(int) const {} };
template <bool INV, typename Hook = FftNoHook>
__device__ __forceinline__ void dft16(double (&xr)[16], double (&xi)[16], const FftConsts fc, Hook hook = Hook())
{
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
        const int step = 8 / half;
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
#pragma unroll
            for (int a = 0; a < half; ++a) {
                const int m = a * step;
                const int p = blk + a, q = p + half;
                double ur = xr[p] + xr[q], ui = xi[p] + xi[q];
                double dr = xr[p] - xr[q], di = xi[p] - xi[q];
                xr[p] = ur; xi[p] = ui;
                if (m == 0) { xr[q] = dr; xi[q] = di; }
                else if (m == 4) {
                    if (!INV) { xr[q] = -di; xi[q] = dr; }
                    else      { xr[q] = di;  xi[q] = -dr; }
                } else {
                    double wr, wi;
                    if (m == 1)      { wr = fc.c1;  wi = fc.s1; }
                    else if (m == 2) { wr = fc.h;   wi = fc.h;  }
                    else if (m == 3) { wr = fc.s1;  wi = fc.c1; }
                    else if (m == 5) { wr = -fc.s1; wi = fc.c1; }
                    else if (m == 6) { wr = -fc.h;  wi = fc.h;  }
                    else             { wr = -fc.c1; wi = fc.s1; }
                    if (!INV) cmul(dr, di, wr, wi); else cmulc(dr, di, wr, wi);
                    xr[q] = dr; xi[q] = di;
                }
            }
        }
        hook(half == 8 ? 0 : half == 4 ? 1 : half == 2 ? 2 : 3);
    }
    // bit-reversal to natural order (compile-time register renaming)
#define FFT_SWAP(i, j) { double t0 = xr[i]; xr[i] = xr[j]; xr[j] = t0; double t1 = xi[i]; xi[i] = xi[j]; xi[j] = t1; }
    FFT_SWAP(1, 8) FFT_SWAP(2, 4) FFT_SWAP(3, 12) FFT_SWAP(5, 10) FFT_SWAP(7, 14) FFT_SWAP(11, 13)
#undef FFT_SWAP
}

// LDS visibility between the lanes of ONE wavefront (a group never spans wavefronts).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that waits for this wave's LDS traffic only (a plain __syncthreads() also drains
// vmcnt, which would serialise global loads that were issued early on purpose).
__device__ __forceinline__ void wg_barrier_lds_only()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16 x 16 transpose of the group's complex tile through LDS: lane `b` gives x[row] for every
// row and receives tile[b][col] for every col.
__device__ __forceinline__ void group_transpose(double (&xr)[16], double (&xi)[16], double *tile, int b)
{
#if FFT_XPOSE_PRIO
    __builtin_amdgcn_s_setprio(FFT_XPOSE_PRIO);
#endif
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        double2 v; v.x = xr[k1]; v.y = xi[k1];
        *reinterpret_cast<double2 *>(tile + 2 * (k1 * 17 + b)) = v;
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (b * 17 + c));
        xr[c] = v.x; xi[c] = v.y;
    }
    wave_lds_sync();
#if FFT_XPOSE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// Forward negacyclic transform.  In: xr[a] = p[16a+b], xi[a] = p[256+16a+b] (already doubles).
// Out: lane k1 (= b) holds X[k1 + 16*k2] in (xr[k2], xi[k2]).
// psi: LDS table psi[j] (j < 256) as double2; tw: LDS table tw[k1*16+b] = w256^(k1*b).
// Split in two halves: the head touches only registers and the read-only tables, the tail is the
// first to write the group's tile (a caller may put a workgroup barrier between them).
__device__ __forceinline__ void nega_fwd_head(double (&xr)[16], double (&xi)[16], const double2 *psi, const double2 *tw,
                                              int b, const FftConsts fc)
{
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        double2 w = psi[16 * a + b];
        cmul(xr[a], xi[a], w.x, w.y);
    }
    dft16<false>(xr, xi, fc);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        double2 w = tw[16 * k1 + b];
        cmul(xr[k1], xi[k1], w.x, w.y);
    }
}

__device__ __forceinline__ void nega_fwd_tail(double (&xr)[16], double (&xi)[16], double *tile, int b, const FftConsts fc)
{
    group_transpose(xr, xi, tile, b);
    dft16<false>(xr, xi, fc);
}

__device__ __forceinline__ void nega_fwd(double (&xr)[16], double (&xi)[16], const double2 *psi, const double2 *tw,
                                         double *tile, int b, const FftConsts fc)
{
    nega_fwd_head(xr, xi, psi, tw, b, fc);
    nega_fwd_tail(xr, xi, tile, b, fc);
}

// Inverse (unscaled, untwisted by conj psi).  In: lane k1 holds F[k1 + 16*k2] in index k2.
// Out: xr[a] = Re z[16a+b] * conj(psi), xi[a] = Im (i.e. real values for coefficients 16a+b and 256+16a+b),
// still multiplied by 256.
__device__ __forceinline__ void nega_inv(double (&xr)[16], double (&xi)[16], const double2 *psi, const double2 *tw,
                                         double *tile, int b, const FftConsts fc)
{
    dft16<true>(xr, xi, fc);
#pragma unroll
    for (int c = 1; c < 16; ++c) {
        double2 w = tw[16 * c + b];
        cmulc(xr[c], xi[c], w.x, w.y);
    }
    group_transpose(xr, xi, tile, b);
    dft16<true>(xr, xi, fc);
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        double2 w = psi[16 * a + b];
        cmulc(xr[a], xi[a], w.x, w.y);
    }
}

// ---- the same transforms with the table reads batched -----------------------------------------------------------------
// With one or two waves per SIMD nothing hides an LDS round trip but the wave's own instruction stream, and the compiler
// keeps a single table read in flight when registers are tight (each of the 31 reads of a transform then exposes ~100
// cycles).  These forms read eight table entries at a time into a buffer, one step ahead of their use.  Identical arithmetic.
__device__ __forceinline__ void fft_tw_load8(double2 (&w)[8], const double2 *tab, int base, int stride)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = tab[base + stride * k];
}
template <bool CONJ, int N>
__device__ __forceinline__ void fft_tw_mul(double *xr, double *xi, const double2 (&w)[8], int first = 0)
{
#pragma unroll
    for (int k = first; k < N; ++k) {
        if (!CONJ) cmul(xr[k], xi[k], w[k].x, w[k].y); else cmulc(xr[k], xi[k], w[k].x, w[k].y);
    }
}

// forward: w0 must already hold psi[16a + b], a = 0..7 (fft_tw_load8(w0, psi, b, 16), issued before the digits were made)
__device__ __forceinline__ void nega_fwd_batched(double (&xr)[16], double (&xi)[16], double2 (&w0)[8], double2 (&w1)[8], const double2 *psi,
                                                 const double2 *tw, double *tile, int b, const FftConsts fc)
{
    fft_tw_load8(w1, psi, 128 + b, 16);                 // a = 8..15
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<false, 8>(xr, xi, w0);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_load8(w0, tw, 16 + b, 16);                   // w256^(k1 b), k1 = 1..8
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<false, 8>(xr + 8, xi + 8, w1);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_load8(w1, tw, 128 + b, 16);                  // k1 = 8..15 (entry 0 unused)
    __builtin_amdgcn_sched_barrier(0);
    dft16<false>(xr, xi, fc);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<false, 8>(xr + 1, xi + 1, w0);
    fft_tw_mul<false, 8>(xr + 8, xi + 8, w1, 1);
    nega_fwd_tail(xr, xi, tile, b, fc);
}

__device__ __forceinline__ void nega_inv_batched(double (&xr)[16], double (&xi)[16], double2 (&w0)[8], double2 (&w1)[8], const double2 *psi,
                                                 const double2 *tw, double *tile, int b, const FftConsts fc)
{
    fft_tw_load8(w0, tw, 16 + b, 16);                   // k1 = 1..8
    fft_tw_load8(w1, tw, 128 + b, 16);                  // k1 = 8..15
    __builtin_amdgcn_sched_barrier(0);
    dft16<true>(xr, xi, fc);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<true, 8>(xr + 1, xi + 1, w0);
    fft_tw_mul<true, 8>(xr + 8, xi + 8, w1, 1);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_load8(w0, psi, b, 16);
    fft_tw_load8(w1, psi, 128 + b, 16);
    __builtin_amdgcn_sched_barrier(0);
    group_transpose(xr, xi, tile, b);
    dft16<true>(xr, xi, fc);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<true, 8>(xr, xi, w0);
    fft_tw_mul<true, 8>(xr + 8, xi + 8, w1);
}

__device__ __forceinline__ uint64_t torus_from_double(double v)
{
    double w = v * 0x1p-72;
    w -= __builtin_rint(w);
    double r = __builtin_rint(w * 0x1p64);
    if (r >= 0x1p63) r -= 0x1p64;
    return (uint64_t)(long long)r;      // (a hand-split floor/fma conversion measured 5 % slower than the compiler's)
}

__device__ __forceinline__ double double_from_torus(uint64_t x) { return (double)(long long)x; }

__device__ __forceinline__ int mod_switch_1024(uint64_t x) { return (int)(((x + (1ULL << 53)) >> 54) & 1023); }

// generic signed balanced decomposition; dig[l] = digit of level l+1
template <int BASE_LOG, int LEVELS>
__device__ __forceinline__ void decompose_all(uint64_t x, int (&dig)[LEVELS])
{
    constexpr int R = 64 - BASE_LOG * LEVELS;
    uint64_t st = (x >> R) + ((x >> (R - 1)) & 1);
    st &= (1ULL << (BASE_LOG * LEVELS)) - 1;
    constexpr uint64_t MASK = (1ULL << BASE_LOG) - 1;
#pragma unroll
    for (int l = LEVELS - 1; l >= 0; --l) {
        uint64_t d = st & MASK;
        st >>= BASE_LOG;
        uint64_t carry = (((d - 1) | st) & d) >> (BASE_LOG - 1);
        st += carry;
        dig[l] = (int)((long long)d - (long long)(carry << BASE_LOG));
    }
}

// Streaming form used by the external product: the first call consumes x and leaves a 32-bit
// state; later calls peel one level each (least significant level first).  Requires
// BASE_LOG*(LEVELS-1) <= 32.
template <int BASE_LOG, int LEVELS>
__device__ __forceinline__ int decompose_first(uint64_t x, uint32_t &state)
{
    constexpr int R = 64 - BASE_LOG * LEVELS;
    uint64_t y = (x >> R) + ((x >> (R - 1)) & 1);
    uint32_t d = (uint32_t)y & ((1u << BASE_LOG) - 1);
    uint32_t st;
    if (LEVELS == 1) st = 0;
    else st = (uint32_t)(y >> BASE_LOG) & (uint32_t)((1ULL << (BASE_LOG * (LEVELS - 1))) - 1);
    uint32_t carry = (((d - 1) | st) & d) >> (BASE_LOG - 1);
    state = st + carry;
    return (int)d - (int)(carry << BASE_LOG);
}

template <int BASE_LOG>
__device__ __forceinline__ int decompose_next(uint32_t &state)
{
    uint32_t d = state & ((1u << BASE_LOG) - 1);
    uint32_t st = state >> BASE_LOG;
    uint32_t carry = (((d - 1) | st) & d) >> (BASE_LOG - 1);
    state = st + carry;
    return (int)d - (int)(carry << BASE_LOG);
}
