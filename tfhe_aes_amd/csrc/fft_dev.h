// fft_dev.h -- device-side negacyclic f64 FFT (N = 512 -> 256 complex points), gadget
// decomposition and torus conversions for gfx950.
//
// Mapping: one polynomial per GROUP of 16 lanes (4 polynomials per 64-lane wavefront), 16 complex
// points per lane, so the 256-point transform is two in-register DFT16 passes with ONE transpose
// through LDS (a 16 x 17 padded tile per group: conflict-free ds_write_b128 / ds_read_b128).
//
// The arithmetic is the canonical one stated in oracle/fheaes_oracle.c ("canonical form, v2") and DESIGN.md:
//   z_j = p_j + i p_{j+256}, j = 16a + b (lane b);  X_k = sum_j z_j e^{2 pi i j (k + 1/4)/256}, k = k1 + 16 k2
//   pass 1   DFT16 over a with frequency offset 1/4 (radix-2 DIT; the stage twiddles e^{2 pi i (k + 1/4)/n} are the SAME in every
//            lane: compile-time constants, no table read -- the negacyclic twist costs no pass of its own),
//   twiddle  T[k1][b] = psi^(b (4 k1 + 1)) (cmul; the one table, 16 entries per lane, read a whole pass ahead),
//   pass 2   transpose, plain DFT16 over b (radix-2 DIT, twiddles 1 and +-i are additions).
//   inverse  plain conjugate DFT16, conj T, transpose, plain conjugate DFT16, conj psi^(16a) (constants).
// Non-trivial DIT butterfly, 6 fused operations:  u = p + w q as two fma per component, v = 2p - u as one.
// No contraction beyond the written fma (the translation unit is compiled with -ffp-contract=off).
// Round 3: the round-2 form (twist pass + radix-2 DIF + separate twiddle pass, 460 instructions and 31 table reads per
// transform) had the table reads of its first half on the critical path of the blind rotation: this form has 404 / 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_consts.h"

#define FHE_N 512
#ifndef FFT_XPOSE_PRIO
#define FFT_XPOSE_PRIO 2   /* wave priority while a transpose's LDS writes/reads are being issued (measured -0.7 %) */
#endif
#ifndef FFT_CHUNK
#define FFT_CHUNK 4            /* butterflies issued together (see dft16) */
#endif
#ifndef FFT_CHUNK_BARRIERS
#define FFT_CHUNK_BARRIERS 1
#endif
#define FHE_H 256
#define GROUP_TILE_BYTES 4352          /* 16 rows x 17 complex x 16 B */
#define GROUP_TILE_DOUBLES (GROUP_TILE_BYTES / 8)
// the twiddle table T in LDS / HBM: entry (k1, b) at index 17 k1 + b -- the forward transform reads a COLUMN (lane b, k1 = 0..15),
// the inverse a ROW (lane k1, b = 0..15); with rows of 17 both are conflict-free 16-byte reads
#define FHE_TW_STRIDE 17
#define FHE_TW_ENTRIES (16 * FHE_TW_STRIDE)

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

// DFT of length 16 in registers, radix-2 decimation in time, natural order in and out.
// INV = false: kernel e^{+2 pi i a (k + phi)/16}, phi = OFFSET ? 1/4 : 0; INV = true: conjugate kernel (phi = 0 only).
// `hook(stage)` runs after butterfly stage 0..3 (a caller interleaves independent memory instructions there);
// `chunk_hook(stage, c0)` after every chunk of butterflies c0 .. c0+CHUNK-1 of a stage (CHUNK: template parameter, default FFT_CHUNK): in the LAST stage (3) butterfly k
// has just produced the final outputs k and k + 8, which live in registers fft_reg(k), fft_reg(k + 8) until dft16 returns
// (natural order only after the renaming at its end) -- a caller can store them while the rest of the stage computes.
struct FftNoHook { __device__ __forceinline__ void operator()(int) const {} };
struct FftNoChunkHook { __device__ __forceinline__ void operator()(int, int) const {} };
__device__ __forceinline__ constexpr int fft_reg(int k)       // register that holds logical position k inside dft16
{
    return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
}
template <bool INV, bool OFFSET, int CHUNK = FFT_CHUNK, typename Hook = FftNoHook, typename ChunkHook = FftNoChunkHook>
__device__ __forceinline__ void dft16(double (&xr)[16], double (&xi)[16], Hook hook = Hook(), ChunkHook chunk_hook = ChunkHook())
{
    static_assert(CHUNK == 1 || CHUNK == 2 || CHUNK == 4 || CHUNK == 8, "butterflies per chunk");
    static_assert(!(INV && OFFSET), "the inverse applies its untwist after the transform");
    constexpr int BR[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};   // logical position -> register
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int n = 2 << st, half = n / 2;
        // The eight butterflies of a stage, FFT_CHUNK at a time, each chunk in three steps (first fma of every u, second fma,
        // then every v): a dependent f64 instruction issues 8 cycles after its producer, and a wave that has the SIMD to itself
        // (the other one waiting for memory) would otherwise stand still for half of that behind every fused pair.
#pragma unroll
        for (int c0 = 0; c0 < 8; c0 += CHUNK) {
            double tr[CHUNK], ti[CHUNK];
#pragma unroll
            for (int j = 0; j < CHUNK; ++j) {
                const int i = c0 + j, blk = (i / half) * n, k = i % half;
                const int e = OFFSET ? (64 * k + 16) / n : 64 * k / n;      // twiddle psi^(16 e), e in 0..31
                const int P = BR[blk + k], Q = BR[blk + k + half];
                if (e != 0 && e != 16) {
                    const double c = FHE_PSI16_RE[e];
                    tr[j] = __builtin_fma(c, xr[Q], xr[P]);
                    ti[j] = __builtin_fma(c, xi[Q], xi[P]);
                }
            }
            if (FFT_CHUNK_BARRIERS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CHUNK; ++j) {
                const int i = c0 + j, blk = (i / half) * n, k = i % half;
                const int e = OFFSET ? (64 * k + 16) / n : 64 * k / n;
                const int Q = BR[blk + k + half];
                if (e != 0 && e != 16) {
                    const double s = INV ? -FHE_PSI16_IM[e] : FHE_PSI16_IM[e];
                    tr[j] = __builtin_fma(-s, xi[Q], tr[j]);
                    ti[j] = __builtin_fma(s, xr[Q], ti[j]);
                }
            }
            if (FFT_CHUNK_BARRIERS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CHUNK; ++j) {
                const int i = c0 + j, blk = (i / half) * n, k = i % half;
                const int e = OFFSET ? (64 * k + 16) / n : 64 * k / n;
                const int P = BR[blk + k], Q = BR[blk + k + half];
                const double pr = xr[P], pi = xi[P], qr = xr[Q], qi = xi[Q];
                if (e == 0) {
                    xr[P] = pr + qr; xi[P] = pi + qi; xr[Q] = pr - qr; xi[Q] = pi - qi;
                } else if (e == 16) {                                        // w = +i (forward), -i (inverse)
                    if (!INV) { xr[P] = pr - qi; xi[P] = pi + qr; xr[Q] = pr + qi; xi[Q] = pi - qr; }
                    else      { xr[P] = pr + qi; xi[P] = pi - qr; xr[Q] = pr - qi; xi[Q] = pi + qr; }
                } else {                                                     // u = p + w q (in tr/ti), v = 2p - u
                    xr[Q] = __builtin_fma(2.0, pr, -tr[j]);
                    xi[Q] = __builtin_fma(2.0, pi, -ti[j]);
                    xr[P] = tr[j]; xi[P] = ti[j];
                }
            }
            if (FFT_CHUNK_BARRIERS) __builtin_amdgcn_sched_barrier(0);
            chunk_hook(st, c0);
        }
        hook(st);
    }
    // logical position k lives in register BR[k]: rename to natural order (compile-time register renaming)
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

// Table reads in batches of eight, issued a whole pass ahead of their use: with one or two waves per SIMD nothing hides an LDS
// round trip but the wave's own instruction stream.
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
// the forward transform's table column of lane b: w0 = T[0..7][b], w1 = T[8..15][b]
__device__ __forceinline__ void fft_fwd_table(double2 (&w0)[8], double2 (&w1)[8], const double2 *tw, int b)
{
    fft_tw_load8(w0, tw, b, FHE_TW_STRIDE);
    fft_tw_load8(w1, tw, 8 * FHE_TW_STRIDE + b, FHE_TW_STRIDE);
}
// the inverse transform's table row of lane k1: w0 = T[k1][0..7] (entry 0 unused), w1 = T[k1][8..15]
__device__ __forceinline__ void fft_inv_table(double2 (&w0)[8], double2 (&w1)[8], const double2 *tw, int k1)
{
    fft_tw_load8(w0, tw, FHE_TW_STRIDE * k1, 1);
    fft_tw_load8(w1, tw, FHE_TW_STRIDE * k1 + 8, 1);
}

// Forward negacyclic transform.  In: xr[a] = p[16a+b], xi[a] = p[256+16a+b] (already doubles), lane b.
// Out: lane k1 (= b) holds X[k1 + 16*k2] in (xr[k2], xi[k2]).
// Split in two halves: the head touches only registers (and the table entries the caller has read with fft_fwd_table), the
// tail is the first to write the group's tile (a caller may put a workgroup barrier between them).
__device__ __forceinline__ void nega_fwd_head(double (&xr)[16], double (&xi)[16], const double2 (&w0)[8], const double2 (&w1)[8])
{
    dft16<false, true>(xr, xi);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<false, 8>(xr, xi, w0);
    fft_tw_mul<false, 8>(xr + 8, xi + 8, w1);
}

__device__ __forceinline__ void nega_fwd_tail(double (&xr)[16], double (&xi)[16], double *tile, int b)
{
    group_transpose(xr, xi, tile, b);
    dft16<false, false>(xr, xi);
}

__device__ __forceinline__ void nega_fwd(double (&xr)[16], double (&xi)[16], const double2 *tw, double *tile, int b)
{
    double2 w0[8], w1[8];
    fft_fwd_table(w0, w1, tw, b);
    __builtin_amdgcn_sched_barrier(0);
    nega_fwd_head(xr, xi, w0, w1);
    nega_fwd_tail(xr, xi, tile, b);
}

// Inverse (unscaled).  In: lane k1 holds F[k1 + 16*k2] in index k2.
// Out: xr[a], xi[a] = the real values of coefficients 16a+b and 256+16a+b (lane b), still multiplied by 256.
__device__ __forceinline__ void nega_inv(double (&xr)[16], double (&xi)[16], const double2 *tw, double *tile, int b)
{
    double2 w0[8], w1[8];
    fft_inv_table(w0, w1, tw, b);
    __builtin_amdgcn_sched_barrier(0);
    dft16<true, false>(xr, xi);
    __builtin_amdgcn_sched_barrier(0);
    fft_tw_mul<true, 8>(xr, xi, w0, 1);
    fft_tw_mul<true, 8>(xr + 8, xi + 8, w1);
    group_transpose(xr, xi, tile, b);
    dft16<true, false>(xr, xi);
#pragma unroll
    for (int a = 1; a < 16; ++a) cmulc(xr[a], xi[a], FHE_PSI16_RE[a], FHE_PSI16_IM[a]);
}

// Back-conversion of an inverse-transform output to the torus.  Canonical definition (oracle/fheaes_oracle.c):
//   w = v * 2^-72; w -= rint(w); r = rint(w * 2^64); result = (uint64)(int64) r   (r = +2^63 wraps to -2^63).
// Computed here without a 64-bit float->integer conversion.  With H = rint(w' * 2^32) and l = rint((w' - H 2^-32) * 2^64)
// one has r = H * 2^32 + l exactly (H * 2^32 is an even integer, so the tie rule is preserved), and both roundings come out
// of magic-number additions: a double in [2^20, 2^21) has ulp 2^-32, one in [2^-12, 2^-11) ulp 2^-64, so the low mantissa
// bits of  w' + 1.5 * 2^20  and of  (w' - H 2^-32) + 1.5 * 2^-12  are H and l as two's complement integers.
// 7 f64 instructions + 2 integer ones against 13 + 3 for the compiler's expansion of the definition; identical for every
// double (checked against the definition on the CPU, 6e7 values around the tie / wrap / tiny cases).
// torus_acc(acc, v) = acc + torus_from_double(v).
typedef uint32_t fhe_u32x2 __attribute__((ext_vector_type(2)));
// torus_acc_scaled(acc, w): the same with w = v * 2^-72 already formed by the caller.  The scaling by a power of two commutes with every
// rounding of the inverse transform's last constant multiply (no result is anywhere near the subnormal range), so a caller may fold
// it into those constants: cmulc(x, 2^-72 c) == 2^-72 cmulc(x, c) bit for bit.
__device__ __forceinline__ uint64_t torus_acc_scaled(uint64_t acc, double w)
{
    w -= __builtin_rint(w);
    const double C_HI = 0x1.8p20, C_LO = 0x1.8p-12;
    const double mh = w + C_HI;
    const double hf = mh - C_HI;
    const double lt = w - hf;
    const double ml = lt + C_LO;
    const fhe_u32x2 bh = __builtin_bit_cast(fhe_u32x2, mh), bl = __builtin_bit_cast(fhe_u32x2, ml);
    fhe_u32x2 r;
    r[0] = bl[0];
    r[1] = bh[0] + bl[1] + 0xC0C80000u;
    return acc + __builtin_bit_cast(uint64_t, r);
}
__device__ __forceinline__ uint64_t torus_acc(uint64_t acc, double v)
{
#ifdef FHE_TORUS_CONV_OLD
    double w = v * 0x1p-72;
    w -= __builtin_rint(w);
    double r = __builtin_rint(w * 0x1p64);
    if (r >= 0x1p63) r -= 0x1p64;
    return acc + (uint64_t)(long long)r;
#else
    double w = v * 0x1p-72;
    w -= __builtin_rint(w);                                  // exact, |w| <= 1/2
    const double C_HI = 0x1.8p20, C_LO = 0x1.8p-12;
    const double mh = w + C_HI;                              // C_HI + rint(w * 2^32) * 2^-32
    const double hf = mh - C_HI;
    const double lt = w - hf;                                // exact, |lt| <= 2^-33
    const double ml = lt + C_LO;                             // C_LO + rint(lt * 2^64) * 2^-64
    const fhe_u32x2 bh = __builtin_bit_cast(fhe_u32x2, mh), bl = __builtin_bit_cast(fhe_u32x2, ml);
    fhe_u32x2 r;
    r[0] = bl[0];
    r[1] = bh[0] + bl[1] + 0xC0C80000u;                      // - 0x3F380000: the exponent word of C_LO
    return acc + __builtin_bit_cast(uint64_t, r);
#endif
}
__device__ __forceinline__ uint64_t torus_from_double(double v) { return torus_acc(0, v); }

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

// ---- gadget decomposition of the EXTERNAL PRODUCTS (blind rotation, CMUX): canonical form v3 (round 5) ---------------------------
// After the closest-representable rounding (x + 2^(R-1), R = 64 - BASE_LOG LEVELS: the same rounding as tfhe-rs' SignedDecomposer)
// the digits are taken by the OFFSET rule of the original TFHE library (tGswTorus32PolynomialDecompH): with
//     z = x_rounded + sum_l (B/2) 2^(64 - BASE_LOG (l + 1))            (one 64-bit addition resolves every carry)
// digit_l = (bits [64 - BASE_LOG (l+1), 64 - BASE_LOG l) of z) - B/2, in [-B/2, B/2).  It recomposes to the same closest representable
// value as the tfhe-rs rule (decompose_all above, which the key switches keep) and gives the same digits except where a digit is
// exactly +-B/2: a tie always carries here, tfhe-rs lets the next digit's top bit decide.  Why: the sequential
// rule costs five dependent integer instructions per digit; this one costs one (a signed bit-field extract of z), and the kernel
// is bound by vector issue and by the socket's power cap, so instructions are time.  Same noise: |digit| <= B/2 either way.
//
// Streaming form: the first call consumes the rounded input and leaves the remaining (LEVELS - 1) BASE_LOG <= 32 bits of z in a
// 32-bit state; later calls peel one level each, least significant level first.
template <int BASE_LOG, int LEVELS>
__device__ __forceinline__ constexpr uint64_t decompose_offset()
{
    uint64_t o = 0;
    for (int l = 0; l < LEVELS; ++l) o += (1ull << (BASE_LOG - 1)) << (64 - BASE_LOG * (l + 1));
    return o;
}

template <int BASE_LOG>
__device__ __forceinline__ int decompose_next(uint32_t &state)
{
    const int digit = (int)(state & ((1u << BASE_LOG) - 1u)) - (1 << (BASE_LOG - 1));
    state >>= BASE_LOG;
    return digit;
}

// First peel from xr = x + 2^(R-1) (the rounding addition already done by the caller, wrapping), R = 64 - BASE_LOG*LEVELS.
template <int BASE_LOG, int LEVELS>
__device__ __forceinline__ int decompose_first_rounded(uint64_t xr, uint32_t &state)
{
    static_assert(BASE_LOG * (LEVELS - 1) <= 32, "the remaining digits must fit the 32-bit state");
    constexpr int R = 64 - BASE_LOG * LEVELS;
    const uint64_t z = xr + decompose_offset<BASE_LOG, LEVELS>();
    state = LEVELS == 1 ? 0u : (uint32_t)(z >> (R + BASE_LOG));                  // exactly (LEVELS-1) BASE_LOG bits
    return (int)((uint32_t)(z >> R) & ((1u << BASE_LOG) - 1u)) - (1 << (BASE_LOG - 1));
}

template <int BASE_LOG, int LEVELS>
__device__ __forceinline__ int decompose_first(uint64_t x, uint32_t &state)
{
    return decompose_first_rounded<BASE_LOG, LEVELS>(x + (1ULL << (64 - BASE_LOG * LEVELS - 1)), state);
}

// The same digits for BASE_LOG = 8, LEVELS = 5 with the state kept as SIGNED bytes (z's upper word with the top bit of every byte
// flipped: byte ^ 0x80 = byte - 128 mod 256), so that a digit is ONE signed bit-field extract at a (wave-uniform) bit offset and the
// state is never shifted: decompose8x5_first returns the least significant digit (level 4), decompose8x5_at(state, 8 j) the digit of
// level 3 - j.
__device__ __forceinline__ int decompose8x5_first_z(uint64_t z, uint32_t &state)      // z = rounded input + decompose_offset<8, 5>()
{
    state = (uint32_t)(z >> 32) ^ 0x80808080u;
    return (int)((uint32_t)z ^ 0x80000000u) >> 24;
}
__device__ __forceinline__ int decompose8x5_first(uint64_t xr, uint32_t &state)
{
    return decompose8x5_first_z(xr + decompose_offset<8, 5>(), state);
}
__device__ __forceinline__ int decompose8x5_at(uint32_t state, unsigned bit)
{
    return __builtin_amdgcn_sbfe((int)state, bit, 8u);
}
