// fft32_dev.h -- the canonical 256-point negacyclic f64 FFT (same arithmetic as fft_dev.h and
// oracle/fheaes_oracle.c: fold + twist by psi^j, DFT16 over the row index (radix-2 DIF), twiddle
// w256^(k*col), transpose, DFT16) mapped onto 32 lanes per polynomial with 8 complex points per lane.
//
// Why: the blind rotation keeps, per polynomial, accumulator + decomposition state + multiply-accumulate
// sums + the transform's working set in registers.  With 16 points per lane that is ~245 VGPRs (2 waves
// per SIMD and spills); with 8 points per lane it is ~110, so 4 waves per SIMD fit and the f64 vector
// unit -- the roof of this kernel -- stays fed while other waves wait on LDS, barriers or key loads.
//
// Eight points per lane hold 3 of an index' 4 bits, so the 8 radix-2 stages of the two DFT16 passes run as
// three in-register steps with TWO trips through the polynomial's 4 KB LDS tile between them:
//
//   step A  (lane = (i, cp): rows {i, i+4, i+8, i+12} x columns {cp, cp+8})    row stages 1, 2
//   trip 1
//   step B  (lane = (u, c):  row positions {4u..4u+3} x columns {c, c+8})      row stages 3, 4; twiddle
//                                                                               w256^(k*col); column stage 1
//   trip 2
//   step C  (lane = (k, blk): row k, column positions {8 blk .. 8 blk + 7})    column stages 2, 3, 4
//
// Every butterfly computes exactly the expression the canonical 16 x 16 form computes for that element
// (cmul / cmulc with one fma per component).  Steps A and B multiply by a per-lane twiddle w16^m read from
// a table, also where the canonical form has m = 0 (no multiply) or m = 4 (swap and negate): x * (1, 0)
// and x * (0, 1) through cmul reproduce those bit for bit except for the sign of a zero, which no
// integer result of the kernels depends on.
//
// The step functions take the lane index and plain arrays, and touch LDS only through the tile pointer,
// so tests/host/fft32_host.cpp runs them lane by lane on the CPU against the oracle (bitwise).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define FFT32_FN __device__ __forceinline__
#else
#define FFT32_FN static inline
#endif

struct Fft32Cplx {
    double x, y;
};   // same layout as double2

struct Fft32Consts {        // w16^1 = (c1, s1), w16^2 = (h, h)
    double c1, s1, h;
};

#define FFT32_TILE_BYTES 4096
#define FFT32_TILE_CPLX 256

FFT32_FN void f32_cmul(double &xr, double &xi, double wr, double wi)
{
    double t = xi * wi;
    double u = xi * wr;
    double re = __builtin_fma(xr, wr, -t);
    double im = __builtin_fma(xr, wi, u);
    xr = re; xi = im;
}

FFT32_FN void f32_cmulc(double &xr, double &xi, double wr, double wi)
{
    double t = xi * wi;
    double u = xr * wi;
    double re = __builtin_fma(xr, wr, t);
    double im = __builtin_fma(xi, wr, -u);
    xr = re; xi = im;
}

// DIF butterfly with a general twiddle: (P, Q) <- (P + Q, (P - Q) * w)   [INV: * conj(w)]
template <bool INV>
FFT32_FN void f32_bfly(double &pr, double &pi, double &qr, double &qi, double wr, double wi)
{
    double ur = pr + qr, ui = pi + qi;
    double dr = pr - qr, di = pi - qi;
    if (!INV) f32_cmul(dr, di, wr, wi); else f32_cmulc(dr, di, wr, wi);
    pr = ur; pi = ui; qr = dr; qi = di;
}

// DIF butterfly with twiddle w16^m for a compile-time m in {0, 2, 4, 6}, exactly as the canonical dft16
template <bool INV, int M>
FFT32_FN void f32_bfly_c(double &pr, double &pi, double &qr, double &qi, const Fft32Consts fc)
{
    double ur = pr + qr, ui = pi + qi;
    double dr = pr - qr, di = pi - qi;
    pr = ur; pi = ui;
    if (M == 0) { qr = dr; qi = di; }
    else if (M == 4) {
        if (!INV) { qr = -di; qi = dr; } else { qr = di; qi = -dr; }
    } else {
        double wr = (M == 2) ? fc.h : -fc.h, wi = fc.h;
        if (!INV) f32_cmul(dr, di, wr, wi); else f32_cmulc(dr, di, wr, wi);
        qr = dr; qi = di;
    }
}

FFT32_FN int f32_bitrev4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

// ---- lane coordinates -------------------------------------------------------------------------
// step A element e = 2 r + s: row i + 4 r, column cp + 8 s
FFT32_FN int f32_a_row(int L, int e) { return (L >> 3) + 4 * (e >> 1); }
FFT32_FN int f32_a_col(int L, int e) { return (L & 7) + 8 * (e & 1); }
// step B element e = 2 r + s: row position 4 u + r, column c + 8 s
FFT32_FN int f32_b_pos(int L, int e) { return 4 * (L >> 3) + (e >> 1); }
FFT32_FN int f32_b_col(int L, int e) { return (L & 7) + 8 * (e & 1); }
// step C element q: row k = L & 15, column position 8 blk + q; after the step it is column frequency kappa
FFT32_FN int f32_c_row(int L) { return L & 15; }
FFT32_FN int f32_c_kappa(int L, int q) { return f32_bitrev4(8 * (L >> 4) + q); }

// tile slots (16-byte units); the XOR terms make the wave-wide 16-byte accesses conflict-free
FFT32_FN int f32_slot1(int pos, int col) { return pos * 16 + (col ^ (pos & 8)); }
FFT32_FN int f32_slot2(int k, int cpos) { return k * 16 + (cpos ^ (k & 7)); }

// ---- per-lane bases: every tile / table access of a step is (one per-lane base) + (compile-time constant) ----------
// (so the compiler has nothing to hoist out of the 669-iteration loop but the bases themselves, which are each 1-3
//  integer instructions from the lane index)
FFT32_FN int f32_bitrev2(int x) { return ((x & 1) << 1) | ((x >> 1) & 1); }
FFT32_FN int f32_bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
// step A: psi / z index of element e = 2 r + s is a_base + 64 r + 8 s; trip-1 slot is a_base + 64 r + 8 (s ^ (r >> 1))
FFT32_FN int f32_a_base(int L) { return 16 * (L >> 3) + (L & 7); }
// step B: trip-1 slot of element e = 2 r + s is (s ? b_rd ^ 8 : b_rd) + 16 r
FFT32_FN int f32_b_rd(int L) { const int u = L >> 3; return 64 * u + (L & 7) + 8 * (u >> 1); }
// step B: twiddle index of element e is b_tw + 64 bitrev2(r) + 8 s
FFT32_FN int f32_b_tw(int L) { return 16 * f32_bitrev2(L >> 3) + (L & 7); }
// step B: trip-2 slot of element e is ((r >> 1) ? b_wr ^ 4 : b_wr) + 64 bitrev2(r) + 8 s
FFT32_FN int f32_b_wr(int L) { const int bu = f32_bitrev2(L >> 3); return 16 * bu + ((L & 7) ^ bu); }
// step C: trip-2 slot of element q is c_rd ^ q
FFT32_FN int f32_c_rd(int L) { return 16 * (L & 15) + 8 * (L >> 4) + (L & 7); }
// step C: natural index (point, or 16 a + b coefficient pair) of element q is c_out + 32 bitrev3(q)
FFT32_FN int f32_c_out(int L) { return (L & 15) + 16 * (L >> 4); }

// ---- the three steps ---------------------------------------------------------------------------
// w16: table of w16^m, m = 0..7, as (re, im)
template <bool INV>
FFT32_FN void f32_step_a(double (&xr)[8], double (&xi)[8], const Fft32Cplx *w16, int L)
{
    const int i = L >> 3;
    const Fft32Cplx w0 = w16[i], w1 = w16[i + 4], w2 = w16[2 * i];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // row stage 1: (i, i+8) with w16^i, (i+4, i+12) with w16^(i+4)
        f32_bfly<INV>(xr[0 + s], xi[0 + s], xr[4 + s], xi[4 + s], w0.x, w0.y);
        f32_bfly<INV>(xr[2 + s], xi[2 + s], xr[6 + s], xi[6 + s], w1.x, w1.y);
        // row stage 2: (i, i+4) and (i+8, i+12) with w16^(2i)
        f32_bfly<INV>(xr[0 + s], xi[0 + s], xr[2 + s], xi[2 + s], w2.x, w2.y);
        f32_bfly<INV>(xr[4 + s], xi[4 + s], xr[6 + s], xi[6 + s], w2.x, w2.y);
    }
}

FFT32_FN void f32_trip1_write(const double (&xr)[8], const double (&xi)[8], Fft32Cplx *tile, int L)
{
    Fft32Cplx *base = tile + f32_a_base(L);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = e >> 1, s = e & 1;
        Fft32Cplx v; v.x = xr[e]; v.y = xi[e];
        base[64 * r + 8 * (s ^ (r >> 1))] = v;
    }
}

FFT32_FN void f32_trip1_read(double (&xr)[8], double (&xi)[8], const Fft32Cplx *tile, int L)
{
    const int b0 = f32_b_rd(L);
    const Fft32Cplx *base0 = tile + b0, *base1 = tile + (b0 ^ 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const Fft32Cplx v = ((e & 1) ? base1 : base0)[16 * (e >> 1)];
        xr[e] = v.x; xi[e] = v.y;
    }
}

// tw: table tw[16 k + col] = w256^(k * col)
template <bool INV>
FFT32_FN void f32_step_b(double (&xr)[8], double (&xi)[8], const Fft32Cplx *tw, const Fft32Cplx *w16, const Fft32Consts fc, int L)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // row stage 3: positions (4u, 4u+2) with w16^0, (4u+1, 4u+3) with w16^4
        f32_bfly_c<INV, 0>(xr[0 + s], xi[0 + s], xr[4 + s], xi[4 + s], fc);
        f32_bfly_c<INV, 4>(xr[2 + s], xi[2 + s], xr[6 + s], xi[6 + s], fc);
        // row stage 4: (4u, 4u+1), (4u+2, 4u+3)
        f32_bfly_c<INV, 0>(xr[0 + s], xi[0 + s], xr[2 + s], xi[2 + s], fc);
        f32_bfly_c<INV, 0>(xr[4 + s], xi[4 + s], xr[6 + s], xi[6 + s], fc);
    }
    // position 4u + r now holds row frequency k = bitrev4(4u + r) = bitrev2(u) + 4 bitrev2(r); twiddle w256^(k * col)
    const Fft32Cplx *twb = tw + f32_b_tw(L);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const Fft32Cplx w = twb[64 * f32_bitrev2(e >> 1) + 8 * (e & 1)];
        if (!INV) f32_cmul(xr[e], xi[e], w.x, w.y); else f32_cmulc(xr[e], xi[e], w.x, w.y);
    }
    // column stage 1: (c, c+8) with w16^c
    const Fft32Cplx wc = w16[L & 7];
#pragma unroll
    for (int r = 0; r < 4; ++r) f32_bfly<INV>(xr[2 * r], xi[2 * r], xr[2 * r + 1], xi[2 * r + 1], wc.x, wc.y);
}

FFT32_FN void f32_trip2_write(const double (&xr)[8], const double (&xi)[8], Fft32Cplx *tile, int L)
{
    const int w0 = f32_b_wr(L);
    Fft32Cplx *base0 = tile + w0, *base1 = tile + (w0 ^ 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = e >> 1, s = e & 1;
        Fft32Cplx v; v.x = xr[e]; v.y = xi[e];
        ((r >> 1) ? base1 : base0)[64 * f32_bitrev2(r) + 8 * s] = v;
    }
}

FFT32_FN void f32_trip2_read(double (&xr)[8], double (&xi)[8], const Fft32Cplx *tile, int L)
{
    const int s0 = f32_c_rd(L);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const Fft32Cplx v = tile[s0 ^ q];
        xr[q] = v.x; xi[q] = v.y;
    }
}

// column stages 2..4 on one block of 8 positions; result left in DIF order: element q is column frequency
// f32_c_kappa(L, q)
template <bool INV>
FFT32_FN void f32_step_c(double (&xr)[8], double (&xi)[8], const Fft32Consts fc)
{
    f32_bfly_c<INV, 0>(xr[0], xi[0], xr[4], xi[4], fc);
    f32_bfly_c<INV, 2>(xr[1], xi[1], xr[5], xi[5], fc);
    f32_bfly_c<INV, 4>(xr[2], xi[2], xr[6], xi[6], fc);
    f32_bfly_c<INV, 6>(xr[3], xi[3], xr[7], xi[7], fc);
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
        f32_bfly_c<INV, 0>(xr[b + 0], xi[b + 0], xr[b + 2], xi[b + 2], fc);
        f32_bfly_c<INV, 4>(xr[b + 1], xi[b + 1], xr[b + 3], xi[b + 3], fc);
    }
#pragma unroll
    for (int b = 0; b < 8; b += 2) f32_bfly_c<INV, 0>(xr[b], xi[b], xr[b + 1], xi[b + 1], fc);
}
