// CPU check of tfhe_aes_amd/csrc/fft32_dev.h: the 32-lane / 8-points-per-lane mapping of the canonical
// negacyclic FFT is run lane by lane (an array stands in for the LDS tile) and compared with the oracle's
// 16 x 16 form, bit for bit up to the sign of zeros.  Built and run by tests/test_fft32_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include "fft32_dev.h"

extern "C" {
void orc_get_twiddles(double *psi_interleaved);
void orc_fft_fwd_int(const int64_t *poly, double *out_interleaved);
void orc_fft_fwd_torus(const uint64_t *poly, double *out_interleaved);
void orc_fft_inv_add(const double *f_interleaved, uint64_t *acc);
}

static Fft32Cplx PSI[512], TW[256], W16[8];
static Fft32Consts FC;

static void tables()
{
    static double psi[1024];
    orc_get_twiddles(psi);
    for (int j = 0; j < 512; ++j) { PSI[j].x = psi[2 * j]; PSI[j].y = psi[2 * j + 1]; }
    auto w256 = [&](int m, double &re, double &im) {
        int e = 4 * (m & 255);
        if (e < 512) { re = PSI[e].x; im = PSI[e].y; } else { re = -PSI[e - 512].x; im = -PSI[e - 512].y; }
    };
    for (int k = 0; k < 16; ++k) for (int c = 0; c < 16; ++c) w256(k * c, TW[16 * k + c].x, TW[16 * k + c].y);
    FC.c1 = PSI[64].x; FC.s1 = PSI[64].y; FC.h = PSI[128].x;
    const double t[8][2] = {{1, 0}, {FC.c1, FC.s1}, {FC.h, FC.h}, {FC.s1, FC.c1}, {0, 1}, {-FC.s1, FC.c1}, {-FC.h, FC.h}, {-FC.c1, FC.s1}};
    for (int m = 0; m < 8; ++m) { W16[m].x = t[m][0]; W16[m].y = t[m][1]; }
}

static inline uint64_t torus_from_double(double v)
{
    double w = v * 0x1p-72;
    w -= rint(w);
    double r = rint(w * 0x1p64);
    if (r >= 0x1p63) r -= 0x1p64;
    return (uint64_t)(int64_t)r;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <bool INV>
static void core(double xr[32][8], double xi[32][8])
{
    Fft32Cplx tile[FFT32_TILE_CPLX];
    for (int L = 0; L < 32; ++L) f32_step_a<INV>(xr[L], xi[L], W16, L);
    for (int L = 0; L < 32; ++L) f32_trip1_write(xr[L], xi[L], tile, L);
    for (int L = 0; L < 32; ++L) f32_trip1_read(xr[L], xi[L], tile, L);
    for (int L = 0; L < 32; ++L) f32_step_b<INV>(xr[L], xi[L], TW, W16, FC, L);
    for (int L = 0; L < 32; ++L) f32_trip2_write(xr[L], xi[L], tile, L);
    for (int L = 0; L < 32; ++L) f32_trip2_read(xr[L], xi[L], tile, L);
    for (int L = 0; L < 32; ++L) f32_step_c<INV>(xr[L], xi[L], FC);
}

int main()
{
    tables();
    int bad = 0;
    for (int trial = 0; trial < 200; ++trial) {
        // ---- forward on digit-like and torus-like inputs
        int64_t poly[512];
        double want[512];
        const bool small = trial & 1;
        for (int j = 0; j < 512; ++j) poly[j] = small ? (int64_t)(rnd() % 257) - 128 : (int64_t)rnd();
        if (trial == 0) memset(poly, 0, sizeof poly);
        orc_fft_fwd_int(poly, want);
        double xr[32][8], xi[32][8];
        for (int L = 0; L < 32; ++L) for (int e = 0; e < 8; ++e) {
            int j = 16 * f32_a_row(L, e) + f32_a_col(L, e);
            xr[L][e] = (double)poly[j]; xi[L][e] = (double)poly[j + 256];
            f32_cmul(xr[L][e], xi[L][e], PSI[j].x, PSI[j].y);
        }
        core<false>(xr, xi);
        double got[512];
        for (int L = 0; L < 32; ++L) for (int q = 0; q < 8; ++q) {
            int p = f32_c_row(L) + 16 * f32_c_kappa(L, q);
            got[2 * p] = xr[L][q]; got[2 * p + 1] = xi[L][q];
        }
        for (int t = 0; t < 512; ++t) if (!(got[t] == want[t])) { if (bad < 10) printf("fwd trial %d word %d: got %a want %a\n", trial, t, got[t], want[t]); ++bad; }
        // ---- inverse of that image
        uint64_t acc_want[512];
        memset(acc_want, 0, sizeof acc_want);
        orc_fft_inv_add(want, acc_want);
        for (int L = 0; L < 32; ++L) for (int e = 0; e < 8; ++e) {
            int p = f32_a_col(L, e) + 16 * f32_a_row(L, e);      // row = k2, column = k1
            xr[L][e] = want[2 * p]; xi[L][e] = want[2 * p + 1];
        }
        core<true>(xr, xi);
        for (int L = 0; L < 32; ++L) for (int q = 0; q < 8; ++q) {
            int j = 16 * f32_c_kappa(L, q) + f32_c_row(L);
            double ur = xr[L][q], ui = xi[L][q];
            f32_cmulc(ur, ui, PSI[j].x, PSI[j].y);
            uint64_t lo = torus_from_double(ur), hi = torus_from_double(ui);
            if (lo != acc_want[j] || hi != acc_want[j + 256]) { if (bad < 10) printf("inv trial %d coef %d\n", trial, j); ++bad; }
        }
    }
    // ---- (per-lane base) + (constant) forms agree with the semantic maps
    for (int L = 0; L < 32; ++L) for (int e = 0; e < 8; ++e) {
        const int r = e >> 1, sbit = e & 1;
        int ok = 1;
        ok &= 16 * f32_a_row(L, e) + f32_a_col(L, e) == f32_a_base(L) + 64 * r + 8 * sbit;
        ok &= f32_slot1(f32_a_row(L, e), f32_a_col(L, e)) == f32_a_base(L) + 64 * r + 8 * (sbit ^ (r >> 1));
        ok &= f32_slot1(f32_b_pos(L, e), f32_b_col(L, e)) == (sbit ? f32_b_rd(L) ^ 8 : f32_b_rd(L)) + 16 * r;
        const int k = f32_bitrev4(f32_b_pos(L, e));
        ok &= 16 * k + f32_b_col(L, e) == f32_b_tw(L) + 64 * f32_bitrev2(r) + 8 * sbit;
        ok &= f32_slot2(k, f32_b_col(L, e)) == ((r >> 1) ? f32_b_wr(L) ^ 4 : f32_b_wr(L)) + 64 * f32_bitrev2(r) + 8 * sbit;
        ok &= f32_slot2(f32_c_row(L), 8 * (L >> 4) + e) == (f32_c_rd(L) ^ e);
        ok &= f32_c_row(L) + 16 * f32_c_kappa(L, e) == f32_c_out(L) + 32 * f32_bitrev3(e);
        if (!ok) { printf("address form mismatch at lane %d element %d\n", L, e); ++bad; }
    }
    // ---- slot maps are bijections of the tile
    {
        int seen1[256] = {0}, seen2[256] = {0};
        for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) { seen1[f32_slot1(a, b)]++; seen2[f32_slot2(a, b)]++; }
        for (int s = 0; s < 256; ++s) if (seen1[s] != 1 || seen2[s] != 1) { printf("slot map not a bijection at %d\n", s); ++bad; }
    }
    printf("fft32 host check: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad ? 1 : 0;
}
