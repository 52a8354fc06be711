/*
 * oracle/fheaes_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, gcc) of the WoPBS S-Box hot path of
 * rostin79s/TFHE-AES and of the AES schedule that drives it.  It is the
 * CHECKER for the HIP engine: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product never links it.
 *
 * What it follows (paths relative to /root/reference):
 *   - pipeline order / list orders / shapes : src/server/sbox/many_wopbs.rs:31-283
 *   - LUT encoding                          : src/server/sbox/gen_lut.rs:9-42
 *   - LUT sets and their order              : src/server/sbox/sbox.rs:20-97
 *   - parameters                            : src/client/client.rs:31-57
 *   - byte / bit order                      : src/client/client.rs:123-138
 *   - AES schedule                          : src/server/server.rs:39-282
 *   - linear layers                         : src/server/encrypt/ and src/server/decrypt/ (all .rs),
 *                                             src/server/key_expansion/key_expansion_utils.rs
 * The arithmetic below those call sites lives in the third-party crate
 * tfhe 0.11.2 (+ tfhe-fft 0.7.0), which is NOT in /root/reference
 * (Cargo.lock:546-549, :580-583) and cannot be built here (no Rust toolchain).
 * Its published algorithms are restated from SURVEY.md section 8 (rows a8, a11-a15)
 * and Appendix A: LWE keyswitch, PBS blind rotation, private functional packing
 * keyswitch, circuit bootstrap, vertical packing (CMUX blind rotation), signed
 * balanced gadget decomposition, modulus switch, sample extraction.
 *
 * PARITY STATUS: at the ciphertext level "parity unpinned" against tfhe-rs
 * (the reference holds no ciphertext fixtures and its f64 FFT is not bit-reproducible
 * across CPUs).  What pins this oracle:
 *   (1) the reference's own known-answer tests are plaintext-level
 *       (src/main.rs:78-95 = NIST SP 800-38A F.1.1): decrypt(oracle FHE-AES) == AES;
 *   (2) the FFT negacyclic product is checked against an exact schoolbook product
 *       mod 2^64 (orc_negacyclic_mul_exact) to within the f64 rounding bound;
 *   (3) every LUT / table against FIPS-197.
 *
 * CANONICAL ARITHMETIC (the HIP engine must reproduce these bit for bit):
 *   - all torus arithmetic is uint64 wrapping;
 *   - polynomial products in Z[X]/(X^512+1) use a 256-point complex f64 FFT of the folded
 *     polynomial, decomposed 16 x 16 with the negacyclic twist merged into the first pass
 *     (see "256-point transform ... canonical form, v2" below): offset-1/4 DFT16 (radix-2 DIT,
 *     fused butterflies), twiddle by psi^(b(4 k1+1)), transpose, plain DFT16 (radix-2 DIT);
 *   - complex multiply  cmul(x,w):  re = fma(xr,wr,-(xi*wi)); im = fma(xr,wi, xi*wr)
 *     conjugate multiply cmulc(x,w): re = fma(xr,wr,  xi*wi ); im = fma(xi,wr,-(xr*wi));
 *   - pointwise multiply-accumulate is one sequential chain per (output poly, point),
 *     levels from the LEAST significant (l = L-1) to the most significant (l = 0), rows r
 *     ascending inside a level:  re = fma(dr,br,re); re = fma(-di,bi,re);
 *                             im = fma(dr,bi,im); im = fma( di,br,im);
 *   - gadget decomposition: the key switches (K1, K3) use tfhe-rs' SignedDecomposer rule (decompose); the external products
 *     (blind rotation, CMUX) use the same closest-representable rounding followed by the OFFSET rule of the original TFHE
 *     library (decompose_offset; canonical form v3, round 5): same recomposed value, digits in [-B/2, B/2), ties always carry;
 *   - back-conversion: w = v*2^-72; w -= rint(w); r = rint(w*2^64) -> int64, wrapping add;
 *   - twiddles: psi^j = exp(i*pi*j/512) from long-double half-angle recurrences (see
 *     init_twiddles), so they do not depend on libm;
 *   - compile with -ffp-contract=off: only the explicit fma() calls fuse.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NPOLY 512
#define HALF 256

typedef struct {
    int32_t n;             /* small LWE dimension                       (669) */
    int32_t k;             /* GLWE dimension                            (4)   */
    int32_t N;             /* polynomial size, must be 512                     */
    int32_t pbs_base_log;  /* 8  */
    int32_t pbs_level;     /* 5  */
    int32_t ks_base_log;   /* 2  */
    int32_t ks_level;      /* 6  */
    int32_t pfks_base_log; /* 12 */
    int32_t pfks_level;    /* 3  */
    int32_t cbs_base_log;  /* 15 */
    int32_t cbs_level;     /* 1  */
} orc_params;

#define BIG(p) ((p)->k * NPOLY)
#define K1(p) ((p)->k + 1)

/* ------------------------------------------------------------------------- */
/* twiddles                                                                   */
/* ------------------------------------------------------------------------- */
static double PSI_RE[NPOLY], PSI_IM[NPOLY];   /* psi^j = exp(i pi j / 512), j < 512 */
static double W256_RE[HALF], W256_IM[HALF];   /* exp(2 pi i m / 256) = psi^(4m)      */
static int g_init_done = 0;
static void init_tw_table(void);

static void init_twiddles(void)
{
    if (g_init_done) return;
    long double bc[8], bs[8];
    bc[7] = sqrtl(0.5L); bs[7] = bc[7];                    /* psi^128 = e^{i pi/4} */
    for (int m = 6; m >= 0; --m) {                         /* psi^(2^m) by half-angle */
        long double c = sqrtl((1.0L + bc[m + 1]) / 2.0L);
        long double s = bs[m + 1] / (2.0L * c);
        bc[m] = c; bs[m] = s;
    }
    for (int j = 0; j <= 128; ++j) {
        long double pr = 1.0L, pi = 0.0L;
        for (int m = 0; m < 8; ++m) if ((j >> m) & 1) {
            long double nr = pr * bc[m] - pi * bs[m];
            long double ni = pr * bs[m] + pi * bc[m];
            pr = nr; pi = ni;
        }
        PSI_RE[j] = (double)pr; PSI_IM[j] = (double)pi;
    }
    PSI_RE[0] = 1.0; PSI_IM[0] = 0.0;
    PSI_IM[128] = PSI_RE[128];
    for (int j = 129; j <= 256; ++j) { PSI_RE[j] = PSI_IM[256 - j]; PSI_IM[j] = PSI_RE[256 - j]; }
    for (int j = 257; j < 512; ++j) { PSI_RE[j] = -PSI_RE[512 - j]; PSI_IM[j] = PSI_IM[512 - j]; }
    for (int m = 0; m < HALF; ++m) {
        int e = 4 * m;
        if (e < 512) { W256_RE[m] = PSI_RE[e]; W256_IM[m] = PSI_IM[e]; }
        else { W256_RE[m] = -PSI_RE[e - 512]; W256_IM[m] = -PSI_IM[e - 512]; }
    }
    init_tw_table();
    g_init_done = 1;
}

__attribute__((constructor)) static void orc_ctor(void) { init_twiddles(); }

void orc_get_twiddles(double *psi_interleaved /* [512][2] */)
{
    init_twiddles();
    for (int j = 0; j < NPOLY; ++j) { psi_interleaved[2 * j] = PSI_RE[j]; psi_interleaved[2 * j + 1] = PSI_IM[j]; }
}

/* ------------------------------------------------------------------------- */
/* 256-point transform of the folded polynomial, 16 x 16 (canonical form, v2)  */
/* ------------------------------------------------------------------------- */
/* z_j = p_j + i p_{j+256} (j = 16a + b).  The negacyclic twist psi^j is NOT a separate pass:
 *   X_k = sum_j z_j e^{2 pi i j (k + 1/4) / 256},   k = k1 + 16 k2
 *       = sum_b e^{2 pi i b k2 / 16} . T[k1][b] . sum_a z_{16a+b} e^{2 pi i a (k1 + 1/4) / 16}
 * with T[k1][b] = e^{2 pi i b (k1 + 1/4) / 256} = psi^(b (4 k1 + 1)):
 *   pass 1  DFT16 over a with frequency offset 1/4 (radix-2 DIT; stage n uses e^{2 pi i (k + 1/4)/n}: constants),
 *   then    multiply by T[k1][b] (cmul, all 256 entries),
 *   pass 2  plain DFT16 over b (radix-2 DIT; twiddles 1 and +-i are additions).
 * Inverse: plain conjugate DFT16 over k2, multiply by conj T[k1][b] (b >= 1), plain conjugate DFT16 over k1,
 * multiply by conj psi^(16a) (a >= 1).
 * DIT butterfly with a non-trivial twiddle w = (c, s), 6 fused operations:
 *   ur = fma(-s, qi, fma(c, qr, pr));  ui = fma(s, qr, fma(c, qi, pi));      u = p + w q
 *   vr = fma(2, pr, -ur);              vi = fma(2, pi, -ui);                 v = p - w q = 2p - u
 * (conjugate twiddle: s -> -s).  The HIP kernels (csrc/fft_dev.h) evaluate the same expression tree. */
static const int BITREV4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
static double TW_RE[16][16], TW_IM[16][16];      /* T[k1][b] = psi^(b (4 k1 + 1)) */

static void init_tw_table(void)
{
    for (int k1 = 0; k1 < 16; ++k1) for (int b = 0; b < 16; ++b) {
        int e = (b * (4 * k1 + 1)) & 1023;
        if (e < 512) { TW_RE[k1][b] = PSI_RE[e]; TW_IM[k1][b] = PSI_IM[e]; }
        else { TW_RE[k1][b] = -PSI_RE[e - 512]; TW_IM[k1][b] = -PSI_IM[e - 512]; }
    }
}

/* DFT of length 16 over the FIRST index of x[16][16] for each column, in place, natural order in and out.
 * inverse=0: kernel e^{+2 pi i a (k + phi)/16}, phi = offset ? 1/4 : 0; inverse=1: conjugate kernel (offset must be 0). */
static void dft16_rows(double xr[16][16], double xi[16][16], int inverse, int offset)
{
    double tr[16][16], ti[16][16];
    for (int p = 0; p < 16; ++p) { memcpy(tr[p], xr[BITREV4[p]], sizeof tr[p]); memcpy(ti[p], xi[BITREV4[p]], sizeof ti[p]); }
    for (int n = 2; n <= 16; n <<= 1) {
        int half = n / 2;
        for (int blk = 0; blk < 16; blk += n) for (int k = 0; k < half; ++k) {
            int e = offset ? (64 * k + 16) / n : 64 * k / n;       /* twiddle = psi^(16 e), e in 0..31 */
            double c = PSI_RE[16 * e], s = PSI_IM[16 * e];
            if (inverse) s = -s;
            int P = blk + k, Q = P + half;
            for (int col = 0; col < 16; ++col) {
                double pr = tr[P][col], pi = ti[P][col], qr = tr[Q][col], qi = ti[Q][col];
                double ur, ui, vr, vi;
                if (e == 0) { ur = pr + qr; ui = pi + qi; vr = pr - qr; vi = pi - qi; }
                else if (e == 16) {                                 /* w = +i (forward) / -i (inverse) */
                    if (!inverse) { ur = pr - qi; ui = pi + qr; vr = pr + qi; vi = pi - qr; }
                    else          { ur = pr + qi; ui = pi - qr; vr = pr - qi; vi = pi + qr; }
                } else {
                    ur = fma(-s, qi, fma(c, qr, pr));
                    ui = fma(s, qr, fma(c, qi, pi));
                    vr = fma(2.0, pr, -ur);
                    vi = fma(2.0, pi, -ui);
                }
                tr[P][col] = ur; ti[P][col] = ui; tr[Q][col] = vr; ti[Q][col] = vi;
            }
        }
    }
    memcpy(xr, tr, sizeof tr); memcpy(xi, ti, sizeof ti);
}

static void transpose16(double x[16][16])
{
    for (int a = 0; a < 16; ++a) for (int b = a + 1; b < 16; ++b) { double t = x[a][b]; x[a][b] = x[b][a]; x[b][a] = t; }
}

/* in: z[16a+b] (untwisted fold); out: X[k] natural order, X_k = sum_j z_j e^{+2 pi i j(k+1/4)/256} */
static void fft256_fwd(double zr[16][16], double zi[16][16])
{
    dft16_rows(zr, zi, 0, 1);                                /* [k1][b] */
    for (int k1 = 0; k1 < 16; ++k1) for (int b = 0; b < 16; ++b) {
        double wr = TW_RE[k1][b], wi = TW_IM[k1][b];
        double xr = zr[k1][b], xi = zi[k1][b];
        zr[k1][b] = fma(xr, wr, -(xi * wi));
        zi[k1][b] = fma(xr, wi, xi * wr);
    }
    transpose16(zr); transpose16(zi);                        /* [b][k1] */
    dft16_rows(zr, zi, 0, 0);                                /* [k2][k1] = X[k1+16k2] */
}

/* in: X[k] natural; out: z[16a+b] = sum_k X_k e^{-2 pi i j(k+1/4)/256} (unscaled; the untwist is included) */
static void fft256_inv(double zr[16][16], double zi[16][16])
{
    dft16_rows(zr, zi, 1, 0);                                /* [b][k1] */
    for (int b = 1; b < 16; ++b) for (int k1 = 0; k1 < 16; ++k1) {
        double wr = TW_RE[k1][b], wi = TW_IM[k1][b];
        double xr = zr[b][k1], xi = zi[b][k1];
        zr[b][k1] = fma(xr, wr, xi * wi);
        zi[b][k1] = fma(xi, wr, -(xr * wi));
    }
    transpose16(zr); transpose16(zi);                        /* [k1][b] */
    dft16_rows(zr, zi, 1, 0);                                /* [a][b] */
    for (int a = 1; a < 16; ++a) for (int b = 0; b < 16; ++b) {
        double wr = PSI_RE[16 * a], wi = PSI_IM[16 * a];
        double xr = zr[a][b], xi = zi[a][b];
        zr[a][b] = fma(xr, wr, xi * wi);
        zi[a][b] = fma(xi, wr, -(xr * wi));
    }
}

/* forward negacyclic transform of a real polynomial given as doubles */
static void nega_fwd(const double *p /*[512]*/, double *fr /*[256]*/, double *fi)
{
    double zr[16][16], zi[16][16];
    double *r = &zr[0][0], *im = &zi[0][0];
    for (int j = 0; j < HALF; ++j) { r[j] = p[j]; im[j] = p[j + HALF]; }
    fft256_fwd(zr, zi);
    memcpy(fr, r, HALF * sizeof(double)); memcpy(fi, im, HALF * sizeof(double));
}

static inline uint64_t torus_from_double(double v)
{
    double w = v * 0x1p-72;            /* 1/256 (FFT scale) * 2^-64 (to torus units) */
    w -= rint(w);                      /* exact */
    double r = rint(w * 0x1p64);       /* |r| <= 2^63 */
    if (r >= 0x1p63) r -= 0x1p64;
    return (uint64_t)(int64_t)r;
}

/* inverse negacyclic transform, rounded to the torus and ADDED into acc[512] */
static void nega_inv_add(const double *fr, const double *fi, uint64_t *acc)
{
    double zr[16][16], zi[16][16];
    double *r = &zr[0][0], *im = &zi[0][0];
    memcpy(r, fr, HALF * sizeof(double)); memcpy(im, fi, HALF * sizeof(double));
    fft256_inv(zr, zi);
    for (int j = 0; j < HALF; ++j) {
        acc[j] += torus_from_double(r[j]);
        acc[j + HALF] += torus_from_double(im[j]);
    }
}

static void torus_poly_fwd(const uint64_t *p, double *fr, double *fi)
{
    double d[NPOLY];
    for (int j = 0; j < NPOLY; ++j) d[j] = (double)(int64_t)p[j];
    nega_fwd(d, fr, fi);
}

/* exported helpers for tests ------------------------------------------------ */
void orc_fft_fwd_int(const int64_t *poly, double *out_interleaved /*[256][2]*/)
{
    double d[NPOLY], fr[HALF], fi[HALF];
    for (int j = 0; j < NPOLY; ++j) d[j] = (double)poly[j];
    nega_fwd(d, fr, fi);
    for (int t = 0; t < HALF; ++t) { out_interleaved[2 * t] = fr[t]; out_interleaved[2 * t + 1] = fi[t]; }
}

/* inverse transform of one Fourier image (natural order, interleaved), rounded to the torus and added into acc[512] */
void orc_fft_inv_add(const double *f_interleaved /*[256][2]*/, uint64_t *acc /*[512]*/)
{
    double fr[HALF], fi[HALF];
    for (int t = 0; t < HALF; ++t) { fr[t] = f_interleaved[2 * t]; fi[t] = f_interleaved[2 * t + 1]; }
    nega_inv_add(fr, fi, acc);
}

void orc_fft_fwd_torus(const uint64_t *poly, double *out_interleaved)
{
    double fr[HALF], fi[HALF];
    torus_poly_fwd(poly, fr, fi);
    for (int t = 0; t < HALF; ++t) { out_interleaved[2 * t] = fr[t]; out_interleaved[2 * t + 1] = fi[t]; }
}

/* out = small (*) torus  in Z_2^64[X]/(X^512+1), through the canonical FFT */
void orc_negacyclic_mul_fft(const int64_t *small, const uint64_t *torus, uint64_t *out)
{
    double d[NPOLY], ar[HALF], ai[HALF], br[HALF], bi[HALF], cr[HALF], ci[HALF];
    for (int j = 0; j < NPOLY; ++j) d[j] = (double)small[j];
    nega_fwd(d, ar, ai);
    torus_poly_fwd(torus, br, bi);
    for (int t = 0; t < HALF; ++t) {
        double re = 0.0, im = 0.0;
        re = fma(ar[t], br[t], re); re = fma(-ai[t], bi[t], re);
        im = fma(ar[t], bi[t], im); im = fma(ai[t], br[t], im);
        cr[t] = re; ci[t] = im;
    }
    memset(out, 0, NPOLY * sizeof(uint64_t));
    nega_inv_add(cr, ci, out);
}

/* exact schoolbook product (independent check of the FFT path) */
void orc_negacyclic_mul_exact(const int64_t *small, const uint64_t *torus, uint64_t *out)
{
    memset(out, 0, NPOLY * sizeof(uint64_t));
    for (int i = 0; i < NPOLY; ++i) {
        uint64_t s = (uint64_t)small[i];
        if (!s) continue;
        for (int j = 0; j < NPOLY; ++j) {
            int t = i + j;
            uint64_t v = s * torus[j];
            if (t < NPOLY) out[t] += v; else out[t - NPOLY] -= v;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* gadget decomposition (tfhe-rs SignedDecomposer)       */
/* ------------------------------------------------------------------------- */
/* dig[l] for l = 0..level-1 is the digit of level l+1 (weight 2^(64-b(l+1))) */
static inline void decompose(uint64_t x, int b, int level, int32_t *dig)
{
    int r = 64 - b * level;
    uint64_t st = (x >> r) + ((x >> (r - 1)) & 1);
    st &= (b * level == 64) ? ~0ULL : ((1ULL << (b * level)) - 1);
    uint64_t mask = (1ULL << b) - 1;
    for (int l = level - 1; l >= 0; --l) {
        uint64_t d = st & mask;
        st >>= b;
        uint64_t carry = (((d - 1) | st) & d) >> (b - 1);
        st += carry;
        dig[l] = (int32_t)((int64_t)d - (int64_t)(carry << b));
    }
}

void orc_decompose(uint64_t x, int base_log, int level, int32_t *out) { decompose(x, base_log, level, out); }

/* Decomposition of the EXTERNAL PRODUCTS (blind rotation, CMUX) -- canonical form v3 (round 5): the same
 * closest-representable rounding as above, then the OFFSET rule of the original TFHE library (tGswTorus32PolynomialDecompH):
 *     z = (x + 2^(r-1)) + sum_l (B/2) 2^(64 - b (l+1)),    dig[l] = (bits [64 - b (l+1), 64 - b l) of z) - B/2   in [-B/2, B/2).
 * It recomposes to the same value mod 2^64 as decompose() and gives the same digits except where a digit is exactly +-B/2 (a tie
 * always carries here; the tfhe-rs rule lets the next digit's top bit decide).  The key switches (K1, K3) keep decompose(): their
 * integer results could be compared with tfhe-rs word for word; behind the f64 FFT of an external product no such comparison
 * exists (ciphertext-level parity unpinned), and on the GPU this rule is one instruction per digit instead of five. */
static inline void decompose_offset(uint64_t x, int b, int level, int32_t *dig)
{
    int r = 64 - b * level;
    uint64_t z = x + (r > 0 ? 1ULL << (r - 1) : 0);
    for (int l = 0; l < level; ++l) z += (1ULL << (b - 1)) << (64 - b * (l + 1));
    uint64_t mask = (1ULL << b) - 1;
    for (int l = 0; l < level; ++l) dig[l] = (int32_t)((z >> (64 - b * (l + 1))) & mask) - (int32_t)(1 << (b - 1));
}

void orc_decompose_offset(uint64_t x, int base_log, int level, int32_t *out) { decompose_offset(x, base_log, level, out); }

static inline int mod_switch(uint64_t x) { return (int)(((x + (1ULL << 53)) >> 54) & 1023); }
int orc_mod_switch(uint64_t x) { return mod_switch(x); }

/* out = p * X^t, t in [0,1024) */
static void poly_mul_monomial(const uint64_t *p, int t, uint64_t *out)
{
    for (int j = 0; j < NPOLY; ++j) {
        int e = j + t;
        int idx = e & (NPOLY - 1);
        int neg = (e >> 9) & 1;
        out[idx] = neg ? (uint64_t)0 - p[j] : p[j];
    }
}

/* ------------------------------------------------------------------------- */
/* external product                                                           */
/* ------------------------------------------------------------------------- */
/* GGSW in the oracle's private planar Fourier format:
 *   gf[((l*k1 + r)*k1 + c)*512 + 0..255] = re, +256..511 = im   (row (l,r), column c) */
static void ext_product_add(int k1, int level, int base_log, const double *gf,
                            const uint64_t *d /*[k1][512]*/, uint64_t *acc /*[k1][512]*/)
{
    int rows = level * k1;
    double *D = (double *)malloc((size_t)rows * NPOLY * sizeof(double));   /* [l][r] -> re[256], im[256] */
    double dig[8][NPOLY];
    int32_t dg[8];
    for (int r = 0; r < k1; ++r) {
        for (int j = 0; j < NPOLY; ++j) {
            decompose_offset(d[r * NPOLY + j], base_log, level, dg);
            for (int l = 0; l < level; ++l) dig[l][j] = (double)dg[l];
        }
        for (int l = 0; l < level; ++l) {
            double *o = D + (size_t)(l * k1 + r) * NPOLY;
            nega_fwd(dig[l], o, o + HALF);
        }
    }
    double fr[HALF], fi[HALF];
    for (int c = 0; c < k1; ++c) {
        for (int t = 0; t < HALF; ++t) { fr[t] = 0.0; fi[t] = 0.0; }
        for (int l = level - 1; l >= 0; --l) for (int r = 0; r < k1; ++r) {   /* least significant level first */
            int lr = l * k1 + r;
            const double *dr = D + (size_t)lr * NPOLY, *di = dr + HALF;
            const double *br = gf + ((size_t)lr * k1 + c) * NPOLY, *bi = br + HALF;
            for (int t = 0; t < HALF; ++t) {
                double re = fr[t], im = fi[t];
                re = fma(dr[t], br[t], re); re = fma(-di[t], bi[t], re);
                im = fma(dr[t], bi[t], im); im = fma(di[t], br[t], im);
                fr[t] = re; fi[t] = im;
            }
        }
        nega_inv_add(fr, fi, acc + (size_t)c * NPOLY);
    }
    free(D);
}

/* standard-domain GGSW [l][r][c][512] u64 -> private planar Fourier */
static void ggsw_to_fourier_planar(int k1, int level, const uint64_t *g, double *gf)
{
    int polys = level * k1 * k1;
    for (int q = 0; q < polys; ++q) torus_poly_fwd(g + (size_t)q * NPOLY, gf + (size_t)q * NPOLY, gf + (size_t)q * NPOLY + HALF);
}

/* exported: canonical interleaved natural-order Fourier image of `polys` torus polynomials:
 * out[q][t][2].  This is the byte layout the engine keeps in HBM for BSK / GGSW. */
void orc_polys_to_fourier(const uint64_t *g, int64_t polys, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < polys; ++q) {
        double fr[HALF], fi[HALF];
        torus_poly_fwd(g + (size_t)q * NPOLY, fr, fi);
        double *o = out + (size_t)q * NPOLY;
        for (int t = 0; t < HALF; ++t) { o[2 * t] = fr[t]; o[2 * t + 1] = fi[t]; }
    }
}

/* acc(glwe) += ggsw (x) d, with ggsw given in the standard domain (exported for tests) */
void orc_external_product_add(const orc_params *p, int level, int base_log,
                              const uint64_t *ggsw_std, const uint64_t *d, uint64_t *acc)
{
    int k1 = K1(p);
    double *gf = (double *)malloc((size_t)level * k1 * k1 * NPOLY * sizeof(double));
    ggsw_to_fourier_planar(k1, level, ggsw_std, gf);
    ext_product_add(k1, level, base_log, gf, d, acc);
    free(gf);
}

/* ------------------------------------------------------------------------- */
/* keys in oracle-private form                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
    orc_params p;
    const uint64_t *ksk;     /* [kN][ks_level][n+1]                   borrowed */
    const uint64_t *pfpksk;  /* [k+1][kN+1][pfks_level][(k+1)N]       borrowed */
    double *bskf;            /* [n][pbs_level][k1][k1] planar Fourier owned    */
} orc_keys;

orc_keys *orc_keys_create(const orc_params *p, const uint64_t *ksk, const uint64_t *bsk_std, const uint64_t *pfpksk)
{
    init_twiddles();
    if (p->N != NPOLY) return NULL;
    orc_keys *K = (orc_keys *)calloc(1, sizeof *K);
    K->p = *p; K->ksk = ksk; K->pfpksk = pfpksk;
    int k1 = K1(p);
    size_t polys = (size_t)p->n * p->pbs_level * k1 * k1;
    K->bskf = (double *)malloc(polys * NPOLY * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)polys; ++q)
        torus_poly_fwd(bsk_std + (size_t)q * NPOLY, K->bskf + (size_t)q * NPOLY, K->bskf + (size_t)q * NPOLY + HALF);
    return K;
}

void orc_keys_destroy(orc_keys *K) { if (K) { free(K->bskf); free(K); } }

/* ------------------------------------------------------------------------- */
/* K1: LWE keyswitch big -> small  (many_wopbs.rs:161-202)        */
/* ------------------------------------------------------------------------- */
void orc_keyswitch(const orc_keys *K, const uint64_t *in /*[kN+1]*/, uint64_t *out /*[n+1]*/)
{
    const orc_params *p = &K->p;
    int big = BIG(p), n1 = p->n + 1, L = p->ks_level;
    int32_t dg[16];
    memset(out, 0, (size_t)n1 * sizeof(uint64_t));
    out[p->n] = in[big];
    for (int i = 0; i < big; ++i) {
        decompose(in[i], p->ks_base_log, L, dg);
        for (int l = 0; l < L; ++l) {
            uint64_t d = (uint64_t)(int64_t)dg[l];
            if (!d) continue;
            const uint64_t *row = K->ksk + ((size_t)i * L + l) * n1;
            for (int o = 0; o < n1; ++o) out[o] -= d * row[o];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* K2: PBS of the circuit bootstrap          */
/* ------------------------------------------------------------------------- */
static void sample_extract(int k, const uint64_t *glwe, uint64_t *lwe)
{
    for (int m = 0; m < k; ++m) {
        const uint64_t *a = glwe + (size_t)m * NPOLY;
        uint64_t *o = lwe + (size_t)m * NPOLY;
        o[0] = a[0];
        for (int j = 1; j < NPOLY; ++j) o[j] = (uint64_t)0 - a[NPOLY - j];
    }
    lwe[(size_t)k * NPOLY] = glwe[(size_t)k * NPOLY];
}

/* blind rotation of a trivial GLWE whose body has every coefficient = tv_const,
 * then sample-extract coefficient 0.  lwe_in is under the small key (n+1 words). */
static void pbs_const_tv(const orc_keys *K, const uint64_t *lwe_in, uint64_t tv_const, uint64_t *lwe_out)
{
    const orc_params *p = &K->p;
    int k1 = K1(p), n = p->n;
    size_t gsz = (size_t)k1 * NPOLY;
    uint64_t *acc = (uint64_t *)calloc(gsz, sizeof(uint64_t));
    uint64_t *d = (uint64_t *)malloc(gsz * sizeof(uint64_t));
    uint64_t tv[NPOLY];
    for (int j = 0; j < NPOLY; ++j) tv[j] = tv_const;
    int bt = mod_switch(lwe_in[n]);
    poly_mul_monomial(tv, (1024 - bt) & 1023, acc + (size_t)p->k * NPOLY);
    size_t ggsw_stride = (size_t)p->pbs_level * k1 * k1 * NPOLY;
    for (int i = 0; i < n; ++i) {
        int at = mod_switch(lwe_in[i]);
        if (at == 0) continue;
        for (int r = 0; r < k1; ++r) {
            poly_mul_monomial(acc + (size_t)r * NPOLY, at, d + (size_t)r * NPOLY);
            for (int j = 0; j < NPOLY; ++j) d[(size_t)r * NPOLY + j] -= acc[(size_t)r * NPOLY + j];
        }
        ext_product_add(k1, p->pbs_level, p->pbs_base_log, K->bskf + (size_t)i * ggsw_stride, d, acc);
    }
    sample_extract(p->k, acc, lwe_out);
    free(acc); free(d);
}

/* homomorphic shift + PBS for CBS level `lvl` (1-based): output LWE(kN) of bit * 2^(64 - cbs_base_log*lvl) */
void orc_cbs_pbs(const orc_keys *K, const uint64_t *lwe_small, int lvl, uint64_t *lwe_out /*[kN+1]*/)
{
    const orc_params *p = &K->p;
    int n = p->n;
    uint64_t *tmp = (uint64_t *)malloc((size_t)(n + 1) * sizeof(uint64_t));
    memcpy(tmp, lwe_small, (size_t)(n + 1) * sizeof(uint64_t));
    tmp[n] += 1ULL << 62;
    uint64_t half_delta = 1ULL << (64 - p->cbs_base_log * lvl - 1);
    pbs_const_tv(K, tmp, (uint64_t)0 - half_delta, lwe_out);
    lwe_out[BIG(p)] += half_delta;
    free(tmp);
}

/* batched forms for the parity tests (one independent unit per OpenMP task) */
void orc_cbs_pbs_batch(const orc_keys *K, const uint64_t *lwe_small, int64_t m, int lvl, uint64_t *lwe_out)
{
    const orc_params *p = &K->p;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < m; ++q)
        orc_cbs_pbs(K, lwe_small + (size_t)q * (p->n + 1), lvl, lwe_out + (size_t)q * (BIG(p) + 1));
}

void orc_keyswitch_batch(const orc_keys *K, const uint64_t *in, int64_t m, uint64_t *out)
{
    const orc_params *p = &K->p;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < m; ++q)
        orc_keyswitch(K, in + (size_t)q * (BIG(p) + 1), out + (size_t)q * (p->n + 1));
}

/* ------------------------------------------------------------------------- */
/* K3: private functional packing keyswitch                    */
/* ------------------------------------------------------------------------- */
void orc_pfpks(const orc_keys *K, int r, const uint64_t *lwe_in /*[kN+1]*/, uint64_t *glwe_out /*[(k+1)N]*/)
{
    const orc_params *p = &K->p;
    int big1 = BIG(p) + 1, L = p->pfks_level;
    size_t gsz = (size_t)K1(p) * NPOLY;
    const uint64_t *key = K->pfpksk + (size_t)r * big1 * L * gsz;
    int32_t dg[16];
    memset(glwe_out, 0, gsz * sizeof(uint64_t));
    for (int i = 0; i < big1; ++i) {
        decompose(lwe_in[i], p->pfks_base_log, L, dg);
        for (int l = 0; l < L; ++l) {
            uint64_t d = (uint64_t)(int64_t)dg[l];
            if (!d) continue;
            const uint64_t *row = key + ((size_t)i * L + l) * gsz;
            for (size_t o = 0; o < gsz; ++o) glwe_out[o] -= d * row[o];
        }
    }
}

/* circuit bootstrap of one bit: small LWE -> standard-domain GGSW [lvl][r][c][512] */
void orc_circuit_bootstrap(const orc_keys *K, const uint64_t *lwe_small, uint64_t *ggsw_out)
{
    const orc_params *p = &K->p;
    int k1 = K1(p);
    size_t gsz = (size_t)k1 * NPOLY;
    uint64_t *lwe = (uint64_t *)malloc((size_t)(BIG(p) + 1) * sizeof(uint64_t));
    for (int lvl = 1; lvl <= p->cbs_level; ++lvl) {
        orc_cbs_pbs(K, lwe_small, lvl, lwe);
        for (int r = 0; r < k1; ++r)
            orc_pfpks(K, r, lwe, ggsw_out + ((size_t)(lvl - 1) * k1 + r) * gsz);
    }
    free(lwe);
}

/* ------------------------------------------------------------------------- */
/* K5: vertical packing (many_wopbs.rs:267-279)                  */
/* ------------------------------------------------------------------------- */
/* ggswf: nbits planar-Fourier GGSWs, index j = input bit j (weight 2^j). lut: one poly. */
/* vertical_packing (many_wopbs.rs:277).  `lut` holds max(2^nbits, 512) entries of ONE output bit, i.e.
 * P = max(1, 2^(nbits-9)) polynomials (gen_lut.rs:19-39: entry idx belongs to polynomial idx / 512).  For nbits > 9 a CMUX tree
 * over input bits 9..nbits-1 (bit 9 at the leaves, the most significant bit at the root) selects the polynomial
 * value >> 9: cmux(g, ct0, ct1) = ct0 + g (x) (ct1 - ct0); the blind rotation over bits 0..8 then selects the coefficient.
 * For nbits <= 9 the tree is degenerate (one polynomial), which is all the AES path uses. */
static void vertical_packing(const orc_params *p, const double *ggswf, int nbits, const uint64_t *lut, uint64_t *lwe_out)
{
    int k1 = K1(p);
    size_t gsz = (size_t)k1 * NPOLY;
    size_t gstride = (size_t)p->cbs_level * k1 * k1 * NPOLY;
    int tree_bits = nbits > 9 ? nbits - 9 : 0;
    size_t polys = (size_t)1 << tree_bits;
    uint64_t *tree = (uint64_t *)calloc(polys * gsz, sizeof(uint64_t));
    uint64_t *ct1 = (uint64_t *)malloc(gsz * sizeof(uint64_t));
    for (size_t j = 0; j < polys; ++j) memcpy(tree + j * gsz + (size_t)p->k * NPOLY, lut + j * NPOLY, NPOLY * sizeof(uint64_t));
    for (int tb = 0; tb < tree_bits; ++tb) {
        size_t nodes = polys >> (tb + 1);
        for (size_t m = 0; m < nodes; ++m) {
            uint64_t *a = tree + (2 * m) * gsz, *b = tree + (2 * m + 1) * gsz;
            for (size_t c = 0; c < gsz; ++c) ct1[c] = b[c] - a[c];
            ext_product_add(k1, p->cbs_level, p->cbs_base_log, ggswf + (size_t)(9 + tb) * gstride, ct1, a);
            if (m) memcpy(tree + m * gsz, a, gsz * sizeof(uint64_t));
        }
    }
    uint64_t *ct0 = tree;
    int rot_bits = nbits < 9 ? nbits : 9;
    for (int j = 0; j < rot_bits; ++j) {
        int deg = 1 << j;
        for (int r = 0; r < k1; ++r) {
            poly_mul_monomial(ct0 + (size_t)r * NPOLY, (1024 - deg) & 1023, ct1 + (size_t)r * NPOLY);
            for (int c = 0; c < NPOLY; ++c) ct1[(size_t)r * NPOLY + c] -= ct0[(size_t)r * NPOLY + c];
        }
        ext_product_add(k1, p->cbs_level, p->cbs_base_log, ggswf + (size_t)j * gstride, ct1, ct0);
    }
    sample_extract(p->k, ct0, lwe_out);
    free(tree); free(ct1);
}

/* ------------------------------------------------------------------------- */
/* many_wopbs_without_padding (many_wopbs.rs:31-116), batched                  */
/* ------------------------------------------------------------------------- */
/* lwe_in : [n_inputs][nbits][kN+1]   bit j of input i (block j, LSB first)
 * luts   : [n_lut_sets][n_luts][nbits][W], W = max(2^nbits, 512) (gen_lut.rs:19-23); input i uses set (lut_per_input ? i : 0)
 * lwe_out: [n_inputs][n_luts][nbits][kN+1]
 * dbg_small / dbg_pbs / dbg_ggsw: optional dumps of intermediates (may be NULL):
 *   dbg_small [n_inputs][nbits][n+1], dbg_pbs [n_inputs][nbits][kN+1] (cbs level 1 only),
 *   dbg_ggsw [n_inputs][nbits][cbs_level][k1][k1][512] */
void orc_wopbs_batch(const orc_keys *K, const uint64_t *lwe_in, int n_inputs, int nbits,
                     const uint64_t *luts, int n_luts, int lut_per_input, uint64_t *lwe_out,
                     uint64_t *dbg_small, uint64_t *dbg_pbs, uint64_t *dbg_ggsw)
{
    const orc_params *p = &K->p;
    int k1 = K1(p), big1 = BIG(p) + 1, n1 = p->n + 1;
    size_t gstd = (size_t)p->cbs_level * k1 * k1 * NPOLY;
    int64_t total_bits = (int64_t)n_inputs * nbits;
    double *ggswf = (double *)malloc((size_t)total_bits * gstd * sizeof(double));
#pragma omp parallel
    {
        uint64_t *small = (uint64_t *)malloc((size_t)n1 * sizeof(uint64_t));
        uint64_t *g = (uint64_t *)malloc(gstd * sizeof(uint64_t));
        uint64_t *tmp = (uint64_t *)malloc((size_t)big1 * sizeof(uint64_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < total_bits; ++q) {
            orc_keyswitch(K, lwe_in + (size_t)q * big1, small);
            if (dbg_small) memcpy(dbg_small + (size_t)q * n1, small, (size_t)n1 * sizeof(uint64_t));
            if (dbg_pbs) { orc_cbs_pbs(K, small, 1, tmp); memcpy(dbg_pbs + (size_t)q * big1, tmp, (size_t)big1 * sizeof(uint64_t)); }
            orc_circuit_bootstrap(K, small, g);
            if (dbg_ggsw) memcpy(dbg_ggsw + (size_t)q * gstd, g, gstd * sizeof(uint64_t));
            ggsw_to_fourier_planar(k1, p->cbs_level, g, ggswf + (size_t)q * gstd);
        }
        free(small); free(g); free(tmp);
        int64_t total_out = (int64_t)n_inputs * n_luts * nbits;
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < total_out; ++q) {
            int64_t i = q / ((int64_t)n_luts * nbits);
            int64_t rem = q % ((int64_t)n_luts * nbits);
            int64_t set = lut_per_input ? i : 0;
            const size_t lut_words = nbits > 9 ? ((size_t)1 << nbits) : NPOLY;
            const uint64_t *lut = luts + ((size_t)set * n_luts * nbits + (size_t)rem) * lut_words;
            vertical_packing(p, ggswf + (size_t)i * nbits * gstd, nbits, lut, lwe_out + (size_t)q * big1);
        }
    }
    free(ggswf);
}

/* ------------------------------------------------------------------------- */
/* LUTs (gen_lut.rs:9-42) and AES tables (tables/table.rs, sbox.rs:20-42)      */
/* ------------------------------------------------------------------------- */
static uint8_t SBOX_T[256], INV_SBOX_T[256];
static int g_tables_done = 0;

static uint8_t gf_mul(uint8_t a, uint8_t b)
{
    uint8_t r = 0;
    for (int i = 0; i < 8; ++i) { if (b & 1) r ^= a; uint8_t hi = a & 0x80; a <<= 1; if (hi) a ^= 0x1B; b >>= 1; }
    return r;
}

static void init_tables(void)
{
    if (g_tables_done) return;
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; ++y) if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, v = inv;
        for (int i = 0; i < 4; ++i) { v = (uint8_t)((v << 1) | (v >> 7)); s ^= v; }
        s ^= 0x63;
        SBOX_T[x] = s; INV_SBOX_T[s] = (uint8_t)x;
    }
    g_tables_done = 1;
}

void orc_get_tables(uint8_t *sbox, uint8_t *inv_sbox) { init_tables(); memcpy(sbox, SBOX_T, 256); memcpy(inv_sbox, INV_SBOX_T, 256); }

/* gen_lut.rs:9-42: lut[b][idx] = ((f(idx & (2^nb-1)) >> b) & 1) << 63, idx < W = max(2^nb, 512) */
void orc_gen_lut(int nb_block, const uint64_t *f_table /*[2^nb]*/, uint64_t *out /*[nb][W]*/)
{
    size_t W = nb_block > 9 ? ((size_t)1 << nb_block) : NPOLY;
    for (size_t idx = 0; idx < W; ++idx) {
        uint64_t v = f_table[idx & (((size_t)1 << nb_block) - 1)];
        for (int b = 0; b < nb_block; ++b) out[(size_t)b * W + idx] = ((v >> b) & 1ULL) << 63;
    }
}

enum { LUTSET_ENC_ROUND = 0, LUTSET_SBOX = 1, LUTSET_INV_SBOX = 2, LUTSET_DEC_MUL = 3, LUTSET_IDENTITY = 4 };

/* builds the 8-bit LUT sets of sbox.rs:46-97 / server.rs:118-119; returns n_luts */
static int build_lutset(int which, uint64_t *out /*[<=4][8][512]*/)
{
    init_tables();
    uint64_t f[4][256];
    int n = 0;
    for (int x = 0; x < 256; ++x) {
        uint8_t s = SBOX_T[x];
        switch (which) {
        case LUTSET_ENC_ROUND: f[0][x] = s; f[1][x] = gf_mul(s, 2); f[2][x] = gf_mul(s, 3); n = 3; break;
        case LUTSET_SBOX: f[0][x] = s; n = 1; break;
        case LUTSET_INV_SBOX: f[0][x] = INV_SBOX_T[x]; n = 1; break;
        case LUTSET_DEC_MUL: f[0][x] = gf_mul((uint8_t)x, 9); f[1][x] = gf_mul((uint8_t)x, 11);
                             f[2][x] = gf_mul((uint8_t)x, 13); f[3][x] = gf_mul((uint8_t)x, 14); n = 4; break;
        default: f[0][x] = (uint64_t)x; n = 1; break;
        }
    }
    for (int i = 0; i < n; ++i) orc_gen_lut(8, f[i], out + (size_t)i * 8 * NPOLY);
    return n;
}

int orc_build_lutset(int which, uint64_t *out) { return build_lutset(which, out); }

/* ------------------------------------------------------------------------- */
/* AES schedule (server.rs:39-282)                                            */
/* state: [16][8][kN+1] (byte index = 4*col+row, bit j = block j, LSB first)   */
/* ------------------------------------------------------------------------- */
static void lwe_add(uint64_t *dst, const uint64_t *src, size_t words) { for (size_t i = 0; i < words; ++i) dst[i] += src[i]; }

static void add_round_key(const orc_params *p, uint64_t *state, const uint64_t *rk)
{
    lwe_add(state, rk, (size_t)16 * 8 * (BIG(p) + 1));
}

/* apply a LUT set to the 16 bytes of `state`; out[16][n_luts][8][big1] */
static void lut_bytes(const orc_keys *K, int which, const uint64_t *state, int nbytes, uint64_t *out)
{
    uint64_t *luts = (uint64_t *)malloc((size_t)4 * 8 * NPOLY * sizeof(uint64_t));
    int nl = build_lutset(which, luts);
    orc_wopbs_batch(K, state, nbytes, 8, luts, nl, 0, out, NULL, NULL, NULL);
    free(luts);
}

static const int MC_ENC[4][4] = {{1, 2, 0, 0}, {0, 1, 2, 0}, {0, 0, 1, 2}, {2, 0, 0, 1}};   /* lut index per (out row, in row): 0=S,1=2S,2=3S */
static const int MC_DEC[4][4] = {{3, 1, 2, 0}, {0, 3, 1, 2}, {2, 0, 3, 1}, {1, 2, 0, 3}};   /* 0=9x,1=11x,2=13x,3=14x */

void orc_aes_encrypt(const orc_keys *K, const uint64_t *round_keys /*[11][16][8][big1]*/, uint64_t *state)
{
    const orc_params *p = &K->p;
    size_t lw = (size_t)BIG(p) + 1, bytew = 8 * lw, statew = 16 * bytew;
    uint64_t *mul = (uint64_t *)malloc((size_t)16 * 3 * bytew * sizeof(uint64_t));
    uint64_t *ns = (uint64_t *)malloc(statew * sizeof(uint64_t));
    add_round_key(p, state, round_keys);
    for (int round = 1; round < 10; ++round) {
        lut_bytes(K, LUTSET_ENC_ROUND, state, 16, mul);
        memset(ns, 0, statew * sizeof(uint64_t));
        for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) {
            uint64_t *o = ns + (size_t)(4 * col + row) * bytew;
            for (int r2 = 0; r2 < 4; ++r2) {
                int src = 4 * ((col + r2) & 3) + r2;              /* ShiftRows folded in (mix_columns.rs:9-21) */
                lwe_add(o, mul + ((size_t)src * 3 + MC_ENC[row][r2]) * bytew, bytew);
            }
        }
        memcpy(state, ns, statew * sizeof(uint64_t));
        add_round_key(p, state, round_keys + (size_t)round * statew);
    }
    lut_bytes(K, LUTSET_SBOX, state, 16, ns);
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row)
        memcpy(state + (size_t)(4 * col + row) * bytew, ns + (size_t)(4 * ((col + row) & 3) + row) * bytew, bytew * sizeof(uint64_t));
    add_round_key(p, state, round_keys + (size_t)10 * statew);
    free(mul); free(ns);
}

static void inv_shift_rows(uint64_t *state, uint64_t *tmp, size_t bytew)
{
    memcpy(tmp, state, 16 * bytew * sizeof(uint64_t));
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row)
        memcpy(state + (size_t)(4 * col + row) * bytew, tmp + (size_t)(4 * ((col - row) & 3) + row) * bytew, bytew * sizeof(uint64_t));
}

void orc_aes_decrypt(const orc_keys *K, const uint64_t *round_keys, uint64_t *state)
{
    const orc_params *p = &K->p;
    size_t lw = (size_t)BIG(p) + 1, bytew = 8 * lw, statew = 16 * bytew;
    uint64_t *mul = (uint64_t *)malloc((size_t)16 * 4 * bytew * sizeof(uint64_t));
    uint64_t *ns = (uint64_t *)malloc(statew * sizeof(uint64_t));
    add_round_key(p, state, round_keys + (size_t)10 * statew);
    for (int round = 10; round >= 2; --round) {
        inv_shift_rows(state, ns, bytew);
        lut_bytes(K, LUTSET_INV_SBOX, state, 16, ns);
        memcpy(state, ns, statew * sizeof(uint64_t));
        add_round_key(p, state, round_keys + (size_t)(round - 1) * statew);
        lut_bytes(K, LUTSET_DEC_MUL, state, 16, mul);
        memset(state, 0, statew * sizeof(uint64_t));
        for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) {
            uint64_t *o = state + (size_t)(4 * col + row) * bytew;
            for (int r2 = 0; r2 < 4; ++r2)
                lwe_add(o, mul + ((size_t)(4 * col + r2) * 4 + MC_DEC[row][r2]) * bytew, bytew);
        }
    }
    inv_shift_rows(state, ns, bytew);
    lut_bytes(K, LUTSET_INV_SBOX, state, 16, ns);
    memcpy(state, ns, statew * sizeof(uint64_t));
    add_round_key(p, state, round_keys);
    free(mul); free(ns);
}

static const uint8_t RCON[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};

/* key: [16][8][big1]; round_keys out: [11][16][8][big1].  RCON enters as a trivial
 * (noise-free) encoding instead of the reference's public-key encryption (server.rs:138-143). */
void orc_aes_key_expansion(const orc_keys *K, const uint64_t *key, uint64_t *round_keys)
{
    const orc_params *p = &K->p;
    size_t lw = (size_t)BIG(p) + 1, bytew = 8 * lw, wordw = 4 * bytew;
    uint64_t *w = round_keys;                                       /* word i at w + i*wordw */
    uint64_t *temp = (uint64_t *)malloc(wordw * sizeof(uint64_t));
    uint64_t *t2 = (uint64_t *)malloc(wordw * sizeof(uint64_t));
    memcpy(w, key, 4 * wordw * sizeof(uint64_t));
    for (int i = 4; i < 44; ++i) {
        memcpy(temp, w + (size_t)(i - 1) * wordw, wordw * sizeof(uint64_t));
        if (i % 4 == 0) {
            for (int j = 0; j < 4; ++j) memcpy(t2 + (size_t)j * bytew, temp + (size_t)((j + 1) & 3) * bytew, bytew * sizeof(uint64_t));
            lut_bytes(K, LUTSET_SBOX, t2, 4, temp);
            uint8_t rc = RCON[i / 4 - 1];
            for (int b = 0; b < 8; ++b) temp[(size_t)b * lw + BIG(p)] += (uint64_t)((rc >> b) & 1) << 63;
        }
        lwe_add(temp, w + (size_t)(i - 4) * wordw, wordw);
        lut_bytes(K, LUTSET_IDENTITY, temp, 4, w + (size_t)i * wordw);
    }
    free(temp); free(t2);
}

/* CTR counter add (server.rs:172-274) with the first-byte carry computed from i & 0xFF
 * (the reference's `x + i as u64 > 255` at server.rs:182 is wrong for i >= 256). */
void orc_add_scalar(const orc_keys *K, uint64_t *state, uint64_t i_hi, uint64_t i_lo)
{
    const orc_params *p = &K->p;
    size_t lw = (size_t)BIG(p) + 1, bytew = 8 * lw;
    uint8_t ib[16];
    for (int j = 0; j < 8; ++j) { ib[15 - j] = (uint8_t)(i_lo >> (8 * j)); ib[7 - j] = (uint8_t)(i_hi >> (8 * j)); }
    uint64_t *luts = (uint64_t *)malloc((size_t)2 * 9 * NPOLY * sizeof(uint64_t));
    uint64_t *in9 = (uint64_t *)malloc((size_t)9 * lw * sizeof(uint64_t));
    uint64_t *out = (uint64_t *)malloc((size_t)2 * 9 * lw * sizeof(uint64_t));
    uint64_t *carry = (uint64_t *)malloc(lw * sizeof(uint64_t));
    uint64_t f[512], g[512];
    for (int x = 0; x < 256; ++x) { f[x] = (uint64_t)((x + ib[15]) & 0xFF); g[x] = (x + ib[15] > 255) ? 1 : 0; }
    orc_gen_lut(8, f, luts); orc_gen_lut(8, g, luts + (size_t)8 * NPOLY);
    orc_wopbs_batch(K, state + (size_t)15 * bytew, 1, 8, luts, 2, 0, out, NULL, NULL, NULL);
    memcpy(state + (size_t)15 * bytew, out, bytew * sizeof(uint64_t));
    memcpy(carry, out + (size_t)8 * lw, lw * sizeof(uint64_t));
    for (int index = 14; index >= 0; --index) {
        memcpy(in9, state + (size_t)index * bytew, bytew * sizeof(uint64_t));
        memcpy(in9 + (size_t)8 * lw, carry, lw * sizeof(uint64_t));
        for (int x = 0; x < 512; ++x) {
            int s = (x & 0xFF) + ((x >> 8) & 1) + ib[index];
            f[x] = (uint64_t)(s & 0xFF); g[x] = s > 255 ? 1 : 0;
        }
        orc_gen_lut(9, f, luts); orc_gen_lut(9, g, luts + (size_t)9 * NPOLY);
        orc_wopbs_batch(K, in9, 1, 9, luts, 2, 0, out, NULL, NULL, NULL);
        memcpy(state + (size_t)index * bytew, out, bytew * sizeof(uint64_t));
        memcpy(carry, out + (size_t)9 * lw, lw * sizeof(uint64_t));
    }
    free(luts); free(in9); free(out); free(carry);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
