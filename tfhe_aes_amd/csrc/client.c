/*
 * client.c -- host-side Client of the FHE-AES engine (libfheaes_client.so, plain C + OpenMP).
 *
 * Mirrors the reference's Client (src/client/client.rs:59-218): key generation
 * (gen_keys_radix + WopbsKey::new_wopbs_key_only_for_wopbs, client.rs:106-107), per-bit
 * encryption without padding (client.rs:128,137) and decryption (client.rs:154).  It exists
 * so that synthetic inputs of the reference's shape can be produced without tfhe-rs; it is
 * NOT on the hot path and contains no FFT: secrets are binary, so every key row is a
 * shift-add negacyclic product.
 *
 * Key layouts (shared with include/fheaes.h; level index 0 = most significant level):
 *   small key  s  [n]  bits;  GLWE key S [k][N] bits;  big LWE key = S flattened (kN bits)
 *   KSK    [kN][ks_level][n+1]              row (i,l) = LWE_s ( S_flat[i] * 2^(64-b(l+1)) )
 *   BSK    [n][pbs_level][k+1][k+1][N]      GGSW(s_i): row (l,r) = GLWE_S( s_i * f_r(2^(64-b(l+1))) ), f_r as below
 *   PFPKSK [k+1][kN+1][pfks_level][(k+1)N]  row (r,i,l) = GLWE_S( f_r(sigma_i * 2^(64-b(l+1))) ),
 *          sigma_i = S_flat[i] (i<kN), sigma_kN = -1;  f_r(x) = -x*S_r(X) (r<k), f_k(x) = x
 * 
 *
 * Randomness: ChaCha20 (RFC 8439 block function, 20 rounds) under two independent 256-bit keys:
 *   secret key  secret keys and every noise sample: one stream per key ciphertext / encryption, nonce = (tag, index),
 *               so the output does not depend on the thread count; Gaussian noise by the Marsaglia polar method with a
 *               series logarithm (+ - * / sqrt only: no libm dependence, bit-reproducible);
 *   mask key    PUBLIC: every mask word of the evaluation keys.  Mask word j of key ciphertext q of key `tag` is 64-bit
 *               word j % 8 of ChaCha20 block j / 8 under (mask key, nonce = (tag, q)): counter-based, random access, so
 *               the engine regenerates all masks on the GPU from (mask key, bodies): fheaes_upload_keys_seeded ships
 *               0.19 GB instead of 1.04 GB (include/fheaes.h).  This is the role tfhe-csprng's public seeds play for
 *               tfhe-rs' Seeded* containers (different generator: theirs is AES-CTR; streams are not interchangeable).
 * The Python Client draws both keys, and a fresh key per encryption call, from os.urandom unless it is given an explicit
 * test seed (client.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../include/fheaes.h"

#define NPOLY 512

/* ---------------------------------------------------------------- PRNG: ChaCha20 */
#define MASK_TAG_KSK 3
#define MASK_TAG_BSK 4
#define MASK_TAG_PFPKSK 5

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7)

/* RFC 8439 section 2.3: state = "expand 32-byte k" | key[8] | counter | nonce[3]; out = 16 words (must stay identical to
 * fheaes_chacha20_block in csrc/kern_linear.h and to client.chacha20_blocks in client.py) */
static void chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16])
{
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 10; ++r) {
        CHACHA_QR(x[0], x[4], x[8], x[12]); CHACHA_QR(x[1], x[5], x[9], x[13]);
        CHACHA_QR(x[2], x[6], x[10], x[14]); CHACHA_QR(x[3], x[7], x[11], x[15]);
        CHACHA_QR(x[0], x[5], x[10], x[15]); CHACHA_QR(x[1], x[6], x[11], x[12]);
        CHACHA_QR(x[2], x[7], x[8], x[13]); CHACHA_QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}
void fheaes_client_chacha20_block(const uint32_t *key8, uint32_t counter, const uint32_t *nonce3, uint32_t *out16) { chacha20_block(key8, counter, nonce3, out16); }

/* one sequential stream: (key, tag, index) */
typedef struct { uint32_t key[8], nonce[3], counter, buf[16]; int pos; } rng_t;

static void rng_seed(rng_t *r, const uint32_t *key8, uint64_t tag, uint64_t index)
{
    memcpy(r->key, key8, 32);
    r->nonce[0] = (uint32_t)tag; r->nonce[1] = (uint32_t)index; r->nonce[2] = (uint32_t)(index >> 32);
    r->counter = 0; r->pos = 16;
}

static inline uint64_t rng_next(rng_t *r)
{
    if (r->pos >= 16) { chacha20_block(r->key, r->counter++, r->nonce, r->buf); r->pos = 0; }
    uint64_t v = (uint64_t)r->buf[r->pos] | ((uint64_t)r->buf[r->pos + 1] << 32);
    r->pos += 2;
    return v;
}

/* ---- the public mask stream: random access, 8 words per block ---- */
typedef struct { const uint32_t *key; uint32_t nonce[3]; uint32_t buf[16]; uint32_t have; } mask_t;
static void mask_seed_ct(mask_t *m, const uint32_t *mask_key8, uint64_t tag, uint64_t ct)
{
    m->key = mask_key8; m->nonce[0] = (uint32_t)tag; m->nonce[1] = (uint32_t)ct; m->nonce[2] = (uint32_t)(ct >> 32); m->have = 0xFFFFFFFFu;
}
static inline uint64_t mask_word(mask_t *m, uint64_t j)
{
    const uint32_t blk = (uint32_t)(j >> 3), w = (uint32_t)(j & 7);
    if (m->have != blk) { chacha20_block(m->key, blk, m->nonce, m->buf); m->have = blk; }
    return (uint64_t)m->buf[2 * w] | ((uint64_t)m->buf[2 * w + 1] << 32);
}
uint64_t fheaes_client_mask_word(const uint32_t *mask_key8, uint64_t tag, uint64_t ct, uint64_t j)
{
    mask_t m;
    mask_seed_ct(&m, mask_key8, tag, ct);
    return mask_word(&m, j);
}

/* ln(x) for x in (0,1], by exponent split + atanh series; deterministic IEEE ops only */
static double det_log(double x)
{
    int e = 0;
    while (x < 0.70710678118654752) { x *= 2.0; --e; }
    while (x >= 1.4142135623730951) { x *= 0.5; ++e; }
    double z = (x - 1.0) / (x + 1.0), z2 = z * z, term = z, sum = 0.0;
    for (int kk = 1; kk <= 41; kk += 2) { sum += term / (double)kk; term *= z2; }
    return 2.0 * sum + (double)e * 0.69314718055994531;
}

/* one N(0,1) sample */
static double rng_gauss(rng_t *r)
{
    for (;;) {
        double u = (double)(int64_t)(rng_next(r) >> 11) * 0x1p-52 - 1.0;
        double v = (double)(int64_t)(rng_next(r) >> 11) * 0x1p-52 - 1.0;
        double s = u * u + v * v;
        if (s > 0.0 && s < 1.0) return u * sqrt(-2.0 * det_log(s) / s);
    }
}

static inline uint64_t noise_word(rng_t *r, double sigma)
{
    if (sigma <= 0.0) return 0;
    double e = rint(rng_gauss(r) * sigma * 0x1p64);
    return (uint64_t)(int64_t)e;
}

/* ---------------------------------------------------------------- helpers */
/* body += A (*) S for a binary S given as the list of its set positions */
static void nega_mac_binary(const uint64_t *a, const int *pos, int npos, uint64_t *body)
{
    for (int q = 0; q < npos; ++q) {
        int t = pos[q];
        for (int j = 0; j < NPOLY - t; ++j) body[j + t] += a[j];
        for (int j = NPOLY - t; j < NPOLY; ++j) body[j + t - NPOLY] -= a[j];
    }
}

static int build_positions(const uint8_t *bits, int *pos)
{
    int c = 0;
    for (int j = 0; j < NPOLY; ++j) if (bits[j]) pos[c++] = j;
    return c;
}

/* fresh GLWE_S(0): mask = public stream `m`, body = sum A_m S_m + e (noise from the secret stream r) */
static void glwe_encrypt_zero(rng_t *r, mask_t *m, int k, const int *pos, const int *npos, double sigma, uint64_t *ct)
{
    uint64_t *body = ct + (size_t)k * NPOLY;
    for (int j = 0; j < k * NPOLY; ++j) ct[j] = mask_word(m, (uint64_t)j);
    for (int j = 0; j < NPOLY; ++j) body[j] = noise_word(r, sigma);
    for (int m = 0; m < k; ++m) nega_mac_binary(ct + (size_t)m * NPOLY, pos + (size_t)m * NPOLY, npos[m], body);
}

/* ---------------------------------------------------------------- API */
void fheaes_client_gen_secret_keys(const fheaes_params *p, const uint32_t *seed /*[8]*/, uint8_t *lwe_sk /*[n]*/, uint8_t *glwe_sk /*[k*N]*/)
{
    rng_t r;
    rng_seed(&r, seed, 1, 0);
    for (uint32_t i = 0; i < p->lwe_dimension; ++i) lwe_sk[i] = (uint8_t)(rng_next(&r) >> 63);
    rng_seed(&r, seed, 2, 0);
    for (uint32_t i = 0; i < p->glwe_dimension * NPOLY; ++i) glwe_sk[i] = (uint8_t)(rng_next(&r) >> 63);
}

void fheaes_client_gen_ksk(const fheaes_params *p, const uint32_t *seed, const uint32_t *mask_seed, const uint8_t *lwe_sk, const uint8_t *glwe_sk,
                           double sigma_lwe, uint64_t *ksk)
{
    int n = (int)p->lwe_dimension, big = (int)(p->glwe_dimension * NPOLY), L = (int)p->ks_level, b = (int)p->ks_base_log;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < big; ++i) {
        rng_t r;
        rng_seed(&r, seed, 3, (uint64_t)i);
        for (int l = 0; l < L; ++l) {
            uint64_t *row = ksk + ((size_t)i * L + l) * (n + 1);
            mask_t mk;
            mask_seed_ct(&mk, mask_seed, MASK_TAG_KSK, (uint64_t)i * L + l);
            uint64_t body = noise_word(&r, sigma_lwe);
            for (int j = 0; j < n; ++j) { row[j] = mask_word(&mk, (uint64_t)j); if (lwe_sk[j]) body += row[j]; }
            if (glwe_sk[i]) body += 1ULL << (64 - b * (l + 1));
            row[n] = body;
        }
    }
}

void fheaes_client_gen_bsk(const fheaes_params *p, const uint32_t *seed, const uint32_t *mask_seed, const uint8_t *lwe_sk, const uint8_t *glwe_sk,
                           double sigma_glwe, uint64_t *bsk)
{
    int n = (int)p->lwe_dimension, k = (int)p->glwe_dimension, k1 = k + 1, L = (int)p->pbs_level, b = (int)p->pbs_base_log;
    size_t gsz = (size_t)k1 * NPOLY;
    int *pos = (int *)malloc((size_t)k * NPOLY * sizeof(int));
    int npos[16];
    for (int m = 0; m < k; ++m) npos[m] = build_positions(glwe_sk + (size_t)m * NPOLY, pos + (size_t)m * NPOLY);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        rng_t r;
        rng_seed(&r, seed, 4, (uint64_t)i);
        for (int l = 0; l < L; ++l) for (int rr = 0; rr < k1; ++rr) {
            const size_t q = ((size_t)i * L + l) * k1 + rr;
            uint64_t *ct = bsk + q * gsz;
            mask_t mk;
            mask_seed_ct(&mk, mask_seed, MASK_TAG_BSK, (uint64_t)q);
            glwe_encrypt_zero(&r, &mk, k, pos, npos, sigma_glwe, ct);
            if (!lwe_sk[i]) continue;
            /* row (l, r) carries s_i * g_l on component r.  The message goes into the BODY (row r < k: -g_l * S_r(X),
             * row k: +g_l), never into a mask polynomial: same phase as adding g_l to mask coefficient 0 of polynomial r,
             * and every mask word stays the public stream (seeded upload regenerates it). */
            const uint64_t g = 1ULL << (64 - b * (l + 1));
            uint64_t *body = ct + (size_t)k * NPOLY;
            if (rr == k) body[0] += g;
            else for (int j = 0; j < NPOLY; ++j) if (glwe_sk[(size_t)rr * NPOLY + j]) body[j] -= g;
        }
    }
    free(pos);
}

void fheaes_client_gen_pfpksk(const fheaes_params *p, const uint32_t *seed, const uint32_t *mask_seed, const uint8_t *glwe_sk, double sigma_pfks, uint64_t *pfpksk)
{
    int k = (int)p->glwe_dimension, k1 = k + 1, L = (int)p->pfks_level, b = (int)p->pfks_base_log;
    int big = k * NPOLY, big1 = big + 1;
    size_t gsz = (size_t)k1 * NPOLY;
    int *pos = (int *)malloc((size_t)k * NPOLY * sizeof(int));
    int npos[16];
    for (int m = 0; m < k; ++m) npos[m] = build_positions(glwe_sk + (size_t)m * NPOLY, pos + (size_t)m * NPOLY);
#pragma omp parallel for schedule(static) collapse(2)
    for (int rr = 0; rr < k1; ++rr) for (int i = 0; i < big1; ++i) {
        rng_t r;
        rng_seed(&r, seed, 5, (uint64_t)rr * (uint64_t)big1 + (uint64_t)i);
        /* sigma_i in {0, 1, -1} */
        int sig = (i < big) ? (int)glwe_sk[i] : -1;
        for (int l = 0; l < L; ++l) {
            const size_t q = ((size_t)rr * big1 + i) * L + l;
            uint64_t *ct = pfpksk + q * gsz;
            uint64_t *body = ct + (size_t)k * NPOLY;
            mask_t mk;
            mask_seed_ct(&mk, mask_seed, MASK_TAG_PFPKSK, (uint64_t)q);
            glwe_encrypt_zero(&r, &mk, k, pos, npos, sigma_pfks, ct);
            if (sig == 0) continue;
            uint64_t g = 1ULL << (64 - b * (l + 1));
            uint64_t x = (sig > 0) ? g : (uint64_t)0 - g;            /* sigma_i * g_l */
            if (rr == k) body[0] += x;                               /* f_k(x) = x */
            else for (int j = 0; j < NPOLY; ++j) if (glwe_sk[(size_t)rr * NPOLY + j]) body[j] -= x;  /* -x * S_r(X) */
        }
    }
    free(pos);
}

/* encrypt_without_padding (client.rs:128): `count` bits -> LWE under the big key, bit at the MSB */
void fheaes_client_encrypt_bits(const fheaes_params *p, const uint32_t *seed /*[8]: one fresh key per call*/, const uint8_t *glwe_sk, double sigma,
                                const uint8_t *bits, uint64_t count, uint64_t *lwe_out)
{
    int big = (int)(p->glwe_dimension * NPOLY);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)count; ++q) {
        rng_t r;
        rng_seed(&r, seed, 6, (uint64_t)q);
        uint64_t *ct = lwe_out + (size_t)q * (big + 1);
        uint64_t body = noise_word(&r, sigma);
        for (int j = 0; j < big; ++j) { ct[j] = rng_next(&r); if (glwe_sk[j]) body += ct[j]; }
        body += (uint64_t)(bits[q] & 1) << 63;
        ct[big] = body;
    }
}

/* decrypt_without_padding (client.rs:154): bit = round(phase / 2^63); phase_out (optional) = raw phases */
void fheaes_client_decrypt_bits(const fheaes_params *p, const uint8_t *glwe_sk, const uint64_t *lwe_in, uint64_t count,
                                uint8_t *bits_out, uint64_t *phase_out)
{
    int big = (int)(p->glwe_dimension * NPOLY);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)count; ++q) {
        const uint64_t *ct = lwe_in + (size_t)q * (big + 1);
        uint64_t ph = ct[big];
        for (int j = 0; j < big; ++j) if (glwe_sk[j]) ph -= ct[j];
        if (phase_out) phase_out[q] = ph;
        bits_out[q] = (uint8_t)(((ph + (1ULL << 62)) >> 63) & 1);
    }
}

/* phases of small-key LWEs / GLWE bodies, for noise diagnostics in tests */
void fheaes_client_phase_small(const fheaes_params *p, const uint8_t *lwe_sk, const uint64_t *lwe_in, uint64_t count, uint64_t *phase_out)
{
    int n = (int)p->lwe_dimension;
    for (uint64_t q = 0; q < count; ++q) {
        const uint64_t *ct = lwe_in + (size_t)q * (n + 1);
        uint64_t ph = ct[n];
        for (int j = 0; j < n; ++j) if (lwe_sk[j]) ph -= ct[j];
        phase_out[q] = ph;
    }
}

void fheaes_client_glwe_phase(const fheaes_params *p, const uint8_t *glwe_sk, const uint64_t *glwe_in, uint64_t count, uint64_t *phase_out /*[count][N]*/)
{
    int k = (int)p->glwe_dimension;
    size_t gsz = (size_t)(k + 1) * NPOLY;
    int *pos = (int *)malloc((size_t)k * NPOLY * sizeof(int));
    int npos[16];
    for (int m = 0; m < k; ++m) npos[m] = build_positions(glwe_sk + (size_t)m * NPOLY, pos + (size_t)m * NPOLY);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)count; ++q) {
        const uint64_t *ct = glwe_in + (size_t)q * gsz;
        uint64_t tmp[NPOLY];
        memset(tmp, 0, sizeof tmp);
        for (int m = 0; m < k; ++m) nega_mac_binary(ct + (size_t)m * NPOLY, pos + (size_t)m * NPOLY, npos[m], tmp);
        for (int j = 0; j < NPOLY; ++j) phase_out[(size_t)q * NPOLY + j] = ct[(size_t)k * NPOLY + j] - tmp[j];
    }
    free(pos);
}
