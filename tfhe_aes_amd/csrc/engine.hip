// engine.hip -- C ABI (include/fheaes.h) and host-side schedule of the MI355X FHE-AES engine.
//
// The host code here is the native counterpart of the reference's Server
// (src/server/server.rs:24-282) and S-Box front end (src/server/sbox/sbox.rs:46-97,
// src/server/sbox/many_wopbs.rs:31-116): it owns the device copies of the keys, the workspace,
// the precomputed AES LUT sets, and enqueues the kernels of kern_*.h on one HIP stream.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "knobs.h"
#include "fheaes.h"
#include "fft_dev.h"
#include "kern_extprod.h"
#include "kern_blindrot_latency.h"
#include "kern_blindrot16.h"
#include "kern_blindrot_pair.h"
#include "ks_launch.h"                 // kern_keyswitch.h: the kernels themselves, or (two-unit product build) their argument block + launch functions
#include "kern_linear.h"

#define FHEAES_VERSION_STR "fheaes-mi355x 0.3 (gfx950)" FHEAES_BUILD_KIND      /* " dev" when built with developer knobs (knobs.h) */
#define MAX_CHUNK_BITS 32768ull
#define MAX_WOPBS_BITS 16u            /* widest radix input of many_wopbs_without_padding (LUT of 2^16 entries per output bit) */

namespace {

std::string g_create_error;
// fheaes_last_error(): a context may be shared between threads (every call takes its lock), so the message a caller reads must
// not be one that another thread is overwriting.  fail() keeps a per-thread copy; the pointer fheaes_last_error returns is valid
// until the same thread's next call into the library.
thread_local std::string tl_error;
thread_local uint64_t tl_error_ctx_id = 0;           // id of the context tl_error belongs to (ids are never reused: a new context at a
                                                     // destroyed one's address does not inherit its message)
std::atomic<uint64_t> g_next_ctx_id{1};

// ---------------------------------------------------------------------------------------------
// host tables
// ---------------------------------------------------------------------------------------------
struct HostTwiddles {
    double psi_re[FHE_N], psi_im[FHE_N];
    HostTwiddles()
    {
        // psi^j = exp(i pi j/512): half-angle recurrences in long double, products of the
        // binary powers, octant symmetry (same specification as oracle/fheaes_oracle.c).
        long double bc[8], bs[8];
        bc[7] = sqrtl(0.5L); bs[7] = bc[7];
        for (int m = 6; m >= 0; --m) {
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
            psi_re[j] = (double)pr; psi_im[j] = (double)pi;
        }
        psi_re[0] = 1.0; psi_im[0] = 0.0;
        psi_im[128] = psi_re[128];
        for (int j = 129; j <= 256; ++j) { psi_re[j] = psi_im[256 - j]; psi_im[j] = psi_re[256 - j]; }
        for (int j = 257; j < 512; ++j) { psi_re[j] = -psi_re[512 - j]; psi_im[j] = psi_im[512 - j]; }
    }
    // psi^e, e mod 1024
    void pow(int e, double &re, double &im) const
    {
        e &= 1023;
        if (e < 512) { re = psi_re[e]; im = psi_im[e]; }
        else { re = -psi_re[e - 512]; im = -psi_im[e - 512]; }
    }
};

const HostTwiddles &twiddles()
{
    static HostTwiddles t;
    return t;
}

// AES tables from the field definition (tables/table.rs, sbox.rs:20-42)
struct AesTables {
    uint8_t sbox[256], inv[256];
    static uint8_t mul(uint8_t a, uint8_t b)
    {
        uint8_t r = 0;
        for (int i = 0; i < 8; ++i) { if (b & 1) r ^= a; uint8_t hi = a & 0x80; a = (uint8_t)(a << 1); if (hi) a ^= 0x1B; b >>= 1; }
        return r;
    }
    AesTables()
    {
        for (int x = 0; x < 256; ++x) {
            uint8_t y = 0;
            if (x) for (int c = 1; c < 256; ++c) if (mul((uint8_t)x, (uint8_t)c) == 1) { y = (uint8_t)c; break; }
            uint8_t s = y, v = y;
            for (int i = 0; i < 4; ++i) { v = (uint8_t)((v << 1) | (v >> 7)); s ^= v; }
            s ^= 0x63;
            sbox[x] = s; inv[s] = (uint8_t)x;
        }
    }
};

const AesTables &aes_tables()
{
    static AesTables t;
    return t;
}

enum { LUTSET_ENC_ROUND = 0, LUTSET_SBOX, LUTSET_INV_SBOX, LUTSET_DEC_MUL, LUTSET_IDENTITY, LUTSET_COUNT };

// words of one (LUT, output bit) row: gen_lut.rs:19-23, lut_size = max(2^nb_block, polynomial_size)
inline uint64_t lut_row_words(uint32_t nb) { return nb > 9 ? (1ull << nb) : (uint64_t)FHE_N; }

void gen_lut_host(uint32_t nb, const uint64_t *f, uint64_t *out)
{
    const uint64_t W = lut_row_words(nb);
    for (uint64_t idx = 0; idx < W; ++idx) {
        uint64_t v = f[idx & ((1ull << nb) - 1)];
        for (uint32_t b = 0; b < nb; ++b) out[(size_t)b * W + idx] = ((v >> b) & 1ull) << 63;
    }
}

int build_lutset_host(int which, std::vector<uint64_t> &out)
{
    const AesTables &T = aes_tables();
    uint64_t f[4][256];
    int n = 1;
    for (int x = 0; x < 256; ++x) {
        uint8_t s = T.sbox[x];
        switch (which) {
        case LUTSET_ENC_ROUND: f[0][x] = s; f[1][x] = AesTables::mul(s, 2); f[2][x] = AesTables::mul(s, 3); n = 3; break;
        case LUTSET_SBOX: f[0][x] = s; break;
        case LUTSET_INV_SBOX: f[0][x] = T.inv[x]; break;
        case LUTSET_DEC_MUL:
            f[0][x] = AesTables::mul((uint8_t)x, 9); f[1][x] = AesTables::mul((uint8_t)x, 11);
            f[2][x] = AesTables::mul((uint8_t)x, 13); f[3][x] = AesTables::mul((uint8_t)x, 14); n = 4; break;
        default: f[0][x] = (uint64_t)x; break;
        }
    }
    out.assign((size_t)n * 8 * FHE_N, 0);
    for (int i = 0; i < n; ++i) gen_lut_host(8, f[i], out.data() + (size_t)i * 8 * FHE_N);
    return n;
}

const int MC_ENC[4][4] = {{1, 2, 0, 0}, {0, 1, 2, 0}, {0, 0, 1, 2}, {2, 0, 0, 1}};
const int MC_DEC[4][4] = {{3, 1, 2, 0}, {0, 3, 1, 2}, {2, 0, 3, 1}, {1, 2, 0, 3}};

GatherTable table_enc_round()
{
    GatherTable t{}; t.terms = 4;
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) for (int r2 = 0; r2 < 4; ++r2) {
        t.src[4 * col + row][r2] = (int8_t)(4 * ((col + r2) & 3) + r2);
        t.lut[4 * col + row][r2] = (int8_t)MC_ENC[row][r2];
    }
    return t;
}
GatherTable table_shift_rows(bool inverse)
{
    GatherTable t{}; t.terms = 1;
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) {
        t.src[4 * col + row][0] = (int8_t)(4 * ((inverse ? col - row : col + row) & 3) + row);
        t.lut[4 * col + row][0] = 0;
    }
    return t;
}
GatherTable table_dec_mix()
{
    GatherTable t{}; t.terms = 4;
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) for (int r2 = 0; r2 < 4; ++r2) {
        t.src[4 * col + row][r2] = (int8_t)(4 * col + r2);
        t.lut[4 * col + row][r2] = (int8_t)MC_DEC[row][r2];
    }
    return t;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct fheaes_ctx {
    fheaes_params p{};
    const uint64_t id = g_next_ctx_id.fetch_add(1);
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    // The reference shares one `&Server` between rayon worker threads (main.rs:55-61).  A context is one GPU stream and one
    // workspace, so concurrent calls are made SAFE by serialising them here (batched calls are the way to use the GPU; this
    // only guarantees that a drop-in that keeps the per-block thread pool does not corrupt the workspace).
    std::recursive_mutex mu;
    // shapes
    uint32_t n = 0, k = 0, k1 = 0, big = 0, big1 = 0;
    uint32_t cu_count = 256;             // compute units of the device (MI355X: 256)
    // how the last fheaes_clone_keys INTO this context moved the key images (fheaes_clone_info)
    int clone_path = FHEAES_CLONE_NONE;
    uint64_t clone_bytes = 0;
    double clone_seconds = 0.0;
    // noise guard (the reference runs tfhe-rs with `noise-asserts` and MaxNoiseLevel::new(5), Cargo.toml:7, client.rs:92): the linear
    // layers count how many nominal-noise ciphertexts (fresh WoPBS outputs, round keys, client encryptions) they sum into one
    uint32_t noise_level_seen = 0;
    int k2_home = -1;                    // blind rotation: 1 = the LDS-home form runs two workgroups per CU here (queried once), 0 = parked form
    int k2_pair_ok = -1;                 // 1 = the paired kernel (159,504 B of LDS per workgroup) can be resident on a CU here (queried once)
    int k2_park_claim = 1;               // paired kernel's parking slots: 1 = claimed from a shared pool (kern_blindrot_pair.h), 0 = one private slot per workgroup
    // keys
    int8_t *ksk_frag = nullptr, *pfpksk_frag = nullptr;      // balanced key bytes in MFMA B-fragment order
    uint32_t ks_ksteps = 0, ks_coltiles = 0, pf_ksteps = 0, pf_coltiles = 0;
    size_t ksk_frag_bytes = 0, pfpksk_frag_bytes = 0, bskf_bytes = 0;
    double2 *bskf = nullptr;
    bool have_keys = false;
    // tables
    double2 *tw_d = nullptr;            // the transform's table T[17 k1 + b] = psi^(b (4 k1 + 1)) (fft_dev.h)
    uint64_t *lutset_d[LUTSET_COUNT] = {};
    int lutset_n[LUTSET_COUNT] = {};
    // workspace
    DevBuf ws_small, ws_pbs, ws_ggsw, ws_ggswf, ws_vp, ws_tmp_a, ws_tmp_b, ws_luts, ws_misc, ws_digits, ws_park, ws_park_owner, ws_tree;
    DevBuf stage[4];                     // host-memspace calls stage their arguments here (grow-only, reused)
    // pinned host staging for the counter bytes of add_scalar; `pin_ev` marks the last copy out of it
    uint8_t *pin = nullptr;
    size_t pin_bytes = 0;
    hipEvent_t pin_ev = nullptr;
    // profiling
    bool prof = false;
    struct Pending { hipEvent_t a, b; int stage; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    double stage_ms[FHEAES_STAGE_COUNT] = {};
    uint64_t stage_launches[FHEAES_STAGE_COUNT] = {};
    uint64_t stage_units[FHEAES_STAGE_COUNT] = {};

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        tl_error = buf;
        tl_error_ctx_id = id;
        return code;
    }
};

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return (ctx)->fail(FHEAES_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define TRY(expr)                  \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != FHEAES_OK) return rc__; \
    } while (0)

struct CtxLock {
    std::unique_lock<std::recursive_mutex> lk;
    explicit CtxLock(const fheaes_ctx *c) { if (c) lk = std::unique_lock<std::recursive_mutex>(const_cast<fheaes_ctx *>(c)->mu); }
};

namespace {

int ensure(fheaes_ctx *c, DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes) return FHEAES_OK;
    if (b.p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { b.p = nullptr; return c->fail(FHEAES_ERR_NOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    b.bytes = bytes;
    return FHEAES_OK;
}

// ---- profiling -------------------------------------------------------------------------------
int prof_flush(fheaes_ctx *c)
{
    for (auto &pe : c->pending) {
        HIP_TRY(c, hipEventSynchronize(pe.b));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, pe.a, pe.b));
        c->stage_ms[pe.stage] += ms;
        c->free_events.push_back(pe.a);
        c->free_events.push_back(pe.b);
    }
    c->pending.clear();
    return FHEAES_OK;
}

struct StageScope {
    fheaes_ctx *c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    StageScope(fheaes_ctx *ctx, int st, uint64_t units) : c(ctx), stage(st), on(ctx->prof)
    {
        c->stage_launches[stage] += 1;
        c->stage_units[stage] += units;
        if (!on) return;
        if (c->pending.size() > 4096) prof_flush(c);
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->free_events.empty()) { e = c->free_events.back(); c->free_events.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = get(); b = get();
        if (a) (void)hipEventRecord(a, c->stream);
    }
    ~StageScope()
    {
        if (!on || !a || !b) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({a, b, stage});
    }
};

#ifdef EP_STAMPS
// developer build: per-phase cycle counts written by the blind-rotation kernels
struct StampReport {
    fheaes_ctx *c; unsigned long long *d = nullptr; size_t waves; const char *const *names; int per_wg;
    StampReport(fheaes_ctx *ctx, size_t waves_, const char *const *names_, int per_wg_ = 0) : c(ctx), waves(waves_), names(names_), per_wg(per_wg_)
    {
        (void)hipMalloc((void **)&d, waves * EP_NPH * 8);
        (void)hipMemsetAsync(d, 0, waves * EP_NPH * 8, c->stream);
    }
    ~StampReport()
    {
        std::vector<unsigned long long> h(waves * EP_NPH);
        (void)hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(d);
        double tot[EP_NPH] = {}, all = 0;
        for (size_t w = 0; w < waves; ++w) for (int i = 0; i < EP_NPH; ++i) tot[i] += (double)h[w * EP_NPH + i];
        for (int i = 0; i < EP_NPH; ++i) all += tot[i];
        fprintf(stderr, "K2 phase cycles per wave per iteration (s_memtime ticks, avg over %zu waves):\n", waves);
        for (int i = 0; i < EP_NPH; ++i) fprintf(stderr, "  %-32s %9.0f  %5.1f %%\n", names[i], tot[i] / ((double)waves * c->n), 100.0 * tot[i] / all);
        fprintf(stderr, "  %-32s %9.0f\n", "total", all / ((double)waves * c->n));
        if (per_wg) {                                   // the same per wave slot of a workgroup (who is the slowest at each barrier)
            fprintf(stderr, "  per wave of a workgroup:      ");
            for (int w = 0; w < per_wg; ++w) fprintf(stderr, " %7d", w);
            fprintf(stderr, "\n");
            for (int i = 0; i < EP_NPH; ++i) {
                fprintf(stderr, "  %-30s", names[i]);
                for (int w = 0; w < per_wg; ++w) {
                    double t = 0;
                    for (size_t g = w; g < waves; g += per_wg) t += (double)h[g * EP_NPH + i];
                    fprintf(stderr, " %7.0f", t / ((double)(waves / per_wg) * c->n));
                }
                fprintf(stderr, "\n");
            }
        }
    }
};
#endif

// ---- how a blind-rotation batch is cut into workgroups (fheaes_k2_launch_plan) -------------------------------------------------
#ifndef LATENCY_BATCH_BITS
#define LATENCY_BATCH_BITS 256ull      /* at most one 512-thread workgroup per CU */
#endif
#ifndef PBS_BALANCE
#define PBS_BALANCE 1
#endif
#ifndef PBS_SMALL_R2
#define PBS_SMALL_R2 1
#endif
#ifndef K2_PAIR
#define K2_PAIR 1                      /* batches above K2_PAIR_MIN_BITS take the paired form (kern_blindrot_pair.h): one 512-thread workgroup per CU */
#endif
#ifndef K2_PAIR_MIN_BITS
#define K2_PAIR_MIN_BITS 768ull
#endif
#ifndef K2_PAIR_TAIL4
#define K2_PAIR_TAIL4 1                /* 1: the paired form fills its last generation with four-ciphertext units on every CU; 0: six-ciphertext units only
                                          (the last generation then covers fewer CUs) -- measured at 4,096 bits, see DESIGN.md */
#endif
struct K2Plan { int form; uint64_t units_main; uint32_t r_main; uint64_t units_tail; uint32_t r_tail; };
K2Plan k2_plan(uint64_t m, uint32_t cu_count, uint32_t k1, bool allow_pair = true)
{
    K2Plan pl{};
    if (m <= LATENCY_BATCH_BITS) { pl.form = 0; pl.units_main = m; pl.r_main = 1; return pl; }
    if (K2_PAIR && allow_pair && k1 == 5 && m > K2_PAIR_MIN_BITS) {
        // paired form: units of 6 and of 4 ciphertexts, one workgroup per CU, a whole number of generations that covers the batch
        // (16,384 bits = 2,560 x 6 + 256 x 4 = 11 generations; 4,096 = 512 x 6 + 256 x 4 = 3; 1,152 = 64 x 6 + 192 x 4 = 1); the
        // smaller units last
        pl.form = 2; pl.r_main = 6; pl.r_tail = 4;
        const uint64_t gens = (m + 6ull * cu_count - 1) / (6ull * cu_count);
        uint64_t nu = gens * cu_count;
        uint64_t four = 6 * nu >= m ? (6 * nu - m) / 2 : 0;      // units that can give up two of their six slots
        if (!K2_PAIR_TAIL4) { pl.units_main = (m + 5) / 6; pl.units_tail = 0; return pl; }
        if (four > nu) four = nu;
        if (four == nu && 4 * nu > m) { nu = (m + 3) / 4; four = nu; }          // less than one generation of 4-ciphertext units
        pl.units_tail = four; pl.units_main = nu - four;
        return pl;
    }
    pl.form = 1;
    pl.r_main = k1 == 5 ? 3 : 8;
    pl.units_main = (m + pl.r_main - 1) / pl.r_main;
    if (PBS_BALANCE && k1 == 5) {
        const uint64_t slots = 2ull * cu_count;
        pl.r_tail = 2;
        if (PBS_SMALL_R2 && (m + 1) / 2 <= cu_count) {
            // at most one two-ciphertext unit per CU: shorter units than three-ciphertext ones, still one per CU
            pl.units_main = 0;
            pl.units_tail = (m + 1) / 2;
        } else if (pl.units_main > slots) {
            // more units than slots (two workgroups per CU): a whole number of generations of 3- and 2-ciphertext units that
            // cover the batch exactly, the 2-ciphertext ones last (see blind_rotate16_kernel)
            const uint64_t nu = slots * ((m + 3 * slots - 1) / (3 * slots));
            if (2 * nu <= m) {
                pl.units_tail = 3 * nu - m;
                pl.units_main = nu - pl.units_tail;
            }
        }
    }
    return pl;
}

// ---- kernel launchers ------------------------------------------------------------------------
int launch_keyswitch(fheaes_ctx *c, const uint64_t *in, uint64_t m, uint64_t *out)
{
    if (m == 0) return FHEAES_OK;
    StageScope sc(c, FHEAES_STAGE_KEYSWITCH, m);
    const uint64_t ct_tiles16 = ((m + KS_CT_TILE - 1) / KS_CT_TILE) * (KS_CT_TILE / 16);
    TRY(ensure(c, c->ws_digits, ct_tiles16 * c->ks_ksteps * 1024));
    int8_t *af = (int8_t *)c->ws_digits.p;
    const uint64_t threads = ct_tiles16 * c->ks_ksteps * 64;
    ks_launch_digits_k1(dim3((unsigned)((threads + 255) / 256)), c->stream, in, (uint64_t)c->big1, c->big, m, c->ks_ksteps, af);
    KeyswitchArgs a{};
    a.afrag = af; a.bfrag = c->ksk_frag; a.ksteps = c->ks_ksteps; a.coltiles = c->ks_coltiles;
    a.in = in; a.in_stride = c->big1; a.body_index = (int32_t)c->big; a.body_col = c->n; a.ncols = c->n + 1;
    a.out = out; a.out_stride = c->n + 1; a.out_z_stride = 0; a.m = m;
#ifndef KS1_LDS
#define KS1_LDS 1                      /* 1 (round 6: 1.87 -> 1.48 ms per 16,384-bit launch, same words): K1 through the LDS-tiled kernel too; 0: one wave = one tile, operands from L2 */
#endif
#if KS1_LDS
    dim3 grid((c->ks_coltiles + KSL_COL_TILES - 1) / KSL_COL_TILES, (unsigned)((m + 16 * KSL_CT_TILES - 1) / (16 * KSL_CT_TILES)), 1);
    ks_launch_mfma_lds(1, grid, c->stream, a);
#else
    dim3 grid((c->ks_coltiles + 3) / 4, (unsigned)((m + KS_CT_TILE - 1) / KS_CT_TILE), 1);
    ks_launch_mfma(1, grid, c->stream, a);
#endif
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

// out: rows of one GGSW level: [m][out_stride] with key r at offset r*(k+1)N
int launch_pfpks(fheaes_ctx *c, const uint64_t *in, uint64_t m, uint64_t *out, uint64_t out_stride)
{
    if (m == 0) return FHEAES_OK;
    StageScope sc(c, FHEAES_STAGE_PFPKS, m);
    const uint32_t gsz = c->k1 * FHE_N;
    const uint64_t ct_tiles16 = ((m + KS_CT_TILE - 1) / KS_CT_TILE) * (KS_CT_TILE / 16);
    TRY(ensure(c, c->ws_digits, ct_tiles16 * c->pf_ksteps * 2 * 1024));
    int8_t *af = (int8_t *)c->ws_digits.p;
    const uint64_t threads = ct_tiles16 * c->pf_ksteps * 64;
    ks_launch_digits_k3(dim3((unsigned)((threads + 255) / 256)), c->stream, in, (uint64_t)c->big1, c->big1, m, c->pf_ksteps, af);
    KeyswitchArgs a{};
    a.afrag = af; a.bfrag = c->pfpksk_frag; a.ksteps = c->pf_ksteps; a.coltiles = c->pf_coltiles;
    a.in = in; a.in_stride = c->big1; a.body_index = -1; a.body_col = 0; a.ncols = gsz;
    a.out = out; a.out_stride = out_stride; a.out_z_stride = gsz; a.m = m;
#ifndef KS_LDS
#define KS_LDS 1
#endif
#if KS_LDS
    dim3 grid((c->pf_coltiles + KSL_COL_TILES - 1) / KSL_COL_TILES, (unsigned)((m + 16 * KSL_CT_TILES - 1) / (16 * KSL_CT_TILES)), c->k1);
    ks_launch_mfma_lds(2, grid, c->stream, a);
#else
    dim3 grid((c->pf_coltiles + 3) / 4, (unsigned)((m + KS_CT_TILE - 1) / KS_CT_TILE), c->k1);
    ks_launch_mfma(2, grid, c->stream, a);
#endif
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

int launch_forward_fourier(fheaes_ctx *c, const uint64_t *in, uint64_t polys, double2 *out, int stage)
{
    if (polys == 0) return FHEAES_OK;
    StageScope sc(c, stage, polys);
    uint64_t wgs = (polys + EP_GROUPS - 1) / EP_GROUPS;
    if (wgs > 8192) wgs = 8192;
    hipLaunchKernelGGL(forward_fourier_kernel, dim3((unsigned)wgs), dim3(EP_THREADS), 0, c->stream, in, out, polys, c->tw_d);
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

// the paired kernel takes nearly all of a CU's LDS: where the runtime cannot place even one such workgroup (a driver that reserves LDS)
// every batch falls back to the 16-form instead of failing the launch
bool k2_pair_allowed(fheaes_ctx *c)
{
    if (c->k2_pair_ok < 0) {
        int per_cu = 0;
        const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blind_rotate_pair_kernel<5, 5, 8, 3, 2>, BRP_THREADS, 0);
        c->k2_pair_ok = (oe == hipSuccess && per_cu >= 1) ? 1 : 0;
        (void)hipGetLastError();
    }
    return c->k2_pair_ok == 1;
}

// the 16-form's LDS-home variant takes exactly half of a CU's 160 KB per workgroup: use it only where the runtime really places two
bool k2_home_allowed(fheaes_ctx *c)
{
    if (c->k1 != 5) return false;
    if (c->k2_home < 0) {
        int per_cu = 0;
        const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blind_rotate16_kernel<5, 5, 8, 3, 2, true>, EP_THREADS, 0);
        c->k2_home = (BR16_W3_LDS_HOME && oe == hipSuccess && per_cu >= 2) ? 1 : 0;
        (void)hipGetLastError();         // a failed query means "fall back", not a failed launch
    }
    return c->k2_home == 1;
}

// bytes of the paired kernel's parking slab for a launch of `grid` workgroups: claimed slots = the shared pool + one private overflow slot
// per workgroup behind it (never touched unless a pool is exhausted), private slots = one per workgroup
size_t k2_pair_park_bytes(const fheaes_ctx *c, uint64_t grid)
{
    return (size_t)((c->k2_park_claim ? BRP_PARK_SLOTS : 0) + grid) * 2 * BRP_PARK_WORDS_PER_HALF * 8;
}

// the kernel a blind-rotation batch of m bits really takes on this context (form of K2Plan after the occupancy fallbacks) and its name
K2Plan k2_context_plan(fheaes_ctx *c, uint64_t m, const char **kernel)
{
    const K2Plan pl = k2_plan(m, c->cu_count, c->k1, k2_pair_allowed(c));
    if (kernel) {
        if (pl.form == 0) *kernel = c->k1 == 5 ? "blind_rotate_latency_kernel<5,5,8>" : "blind_rotate_latency_kernel<2,5,8>";
        else if (pl.form == 2) *kernel = c->k2_park_claim ? "blind_rotate_pair_kernel<5,5,8,3,2> parking=claimed" : "blind_rotate_pair_kernel<5,5,8,3,2> parking=private";
        else if (c->k1 != 5) *kernel = "blind_rotate16_kernel<2,5,8,8,0,false>";
        else *kernel = k2_home_allowed(c) ? "blind_rotate16_kernel<5,5,8,3,2,true>" : "blind_rotate16_kernel<5,5,8,3,2,false>";
    }
    return pl;
}

int launch_cbs_pbs(fheaes_ctx *c, const uint64_t *lwe_small, uint64_t m, uint32_t level, uint64_t *out)
{
    if (m == 0) return FHEAES_OK;
    // the blind-rotation kernels address the Fourier BSK as ONE raw buffer with 32-bit byte offsets
    if ((uint64_t)c->n * c->p.pbs_level * c->k1 * c->k1 * FHE_H * 16 > 0x7FFFFFFFull)
        return c->fail(FHEAES_ERR_INVALID, "bootstrapping key larger than 2 GiB is not supported by the blind-rotation kernels");
    StageScope sc(c, FHEAES_STAGE_BLIND_ROTATE, m);
    ExtProdArgs a{};
    a.ggsw = c->bskf; a.tw = c->tw_d;
    a.out = out; a.count = m; a.iters = c->n; a.lwe_in = lwe_small;
    const uint64_t half_delta = 1ull << (64 - c->p.cbs_base_log * level - 1);
    a.tv_const = (uint64_t)0 - half_delta; a.body_shift = 1ull << 62; a.post_add = half_delta;
    if (m <= LATENCY_BATCH_BITS) {
        // latency regime: one ciphertext per 512-thread workgroup, all levels transformed at once (kern_blindrot_latency.h)
#ifdef EP_STAMPS
        static const char *namesL[EP_NPH] = {"barrier (result) + accumulate", "rotate+decompose (1 coeff x K1)", "read digits + forward fft", "digit stores", "barrier (digits)", "MAC", "barrier (MAC done)",
                                             "products store + next rows", "barrier (products)", "inverse fft -> doubles", "barrier (acc)", "barrier (decomposition)"};
        StampReport rep(c, (size_t)m * 8, namesL, 8);
        a.stamps = rep.d;
#endif
        if (c->k1 == 5) hipLaunchKernelGGL((blind_rotate_latency_kernel<5, 5, 8>), dim3((unsigned)m), dim3(BL_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((blind_rotate_latency_kernel<2, 5, 8>), dim3((unsigned)m), dim3(BL_THREADS), 0, c->stream, a);
        // (257..768 bits: the throughput form below with at most one workgroup per CU, 14.6 ms per launch; the round-1
        //  one-ciphertext-per-workgroup form of kern_extprod.h took 21.6 ms there and the latency form in two waves 16-18 ms)
    } else if (k2_plan(m, c->cu_count, c->k1, k2_pair_allowed(c)).form == 2) {
        // paired throughput form (kern_blindrot_pair.h): one 512-thread workgroup per CU, 6 (or 4) ciphertexts share every key fetch
        const K2Plan pl = k2_plan(m, c->cu_count, c->k1, true);
        const unsigned gridp = (unsigned)(pl.units_main + pl.units_tail);
        a.units_main = (uint32_t)pl.units_main;
        const size_t park_bytes = k2_pair_park_bytes(c, gridp);
        if (park_bytes > 0x7FFFFFFFull) return c->fail(FHEAES_ERR_INVALID, "internal: parking slab of %zu bytes exceeds one raw buffer", park_bytes);
        TRY(ensure(c, c->ws_park, park_bytes));
        a.park = (uint64_t *)c->ws_park.p; a.park_bytes = park_bytes;
        if (c->k2_park_claim) {
            // owner words of the shared slots: all free when a launch starts (every workgroup gives its slot back before it ends; the
            // memset makes that hold even after a launch that was aborted)
            TRY(ensure(c, c->ws_park_owner, BRP_PARK_SLOTS * sizeof(uint32_t)));
            HIP_TRY(c, hipMemsetAsync(c->ws_park_owner.p, 0, BRP_PARK_SLOTS * sizeof(uint32_t), c->stream));
            a.park_owner = (uint32_t *)c->ws_park_owner.p;
        }
#ifdef EP_STAMPS
        static const char *namesP[EP_NPH] = {"stage+rotate+decomp_first", "decomp_next", "fwd head", "pre-level barrier", "fwd tail (transpose+dft16)",
                                             "digit stores+late loads", "exchange barrier", "MAC", "products exchange", "inverse fft", "convert+add", "loop head"};
        StampReport rep(c, (size_t)gridp * 8, namesP);
        a.stamps = rep.d;
#endif
        hipLaunchKernelGGL((blind_rotate_pair_kernel<5, 5, 8, 3, 2>), dim3(gridp), dim3(BRP_THREADS), 0, c->stream, a);
    } else {
        // throughput form (kern_blindrot16.h): accumulator parked in HBM between uses, key rows prefetched across the transform
        const K2Plan pl = k2_plan(m, c->cu_count, c->k1, false);
        const unsigned grid16 = (unsigned)(pl.units_main + pl.units_tail);
        a.units_main = (uint32_t)pl.units_main;
        const size_t park_bytes = (size_t)grid16 * BR16_PARK_WORDS_PER_WG * 8;
        if (park_bytes > 0x7FFFFFFFull) return c->fail(FHEAES_ERR_INVALID, "internal: parking slab of %zu bytes exceeds one raw buffer", park_bytes);
        TRY(ensure(c, c->ws_park, park_bytes));
        a.park = (uint64_t *)c->ws_park.p; a.park_bytes = park_bytes;
#ifdef EP_STAMPS
        static const char *names16[EP_NPH] = {"stage+rotate+decomp_first", "decomp_next", "fwd head", "pre-level barrier", "fwd tail (transpose+dft16)",
                                              "digit stores+late loads", "exchange barrier", "MAC", "products exchange", "inverse fft", "convert+add", "loop head"};
        StampReport rep(c, (size_t)grid16 * 4, names16);
        a.stamps = rep.d;
#endif
        if (c->k1 == 5 && k2_home_allowed(c)) hipLaunchKernelGGL((blind_rotate16_kernel<5, 5, 8, 3, 2, true>), dim3(grid16), dim3(EP_THREADS), 0, c->stream, a);
        else if (c->k1 == 5) hipLaunchKernelGGL((blind_rotate16_kernel<5, 5, 8, 3, 2, false>), dim3(grid16), dim3(EP_THREADS), 0, c->stream, a);
        else hipLaunchKernelGGL((blind_rotate16_kernel<2, 5, 8, 8>), dim3(grid16), dim3(EP_THREADS), 0, c->stream, a);
    }
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

int launch_vertical_packing(fheaes_ctx *c, const double2 *ggswf, uint64_t n_inputs, uint32_t bits, const uint64_t *luts,
                            uint32_t n_luts, int per_input, uint64_t *out)
{
    if (n_inputs == 0) return FHEAES_OK;
    StageScope sc(c, FHEAES_STAGE_VERTICAL_PACKING, n_inputs * n_luts * bits);
    const uint32_t inst_per_input = n_luts * bits;
    const uint64_t W = lut_row_words(bits);
    // ---- inputs wider than 9 bits: CMUX tree over bits 9..bits-1 (kern_extprod.h, cmux_level_kernel), root -> ws_tree ----
    const uint64_t *glwe_root = nullptr;
    if (bits > 9) {
        const uint32_t tree_bits = bits - 9;
        const uint64_t instances = n_inputs * inst_per_input, gsz = (uint64_t)c->k1 * FHE_N;
        // level t writes (2^(tree_bits-1-t)) nodes per instance; ping-pong between the two halves of ws_tree
        const uint64_t half_words = instances * (1ull << (tree_bits - 1)) * gsz;
        TRY(ensure(c, c->ws_tree, 2 * half_words * 8));
        uint64_t *buf[2] = {(uint64_t *)c->ws_tree.p, (uint64_t *)c->ws_tree.p + half_words};
        for (uint32_t t = 0; t < tree_bits; ++t) {
            CmuxArgs a{};
            a.ggsw = ggswf; a.tw = c->tw_d;
            a.luts = t == 0 ? luts : nullptr; a.in = t == 0 ? nullptr : buf[(t - 1) & 1]; a.out = buf[t & 1];
            a.bits = bits; a.bit = 9 + t; a.nodes_out = 1u << (tree_bits - 1 - t);
            a.inst_per_input = inst_per_input; a.lut_per_input = per_input ? 1 : 0; a.lut_words = W;
            const uint64_t jobs = (uint64_t)inst_per_input * a.nodes_out;
            if (c->k1 == 5) {
                constexpr int R = 3;
                a.wg_per_input = (uint32_t)((jobs + R - 1) / R);
                hipLaunchKernelGGL((cmux_level_kernel<5, 15, R>), dim3((unsigned)(n_inputs * a.wg_per_input)), dim3(EP_THREADS), 0, c->stream, a);
            } else {
                constexpr int R = 8;
                a.wg_per_input = (uint32_t)((jobs + R - 1) / R);
                hipLaunchKernelGGL((cmux_level_kernel<2, 15, R>), dim3((unsigned)(n_inputs * a.wg_per_input)), dim3(EP_THREADS), 0, c->stream, a);
            }
            HIP_TRY(c, hipGetLastError());
        }
        glwe_root = buf[(tree_bits - 1) & 1];
    }
    ExtProdArgs a{};
    a.ggsw = ggswf; a.tw = c->tw_d;
    a.out = out; a.count = n_inputs * n_luts * bits; a.iters = bits < 9 ? bits : 9; a.ggsw_per_input = bits;
    a.luts = luts; a.lut_words = W; a.glwe_in = glwe_root;
    a.n_luts = n_luts; a.lut_per_input = per_input ? 1 : 0; a.inst_per_input = inst_per_input;
    if (c->k1 == 5) {
        constexpr int R = 3;
        a.wg_per_input = (a.inst_per_input + R - 1) / R;
        hipLaunchKernelGGL((extprod_rotate_kernel<5, 1, 15, R, true>), dim3((unsigned)(n_inputs * a.wg_per_input)), dim3(EP_THREADS), 0, c->stream, a);
    } else {
        constexpr int R = 8;
        a.wg_per_input = (a.inst_per_input + R - 1) / R;
        hipLaunchKernelGGL((extprod_rotate_kernel<2, 1, 15, R, true>), dim3((unsigned)(n_inputs * a.wg_per_input)), dim3(EP_THREADS), 0, c->stream, a);
    }
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

// static schedule assertion (NOT runtime noise tracking: words carry no metadata): every linear layer of the engine's own AES schedule
// declares here how many WoPBS outputs it sums per word; a table that would sum more than tfhe-rs' noise-asserts allow is refused
int noise_guard(fheaes_ctx *c, uint32_t level, const char *what)
{
    if (level > c->noise_level_seen) c->noise_level_seen = level;
    if (level > FHEAES_MAX_NOISE_LEVEL)
        return c->fail(FHEAES_ERR_INVALID, "%s would sum %u nominal-noise ciphertexts; the parameter set allows %u (MaxNoiseLevel, client.rs:92)", what, level,
                       (unsigned)FHEAES_MAX_NOISE_LEVEL);
    return FHEAES_OK;
}

int launch_gather(fheaes_ctx *c, const uint64_t *src, uint32_t n_luts, const uint64_t *rk, uint64_t *out, uint64_t n_blocks, const GatherTable &t)
{
    if (n_blocks == 0) return FHEAES_OK;
    TRY(noise_guard(c, (uint32_t)t.terms + (rk ? 1u : 0u), "the linear layer (MixColumns / ShiftRows + AddRoundKey)"));
    StageScope sc(c, FHEAES_STAGE_LINEAR, n_blocks);
    const uint32_t bw = 8 * c->big1;
    dim3 grid((bw + 1023) / 1024, 16, (unsigned)n_blocks);
    hipLaunchKernelGGL(gather_add_kernel, grid, dim3(256), 0, c->stream, src, n_luts, rk, out, n_blocks, bw, t);
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

int launch_add_bcast(fheaes_ctx *c, uint64_t *dst, const uint64_t *src, uint64_t words_per_block, uint64_t n_blocks)
{
    if (n_blocks == 0) return FHEAES_OK;
    TRY(noise_guard(c, 2, "the initial AddRoundKey"));
    StageScope sc(c, FHEAES_STAGE_LINEAR, n_blocks);
    uint64_t total = words_per_block * n_blocks;
    unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid), dim3(256), 0, c->stream, dst, src, words_per_block, n_blocks);
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

int launch_add2(fheaes_ctx *c, uint64_t *dst, const uint64_t *a, const uint64_t *b, uint64_t words)
{
    TRY(noise_guard(c, 2, "a key-expansion word sum"));
    StageScope sc(c, FHEAES_STAGE_LINEAR, 1);
    unsigned grid = (unsigned)std::min<uint64_t>((words + 255) / 256, 16384);
    hipLaunchKernelGGL(add2_kernel, dim3(grid), dim3(256), 0, c->stream, dst, a, b, words);
    HIP_TRY(c, hipGetLastError());
    return FHEAES_OK;
}

int check_keys(fheaes_ctx *c)
{
    if (!c) return FHEAES_ERR_INVALID;
    if (!c->have_keys) return c->fail(FHEAES_ERR_NOKEYS, "evaluation keys have not been uploaded");
    return FHEAES_OK;
}

// ---- many_wopbs_without_padding on device buffers ----------------------------------------------
int wopbs_dev(fheaes_ctx *c, const uint64_t *lwe_in, uint64_t n_inputs, uint32_t bits, const uint64_t *luts, uint32_t n_luts,
              int per_input, uint64_t *lwe_out)
{
    if (bits < 1 || bits > MAX_WOPBS_BITS) return c->fail(FHEAES_ERR_INVALID, "bits_per_input must be in 1..%u (got %u)", MAX_WOPBS_BITS, bits);
    if (n_luts < 1) return c->fail(FHEAES_ERR_INVALID, "n_luts must be >= 1");
    if (n_inputs == 0) return FHEAES_OK;
    uint64_t chunk_inputs = std::max<uint64_t>(1, MAX_CHUNK_BITS / bits);
    if (bits > 9) {
        // the CMUX tree keeps 2^(bits-9) GLWEs per (input, LUT, output bit) in flight: bound that workspace to ~2 GiB per chunk
        const uint64_t per_input = (uint64_t)n_luts * bits * (1ull << (bits - 9)) * c->k1 * FHE_N * 8;
        chunk_inputs = std::max<uint64_t>(1, std::min<uint64_t>(chunk_inputs, (2ull << 30) / per_input));
    }
    const uint64_t W = lut_row_words(bits);
    const uint64_t ggsw_words = (uint64_t)c->k1 * c->k1 * FHE_N;    // cbs_level == 1
    const uint64_t cap_bits = std::min<uint64_t>(n_inputs, chunk_inputs) * bits;
    TRY(ensure(c, c->ws_small, cap_bits * (c->n + 1) * 8));
    TRY(ensure(c, c->ws_pbs, cap_bits * c->big1 * 8));
    TRY(ensure(c, c->ws_ggsw, cap_bits * ggsw_words * 8));
    TRY(ensure(c, c->ws_ggswf, cap_bits * ggsw_words * 8));
    for (uint64_t i0 = 0; i0 < n_inputs; i0 += chunk_inputs) {
        const uint64_t ni = std::min<uint64_t>(chunk_inputs, n_inputs - i0);
        const uint64_t m = ni * bits;
        const uint64_t *in = lwe_in + i0 * bits * c->big1;
        TRY(launch_keyswitch(c, in, m, (uint64_t *)c->ws_small.p));
        TRY(launch_cbs_pbs(c, (const uint64_t *)c->ws_small.p, m, 1, (uint64_t *)c->ws_pbs.p));
        TRY(launch_pfpks(c, (const uint64_t *)c->ws_pbs.p, m, (uint64_t *)c->ws_ggsw.p, ggsw_words));
        TRY(launch_forward_fourier(c, (const uint64_t *)c->ws_ggsw.p, m * c->k1 * c->k1, (double2 *)c->ws_ggswf.p, FHEAES_STAGE_GGSW_FFT));
        const uint64_t *l = per_input ? luts + i0 * n_luts * bits * W : luts;
        TRY(launch_vertical_packing(c, (const double2 *)c->ws_ggswf.p, ni, bits, l, n_luts, per_input, lwe_out + i0 * n_luts * bits * c->big1));
    }
    return FHEAES_OK;
}

// Host-memspace calls: arguments are staged through context-owned device buffers (grown on demand, reused by later
// calls -- the per-byte `sbox` call pattern of the reference's Rust side must not pay a hipMalloc/hipFree each time).
struct Staged {
    fheaes_ctx *c;
    int used = 0;
    explicit Staged(fheaes_ctx *ctx) : c(ctx) {}
    ~Staged() { (void)hipStreamSynchronize(c->stream); }      // host pointers are borrowed for the duration of the call only
    int alloc(void **out, size_t bytes)
    {
        if (used >= 4) return c->fail(FHEAES_ERR_INVALID, "internal: too many staged arguments");
        TRY(ensure(c, c->stage[used], bytes ? bytes : 8));
        *out = c->stage[used++].p;
        return FHEAES_OK;
    }
    int in(const void *host, size_t bytes, void **dev)
    {
        TRY(alloc(dev, bytes));
        HIP_TRY(c, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
        return FHEAES_OK;
    }
    int out(void *host, const void *dev, size_t bytes)
    {
        HIP_TRY(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return FHEAES_OK;
    }
};

int supported(const fheaes_params *p, std::string &why)
{
    char buf[256];
    if (p->polynomial_size != FHE_N) { why = "polynomial_size must be 512"; return 0; }
    if (p->glwe_dimension != 4 && p->glwe_dimension != 1) { why = "glwe_dimension must be 4 (PARAM_OPT) or 1 (toy)"; return 0; }
    if (p->pbs_base_log != 8 || p->pbs_level != 5 || p->ks_base_log != 2 || p->ks_level != 6 || p->pfks_base_log != 12 ||
        p->pfks_level != 3 || p->cbs_base_log != 15 || p->cbs_level != 1) {
        snprintf(buf, sizeof buf, "unsupported gadget: kernels are instantiated for pbs(8,5) ks(2,6) pfks(12,3) cbs(15,1)");
        why = buf;
        return 0;
    }
    if (p->lwe_dimension < 1 || p->lwe_dimension > 4096) { why = "lwe_dimension out of range"; return 0; }
    return 1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int fheaes_k2_launch_plan(uint64_t m, uint32_t cu_count, uint32_t k, int *form, uint64_t *units_main, uint32_t *r_main,
                          uint64_t *units_tail, uint32_t *r_tail)
{
    if (!form || !units_main || !r_main || !units_tail || !r_tail || cu_count == 0 || m == 0) return FHEAES_ERR_INVALID;
    const K2Plan pl = k2_plan(m, cu_count, k + 1);
    *form = pl.form; *units_main = pl.units_main; *r_main = pl.r_main; *units_tail = pl.units_tail; *r_tail = pl.r_tail;
    return FHEAES_OK;
}
int fheaes_k2_context_plan(fheaes_ctx *ctx, uint64_t m, int *form, uint64_t *units_main, uint32_t *r_main, uint64_t *units_tail,
                           uint32_t *r_tail, char *kernel, size_t kernel_cap)
{
    if (!ctx) return FHEAES_ERR_INVALID;
    CtxLock lock__(ctx);
    if (!form || !units_main || !r_main || !units_tail || !r_tail || m == 0) return ctx->fail(FHEAES_ERR_INVALID, "k2_context_plan: null output or empty batch");
    const char *name = "";
    const K2Plan pl = k2_context_plan(ctx, m, &name);
    *form = pl.form; *units_main = pl.units_main; *r_main = pl.r_main; *units_tail = pl.units_tail; *r_tail = pl.r_tail;
    if (kernel && kernel_cap) { std::strncpy(kernel, name, kernel_cap - 1); kernel[kernel_cap - 1] = 0; }
    return FHEAES_OK;
}
int fheaes_k2_set_parking(fheaes_ctx *ctx, int claimed)
{
    if (!ctx) return FHEAES_ERR_INVALID;
    CtxLock lock__(ctx);
    if (claimed != 0 && claimed != 1) return ctx->fail(FHEAES_ERR_INVALID, "k2_set_parking: 1 = claimed slots, 0 = one private slot per workgroup (got %d)", claimed);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->k2_park_claim = claimed;
    return FHEAES_OK;
}
const char *fheaes_version(void) { return FHEAES_VERSION_STR; }

const char *fheaes_last_error(const fheaes_ctx *ctx)
{
    if (!ctx) return g_create_error.c_str();
    if (tl_error_ctx_id != ctx->id) {            // this thread has not failed on ctx: hand out a private copy of the context's last message
        CtxLock lock__(ctx);
        tl_error = ctx->err;
        tl_error_ctx_id = ctx->id;
    }
    return tl_error.c_str();
}

int fheaes_get_twiddles(double *psi_out)
{
    if (!psi_out) return FHEAES_ERR_INVALID;
    const HostTwiddles &t = twiddles();
    for (int j = 0; j < FHE_N; ++j) { psi_out[2 * j] = t.psi_re[j]; psi_out[2 * j + 1] = t.psi_im[j]; }
    return FHEAES_OK;
}

int fheaes_gen_lut(uint32_t nb_block, const uint64_t *f_table, uint64_t *lut_out)
{
    if (nb_block < 1 || nb_block > MAX_WOPBS_BITS || !f_table || !lut_out) return FHEAES_ERR_INVALID;
    gen_lut_host(nb_block, f_table, lut_out);
    return FHEAES_OK;
}

int fheaes_create(const fheaes_params *params, int device, fheaes_ctx **out)
{
    if (!params || !out) { g_create_error = "null argument"; return FHEAES_ERR_INVALID; }
    std::string why;
    if (!supported(params, why)) { g_create_error = why; return FHEAES_ERR_INVALID; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_error = std::string("no HIP device: ") + hipGetErrorString(e); return FHEAES_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return FHEAES_ERR_INVALID; }
    fheaes_ctx *c = new fheaes_ctx();
    c->p = *params; c->device = device;
    c->n = params->lwe_dimension; c->k = params->glwe_dimension; c->k1 = c->k + 1; c->big = c->k * FHE_N; c->big1 = c->big + 1;
    auto bail = [&](const char *what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        fheaes_destroy(c);
        return FHEAES_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cu_count = (uint32_t)cus;
    }
    c->stream = c->own_stream;
    // twiddle tables
    const HostTwiddles &t = twiddles();
    std::vector<double2> tw(FHE_TW_ENTRIES);
    for (auto &w : tw) { w.x = 0.0; w.y = 0.0; }
    for (int k1 = 0; k1 < 16; ++k1) for (int b = 0; b < 16; ++b) { double re, im; t.pow(b * (4 * k1 + 1), re, im); tw[FHE_TW_STRIDE * k1 + b].x = re; tw[FHE_TW_STRIDE * k1 + b].y = im; }
    // the kernels' compile-time constants (fft_consts.h, generated from this table) must BE this table
    for (int m = 0; m < 32; ++m)
        if (FHE_PSI16_RE[m] != t.psi_re[16 * m] || FHE_PSI16_IM[m] != t.psi_im[16 * m]) {
            g_create_error = "fft_consts.h does not match the twiddle table (regenerate it: tools/gen_fft_consts.py)";
            fheaes_destroy(c);
            return FHEAES_ERR_INVALID;
        }
    if ((e = hipMalloc((void **)&c->tw_d, FHE_TW_ENTRIES * sizeof(double2))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(c->tw_d, tw.data(), FHE_TW_ENTRIES * sizeof(double2), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    // AES LUT sets, built once (the reference rebuilds them on every call: sbox.rs:54-60, :85-94)
    for (int s = 0; s < LUTSET_COUNT; ++s) {
        std::vector<uint64_t> h;
        c->lutset_n[s] = build_lutset_host(s, h);
        if ((e = hipMalloc((void **)&c->lutset_d[s], h.size() * 8)) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMemcpy(c->lutset_d[s], h.data(), h.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    }
    *out = c;
    return FHEAES_OK;
}

void fheaes_destroy(fheaes_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &pe : c->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto ev : c->free_events) (void)hipEventDestroy(ev);
    void *ptrs[] = {c->ksk_frag, c->pfpksk_frag, c->bskf, c->tw_d, c->ws_digits.p,
                    c->ws_small.p, c->ws_pbs.p, c->ws_ggsw.p, c->ws_ggswf.p, c->ws_vp.p, c->ws_tmp_a.p, c->ws_tmp_b.p, c->ws_luts.p, c->ws_misc.p,
                    c->ws_park.p, c->ws_park_owner.p, c->ws_tree.p};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &b : c->stage) if (b.p) (void)hipFree(b.p);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_ev) (void)hipEventDestroy(c->pin_ev);
    for (int s = 0; s < LUTSET_COUNT; ++s) if (c->lutset_d[s]) (void)hipFree(c->lutset_d[s]);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

size_t fheaes_key_words(const fheaes_ctx *c, int which)
{
    if (!c) return 0;
    switch (which) {
    case FHEAES_KEY_KSK: return (size_t)c->big * c->p.ks_level * (c->n + 1);
    case FHEAES_KEY_BSK: return (size_t)c->n * c->p.pbs_level * c->k1 * c->k1 * FHE_N;
    case FHEAES_KEY_PFPKSK: return (size_t)c->k1 * c->big1 * c->p.pfks_level * c->k1 * FHE_N;
    default: return 0;
    }
}

int fheaes_set_stream(fheaes_ctx *c, void *hip_stream)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return FHEAES_OK;
}

int fheaes_synchronize(fheaes_ctx *c)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FHEAES_OK;
}

int fheaes_reserve(fheaes_ctx *c, uint64_t max_bits)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t bits = std::min<uint64_t>(max_bits, MAX_CHUNK_BITS);
    const uint64_t ggsw_words = (uint64_t)c->k1 * c->k1 * FHE_N;
    TRY(ensure(c, c->ws_small, bits * (c->n + 1) * 8));
    TRY(ensure(c, c->ws_pbs, bits * c->big1 * 8));
    TRY(ensure(c, c->ws_ggsw, bits * ggsw_words * 8));
    TRY(ensure(c, c->ws_ggswf, bits * ggsw_words * 8));
    {
        // the blind rotation's parking slab for the largest launch this reservation covers (64 KB per workgroup)
        const K2Plan pl = k2_plan(bits, c->cu_count, c->k1, k2_pair_allowed(c));
        if (pl.form == 1) TRY(ensure(c, c->ws_park, (size_t)(pl.units_main + pl.units_tail) * BR16_PARK_WORDS_PER_WG * 8));
        if (pl.form == 2) {
            TRY(ensure(c, c->ws_park, k2_pair_park_bytes(c, pl.units_main + pl.units_tail)));
            TRY(ensure(c, c->ws_park_owner, BRP_PARK_SLOTS * sizeof(uint32_t)));
        }
    }
    return FHEAES_OK;
}

// Both uploads end in the same three conversions of the standard-domain words staged in HBM: KSK / PFPKSK -> balanced int8
// byte planes in MFMA fragment order, BSK -> Fourier.  `seeded`: the caller passed bodies only and the masks are
// regenerated on the GPU from the public 256-bit mask key (ChaCha20 stream of csrc/client.c) -- 0.19 GB over PCIe / xGMI
// instead of 1.04 GB.
static int upload_keys_impl(fheaes_ctx *c, const uint64_t *ksk, const uint64_t *bsk, const uint64_t *pfpksk, int memspace, bool seeded, const MaskKey &mask_key)
{
    if (!c || !ksk || !bsk || !pfpksk) return c ? c->fail(FHEAES_ERR_INVALID, "null key pointer") : FHEAES_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t kw = fheaes_key_words(c, FHEAES_KEY_KSK), bw = fheaes_key_words(c, FHEAES_KEY_BSK), pw = fheaes_key_words(c, FHEAES_KEY_PFPKSK);
    c->have_keys = false;
    // K1 / K3 keys: balanced int8 byte planes in MFMA fragment order (same byte count as the uint64 keys)
    const uint32_t rows1 = c->big * c->p.ks_level, ncol1 = c->n + 1;
    const uint32_t rows3 = c->big1 * c->p.pfks_level, ncol3 = c->k1 * FHE_N;
    c->ks_ksteps = (rows1 + KS_KSTEP - 1) / KS_KSTEP; c->ks_coltiles = (ncol1 + 15) / 16;
    c->pf_ksteps = (rows3 + KS_KSTEP - 1) / KS_KSTEP; c->pf_coltiles = (ncol3 + 15) / 16;
    const size_t frag1 = (size_t)c->ks_ksteps * c->ks_coltiles * 8 * 1024;
    const size_t frag3 = (size_t)c->k1 * c->pf_ksteps * c->pf_coltiles * 8 * 1024;
    if (!c->ksk_frag) HIP_TRY(c, hipMalloc((void **)&c->ksk_frag, frag1));
    if (!c->pfpksk_frag) HIP_TRY(c, hipMalloc((void **)&c->pfpksk_frag, frag3));
    if (!c->bskf) HIP_TRY(c, hipMalloc((void **)&c->bskf, bw * 8));
    c->ksk_frag_bytes = frag1; c->pfpksk_frag_bytes = frag3; c->bskf_bytes = bw * 8;
    // stage the standard-domain words in HBM (largest key first), transform, free
    void *tmp = nullptr, *tmp_body = nullptr;
    const size_t tmp_words = std::max(std::max(kw, bw), pw);
    // seeded: key ciphertext counts and body sizes
    const uint64_t cts[3] = {(uint64_t)rows1, (uint64_t)c->n * c->p.pbs_level * c->k1, (uint64_t)c->k1 * rows3};
    const uint32_t mask_w[3] = {c->n, c->big, c->big}, body_w[3] = {1, FHE_N, FHE_N};
    const uint64_t tags[3] = {3, 4, 5};                                  // MASK_TAG_* of csrc/client.c
    size_t body_max = 0;
    for (int i = 0; i < 3; ++i) body_max = std::max(body_max, (size_t)cts[i] * body_w[i]);
    if (memspace != FHEAES_DEVICE || seeded) HIP_TRY(c, hipMalloc(&tmp, tmp_words * 8));
    if (seeded && memspace != FHEAES_DEVICE) {
        hipError_t me = hipMalloc(&tmp_body, body_max * 8);
        if (me != hipSuccess) { (void)hipFree(tmp); return c->fail(FHEAES_ERR_NOMEM, "hipMalloc(%zu): %s", body_max * 8, hipGetErrorString(me)); }
    }
    hipError_t copy_err = hipSuccess;
    // which: 0 KSK, 1 BSK, 2 PFPKSK; returns the device pointer of the full standard-domain key
    auto staged = [&](int which, const uint64_t *src, size_t words) -> const uint64_t * {
        if (!seeded) {
            if (memspace == FHEAES_DEVICE) return src;
            hipError_t ce = hipMemcpyAsync(tmp, src, words * 8, hipMemcpyHostToDevice, c->stream);
            if (ce != hipSuccess && copy_err == hipSuccess) copy_err = ce;
            return (const uint64_t *)tmp;
        }
        const uint64_t *bodies = src;
        if (memspace != FHEAES_DEVICE) {
            hipError_t ce = hipMemcpyAsync(tmp_body, src, (size_t)cts[which] * body_w[which] * 8, hipMemcpyHostToDevice, c->stream);
            if (ce != hipSuccess && copy_err == hipSuccess) copy_err = ce;
            bodies = (const uint64_t *)tmp_body;
        }
        hipLaunchKernelGGL(expand_masks_kernel, dim3(8192), dim3(256), 0, c->stream, (uint64_t *)tmp, bodies, cts[which], mask_w[which], body_w[which],
                           mask_key, (uint32_t)tags[which]);
        return (const uint64_t *)tmp;
    };
    int rc = FHEAES_OK;
    {
        const uint64_t *d = staged(0, ksk, kw);
        const uint64_t threads = (uint64_t)c->ks_ksteps * c->ks_coltiles * 64;
        ks_launch_keybytes(dim3((unsigned)((threads + 255) / 256), 1), c->stream, d, (uint64_t)0, rows1, ncol1, c->ks_ksteps, c->ks_coltiles, c->ksk_frag);
        if (tmp) (void)hipStreamSynchronize(c->stream);
    }
    {
        const uint64_t *d = staged(2, pfpksk, pw);
        const uint64_t threads = (uint64_t)c->pf_ksteps * c->pf_coltiles * 64;
        ks_launch_keybytes(dim3((unsigned)((threads + 255) / 256), c->k1), c->stream, d, (uint64_t)rows3 * ncol3, rows3, ncol3, c->pf_ksteps, c->pf_coltiles,
                           c->pfpksk_frag);
        if (tmp) (void)hipStreamSynchronize(c->stream);
    }
    {
        // BSK: standard domain -> Fourier (the reference holds it in Fourier form already, many_wopbs.rs:34-35)
        const uint64_t *d = staged(1, bsk, bw);
        rc = launch_forward_fourier(c, d, bw / FHE_N, c->bskf, FHEAES_STAGE_GGSW_FFT);
        c->stage_launches[FHEAES_STAGE_GGSW_FFT] = 0; c->stage_units[FHEAES_STAGE_GGSW_FFT] = 0;
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (tmp) (void)hipFree(tmp);
    if (tmp_body) (void)hipFree(tmp_body);
    if (rc != FHEAES_OK) return rc;
    if (copy_err != hipSuccess) return c->fail(FHEAES_ERR_DEVICE, "key upload (host -> device copy): %s", hipGetErrorString(copy_err));
    if (e != hipSuccess) return c->fail(FHEAES_ERR_DEVICE, "key upload: %s", hipGetErrorString(e));
    HIP_TRY(c, hipGetLastError());
    if (c->prof) prof_flush(c);
    c->stage_ms[FHEAES_STAGE_GGSW_FFT] = 0;
    c->have_keys = true;
    return FHEAES_OK;
}

int fheaes_upload_keys(fheaes_ctx *c, const uint64_t *ksk, const uint64_t *bsk, const uint64_t *pfpksk, int memspace)
{
    CtxLock lock__(c);
    return upload_keys_impl(c, ksk, bsk, pfpksk, memspace, false, MaskKey{});
}

int fheaes_upload_keys_seeded(fheaes_ctx *c, const uint32_t *mask_key, const uint64_t *ksk_body, const uint64_t *bsk_body, const uint64_t *pfpksk_body,
                              int memspace)
{
    CtxLock lock__(c);
    if (!c || !mask_key) return c ? c->fail(FHEAES_ERR_INVALID, "null mask key") : FHEAES_ERR_INVALID;
    MaskKey k;
    memcpy(k.k, mask_key, sizeof k.k);                       // the 32-byte key itself is always a HOST array
    return upload_keys_impl(c, ksk_body, bsk_body, pfpksk_body, memspace, true, k);
}

// One upload over PCIe, then device-to-device copies of the CONVERTED key images (int8 fragment planes of KSK / PFPKSK, Fourier
// BSK: 1.04 GB) -- over xGMI when the contexts sit on different GPUs (hipMemcpyPeerAsync), inside HBM when they share one.
int fheaes_clone_keys(fheaes_ctx *dst, fheaes_ctx *src)
{
    if (!dst || !src) return FHEAES_ERR_INVALID;
    if (dst == src) return dst->fail(FHEAES_ERR_INVALID, "fheaes_clone_keys: source and destination are the same context");
    // both locks, in address order (two threads cloning in opposite directions must not deadlock)
    CtxLock l1(dst < src ? dst : src), l2(dst < src ? src : dst);
    if (!src->have_keys) return dst->fail(FHEAES_ERR_NOKEYS, "fheaes_clone_keys: the source context has no keys");
    if (memcmp(&dst->p, &src->p, sizeof(fheaes_params)) != 0) return dst->fail(FHEAES_ERR_INVALID, "fheaes_clone_keys: parameter sets differ");
    HIP_TRY(dst, hipSetDevice(src->device));
    HIP_TRY(dst, hipStreamSynchronize(src->stream));            // the source's conversions are complete
    HIP_TRY(dst, hipSetDevice(dst->device));
    dst->have_keys = false;
    dst->clone_path = FHEAES_CLONE_NONE; dst->clone_bytes = 0; dst->clone_seconds = 0.0;
    // Between two GPUs the copy is a direct xGMI transfer only if peer access is possible AND enabled; otherwise the runtime stages
    // it through host memory.  Ask, enable once per device pair ("already enabled" is fine), and remember which of the two it was.
    int path = FHEAES_CLONE_SAME_DEVICE;
    if (dst->device != src->device) {
        path = FHEAES_CLONE_STAGED;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);           // the current device is dst's
            if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) path = FHEAES_CLONE_PEER;
        }
        (void)hipGetLastError();                                                       // "already enabled" must not poison later HIP_TRY(hipGetLastError())
    }
    const auto t_clone = std::chrono::steady_clock::now();
    uint64_t moved = 0;
    struct { void **d; const void *s; size_t bytes; size_t *have; } img[3] = {
        {(void **)&dst->ksk_frag, src->ksk_frag, src->ksk_frag_bytes, &dst->ksk_frag_bytes},
        {(void **)&dst->pfpksk_frag, src->pfpksk_frag, src->pfpksk_frag_bytes, &dst->pfpksk_frag_bytes},
        {(void **)&dst->bskf, src->bskf, src->bskf_bytes, &dst->bskf_bytes}};
    for (auto &g : img) {
        if (*g.d && *g.have != g.bytes) { HIP_TRY(dst, hipStreamSynchronize(dst->stream)); HIP_TRY(dst, hipFree(*g.d)); *g.d = nullptr; }
        if (!*g.d) {
            hipError_t e = hipMalloc(g.d, g.bytes);
            if (e != hipSuccess) { *g.d = nullptr; return dst->fail(FHEAES_ERR_NOMEM, "hipMalloc(%zu bytes): %s", g.bytes, hipGetErrorString(e)); }
        }
        *g.have = g.bytes;
        if (dst->device == src->device) HIP_TRY(dst, hipMemcpyAsync(*g.d, g.s, g.bytes, hipMemcpyDeviceToDevice, dst->stream));
        else HIP_TRY(dst, hipMemcpyPeerAsync(*g.d, dst->device, g.s, src->device, g.bytes, dst->stream));
        moved += g.bytes;
    }
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    dst->clone_path = path; dst->clone_bytes = moved;
    dst->clone_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_clone).count();
    dst->ks_ksteps = src->ks_ksteps; dst->ks_coltiles = src->ks_coltiles; dst->pf_ksteps = src->pf_ksteps; dst->pf_coltiles = src->pf_coltiles;
    dst->have_keys = true;
    return FHEAES_OK;
}

int fheaes_noise_level_seen(fheaes_ctx *c, uint32_t *max_seen, uint32_t *limit)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    if (max_seen) *max_seen = c->noise_level_seen;
    if (limit) *limit = FHEAES_MAX_NOISE_LEVEL;
    return FHEAES_OK;
}

int fheaes_clone_info(fheaes_ctx *c, int *path, uint64_t *bytes, double *seconds)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    if (path) *path = c->clone_path;
    if (bytes) *bytes = c->clone_bytes;
    if (seconds) *seconds = c->clone_seconds;
    return FHEAES_OK;
}

size_t fheaes_key_body_words(const fheaes_ctx *c, int which)
{
    if (!c) return 0;
    switch (which) {
    case FHEAES_KEY_KSK: return (size_t)c->big * c->p.ks_level;
    case FHEAES_KEY_BSK: return (size_t)c->n * c->p.pbs_level * c->k1 * FHE_N;
    case FHEAES_KEY_PFPKSK: return (size_t)c->k1 * c->big1 * c->p.pfks_level * FHE_N;
    default: return 0;
    }
}

int fheaes_read_bsk_fourier(fheaes_ctx *c, uint32_t i, double *out)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (i >= c->n || !out) return c->fail(FHEAES_ERR_INVALID, "bad GGSW index");
    const size_t words = (size_t)c->p.pbs_level * c->k1 * c->k1 * FHE_N;
    HIP_TRY(c, hipMemcpyAsync(out, (const double *)c->bskf + (size_t)i * words, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FHEAES_OK;
}

// ---- stage-by-stage ---------------------------------------------------------------------------
int fheaes_keyswitch_batch(fheaes_ctx *c, const uint64_t *lwe_in, uint64_t m, uint64_t *lwe_out, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!lwe_in || !lwe_out) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return launch_keyswitch(c, lwe_in, m, lwe_out);
    Staged s(c);
    void *din, *dout;
    TRY(s.in(lwe_in, m * c->big1 * 8, &din));
    TRY(s.alloc(&dout, m * (c->n + 1) * 8));
    TRY(launch_keyswitch(c, (const uint64_t *)din, m, (uint64_t *)dout));
    return s.out(lwe_out, dout, m * (c->n + 1) * 8);
}

int fheaes_cbs_pbs_batch(fheaes_ctx *c, const uint64_t *lwe_small, uint64_t m, uint32_t level, uint64_t *lwe_out, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!lwe_small || !lwe_out) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    if (level < 1 || level > c->p.cbs_level) return c->fail(FHEAES_ERR_INVALID, "cbs level %u out of range", level);
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return launch_cbs_pbs(c, lwe_small, m, level, lwe_out);
    Staged s(c);
    void *din, *dout;
    TRY(s.in(lwe_small, m * (c->n + 1) * 8, &din));
    TRY(s.alloc(&dout, m * c->big1 * 8));
    TRY(launch_cbs_pbs(c, (const uint64_t *)din, m, level, (uint64_t *)dout));
    return s.out(lwe_out, dout, m * c->big1 * 8);
}

int fheaes_pfpks_batch(fheaes_ctx *c, const uint64_t *lwe_in, uint64_t m, uint64_t *ggsw_rows_out, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!lwe_in || !ggsw_rows_out) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t words = (uint64_t)c->k1 * c->k1 * FHE_N;
    if (memspace == FHEAES_DEVICE) return launch_pfpks(c, lwe_in, m, ggsw_rows_out, words);
    Staged s(c);
    void *din, *dout;
    TRY(s.in(lwe_in, m * c->big1 * 8, &din));
    TRY(s.alloc(&dout, m * words * 8));
    TRY(launch_pfpks(c, (const uint64_t *)din, m, (uint64_t *)dout, words));
    return s.out(ggsw_rows_out, dout, m * words * 8);
}

int fheaes_forward_fourier_batch(fheaes_ctx *c, const uint64_t *polys_in, uint64_t polys, double *fourier_out, int memspace)
{
    CtxLock lock__(c);
    if (!c || !polys_in || !fourier_out) return c ? c->fail(FHEAES_ERR_INVALID, "null pointer") : FHEAES_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return launch_forward_fourier(c, polys_in, polys, (double2 *)fourier_out, FHEAES_STAGE_GGSW_FFT);
    Staged s(c);
    void *din, *dout;
    TRY(s.in(polys_in, polys * FHE_N * 8, &din));
    TRY(s.alloc(&dout, polys * FHE_N * 8));
    TRY(launch_forward_fourier(c, (const uint64_t *)din, polys, (double2 *)dout, FHEAES_STAGE_GGSW_FFT));
    return s.out(fourier_out, dout, polys * FHE_N * 8);
}

int fheaes_vertical_packing_batch(fheaes_ctx *c, const double *ggsw_fourier, uint64_t n_inputs, uint32_t bits, const uint64_t *luts,
                                  uint32_t n_luts, int lut_per_input, uint64_t *lwe_out, int memspace)
{
    CtxLock lock__(c);
    if (!c || !ggsw_fourier || !luts || !lwe_out) return c ? c->fail(FHEAES_ERR_INVALID, "null pointer") : FHEAES_ERR_INVALID;
    if (bits < 1 || bits > MAX_WOPBS_BITS || n_luts < 1) return c->fail(FHEAES_ERR_INVALID, "bits must be 1..%u and n_luts >= 1", MAX_WOPBS_BITS);
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return launch_vertical_packing(c, (const double2 *)ggsw_fourier, n_inputs, bits, luts, n_luts, lut_per_input, lwe_out);
    Staged s(c);
    void *dg, *dl, *dout;
    const uint64_t gw = (uint64_t)c->k1 * c->k1 * FHE_N;
    const uint64_t sets = lut_per_input ? n_inputs : 1;
    TRY(s.in(ggsw_fourier, n_inputs * bits * gw * 8, &dg));
    TRY(s.in(luts, sets * n_luts * bits * lut_row_words(bits) * 8, &dl));
    TRY(s.alloc(&dout, n_inputs * n_luts * bits * c->big1 * 8));
    TRY(launch_vertical_packing(c, (const double2 *)dg, n_inputs, bits, (const uint64_t *)dl, n_luts, lut_per_input, (uint64_t *)dout));
    return s.out(lwe_out, dout, n_inputs * n_luts * bits * c->big1 * 8);
}

// ---- plugin API -------------------------------------------------------------------------------
int fheaes_wopbs_batch(fheaes_ctx *c, const uint64_t *lwe_in, uint64_t n_inputs, uint32_t bits, const uint64_t *luts, uint32_t n_luts,
                       int lut_per_input, uint64_t *lwe_out, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!lwe_in || !luts || !lwe_out) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return wopbs_dev(c, lwe_in, n_inputs, bits, luts, n_luts, lut_per_input, lwe_out);
    if (bits < 1 || bits > MAX_WOPBS_BITS || n_luts < 1) return c->fail(FHEAES_ERR_INVALID, "bits must be 1..%u and n_luts >= 1", MAX_WOPBS_BITS);
    Staged s(c);
    void *din, *dl, *dout;
    const uint64_t sets = lut_per_input ? n_inputs : 1;
    TRY(s.in(lwe_in, n_inputs * bits * c->big1 * 8, &din));
    TRY(s.in(luts, sets * n_luts * bits * lut_row_words(bits) * 8, &dl));
    TRY(s.alloc(&dout, n_inputs * n_luts * bits * c->big1 * 8));
    TRY(wopbs_dev(c, (const uint64_t *)din, n_inputs, bits, (const uint64_t *)dl, n_luts, lut_per_input, (uint64_t *)dout));
    return s.out(lwe_out, dout, n_inputs * n_luts * bits * c->big1 * 8);
}

static int many_sbox_dev(fheaes_ctx *c, const uint64_t *bytes, uint64_t n_bytes, int set, uint64_t *out)
{
    return wopbs_dev(c, bytes, n_bytes, 8, c->lutset_d[set], (uint32_t)c->lutset_n[set], 0, out);
}

int fheaes_many_sbox(fheaes_ctx *c, const uint64_t *bytes, uint64_t n_bytes, int inv, uint64_t *out, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!bytes || !out) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    const int set = inv ? LUTSET_DEC_MUL : LUTSET_ENC_ROUND;
    if (memspace == FHEAES_DEVICE) return many_sbox_dev(c, bytes, n_bytes, set, out);
    Staged s(c);
    void *din, *dout;
    const uint64_t bw = 8ull * c->big1;
    TRY(s.in(bytes, n_bytes * bw * 8, &din));
    TRY(s.alloc(&dout, n_bytes * c->lutset_n[set] * bw * 8));
    TRY(many_sbox_dev(c, (const uint64_t *)din, n_bytes, set, (uint64_t *)dout));
    return s.out(out, dout, n_bytes * c->lutset_n[set] * bw * 8);
}

int fheaes_sbox(fheaes_ctx *c, uint64_t *bytes, uint64_t n_bytes, int inv, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!bytes) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    const int set = inv ? LUTSET_INV_SBOX : LUTSET_SBOX;
    const uint64_t bw = 8ull * c->big1;
    if (memspace == FHEAES_DEVICE) {
        TRY(ensure(c, c->ws_vp, n_bytes * bw * 8));
        TRY(many_sbox_dev(c, bytes, n_bytes, set, (uint64_t *)c->ws_vp.p));
        HIP_TRY(c, hipMemcpyAsync(bytes, c->ws_vp.p, n_bytes * bw * 8, hipMemcpyDeviceToDevice, c->stream));
        return FHEAES_OK;
    }
    Staged s(c);
    void *din, *dout;
    TRY(s.in(bytes, n_bytes * bw * 8, &din));
    TRY(s.alloc(&dout, n_bytes * bw * 8));
    TRY(many_sbox_dev(c, (const uint64_t *)din, n_bytes, set, (uint64_t *)dout));
    return s.out(bytes, dout, n_bytes * bw * 8);
}

// ---- Server API -------------------------------------------------------------------------------
static int aes_encrypt_dev(fheaes_ctx *c, const uint64_t *rk, uint64_t *state, uint64_t n_blocks)
{
    const uint64_t bw = 8ull * c->big1, sw = 16 * bw, nbytes = 16 * n_blocks;
    TRY(ensure(c, c->ws_vp, nbytes * 3 * bw * 8));
    uint64_t *vp = (uint64_t *)c->ws_vp.p;
    const GatherTable t_round = table_enc_round(), t_shift = table_shift_rows(false);
    TRY(launch_add_bcast(c, state, rk, sw, n_blocks));                                   // server.rs:42
    for (int round = 1; round < 10; ++round) {                                           // server.rs:44-57
        TRY(many_sbox_dev(c, state, nbytes, LUTSET_ENC_ROUND, vp));
        TRY(launch_gather(c, vp, 3, rk + (uint64_t)round * sw, state, n_blocks, t_round));
    }
    TRY(many_sbox_dev(c, state, nbytes, LUTSET_SBOX, vp));                               // server.rs:59-63
    TRY(launch_gather(c, vp, 1, rk + 10ull * sw, state, n_blocks, t_shift));
    return FHEAES_OK;
}

static int aes_decrypt_dev(fheaes_ctx *c, const uint64_t *rk, uint64_t *state, uint64_t n_blocks)
{
    const uint64_t bw = 8ull * c->big1, sw = 16 * bw, nbytes = 16 * n_blocks;
    TRY(ensure(c, c->ws_vp, nbytes * 4 * bw * 8));
    uint64_t *vp = (uint64_t *)c->ws_vp.p;
    const GatherTable t_inv = table_shift_rows(true), t_mix = table_dec_mix();
    TRY(launch_add_bcast(c, state, rk + 10ull * sw, sw, n_blocks));                      // server.rs:70
    for (int round = 10; round >= 2; --round) {                                          // server.rs:72-96
        // inv_shift_rows commutes with the bytewise S-Box: INV_SBOX first, then the permutation + round key
        TRY(many_sbox_dev(c, state, nbytes, LUTSET_INV_SBOX, vp));
        TRY(launch_gather(c, vp, 1, rk + (uint64_t)(round - 1) * sw, state, n_blocks, t_inv));
        TRY(many_sbox_dev(c, state, nbytes, LUTSET_DEC_MUL, vp));
        TRY(launch_gather(c, vp, 4, nullptr, state, n_blocks, t_mix));
    }
    TRY(many_sbox_dev(c, state, nbytes, LUTSET_INV_SBOX, vp));                           // server.rs:98-104
    TRY(launch_gather(c, vp, 1, rk, state, n_blocks, t_inv));
    return FHEAES_OK;
}

static int aes_crypt(fheaes_ctx *c, const uint64_t *round_keys, uint64_t *state, uint64_t n_blocks, int memspace, bool dec)
{
    TRY(check_keys(c));
    if (!round_keys || !state) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return dec ? aes_decrypt_dev(c, round_keys, state, n_blocks) : aes_encrypt_dev(c, round_keys, state, n_blocks);
    Staged s(c);
    void *drk, *dst;
    const uint64_t sw = 16ull * 8 * c->big1;
    TRY(s.in(round_keys, 11 * sw * 8, &drk));
    TRY(s.in(state, n_blocks * sw * 8, &dst));
    TRY(dec ? aes_decrypt_dev(c, (const uint64_t *)drk, (uint64_t *)dst, n_blocks) : aes_encrypt_dev(c, (const uint64_t *)drk, (uint64_t *)dst, n_blocks));
    return s.out(state, dst, n_blocks * sw * 8);
}

int fheaes_aes_encrypt(fheaes_ctx *c, const uint64_t *round_keys, uint64_t *state, uint64_t n_blocks, int memspace)
{
    CtxLock lock__(c);
    return aes_crypt(c, round_keys, state, n_blocks, memspace, false);
}

int fheaes_aes_decrypt(fheaes_ctx *c, const uint64_t *round_keys, uint64_t *state, uint64_t n_blocks, int memspace)
{
    CtxLock lock__(c);
    return aes_crypt(c, round_keys, state, n_blocks, memspace, true);
}

static int key_expansion_dev(fheaes_ctx *c, const uint64_t *key, uint64_t *w)
{
    static const uint8_t RCON[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
    const uint64_t bw = 8ull * c->big1, ww = 4 * bw;
    TRY(ensure(c, c->ws_tmp_a, ww * 8));
    TRY(ensure(c, c->ws_tmp_b, ww * 8));
    uint64_t *ta = (uint64_t *)c->ws_tmp_a.p, *tb = (uint64_t *)c->ws_tmp_b.p;
    HIP_TRY(c, hipMemcpyAsync(w, key, 4 * ww * 8, hipMemcpyDeviceToDevice, c->stream));             // server.rs:122-128
    for (int i = 4; i < 44; ++i) {                                                                  // server.rs:131-155
        const uint64_t *prev = w + (uint64_t)(i - 1) * ww;
        if (i % 4 == 0) {
            for (int j = 0; j < 4; ++j)                                                             // fhe_rot_word
                HIP_TRY(c, hipMemcpyAsync(ta + (uint64_t)j * bw, prev + (uint64_t)((j + 1) & 3) * bw, bw * 8, hipMemcpyDeviceToDevice, c->stream));
            TRY(many_sbox_dev(c, ta, 4, LUTSET_SBOX, tb));                                          // fhe_sub_word
            hipLaunchKernelGGL(add_const_byte_kernel, dim3(1), dim3(64), 0, c->stream, tb, c->big1, (uint32_t)RCON[i / 4 - 1]);
            HIP_TRY(c, hipGetLastError());
            TRY(launch_add2(c, ta, tb, w + (uint64_t)(i - 4) * ww, ww));
        } else {
            TRY(launch_add2(c, ta, prev, w + (uint64_t)(i - 4) * ww, ww));
        }
        TRY(many_sbox_dev(c, ta, 4, LUTSET_IDENTITY, w + (uint64_t)i * ww));                        // refresh, server.rs:150
    }
    return FHEAES_OK;
}

int fheaes_aes_key_expansion(fheaes_ctx *c, const uint64_t *key, uint64_t *round_keys, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!key || !round_keys) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (memspace == FHEAES_DEVICE) return key_expansion_dev(c, key, round_keys);
    Staged s(c);
    void *dk, *dw;
    const uint64_t sw = 16ull * 8 * c->big1;
    TRY(s.in(key, sw * 8, &dk));
    TRY(s.alloc(&dw, 11 * sw * 8));
    TRY(key_expansion_dev(c, (const uint64_t *)dk, (uint64_t *)dw));
    return s.out(round_keys, dw, 11 * sw * 8);
}

static int add_scalar_dev(fheaes_ctx *c, uint64_t *state, uint64_t n_blocks, const uint64_t *counters)
{
    const uint32_t lw = c->big1;
    // counter bytes, MSB first (server.rs:174-178): addend[byte][blk], staged through a context-owned pinned buffer so
    // that the call only ENQUEUES (fheaes.h: FHEAES_DEVICE calls are not synchronised).  The only wait is for the copy
    // out of that buffer that an earlier add_scalar enqueued.
    const size_t add_bytes = 16 * n_blocks;
    if (c->pin_ev) HIP_TRY(c, hipEventSynchronize(c->pin_ev));
    else HIP_TRY(c, hipEventCreateWithFlags(&c->pin_ev, hipEventDisableTiming));
    if (c->pin_bytes < add_bytes) {
        if (c->pin) { HIP_TRY(c, hipHostFree(c->pin)); c->pin = nullptr; c->pin_bytes = 0; }
        HIP_TRY(c, hipHostMalloc((void **)&c->pin, add_bytes, hipHostMallocDefault));
        c->pin_bytes = add_bytes;
    }
    uint8_t *add = c->pin;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        uint64_t hi = counters[2 * b], lo = counters[2 * b + 1];
        for (int j = 0; j < 8; ++j) { add[(15 - j) * n_blocks + b] = (uint8_t)(lo >> (8 * j)); add[(7 - j) * n_blocks + b] = (uint8_t)(hi >> (8 * j)); }
    }
    TRY(ensure(c, c->ws_misc, 16 * n_blocks + n_blocks * lw * 8 + 64));
    uint8_t *add_d = (uint8_t *)c->ws_misc.p;
    uint64_t *carry = (uint64_t *)((uint8_t *)c->ws_misc.p + ((16 * n_blocks + 63) / 64) * 64);
    HIP_TRY(c, hipMemcpyAsync(add_d, add, add_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->pin_ev, c->stream));
    TRY(ensure(c, c->ws_tmp_a, n_blocks * 9ull * lw * 8));
    TRY(ensure(c, c->ws_tmp_b, n_blocks * 2ull * 9 * lw * 8));
    TRY(ensure(c, c->ws_luts, n_blocks * 2ull * 9 * FHE_N * 8));
    uint64_t *in9 = (uint64_t *)c->ws_tmp_a.p, *res = (uint64_t *)c->ws_tmp_b.p, *luts = (uint64_t *)c->ws_luts.p;
    for (int byte = 15; byte >= 0; --byte) {
        const uint32_t bits = byte == 15 ? 8 : 9;
        dim3 g1((bits * lw + 255) / 256, (unsigned)n_blocks);
        hipLaunchKernelGGL(pack9_kernel, g1, dim3(256), 0, c->stream, (const uint64_t *)state, (const uint64_t *)carry, in9, (uint32_t)byte, lw, n_blocks, bits);
        hipLaunchKernelGGL(counter_lut_kernel, dim3((2 * bits * FHE_N + 255) / 256, (unsigned)n_blocks), dim3(256), 0, c->stream, luts,
                           (const uint8_t *)(add_d + (size_t)byte * n_blocks), bits, n_blocks);
        HIP_TRY(c, hipGetLastError());
        TRY(wopbs_dev(c, in9, n_blocks, bits, luts, 2, 1, res));
        hipLaunchKernelGGL(unpack_sum_carry_kernel, dim3((9 * lw + 255) / 256, (unsigned)n_blocks), dim3(256), 0, c->stream, (const uint64_t *)res, bits, state, carry,
                           (uint32_t)byte, lw, n_blocks);
        HIP_TRY(c, hipGetLastError());
    }
    return FHEAES_OK;
}

int fheaes_add_scalar(fheaes_ctx *c, uint64_t *state, uint64_t n_blocks, const uint64_t *counters_hi_lo, int memspace)
{
    CtxLock lock__(c);
    TRY(check_keys(c));
    if (!state || !counters_hi_lo) return c->fail(FHEAES_ERR_INVALID, "null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_blocks == 0) return FHEAES_OK;
    if (memspace == FHEAES_DEVICE) return add_scalar_dev(c, state, n_blocks, counters_hi_lo);
    Staged s(c);
    void *dst;
    const uint64_t sw = 16ull * 8 * c->big1;
    TRY(s.in(state, n_blocks * sw * 8, &dst));
    TRY(add_scalar_dev(c, (uint64_t *)dst, n_blocks, counters_hi_lo));
    return s.out(state, dst, n_blocks * sw * 8);
}

// ---- measurement ------------------------------------------------------------------------------
int fheaes_profile_enable(fheaes_ctx *c, int on)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    if (!on && c->prof) TRY(prof_flush(c));
    c->prof = on != 0;
    return FHEAES_OK;
}

int fheaes_profile_reset(fheaes_ctx *c)
{
    CtxLock lock__(c);
    if (!c) return FHEAES_ERR_INVALID;
    TRY(prof_flush(c));
    for (int s = 0; s < FHEAES_STAGE_COUNT; ++s) { c->stage_ms[s] = 0; c->stage_launches[s] = 0; c->stage_units[s] = 0; }
    return FHEAES_OK;
}

int fheaes_profile_read(fheaes_ctx *c, int stage, double *total_ms, uint64_t *launches, uint64_t *units)
{
    CtxLock lock__(c);
    if (!c || stage < 0 || stage >= FHEAES_STAGE_COUNT) return FHEAES_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    TRY(prof_flush(c));
    if (total_ms) *total_ms = c->stage_ms[stage];
    if (launches) *launches = c->stage_launches[stage];
    if (units) *units = c->stage_units[stage];
    return FHEAES_OK;
}

}  // extern "C"
