// f64 matrix-core questions for the blind-rotation multiply-accumulate (developer tool, not product):
//  1. issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 on gfx950
//  2. does a wave's f64 MFMA stream overlap (a) v_fma_f64 of the SAME wave, (b) of ANOTHER wave on the same SIMD
//  3. exact semantics: is D = fma chain over k in ascending order, starting from C (bitwise)?
//  4. operand lane layout of the 4x4x4 4-block form
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double4v __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

// MODE bit0: MFMA16 stream, bit1: VALU fma stream, bit2: MFMA4x4 stream.  NV = v_fma_f64 per MFMA slot
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_mix(double *out, double a0)
{
    double4v acc[4];
    double acc4[8];
    double v[8];
    for (int i = 0; i < 4; ++i) acc[i] = (double4v){0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { v[i] = i + threadIdx.x; acc4[i] = i; }
    double a = a0 + threadIdx.x, b = a0 * 0.5 + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE & 4) {
                acc4[2 * i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[2 * i], 0, 0, 0);
                acc4[2 * i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc4[2 * i + 1], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[(i * NV + j) & 7] = __builtin_fma(v[(i * NV + j) & 7], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512-thread workgroup, one per CU: waves 0-3 run stream X, waves 4-7 stream Y (two waves per SIMD).
// X/Y: 0 idle, 1 MFMA16, 2 VALU fma, 4 MFMA 4x4x4
template <int X, int Y>
__global__ __launch_bounds__(512) void k_pair(double *out, double a0)
{
    const int wave = threadIdx.x >> 6;
    double4v acc[4];
    double acc4[8];
    double v[8];
    for (int i = 0; i < 4; ++i) acc[i] = (double4v){0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { v[i] = i + threadIdx.x; acc4[i] = i; }
    double a = a0 + threadIdx.x, b = a0 * 0.5 + threadIdx.x;
    const int role = wave < 4 ? X : Y;
    if (role == 1) {
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 64; ++i) v[i & 7] = __builtin_fma(v[i & 7], a, b);
    } else if (role == 4) {
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_probe16(const double *A, const double *B, const double *C, double *D)
{
    int l = threadIdx.x;
    double4v c;
    for (int j = 0; j < 4; ++j) c[j] = C[((l >> 4) + 4 * j) * 16 + (l & 15)];
    double4v d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[((l >> 4) + 4 * j) * 16 + (l & 15)] = d[j];
}
// raw lane in / lane out for the 4x4x4 form
__global__ void k_probe4(const double *a, const double *b, const double *c, double *d)
{
    int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

template <typename F> double time_ms(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int cus = 256;
    void *buf; CHECK(hipMalloc(&buf, (size_t)cus * 8 * 512 * 8));
    double *o = (double *)buf;
    // slots = ITER * 4 per wave; report cycles per slot per SIMD assuming 2.4 GHz
    auto rep = [&](const char *name, double ms, int waves_per_simd) {
        double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 4) / waves_per_simd;
        printf("%-44s %8.3f ms   %7.1f cyc/slot/wave-on-SIMD (2.4 GHz nominal)\n", name, ms, cyc);
    };
#define RUN1(MODE, NV, label) rep(label " 1w/SIMD", time_ms([&] { hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(cus), dim3(256), 0, 0, o, 1.25); }), 1); \
                              rep(label " 2w/SIMD", time_ms([&] { hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(cus * 2), dim3(256), 0, 0, o, 1.25); }), 2);
    printf("slot = one MFMA16 (2048 flop) and/or two MFMA4 (2 x 512 flop) and/or NV v_fma_f64 (NV x 128 flop)\n");
    RUN1(1, 0, "mfma16 only")
    RUN1(4, 0, "2 x mfma4x4x4 only")
    RUN1(2, 8, "8 v_fma only")
    RUN1(2, 16, "16 v_fma only")
    RUN1(3, 4, "mfma16 + 4 v_fma (same wave)")
    RUN1(3, 8, "mfma16 + 8 v_fma (same wave)")
    RUN1(3, 16, "mfma16 + 16 v_fma (same wave)")
    RUN1(6, 4, "2 mfma4 + 4 v_fma (same wave)")
    RUN1(6, 8, "2 mfma4 + 8 v_fma (same wave)")
    auto repp = [&](const char *name, double ms) { printf("%-44s %8.3f ms\n", name, ms); };
#define RUNP(X, Y, label) repp(label, time_ms([&] { hipLaunchKernelGGL((k_pair<X, Y>), dim3(cus), dim3(512), 0, 0, o, 1.25); }));
    printf("pairs: waves 0-3 | waves 4-7 of a 512-thread workgroup (one per CU); mfma16 stream = %d MFMA, valu = %d v_fma, mfma4 = %d\n", ITER * 4, ITER * 64, ITER * 8);
    RUNP(1, 0, "mfma16 | idle")
    RUNP(2, 0, "valu   | idle")
    RUNP(4, 0, "mfma4  | idle")
    RUNP(1, 1, "mfma16 | mfma16")
    RUNP(2, 2, "valu   | valu")
    RUNP(1, 2, "mfma16 | valu")
    RUNP(4, 2, "mfma4  | valu")
    RUNP(4, 4, "mfma4  | mfma4")

    // ---- exactness, 16x16x4 ----
    std::vector<double> A(64), B(64), C(256), D(256);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(int64_t)s * 0x1p-40; };
    double *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dC, 2048)); CHECK(hipMalloc(&dD, 2048));
    int fwd = 0, rev = 0, other = 0;
    for (int trial = 0; trial < 64; ++trial) {
        for (auto &x : A) x = rnd(); for (auto &x : B) x = rnd(); for (auto &x : C) x = rnd() * 1e6;
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double f = C[i * 16 + j], r = C[i * 16 + j];
            for (int k = 0; k < 4; ++k) f = __builtin_fma(A[i * 4 + k], B[k * 16 + j], f);
            for (int k = 3; k >= 0; --k) r = __builtin_fma(A[i * 4 + k], B[k * 16 + j], r);
            double d = D[i * 16 + j];
            if (d == f) ++fwd; else if (d == r) ++rev; else ++other;
        }
    }
    printf("mfma_f64_16x16x4 probe: %d == fma chain k=0..3 from C, %d == chain k=3..0 (only), %d neither\n", fwd, rev, other);

    // ---- layout + exactness, 4x4x4 4 blocks ----
    // layout discovery: a = 1 at a single lane la, b = 1 at a single lane lb: D gets 1 where A[i][k]*B[k][j] pairs up
    std::vector<double> a(64), b(64), c(64, 0.0), d(64);
    double *da, *db, *dc, *dd;
    CHECK(hipMalloc(&da, 512)); CHECK(hipMalloc(&db, 512)); CHECK(hipMalloc(&dc, 512)); CHECK(hipMalloc(&dd, 512));
    hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
    printf("4x4x4 layout: for a-lane la (block 0, lanes 0..15) x b-lane lb: output lane(s) that become nonzero\n");
    for (int la = 0; la < 16; ++la) {
        printf("  la=%2d:", la);
        for (int lb = 0; lb < 16; ++lb) {
            std::fill(a.begin(), a.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
            a[la] = 1.0; b[lb] = 1.0;
            hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
            int hit = -1, n = 0;
            for (int l = 0; l < 64; ++l) if (d[l] != 0.0) { hit = l; ++n; }
            if (n == 0) printf("  . "); else if (n == 1) printf(" %2d ", hit); else printf(" %2d+", hit);
        }
        printf("\n");
    }
    // cross-block check: a at lane 0, b at lane 16 (different blocks) should give nothing if blocks are independent
    {
        std::fill(a.begin(), a.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
        a[0] = 1.0; b[16] = 1.0;
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        int n = 0; for (int l = 0; l < 64; ++l) if (d[l] != 0.0) ++n;
        printf("4x4x4: a@lane0 x b@lane16 -> %d nonzero outputs (0 = blocks independent)\n", n);
        a[0] = 0; a[16] = 1.0;
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        printf("4x4x4: a@lane16 x b@lane16 -> nonzero at:");
        for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    // exactness under the hypothesis A[i][k] at lane i+4k, B[k][j] at lane j+4k, D[i][j] at lane j+4i (+16*block) -- verified against the table above by eye;
    // test all 4 candidate (a-lane, b-lane, d-lane) conventions numerically
    {
        int best_conv = -1;
        for (int conv = 0; conv < 8; ++conv) {
            int okf = 0, okr = 0, bad = 0;
            uint64_t s2 = 0x9E3779B97F4A7C15ull;
            auto rnd2 = [&]() { s2 ^= s2 << 13; s2 ^= s2 >> 7; s2 ^= s2 << 17; return (double)(int64_t)s2 * 0x1p-40; };
            for (int trial = 0; trial < 16; ++trial) {
                for (auto &x : a) x = rnd2(); for (auto &x : b) x = rnd2(); for (auto &x : c) x = rnd2() * 1e6;
                hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
                hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
                for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
                    auto AL = [&](int ii, int kk) { return 16 * blk + ((conv & 1) ? 4 * ii + kk : ii + 4 * kk); };
                    auto BL = [&](int kk, int jj) { return 16 * blk + ((conv & 2) ? 4 * jj + kk : jj + 4 * kk); };
                    int dl = 16 * blk + ((conv & 4) ? i + 4 * j : j + 4 * i);
                    double f = c[dl], r = c[dl];
                    for (int k = 0; k < 4; ++k) f = __builtin_fma(a[AL(i, k)], b[BL(k, j)], f);
                    for (int k = 3; k >= 0; --k) r = __builtin_fma(a[AL(i, k)], b[BL(k, j)], r);
                    if (d[dl] == f) ++okf; else if (d[dl] == r) ++okr; else ++bad;
                }
            }
            printf("4x4x4 convention %d (A %s, B %s, D %s): %d == chain k asc, %d == chain k desc only, %d neither\n", conv,
                   (conv & 1) ? "4i+k" : "i+4k", (conv & 2) ? "4j+k" : "j+4k", (conv & 4) ? "i+4j" : "j+4i", okf, okr, bad);
            if (bad == 0 && best_conv < 0) best_conv = conv;
        }
        printf("4x4x4 exact convention: %d\n", best_conv);
    }
    return 0;
}
