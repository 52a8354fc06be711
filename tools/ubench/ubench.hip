// Micro-benchmarks of instruction issue rates on gfx950 (developer tool, not part of the product).
// Each kernel runs ITER x UNROLL independent-accumulator instructions per lane; we report lane-ops/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;

__global__ void k_dot4(unsigned *out, unsigned a0, unsigned b0) {
    unsigned acc[16];
    unsigned b = b0 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_udot4(a0 + i, b, acc[i], false);
        b += acc[0];
    }
    unsigned s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64(unsigned long long *out, unsigned a0, unsigned b0) {
    unsigned long long acc[16];
    unsigned b = b0 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += (unsigned long long)(a0 + i) * b;
        b += (unsigned)acc[0];
    }
    unsigned long long s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(double *out, double a0, double b0) {
    double acc[16];
    double b = b0 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a0 + i, b, acc[i]);
        b += 1e-9;
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64f(double *out, double a0, double b0) {
    double acc[16];
    double b = b0 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc[i] + b;
        b += 1e-9;
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addu32(unsigned *out, unsigned a0, unsigned b0) {
    unsigned acc[16];
    unsigned b = b0 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (acc[i] ^ b) + (a0 + i);   // v_xad_u32 / 2 ops
        b += acc[3];
    }
    unsigned s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k_mfma_i8(int *out, int a0) {
    int4v acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (int4v){0, 0, 0, 0};
    int4v a = (int4v){a0 + (int)threadIdx.x, a0, a0 + 1, a0 + 2};
    int4v b = (int4v){a0 * 3, a0 + 7, (int)threadIdx.x, a0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mfma_f64(double *out, double a0) {
    double4v acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (double4v){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = a0 * 0.5 + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// bit-exactness probe: D = C + sum_k a_k b_k of the f64 MFMA vs fma chains in both orders
__global__ void k_mfma_f64_probe(const double *A, const double *B, const double *C, double *D) {
    // 16x16x4: lane l holds A[row=l%16][k=l/16], B[k=l/16][col=l%16]; D: 4 values per lane: rows 4*(l/16)+j, col l%16
    int l = threadIdx.x;
    double4v c;
    for (int j = 0; j < 4; ++j) c[j] = C[(4 * (l / 16) + j) * 16 + (l % 16)];
    double4v d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + (l % 16)], c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (l / 16) + j) * 16 + (l % 16)] = d[j];
}

template <typename F> double time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    int cus = 256, wgs = cus * 8, threads = 256;
    void *buf; CHECK(hipMalloc(&buf, (size_t)wgs * threads * 8));
    double clk_ghz = 2.4;
    auto report = [&](const char *name, double ms, double ops_per_lane) {
        double lane_ops = (double)wgs * threads * ops_per_lane;
        printf("%-14s %8.3f ms  %7.2f Tlane-op/s  %6.1f lane-ops/clk/CU (at %.1f GHz nominal)\n", name, ms, lane_ops / ms / 1e9,
               lane_ops / (ms * 1e-3) / (cus * clk_ghz * 1e9), clk_ghz);
    };
    report("v_dot4_u32_u8", time_ms([&] { hipLaunchKernelGGL(k_dot4, dim3(wgs), dim3(threads), 0, 0, (unsigned *)buf, 3u, 5u); }), 16.0 * ITER);
    report("mad_u64_u32", time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(wgs), dim3(threads), 0, 0, (unsigned long long *)buf, 3u, 5u); }), 16.0 * ITER);
    report("v_fma_f64", time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(wgs), dim3(threads), 0, 0, (double *)buf, 1.5, 2.5); }), 16.0 * ITER);
    report("v_add_f64", time_ms([&] { hipLaunchKernelGGL(k_add64f, dim3(wgs), dim3(threads), 0, 0, (double *)buf, 1.5, 2.5); }), 16.0 * ITER);
    report("u32 xor+add", time_ms([&] { hipLaunchKernelGGL(k_addu32, dim3(wgs), dim3(threads), 0, 0, (unsigned *)buf, 3u, 5u); }), 16.0 * ITER);
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_i8, dim3(wgs), dim3(threads), 0, 0, (int *)buf, 3); });
        double macs = (double)wgs * (threads / 64) * 4.0 * ITER * 16 * 16 * 64;
        printf("%-14s %8.3f ms  %7.1f T i8-MAC/s   %6.1f i8-MAC/clk/CU\n", "mfma_i8 16x16x64", ms, macs / ms / 1e9, macs / (ms * 1e-3) / (cus * clk_ghz * 1e9));
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f64, dim3(wgs), dim3(threads), 0, 0, (double *)buf, 1.25); });
        double fl = (double)wgs * (threads / 64) * 4.0 * ITER * 16 * 16 * 4 * 2;
        printf("%-14s %8.3f ms  %7.2f TFLOP/s f64\n", "mfma_f64 16x16x4", ms, fl / ms / 1e9);
    }
    // probe
    std::vector<double> A(64), B(64), C(256), D(256);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(int64_t)s * 0x1p-40; };
    for (auto &x : A) x = rnd(); for (auto &x : B) x = rnd(); for (auto &x : C) x = rnd();
    double *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dC, 2048)); CHECK(hipMalloc(&dD, 2048));
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma_f64_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int fwd = 0, rev = 0, other = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double f = C[i * 16 + j], r = C[i * 16 + j];
        for (int k = 0; k < 4; ++k) f = __builtin_fma(A[i * 4 + k], B[k * 16 + j], f);
        for (int k = 3; k >= 0; --k) r = __builtin_fma(A[i * 4 + k], B[k * 16 + j], r);
        double d = D[i * 16 + j];
        if (d == f) ++fwd; else if (d == r) ++rev; else ++other;
    }
    printf("mfma_f64 probe: %d outputs == fma chain k=0..3, %d == chain k=3..0, %d neither\n", fwd, rev, other);
    return 0;
}
