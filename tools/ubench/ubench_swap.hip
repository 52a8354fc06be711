// v_permlane16_swap_b32 on gfx950: semantics and issue cost, alone and next to v_fma_f64 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

__global__ void k_sem(unsigned *o)
{
    unsigned l = threadIdx.x;
    uint2v r = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
}

// NS swaps (independent register pairs) + NF v_fma_f64 per loop trip
template <int NS, int NF>
__global__ __launch_bounds__(256) void k_mix(double *out, double a0, unsigned u0)
{
    unsigned x[16], y[16];
    double v[8];
    for (int i = 0; i < 16; ++i) { x[i] = u0 + i + threadIdx.x; y[i] = u0 * 3 + i; }
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    double a = a0 + threadIdx.x, b = a0 * 0.5;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            uint2v r = __builtin_amdgcn_permlane16_swap(x[i & 15], y[i & 15], false, false);
            x[i & 15] = r[0]; y[i & 15] = r[1];
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) v[j & 7] = __builtin_fma(v[j & 7], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += (double)(x[i] ^ y[i]);
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the LDS-crossbar alternative: ds_bpermute_b32 (no LDS storage)
template <int NS, int NF>
__global__ __launch_bounds__(256) void k_bperm(double *out, double a0, unsigned u0)
{
    unsigned x[16];
    double v[8];
    for (int i = 0; i < 16; ++i) x[i] = u0 + i + threadIdx.x;
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    double a = a0 + threadIdx.x, b = a0 * 0.5;
    const int addr = ((threadIdx.x ^ 16) & 63) * 4;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i & 15] = __builtin_amdgcn_ds_bpermute(addr, x[i & 15]);
#pragma unroll
        for (int j = 0; j < NF; ++j) v[j & 7] = __builtin_fma(v[j & 7], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += (double)x[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> double time_ms(F launch)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) launch();
    (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    unsigned *o; (void)hipMalloc((void **)&o, 512);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, o);
    std::vector<unsigned> h(128);
    (void)hipMemcpy(h.data(), o, 512, hipMemcpyDeviceToHost);
    printf("permlane16_swap(vdst = lane, src0 = 100 + lane):\n  vdst':");
    for (int i = 0; i < 64; ++i) printf(" %u", h[i]);
    printf("\n  src0':");
    for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]);
    printf("\n");
    const int cus = 256;
    double *buf; (void)hipMalloc((void **)&buf, (size_t)cus * 2 * 256 * 8);
#define RUN(K, NS, NF, label) { \
        double t1 = time_ms([&] { hipLaunchKernelGGL((K<NS, NF>), dim3(cus), dim3(256), 0, 0, buf, 1.25, 7u); }); \
        double t2 = time_ms([&] { hipLaunchKernelGGL((K<NS, NF>), dim3(cus * 2), dim3(256), 0, 0, buf, 1.25, 7u); }); \
        printf("%-34s 1w/SIMD %7.3f ms (%6.1f nominal cyc/trip)   2w/SIMD %7.3f ms (%6.1f cyc/trip/SIMD)\n", label, t1, t1 * 1e-3 * 2.4e9 / ITER, t2, t2 * 1e-3 * 2.4e9 / ITER); }
    RUN(k_mix, 0, 64, "64 v_fma_f64")
    RUN(k_mix, 16, 0, "16 swaps")
    RUN(k_mix, 32, 0, "32 swaps")
    RUN(k_mix, 16, 64, "16 swaps + 64 v_fma_f64")
    RUN(k_mix, 32, 64, "32 swaps + 64 v_fma_f64")
    RUN(k_bperm, 16, 0, "16 ds_bpermute")
    RUN(k_bperm, 16, 64, "16 ds_bpermute + 64 v_fma_f64")
    RUN(k_bperm, 32, 64, "32 ds_bpermute + 64 v_fma_f64")
    return 0;
}
