// Dependent-issue latency of v_fma_f64 / v_add_f64: NCH independent chains per wave, 1/2/4 waves per SIMD (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void k(double *out, double d0)
{
    double x[NCH];
    for (int i = 0; i < NCH; ++i) x[i] = d0 + i + threadIdx.x;
    double y = d0 * 0.5;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 32 / NCH; ++rep)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y));
                else asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
            }
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> double time_ms(F launch)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 10; ++w) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int w = 0; w < 5; ++w) launch();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    double *buf; (void)hipMalloc((void **)&buf, (size_t)256 * 4 * 256 * 8);
    printf("nominal-2.4GHz cycles per wave-instruction per SIMD; chains = independent accumulators per wave\n%-22s %9s %9s %9s\n", "", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
#define RUN(NCH, ADD, label) { double c[3]; int q = 0; for (int wg : {1, 2, 4}) { double ms = time_ms([&] { hipLaunchKernelGGL((k<NCH, ADD>), dim3(256 * wg), dim3(256), 0, 0, buf, 1.25); }); \
        c[q++] = ms * 1e-3 * 2.4e9 / (ITER * 32.0) / wg; } printf("%-22s %9.2f %9.2f %9.2f\n", label, c[0], c[1], c[2]); }
    RUN(1, false, "fma 1 chain") RUN(2, false, "fma 2 chains") RUN(4, false, "fma 4 chains") RUN(8, false, "fma 8 chains") RUN(16, false, "fma 16 chains")
    RUN(1, true, "add 1 chain") RUN(2, true, "add 2 chains") RUN(4, true, "add 4 chains")
    return 0;
}
