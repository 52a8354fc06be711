// Issue cost of the individual VALU instructions the blind rotation is made of, at 1, 2 and 4 waves per SIMD
// (developer tool).  Each kernel issues ITER x 32 instructions of one kind on 8 independent register chains.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2048;

#define KERNEL(NAME, DECL, BODY)                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned u0, double d0)  \
    {                                                                                   \
        DECL;                                                                           \
        for (int it = 0; it < ITER; ++it) {                                             \
            _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) { BODY; }               \
        }                                                                               \
        unsigned s = 0;                                                                 \
        for (int i = 0; i < 8; ++i) s += (unsigned)x[i];                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                 \
    }

#define U32DECL unsigned x[8]; for (int i = 0; i < 8; ++i) x[i] = u0 + i + threadIdx.x; unsigned y = u0 * 3 + threadIdx.x
#define F64DECL double x[8]; for (int i = 0; i < 8; ++i) x[i] = d0 + i + threadIdx.x; double y = d0 * 0.5
#define EACH(ASM, CONSTR) _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : CONSTR)

KERNEL(k_add_u32, U32DECL, EACH("v_add_u32 %0, %0, %1", "v"(y)))
KERNEL(k_and_b32, U32DECL, EACH("v_and_b32 %0, %0, %1", "v"(y)))
KERNEL(k_lshl_b32, U32DECL, EACH("v_lshlrev_b32 %0, 1, %0", "v"(y)))
KERNEL(k_lshl_or, U32DECL, EACH("v_lshl_or_b32 %0, %0, 1, %1", "v"(y)))
KERNEL(k_bfe_u32, U32DECL, EACH("v_bfe_u32 %0, %0, 3, 8", "v"(y)))
KERNEL(k_cndmask, U32DECL, EACH("v_cndmask_b32 %0, %0, %1, vcc", "v"(y)))
KERNEL(k_mov_b32, U32DECL, EACH("v_mov_b32 %0, %1", "v"(y)))
KERNEL(k_add_co, U32DECL, EACH("v_add_co_u32 %0, vcc, %0, %1", "v"(y)))
KERNEL(k_fma_f32, U32DECL, EACH("v_fma_f32 %0, %0, %1, %1", "v"(y)))
KERNEL(k_pk_fma_f32, F64DECL, EACH("v_pk_fma_f32 %0, %0, %1, %1", "v"(y)))
KERNEL(k_fma_f64, F64DECL, EACH("v_fma_f64 %0, %0, %1, %1", "v"(y)))
KERNEL(k_add_f64, F64DECL, EACH("v_add_f64 %0, %0, %1", "v"(y)))
KERNEL(k_mul_f64, F64DECL, EACH("v_mul_f64 %0, %0, %1", "v"(y)))
KERNEL(k_rndne_f64, F64DECL, EACH("v_rndne_f64 %0, %0", "v"(y)))
KERNEL(k_ldexp_f64, F64DECL, EACH("v_ldexp_f64 %0, %0, 1", "v"(y)))
KERNEL(k_lshr_b64, F64DECL, EACH("v_lshrrev_b64 %0, 1, %0", "v"(y)))
KERNEL(k_lshl_add_u64, F64DECL, EACH("v_lshl_add_u64 %0, %0, 1, %1", "v"(y)))
__global__ __launch_bounds__(256) void k_cvt_f64_i32(unsigned *out, unsigned u0, double d0)
{
    double x[8]; int z[8];
    for (int i = 0; i < 8; ++i) z[i] = u0 + i + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[i]) : "v"(z[i]));
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += (unsigned)x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cvt_i32_f64(unsigned *out, unsigned u0, double d0)
{
    double x[8]; int z[8];
    for (int i = 0; i < 8; ++i) x[i] = d0 + i + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(z[i]) : "v"(x[i]));
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += (unsigned)z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> double time_ms(F launch)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 20; ++w) launch();      // let the clock settle under this load
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int w = 0; w < 10; ++w) launch();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main()
{
    unsigned *buf; (void)hipMalloc((void **)&buf, (size_t)256 * 4 * 256 * 4);
    printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal (ITER*32 instructions per wave)\n%-18s %10s %10s %10s\n", "", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
#define RUN(K) { double c[3]; int k = 0; for (int wg : {1, 2, 4}) { double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(256 * wg), dim3(256), 0, 0, buf, 7u, 1.25); }); \
        c[k++] = ms * 1e-3 * 2.4e9 / (ITER * 32.0) / wg; } printf("%-18s %10.2f %10.2f %10.2f\n", #K, c[0], c[1], c[2]); }
    RUN(k_fma_f64) RUN(k_add_f64) RUN(k_mul_f64) RUN(k_rndne_f64) RUN(k_ldexp_f64) RUN(k_cvt_f64_i32) RUN(k_cvt_i32_f64)
    RUN(k_add_u32) RUN(k_and_b32) RUN(k_lshl_b32) RUN(k_lshl_or) RUN(k_bfe_u32) RUN(k_cndmask) RUN(k_mov_b32) RUN(k_add_co)
    RUN(k_lshr_b64) RUN(k_lshl_add_u64) RUN(k_fma_f32) RUN(k_pk_fma_f32)
    return 0;
}
