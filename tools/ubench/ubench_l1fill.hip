// L1 fill rate of the GGSW-row access pattern (developer tool): every workgroup walks the same 100 KB "levels" of a large
// buffer (so the lines come from L2, as in the blind rotation), each wave requesting NL x 1 KB per level with
// buffer_load_dwordx4, under different address layouts, bursts or loads spaced by vector work, 1 or 2 workgroups per CU.
//   layout 0: level + q * 4 KB + wave * 1 KB + lane * 16      (the engine's Fourier BSK: entry-major, 256 points x 16 B per entry)
//   layout 1: level + wave * NL KB + q * 1 KB + lane * 16     (wave-contiguous)
//   layout 2: level + q * 4 KB + ((wave + q) & 3) * 1 KB + lane * 16   (entry-major, waves skewed over the row)
// Prints bytes per clock per CU (s_memtime-based, averaged over waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 25, LEVELS = 400;
constexpr unsigned LEVEL_BYTES = 25 * 4096;

template <int LAYOUT, int SPACE, int PAD_KB, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(const double *buf, unsigned buf_bytes, unsigned long long *cyc, double *sink)
{
    __shared__ double pad[PAD_KB * 128];
    if (threadIdx.x == 0) pad[0] = 1.0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(buf), 0, (int)buf_bytes, 0x00020000);
    double acc = 0.0, f = 1.0 + lane;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < LEVELS; ++l) {
        const unsigned lb = (unsigned)l * LEVEL_BYTES;
        u32x4 v[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            unsigned off;
            if (LAYOUT == 0) off = q * 4096u + (wave & 3) * 1024u;
            else if (LAYOUT == 1) off = (wave & 3) * (NL * 1024u) + q * 1024u;
            else off = q * 4096u + ((wave + q) & 3) * 1024u;
            v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, lb + off, 0);
            if (SPACE) {
#pragma unroll
                for (int s = 0; s < SPACE; ++s) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f) : "v"(acc));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NL; ++q) acc += __uint_as_float(v[q][0] ^ v[q][3]);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    sink[(blockIdx.x * 256 + threadIdx.x) & (512 * 256 - 1)] = acc + f + pad[0];
}

template <int LAYOUT, int SPACE, int PAD_KB, int THREADS = 256>
void run(const char *label, int wgs, const double *buf, unsigned bytes, unsigned long long *dc, double *sink)
{
    const int grid = 256 * wgs;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<LAYOUT, SPACE, PAD_KB, THREADS>), dim3(grid), dim3(THREADS), 0, 0, buf, bytes, dc, sink);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<LAYOUT, SPACE, PAD_KB, THREADS>), dim3(grid), dim3(THREADS), 0, 0, buf, bytes, dc, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= h.size();
    // s_memtime ticks at 100 MHz on this part; convert with the launch time instead: cycles = ms * clk
    const double bytes_per_cu = (double)wgs * (THREADS / 64) * NL * 1024.0 * LEVELS;
    printf("%-44s waves/WG=%d wg/CU=%d  %8.3f ms  %6.1f B/clk/CU at 2.4 GHz  (%.0f ticks/wave)\n", label, THREADS / 64, wgs, ms, bytes_per_cu / (ms * 1e-3 * 2.4e9), avg);
}

int main()
{
    const unsigned bytes = LEVELS * LEVEL_BYTES;
    double *buf, *sink; unsigned long long *dc;
    (void)hipMalloc((void **)&buf, bytes); (void)hipMemset(buf, 1, bytes);
    (void)hipMalloc((void **)&sink, 512 * 256 * 8); (void)hipMalloc((void **)&dc, 512 * 4 * 8);
    // PAD 72 KB -> two workgroups per CU fit (grid 512), PAD 100 KB -> one per CU (grid 256)
    run<0, 0, 100>("entry-major, burst", 1, buf, bytes, dc, sink);
    run<1, 0, 100>("wave-contiguous, burst", 1, buf, bytes, dc, sink);
    run<2, 0, 100>("entry-major skewed, burst", 1, buf, bytes, dc, sink);
    run<0, 8, 100>("entry-major, 8 fma between loads", 1, buf, bytes, dc, sink);
    run<0, 32, 100>("entry-major, 32 fma between loads", 1, buf, bytes, dc, sink);
    run<1, 32, 100>("wave-contiguous, 32 fma between loads", 1, buf, bytes, dc, sink);
    run<0, 0, 72>("entry-major, burst", 2, buf, bytes, dc, sink);
    run<1, 0, 72>("wave-contiguous, burst", 2, buf, bytes, dc, sink);
    run<2, 0, 72>("entry-major skewed, burst", 2, buf, bytes, dc, sink);
    run<0, 32, 72>("entry-major, 32 fma between loads", 2, buf, bytes, dc, sink);
    run<1, 32, 72>("wave-contiguous, 32 fma between loads", 2, buf, bytes, dc, sink);
    run<0, 0, 100, 512>("entry-major, burst, 8 waves", 1, buf, bytes, dc, sink);
    run<0, 0, 100, 1024>("entry-major, burst, 16 waves", 1, buf, bytes, dc, sink);
    run<0, 0, 100, 128>("entry-major, burst, 2 waves", 1, buf, bytes, dc, sink);
    run<0, 0, 100, 64>("entry-major, burst, 1 wave", 1, buf, bytes, dc, sink);
    run<0, 0, 36, 256>("entry-major, burst, 4 WG/CU", 4, buf, bytes, dc, sink);
    return 0;
}
