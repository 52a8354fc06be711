// Energy per byte / per operation of the data paths the blind rotation uses (developer tool, round 4).
//
// Round 4 found K2 power-capped (1,335 W of a 1,400 W cap; tools/gpu_power.py): under the cap the launch time is
// (energy per launch) / (power), not (cycles) / (clock), so what an operand costs in JOULES decides the design.
// Each mode runs ~1.5 s on all 256 CUs with 8 waves per CU and reads the socket energy counter (rocm_smi) around it:
//   idle      nothing (static power)
//   fma       v_fma_f64 back to back (4 independent chains per lane)
//   mfma i8   v_mfma_i32_16x16x64_i8 back to back on pseudo-random bytes (8 accumulators per wave)
//   lds       ds_read_b128 from a conflict-free 16 KB window; round 5: LDS STORES (b128 random / constant data, 2 x b64, 4 x ds_write_addtid_b32)
//   l1        buffer_load_dwordx4 of the same 8 KB per wave again and again (L1 hits)
//   l2        every workgroup walks the same 2 MB (L2 hits after the first touch; the blind rotation's key rows)
//   mall      every workgroup walks its own slice of 128 MB  (L2 misses, Infinity-Cache sized)
//   hbm       every workgroup walks its own slice of 4 GB    (L2 and Infinity-Cache misses)
//   st+ld     store then re-load a private 64 KB per workgroup (the accumulator parking pattern)
// Output: seconds, joules, watts, and picojoules per byte (or per flop) ABOVE the idle power.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_energy ubench_energy.hip -L/opt/rocm/lib -lrocm_smi64
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static double energy_j()
{
    uint64_t e = 0, ts = 0; float res = 0;
    if (rsmi_dev_energy_count_get(0, &e, &res, &ts) != RSMI_STATUS_SUCCESS) return -1.0;
    return (double)e * res * 1e-6;
}

// MASKED: lanes 16..63 of every wave leave at once, so the same instruction stream runs with a quarter of EXEC set: what does an
// inactive lane cost? (the question behind "switch the idle lane groups of the blind rotation off")
template <bool MASKED>
__global__ __launch_bounds__(512, 1) void k_fma(double *sink, int iters)
{
    if (MASKED && (threadIdx.x & 63) >= 16) return;
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
        }
    }
    sink[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// int8 matrix cores: v_mfma_i32_16x16x64_i8, 8 independent accumulators per wave, operand bytes pseudo-random and lane-dependent
// (the key-switching kernels K1 / K3 multiply balanced random bytes: all-zero operands would make the array look cheaper than it is)
__global__ __launch_bounds__(512, 1) void k_mfma_i8(int *sink, int iters)
{
    i32x4 a[2], b[4], acc[8];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) { x = x * 1664525u + 1013904223u; a[i][j] = (int)x; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { x = x * 1664525u + 1013904223u; b[i][j] = (int)x; }
    for (int i = 0; i < 8; ++i) acc[i] = (i32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u & 1], b[u & 3], acc[u], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = r;
}

__global__ __launch_bounds__(512, 1) void k_lds(double *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) double lds[8 * 1024];          // 64 KB
    for (int i = threadIdx.x; i < 8 * 1024; i += 512) lds[i] = i;
    __syncthreads();
    double acc = 0;
    const double2 *p = reinterpret_cast<const double2 *>(lds) + threadIdx.x;   // 16 B per lane, consecutive: conflict-free
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { double2 v = p[u * 512]; asm volatile("" :: "v"(v.x), "v"(v.y)); acc += 0.0 * v.x; }
        asm volatile("" ::: "memory");
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

// LDS STORES (round 5): MODE 0 = ds_write_b128 of pseudo-random doubles (a fresh value per store: xorshift on the lane's registers),
// 1 = ds_write_b128 of the same two doubles every time (what a stale-data proxy stores), 2 = two ds_write_b64, 3 = four
// ds_write_addtid_b32 (no address VGPR; M0-relative, dword-interleaved across the lanes).  16 B per lane and step in every mode.
template <int MODE>
__global__ __launch_bounds__(512, 1) void k_lds_store(double *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) double lds[8 * 1024];          // 64 KB
    for (int i = threadIdx.x; i < 8 * 1024; i += 512) lds[i] = i;
    __syncthreads();
    unsigned long long a = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 512 * blockIdx.x), b = a ^ 0xD1B54A32D192ED03ull;
    double2 *p = reinterpret_cast<double2 *>(lds) + threadIdx.x;           // 16 B per lane, consecutive: conflict-free
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 1) { a ^= a << 13; a ^= a >> 7; a ^= a << 17; b += a; }    // new bits every store (integer work: priced by MODE 1's difference to a pure loop)
            double2 v; v.x = __builtin_bit_cast(double, a); v.y = __builtin_bit_cast(double, b);
            if (MODE <= 1) p[u * 512] = v;
            else if (MODE == 2) {
                reinterpret_cast<double *>(lds)[u * 1024 + threadIdx.x] = v.x;
                reinterpret_cast<double *>(lds)[u * 1024 + 512 + threadIdx.x] = v.y;
            } else {
                const unsigned w0 = (unsigned)a, w1 = (unsigned)(a >> 32), w2 = (unsigned)b, w3 = (unsigned)(b >> 32);
                const unsigned m0 = (unsigned)(u * 8192 + (threadIdx.x >> 6) * 1024);      // this wave's 1 KB: four planes of 256 B
                asm volatile("s_mov_b32 m0, %4\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                             "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                             :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(__builtin_amdgcn_readfirstlane(m0)) : "memory", "m0");
            }
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    sink[blockIdx.x * 512 + threadIdx.x] = lds[threadIdx.x] + __builtin_bit_cast(double, a);
}
// the integer work of k_lds_store<0> without its stores (what to subtract)
__global__ __launch_bounds__(512, 1) void k_xorshift(double *sink, int iters)
{
    unsigned long long a = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 512 * blockIdx.x), b = a ^ 0xD1B54A32D192ED03ull;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a ^= a << 13; a ^= a >> 7; a ^= a << 17; b += a; asm volatile("" : "+v"(a), "+v"(b)); }
    }
    sink[blockIdx.x * 512 + threadIdx.x] = __builtin_bit_cast(double, a + b);
}

// walk `span` bytes (a multiple of 8 KB per wave step) starting at `base(blockIdx)`; 8 x 1 KB requests per wave per step
template <int MODE>
__global__ __launch_bounds__(512, 1) void k_load(const double *buf, unsigned long long buf_bytes, unsigned long long span, double *sink, int steps)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (MODE == 2) base = (unsigned long long)blockIdx.x * span % buf_bytes;          // own slice
    const char *b = reinterpret_cast<const char *>(buf) + base;
    float acc = 0;
    unsigned long long off = (unsigned long long)wave * 8192;
    for (int s = 0; s < steps; ++s) {
        u32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const u32x4 *>(b + off + q * 1024 + lane * 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += __uint_as_float(v[q][0] ^ v[q][3]);
        asm volatile("" ::: "memory");                            // MODE 0: the loads must be issued again, not hoisted
        if (MODE != 0) { off += 8 * 8192; if (off + 8192 > span) off = (unsigned long long)wave * 8192; }      // MODE 0: the same 8 KB again (L1)
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

// SLOTS x 64 KB per workgroup, visited round robin: every step stores one 64 KB slot and re-loads it (the parking pattern), so the
// time between two visits of a line, and the footprint (G x SLOTS x 64 KB), grow with SLOTS: 1 -> L2-resident, 8 -> 128 MB
// (Infinity-Cache sized), 64 -> 1 GB.  AUX_ST / AUX_LD: cache policy bits (1 = sc0, 2 = nt, 16 = sc1).
template <int AUX_ST, int AUX_LD>
__global__ __launch_bounds__(512, 1) void k_park(double *slab, double *sink, int steps, int slots)
{
    char *b = reinterpret_cast<char *>(slab) + (size_t)blockIdx.x * slots * 65536;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(b, 0, slots * 65536, 0x00020000);
    u32x4 v; v[0] = threadIdx.x; v[1] = 1; v[2] = 2; v[3] = 3;
    unsigned slot = 0;
    for (int s = 0; s < steps; ++s) {
        const unsigned so = slot * 65536u;
#pragma unroll
        for (int q = 0; q < 8; ++q) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, threadIdx.x * 16u, so + q * 8192u, AUX_ST);
        u32x4 w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, threadIdx.x * 16u, so + q * 8192u, AUX_LD);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[0] += w[q][1];
        slot = slot + 1 == (unsigned)slots ? 0 : slot + 1;
    }
    sink[blockIdx.x * 512 + threadIdx.x] = v[0];
}

struct Res { double s, j; };
template <typename F>
static Res timed(F launch)
{
    launch();                                  // warm-up
    (void)hipDeviceSynchronize();
    const double e0 = energy_j();
    const auto t0 = std::chrono::steady_clock::now();
    double s = 0;
    while (s < 1.5) { launch(); (void)hipDeviceSynchronize(); s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    return {s, energy_j() - e0};
}

int main()
{
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("rocm_smi not available\n"); return 1; }
    const int G = 256;
    double *sink, *buf, *slab;
    const unsigned long long big = 4ull << 30;
    (void)hipMalloc((void **)&sink, G * 512 * 8);
    if (hipMalloc((void **)&buf, big) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, big);
    slab = buf;                                   // the parking modes reuse the big buffer
    (void)hipDeviceSynchronize();
    // idle
    const double e0 = energy_j();
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));
    const double idle_w = (energy_j() - e0) / 1.5;
    printf("%-8s %7.1f W\n", "idle", idle_w);
    auto report = [&](const char *name, Res r, double units, const char *unit) {
        printf("%-20s %6.2f s  %8.1f J  %7.1f W   %8.2f pJ/%s above idle  (%.3g %s/s)\n", name, r.s, r.j, r.j / r.s, (r.j - idle_w * r.s) / units * 1e12, unit, units / r.s, unit);
        fflush(stdout);
    };
    {
        double n = 0; const int it = 40000;
        Res r = timed([&] { hipLaunchKernelGGL(k_fma<false>, dim3(G), dim3(512), 0, 0, sink, it); n += (double)G * 512 * it * 64 * 2; });
        report("fma", r, n, "flop");
    }
    {
        double n = 0; const int it = 40000;            // a quarter of the lanes active: flops counted for the active lanes only
        Res r = timed([&] { hipLaunchKernelGGL(k_fma<true>, dim3(G), dim3(512), 0, 0, sink, it); n += (double)G * 128 * it * 64 * 2; });
        report("fma 1/4", r, n, "flop");
    }
    {
        double n = 0; const int it = 40000;            // per wave-instruction 16 x 16 x 64 = 16,384 multiply-adds
        Res r = timed([&] { hipLaunchKernelGGL(k_mfma_i8, dim3(G), dim3(512), 0, 0, reinterpret_cast<int *>(sink), it); n += (double)G * 8 * it * 8 * 16384.0; });
        report("mfma i8", r, n, "MAC");
    }
    {
        double n = 0; const int it = 100000;
        Res r = timed([&] { hipLaunchKernelGGL(k_lds, dim3(G), dim3(512), 0, 0, sink, it); n += (double)G * 512 * it * 8 * 16; });
        report("lds", r, n, "B");
    }
    {
        const int it = 50000;
        const double per = (double)G * 512 * it * 8 * 16;
        { double n = 0; Res r = timed([&] { hipLaunchKernelGGL(k_xorshift, dim3(G), dim3(512), 0, 0, sink, it); n += per; }); report("xorshift (no store)", r, n, "B-equivalent"); }
        { double n = 0; Res r = timed([&] { hipLaunchKernelGGL(k_lds_store<0>, dim3(G), dim3(512), 0, 0, sink, it); n += per; }); report("lds st b128 rnd", r, n, "B"); }
        { double n = 0; Res r = timed([&] { hipLaunchKernelGGL(k_lds_store<1>, dim3(G), dim3(512), 0, 0, sink, it); n += per; }); report("lds st b128 const", r, n, "B"); }
        { double n = 0; Res r = timed([&] { hipLaunchKernelGGL(k_lds_store<2>, dim3(G), dim3(512), 0, 0, sink, it); n += per; }); report("lds st 2xb64 rnd", r, n, "B"); }
        { double n = 0; Res r = timed([&] { hipLaunchKernelGGL(k_lds_store<3>, dim3(G), dim3(512), 0, 0, sink, it); n += per; }); report("lds st 4xaddtid", r, n, "B"); }
    }
    const int steps = 20000;
    const double per_launch = (double)G * 512 * steps * 8 * 16;
    {
        double n = 0;
        Res r = timed([&] { hipLaunchKernelGGL(k_load<0>, dim3(G), dim3(512), 0, 0, buf, big, 65536ull, sink, steps); n += per_launch; });
        report("l1", r, n, "B");
    }
    {
        double n = 0;
        Res r = timed([&] { hipLaunchKernelGGL(k_load<1>, dim3(G), dim3(512), 0, 0, buf, big, 2ull << 20, sink, steps); n += per_launch; });
        report("l2", r, n, "B");
    }
    {
        double n = 0;
        Res r = timed([&] { hipLaunchKernelGGL(k_load<2>, dim3(G), dim3(512), 0, 0, buf, 128ull << 20, 512ull << 10, sink, steps); n += per_launch; });
        report("mall", r, n, "B");
    }
    {
        double n = 0;
        Res r = timed([&] { hipLaunchKernelGGL(k_load<2>, dim3(G), dim3(512), 0, 0, buf, big, 16ull << 20, sink, steps); n += per_launch; });
        report("hbm", r, n, "B");
    }
    // the parking pattern: store + immediate re-load of 64 KB slots; footprint 16 MB (L2), 128 MB (Infinity Cache), 1 GB (HBM)
    auto park = [&](const char *name, auto kern, int slots) {
        double n = 0;
        Res r = timed([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(512), 0, 0, slab, sink, steps, slots); n += 2.0 * G * 65536.0 * steps; });
        report(name, r, n, "B");
    };
    park("sl 16M", k_park<0, 2>, 1);
    park("sl 128M", k_park<0, 2>, 8);
    park("sl 1G", k_park<0, 2>, 64);
    park("128M 0/0", k_park<0, 0>, 8);
    park("128M nt/nt", k_park<2, 2>, 8);
    park("128M sc1/nt", k_park<16, 2>, 8);
    park("128M sc1/sc1", k_park<16, 16>, 8);
    park("128M s01/s01", k_park<17, 17>, 8);
    return 0;
}
