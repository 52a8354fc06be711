// ubench_xcc.hip -- which XCD does a workgroup run on (HW_REG_XCC_ID), and how fast does a word written by one workgroup
// become visible to another one on the SAME XCD (plain store, sc0 load: through that XCD's L2) or on ANOTHER one.
// Background for the pacing of kern_blindrot16.h.   hipcc --offload-arch=gfx950 -O3 -o ubench_xcc ubench_xcc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x;
}
__device__ __forceinline__ uint32_t hw_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x));
    return x;
}

__global__ __launch_bounds__(256, 2) void where(uint32_t *out, uint32_t *cnt)
{
    __shared__ double pad[9000];        // 72 KB: two workgroups per CU
    pad[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        uint32_t x = xcc_id();
        out[3 * blockIdx.x] = x;
        out[3 * blockIdx.x + 1] = hw_id();
        out[3 * blockIdx.x + 2] = atomicAdd(cnt + (x & 7), 1u);
    }
    // keep the workgroup alive for a while so that all 512 are resident together
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 2000000ull) { __builtin_amdgcn_s_sleep(64); }
    if (pad[threadIdx.x] == 1.0) out[0] = 0;
}

// ping: workgroup `a` publishes step s (plain store), workgroup `b` polls with an sc0 load until it sees s and answers in its own word;
// returns cycles per round trip measured on a
#ifndef PEEK
#define PEEK(r, off) __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(0, r, off, 0, 0)      /* executes in this XCD's L2, returns the word */
#endif
__global__ __launch_bounds__(64) void ping(uint32_t *words, const uint32_t *xcd_of, uint32_t want_same, unsigned long long *cyc, uint32_t rounds, uint32_t *pair)
{
    __shared__ double pad[9000];
    pad[threadIdx.x] = 0;
    const uint32_t me = blockIdx.x;
    if (me != pair[0] && me != pair[1]) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(words, 0, 1024, 0x00020000);
    const bool a = me == pair[0];
    unsigned long long t0 = __builtin_readcyclecounter();
    uint32_t spins = 0;
    for (uint32_t s = 1; s <= rounds; ++s) {
        if (a) {
            __builtin_amdgcn_raw_buffer_store_b32(s, r, 0, 0, 0);
            while ((uint32_t)PEEK(r, 256) != s) { if (++spins > 2000000u) break; }
        } else {
            while ((uint32_t)PEEK(r, 0) != s) { if (++spins > 2000000u) break; }
            __builtin_amdgcn_raw_buffer_store_b32(s, r, 256, 0, 0);
        }
        if (spins > 2000000u) break;
    }
    if (a && threadIdx.x == 0) { cyc[0] = __builtin_readcyclecounter() - t0; cyc[1] = spins; }
    if (pad[threadIdx.x] == 1.0) words[100] = 0;
}

int main()
{
    const int G = 512;
    uint32_t *out, *cnt, *words, *pair;
    unsigned long long *cyc;
    hipMalloc(&out, G * 12); hipMalloc(&cnt, 64); hipMalloc(&words, 1024); hipMalloc(&cyc, 16); hipMalloc(&pair, 8);
    hipMemset(cnt, 0, 64);
    hipLaunchKernelGGL(where, dim3(G), dim3(256), 0, 0, out, cnt);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(3 * G), hc(16);
    hipMemcpy(h.data(), out, G * 12, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), cnt, 64, hipMemcpyDeviceToHost);
    printf("XCC_ID register of workgroups 0..23:");
    for (int i = 0; i < 24; ++i) printf(" %x", h[3 * i]);
    printf("\nworkgroups per (XCC_ID & 7):");
    for (int i = 0; i < 8; ++i) printf(" %u", hc[i]);
    printf("\nHW_ID of workgroups 0..7:");
    for (int i = 0; i < 8; ++i) printf(" %08x", h[3 * i + 1]);
    printf("\n");
    // a pair on the same XCD and a pair on different XCDs (64-thread workgroups of a second launch land elsewhere: look the ids up again)
    std::vector<uint32_t> xcd(G);
    for (int i = 0; i < G; ++i) xcd[i] = h[3 * i] & 7;
    for (int same = 1; same >= 0; --same) {
        uint32_t p[2] = {0, 0};
        for (int j = 1; j < G; ++j) if ((xcd[j] == xcd[0]) == (same == 1)) { p[1] = j; break; }
        hipMemcpy(pair, p, 8, hipMemcpyHostToDevice);
        hipMemset(words, 0, 1024);
        // same grid shape as `where` would be needed for the same placement; workgroup -> XCD is round-robin by index, so the ids hold
        hipLaunchKernelGGL(ping, dim3(G), dim3(64), 0, 0, words, (const uint32_t *)out, (uint32_t)same, cyc, 2000u, pair);
        hipDeviceSynchronize();
        unsigned long long c[2];
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        printf("ping-pong workgroups %u (xcd %u) <-> %u (xcd %u): %.0f cycles per round trip (plain store, L2 atomic read), %llu polls in 2000 rounds\n", p[0], xcd[p[0]], p[1],
               xcd[p[1]], (double)c[0] / 2000.0, c[1]);
    }
    return 0;
}
