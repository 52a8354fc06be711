// ubench_cuid.hip -- is (HW_REG_XCC_ID, HW_REG_HW_ID) a collision-free CU slot index for a kernel that fits ONE workgroup per CU?
// 2,560 workgroups of 512 threads with 159 KB of LDS each (the paired blind rotation's footprint) spin for a while and record their
// XCC id, HW id and start / end time; the host prints which HW_ID bits vary, how many distinct (xcc, se, sh, cu) tuples occur, and
// whether two workgroups with the same tuple ever overlapped in time.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_cuid ubench_cuid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned xcc, hw; unsigned long long t0, t1; };
__global__ __launch_bounds__(512, 1) void k(Rec *out, int spin)
{
    __shared__ double lds[159488 / 8];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long t0 = wall_clock64();
    double acc = lds[threadIdx.x];
    for (int i = 0; i < spin; ++i) acc = __builtin_fma(acc, 1.0000001, 0.5);
    lds[threadIdx.x] = acc;
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{xcc, hw, t0, t1 + (lds[1] == 12345.0)};
}
int main()
{
    const int G = 2560;
    Rec *d; (void)hipMalloc((void **)&d, G * sizeof(Rec));
    hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, d, 200000);
    (void)hipDeviceSynchronize();
    std::vector<Rec> r(G);
    (void)hipMemcpy(r.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned or_hw = 0, and_hw = ~0u, or_x = 0, and_x = ~0u;
    for (auto &e : r) { or_hw |= e.hw; and_hw &= e.hw; or_x |= e.xcc; and_x &= e.xcc; }
    printf("HW_ID bits that vary: %08x   XCC_ID bits that vary: %08x\n", or_hw & ~and_hw, or_x & ~and_x);
    // gfx9 layout: WAVE_ID[3:0] SIMD_ID[5:4] PIPE_ID[7:6] CU_ID[11:8] SH_ID[12] SE_ID[15:13]
    std::map<unsigned, std::vector<int>> slots;
    for (int i = 0; i < G; ++i) slots[((r[i].xcc & 15) << 8) | ((r[i].hw >> 8) & 0xFF)].push_back(i);
    printf("distinct (xcc, se, sh, cu) tuples: %zu for %d workgroups\n", slots.size(), G);
    int overlaps = 0; size_t maxper = 0; unsigned maxkey = 0;
    for (auto &kv : slots) {
        auto v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
        for (size_t j = 1; j < v.size(); ++j) if (r[v[j]].t0 < r[v[j - 1]].t1) ++overlaps;
        maxper = std::max(maxper, v.size());
        maxkey = std::max(maxkey, kv.first);
    }
    printf("same-tuple workgroups overlapping in time: %d   (most workgroups on one tuple: %zu; largest key %03x)\n", overlaps, maxper, maxkey);
    std::map<unsigned, int> cu_vals, se_vals;
    for (auto &e : r) { cu_vals[(e.hw >> 8) & 15]++; se_vals[(e.hw >> 12) & 15]++; }
    printf("CU_ID values:"); for (auto &kv : cu_vals) printf(" %u", kv.first); printf("\nSH/SE field values:"); for (auto &kv : se_vals) printf(" %x", kv.first); printf("\n");
    return 0;
}
