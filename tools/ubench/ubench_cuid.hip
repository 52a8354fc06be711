// ubench_cuid.hip -- is (HW_REG_XCC_ID, HW_REG_HW_ID) a collision-free CU slot index for a kernel that fits ONE workgroup per CU?
// Round 5 asked only "are the tuples of co-resident workgroups distinct" (yes).  Round 6 asks the question that matters: does a
// workgroup KEEP its tuple for its whole life?  HIP's own __smid() says "the results vary over time": a queue that is preempted
// (compute wave save/restore) resumes its workgroups wherever the dispatcher places them.
//
// 2,816 workgroups of 512 threads with 159 KB of LDS each (the paired blind rotation's footprint and grid) spin ~70 us x 2,816 / 256
// per launch; every workgroup records XCC id and HW id at its START and at its END, and start / end time.  Launches repeat for
// `seconds` (default 100).  The host prints, per launch that saw one: how many workgroups ended on another (se, sh, cu) than they
// started on, how many changed XCC, and whether two workgroups holding the same START tuple overlapped in time (= a collision of
// slots derived from the start tuple).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_cuid ubench_cuid.hip && ./ubench_cuid [seconds] [spin]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned xcc0, hw0, xcc1, hw1; unsigned long long t0, t1; };
__global__ __launch_bounds__(512, 1) void k(Rec *out, int spin)
{
    __shared__ double lds[159488 / 8];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned xcc0, hw0, xcc1, hw1;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc0));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw0));
    const unsigned long long t0 = wall_clock64();
    double acc = lds[threadIdx.x];
    for (int i = 0; i < spin; ++i) acc = __builtin_fma(acc, 1.0000001, 0.5);
    lds[threadIdx.x] = acc;
    __syncthreads();
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc1) :: "memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw1) :: "memory");
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{xcc0, hw0, xcc1, hw1, t0, t1 + (lds[1] == 12345.0)};
}
int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 100.0;
    const int spin = argc > 2 ? atoi(argv[2]) : 2000000;
    const int G = 2816;
    Rec *d; (void)hipMalloc((void **)&d, G * sizeof(Rec));
    std::vector<Rec> r(G);
    const auto T0 = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(); };
    int launches = 0, events = 0;
    bool first = true;
    while (now() < seconds) {
        const double ta = now();
        hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, d, spin);
        (void)hipDeviceSynchronize();
        const double tb = now();
        (void)hipMemcpy(r.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
        ++launches;
        int moved_cu = 0, moved_xcc = 0;
        for (auto &e : r) {
            if (((e.hw0 ^ e.hw1) >> 8) & 0xFF) ++moved_cu;
            if ((e.xcc0 ^ e.xcc1) & 15) ++moved_xcc;
        }
        // slots as the round-5 kernel derived them: XCC[2:0] | HW_ID[14:8] read at the start
        std::map<unsigned, std::vector<int>> slots;
        for (int i = 0; i < G; ++i) slots[((r[i].xcc0 & 7) << 7) | ((r[i].hw0 >> 8) & 0x7F)].push_back(i);
        int overlaps = 0;
        for (auto &kv : slots) {
            auto v = kv.second;
            std::sort(v.begin(), v.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
            for (size_t j = 1; j < v.size(); ++j) if (r[v[j]].t0 < r[v[j - 1]].t1) ++overlaps;
        }
        if (first) {
            unsigned or_hw = 0, and_hw = ~0u, or_x = 0, and_x = ~0u;
            for (auto &e : r) { or_hw |= e.hw0; and_hw &= e.hw0; or_x |= e.xcc0; and_x &= e.xcc0; }
            std::map<unsigned, int> cu_vals, se_vals;
            for (auto &e : r) { cu_vals[(e.hw0 >> 8) & 15]++; se_vals[(e.hw0 >> 12) & 15]++; }
            printf("launch 0: %.1f ms; HW_ID bits that vary: %08x   XCC_ID bits that vary: %08x; distinct start slots %zu for %d workgroups\n", 1e3 * (tb - ta), or_hw & ~and_hw, or_x & ~and_x, slots.size(), G);
            printf("CU_ID values:"); for (auto &kv : cu_vals) printf(" %u", kv.first);
            printf("\nSH/SE field values (HW_ID[15:12]):"); for (auto &kv : se_vals) printf(" %x", kv.first);
            printf("\n");
            first = false;
        }
        if (moved_cu || moved_xcc || overlaps) {
            ++events;
            printf("t=%.1f s launch %d (%.1f ms): %d workgroups ended on another (se,sh,cu) than they started on, %d on another XCC; %d pairs with the same start slot overlapped in time\n",
                   tb, launches - 1, 1e3 * (tb - ta), moved_cu, moved_xcc, overlaps);
            fflush(stdout);
        }
    }
    printf("%d launches in %.1f s, %d with migrated workgroups\n", launches, now(), events);
    return 0;
}
