// kern_blindrot_pair.h -- K2, blind rotation + sample extract, throughput form of round 4:
// ONE 512-thread workgroup per CU carries 2 x R ciphertexts (R = 3: six, R = 2: four).  Same arithmetic and lane mapping as
// kern_blindrot16.h (bit-identical results); what changes is who shares what.
//
// Why (round 4, tools/gpu_power.py + tools/ubench/ubench_energy.hip): the kernel runs at the socket's power cap (1,335 W of 1,400),
// so a launch takes (joules per launch) / (watts), not (cycles) / (clock) -- and 25 % of the joules were operand movement:
// the Fourier key rows pulled into every CU TWICE per iteration (once per workgroup of kern_blindrot16.h's two), the accumulator parked
// in memory between its two uses (39 J of 305 per launch), and key rows re-fetched from HBM because the 512 resident workgroups
// spread over many iterations.  This form attacks all three:
//   * Two halves of four wavefronts each do what a kern_blindrot16.h workgroup does for the transforms (half h: ciphertexts
//     h R .. h R + R - 1, group g of the half owns polynomial g % K1 of its ciphertext g / K1), but the multiply-accumulate is shared:
//     thread (h, t) owns Fourier point t and the OUTPUT COLUMNS {2h, 2h + 1} of all 2R ciphertexts plus column 4 of its own half's R.
//     A thread therefore needs 15 of the 25 GGSW entries of a level, every entry it fetches serves 2R (or R) ciphertexts, and the
//     CU pulls 60 % of the bytes through its vector-memory path that two independent workgroups pull (the column-4 rows twice, the
//     second time out of L1).  It reads twice the digits from LDS instead (3.6 pJ/B against 12 pJ/B for an L2 hit, 39-100 beyond).
//   * 15 entries in flight instead of 25 frees 40 VGPRs: the upper halves hi[] of the accumulator stay in registers (BRP_RESIDENT_HI),
//     only lo[] is parked -- and the wavefronts 3 and 7 park nothing at all: as in kern_blindrot16.h's HOME form, group 15 of a half
//     mirrors group 14 (same tile, same values), and the LDS that leaves free (2 tiles + what one workgroup per CU does not use) is
//     the permanent home of the six accumulators of those two wavefronts.
//   * One workgroup per CU, all alike, all started together: no "older workgroup wins the SIMD arbiter" asymmetry (kern_blindrot16.h's
//     two workgroups of a CU run at 15 and 33 ms per unit), so the chip's workgroups stay within a few iterations of each other and a
//     GGSW fetched into an XCD's L2 serves that XCD's 32 workgroups before it is evicted.
#pragma once
#include "fft_dev.h"
#include "kern_extprod.h"
#include "kern_blindrot16.h"

#define BRP_THREADS 512
#ifndef BRP_EARLY
#define BRP_EARLY 9          /* GGSW entries (of 15 per thread and level) requested between the instructions of the transform's second half
                                (6 / 8 / 9 / 10 / 12: 218.6 / 215.4 / 215.1 / 214.1 / 216.8 ms per 16,384-bit launch; 12 spills 4 registers) */
#endif
#ifndef BRP_TAIL
#define BRP_TAIL 4           /* ... requested only after the multiply-accumulate has used row 0, into the registers that row has left
                                (0 / 2 / 3 / 4 / 5 / 6 on one box: 214.4 / 210.1 / 211.3 / 209.8 / 210.5 / 213.2 ms; on a slower one 220.4 / 218.8 for 0 / 4) */
#endif
#ifndef BRP_RESIDENT_HI
#define BRP_RESIDENT_HI 1
#endif
#ifndef BRP_W1_LATE
#define BRP_W1_LATE 2           /* 1: second half of the table column requested at the start of the twiddle pass; 2: in two requests of four entries, each
                                into registers the first half has just left (no spill with hi[] resident: 249 VGPRs) */
#endif
// Parking slots (round 6).  The parked half of the accumulator lives in a slab of 64 KB slots.  A workgroup's slot is either
//   * CLAIMED (ExtProdArgs::park_owner != null, the default): the first BRP_PARK_SLOTS = 8 x 128 slots are shared by all generations of
//     all launches -- 128 per XCC (HW_REG_XCC_ID: only that XCC's L2 ever caches a slot's lines, so a slot never needs a cross-XCC
//     hand-over), taken with one compare-and-swap on an owner word at the workgroup's start (the probe starts at the slot its
//     predecessor on this compute unit has just released: HW_REG_HW_ID[14:8] is a HINT, nothing depends on its value) and given back
//     after the workgroup's last parking access has been acknowledged.  The 16 MB the 256 resident workgroups touch stay in the L2s /
//     the Infinity Cache instead of 180 MB per 16,384-bit launch streaming through them (round 5: 203.9 -> 199.2 ms per launch).
//     A workgroup that finds its XCC's 128 slots taken (never observed; it would take four times the resident workgroups)
//     falls back to the private slot BRP_PARK_SLOTS + blockIdx.x behind them;
//     A compute unit's L1 cannot serve a stale line of a re-used slot: a lane reads back only what it has itself stored since it
//     owns the slot (stores update or invalidate the line), and a workgroup streams ~600 KB of key rows per iteration through the
//     32 KB L1, so nothing an earlier owner left there survives even one iteration (27 us), let alone a hand-over;
//   * or PRIVATE (park_owner == null): slot = blockIdx.x, 64 KB per workgroup of the launch (the round-4 layout).
// Round 5 DERIVED the slot from the hardware id registers read once at the start ("this kernel fits one workgroup per CU, so the
// physical CU is a collision-free index").  That is wrong whenever the queue is preempted mid-kernel (compute wave save / restore):
// the workgroups resume on whatever CU the dispatcher gives them (HIP's own __smid(): "the results vary over time"), a resumed
// workgroup keeps the slot of the CU it STARTED on, and the next workgroup dispatched to that CU derives the same slot -- two live
// accumulators in one slot.  On the MI355X boxes of this pool such a preemption happens every ~30 s of sustained load; it corrupted
// the parked halves of 200-260 rows of one launch each time (profiles/r06_park_collision.txt).  Ownership is
// now a fact recorded in memory, not an inference from where a wave happens to run.
#define BRP_PARK_SLOTS 1024
#define BRP_SLOTS_PER_XCC 128
#ifndef BRP_MAC_PRIO
#define BRP_MAC_PRIO 1       /* wave priority during the multiply-accumulate (0 / 1 / 3: 214.2 / 211.8 / 212.0 ms per 16,384-bit launch) */
#endif
#ifndef BRP_CHUNK
#define BRP_CHUNK 1          /* butterflies issued together in this kernel's transforms (fft_dev.h dft16; 1 / 2 / 4 / 8: 211.5 / 214.4 / 214.2 / 223.4 ms) */
#endif
#ifndef BRP_SKIP_IDLE_WAVES
#define BRP_SKIP_IDLE_WAVES 1 /* four-ciphertext units (R = 2): the last wavefront of either half (lane groups 12-15) owns no polynomial -- it is needed for
                                the multiply-accumulate (every thread owns a Fourier point) and for the products exchange, but its rotation, transforms
                                and conversion work on nothing.  1: those two wavefronts run their OWN iteration body (the same barriers, their key
                                rows, the multiply-accumulate, the exchange; a wave-uniform branch at the top of the iteration) -- one generation of
                                eleven in a 16,384-bit launch, one of three in a 4,096-bit one.  The six-ciphertext body is not touched (its one idle
                                group shares a wavefront with three busy ones) */
#endif
#define BRP_HALF_TILES (EP_GROUPS - 1)                                                   /* 15 tiles per half: group 15 shares group 14's */
#define BRP_LDS_DOUBLES(R) (2 * FHE_TW_ENTRIES + 2 * BRP_HALF_TILES * GROUP_TILE_DOUBLES + ((R) == 3 ? 6 * FHE_N : 0))   /* R = 3: 159,488 B */
#define BRP_LDS_EXTRA_DOUBLES 2                                                          /* + the claimed parking slot, broadcast to the eight wavefronts */
#define BRP_PARK_WORDS_PER_HALF (BRP_RESIDENT_HI ? 8 * EP_THREADS * 2 : 16 * EP_THREADS * 2)   /* per half and iteration: 32 KB (lo[] only) or 64 KB */

__device__ __forceinline__ int brp_opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// One unit = 2R ciphertexts starting at inst0, one 512-thread workgroup.
template <int K1, int LEVELS, int BASE_LOG, int R>
__device__ __forceinline__ void blind_rotate_pair_unit(const ExtProdArgs &A, double *lds_all, unsigned *slot_word, const uint64_t inst0, const unsigned unit_index)
{
    static_assert(K1 == 5, "the column split {0,1} / {2,3} / shared 4 is written for k + 1 = 5");
    static_assert(R * K1 < EP_GROUPS, "a half needs at least one idle lane group");
    constexpr int RT = 2 * R;                                   // ciphertexts of the unit
    constexpr int NQ = 3 * K1;                                  // GGSW entries a thread needs per level: K1 rows x (2 own columns + column 4)
    constexpr bool HOME = R == 3;                               // wavefronts 3 and 7 keep their accumulators in LDS
    constexpr int LAST_T = EP_GROUPS - 2, HOME_G0 = 12;         // group 15 of a half shares tile 14; first group of the half's last wavefront

    double2 *tw = reinterpret_cast<double2 *>(lds_all);
    double *lds = lds_all + 2 * FHE_TW_ENTRIES;                 // half 0's 15 tiles, half 1's 15 tiles, then the six homes

    const int tid = threadIdx.x;
    const int hh = __builtin_amdgcn_readfirstlane(tid >> 8);    // wave-uniform: which half
    const int tq0 = tid & 255;
    const int g = tq0 >> 4, b = tq0 & 15;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    const bool home_wave = HOME && __builtin_amdgcn_readfirstlane(tq0) >= 16 * HOME_G0;
    // wave-uniform: this wavefront's four lane groups own no polynomial (R = 2: groups 12..15 of either half)
    const bool idle_wave = BRP_SKIP_IDLE_WAVES && R * K1 <= 12 && __builtin_amdgcn_readfirstlane(tq0) >= 16 * 12;
    double *ldsh = lds + hh * (BRP_HALF_TILES * GROUP_TILE_DOUBLES);                      // this half's tiles (scalar)
    double *ldso = lds + (1 - hh) * (BRP_HALF_TILES * GROUP_TILE_DOUBLES);                // the other half's
    auto tile_of = [&](const int tq) -> double * {
        int gq = tq >> 4;
        gq = gq < LAST_T ? gq : LAST_T;
        return ldsh + gq * GROUP_TILE_DOUBLES;
    };
    auto stage_of = [&](const int tq) -> uint64_t * {
        int gq = tq >> 4;
        gq = gq < LAST_T ? gq : LAST_T;
        if (HOME) {
            const int tile_words = hh * (BRP_HALF_TILES * GROUP_TILE_DOUBLES) + gq * GROUP_TILE_DOUBLES;
            const int home_words = 2 * BRP_HALF_TILES * GROUP_TILE_DOUBLES + (hh * 3 + gq - HOME_G0) * FHE_N;
            return reinterpret_cast<uint64_t *>(lds) + (gq >= HOME_G0 ? home_words : tile_words);
        }
        return reinterpret_cast<uint64_t *>(ldsh + gq * GROUP_TILE_DOUBLES);
    };
    // byte offset of a lane in a parking chunk; lanes that park nothing (idle groups; the home wavefronts) are out of range
    auto park_lane = [&](const int tq) -> unsigned {
        const int gq = tq >> 4;
        return gq < (HOME ? HOME_G0 : R * K1) ? (unsigned)tq * 16u : 0x80000000u;
    };

    ep_load_table(tw, A.tw);
    uint64_t inst = inst0 + (uint64_t)hh * R + r_own;
    const bool valid = inst < A.count;
    if (!valid) inst = A.count - 1;
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    // ---- accumulator init: NEGATED accumulator, coefficients 16a+b and 256+16a+b in lane b (kern_blindrot16.h) -------------------
    uint64_t lo[16], hi[16];
    {
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            int j0 = 16 * a + b, j1 = j0 + 256;
            int e0 = ((j0 - t) & 511) + t, e1 = ((j1 - t) & 511) + t;
            uint64_t v0 = ((e0 >> 9) & 1) ? A.tv_const : (uint64_t)0 - A.tv_const;
            uint64_t v1 = ((e1 >> 9) & 1) ? A.tv_const : (uint64_t)0 - A.tv_const;
            lo[a] = (p_own == K1 - 1) ? v0 : 0;
            hi[a] = (p_own == K1 - 1) ? v1 : 0;
        }
    }
    const __amdgpu_buffer_rsrc_t park_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.park, 0, (int)A.park_bytes, 0x00020000);
    // ---- parking slot: claimed (see the top of this file) or private ----------------------------------------------------------------
    if (A.park_owner && tid == 0) {
        unsigned hw_xcc, hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw_xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        const unsigned base = (hw_xcc & 7u) * BRP_SLOTS_PER_XCC, hint = (hw_id >> 8) & (BRP_SLOTS_PER_XCC - 1);
        unsigned got = BRP_PARK_SLOTS + unit_index;                   // private slot behind the shared ones: only if the XCC's 128 are all owned
        for (unsigned i = 0; i < BRP_SLOTS_PER_XCC; ++i) {
            const unsigned sidx = base + ((hint + i) & (BRP_SLOTS_PER_XCC - 1));
            unsigned expected = 0;
            // relaxed: the previous owner released the slot only after its last parking access was acknowledged (below), and this
            // workgroup's first parking store is issued after the barrier that publishes `got`
            if (__hip_atomic_compare_exchange_strong(A.park_owner + sidx, &expected, unit_index + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                got = sidx;
                break;
            }
        }
        *slot_word = got;
    }
#define BRP_PARK_SLOT(a) ((unsigned)(a) * (EP_THREADS * 16))
    __syncthreads();   // tables and the claimed slot visible
    const unsigned park_slot = A.park_owner ? (unsigned)__builtin_amdgcn_readfirstlane((int)*slot_word) : unit_index;
    const unsigned park_wg = (park_slot * 2u + (unsigned)hh) * (unsigned)(BRP_PARK_WORDS_PER_HALF * 8);       // wave-uniform

    constexpr unsigned GGSW_BYTES = LEVELS * K1 * K1 * FHE_H * 16;
    const __amdgpu_buffer_rsrc_t bsk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(A.ggsw), 0, (int)(A.iters * GGSW_BYTES), 0x00020000);
    const unsigned col_bytes = (unsigned)(2 * hh) * (FHE_H * 16);     // scalar: this half's first output column
    uint64_t a_next = lwe[0];

#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    // coefficient pair a of the (negated) accumulator -> where the next rotation reads it (the group's tile, or its LDS home) and,
    // for the parked groups, lo[] -> the parking slab, two coefficients per 16-byte chunk (hi[] stays in registers)
    auto stage_park = [&](const int a, const int tq) {
        uint64_t *stage = stage_of(tq);
        stage[16 * a + (tq & 15)] = lo[a];
        stage[256 + 16 * a + (tq & 15)] = hi[a];
#if BRP_RESIDENT_HI
        if (a & 1) {
            ep_u32x4 v;
            v[0] = (uint32_t)lo[a - 1]; v[1] = (uint32_t)(lo[a - 1] >> 32); v[2] = (uint32_t)lo[a]; v[3] = (uint32_t)(lo[a] >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(v, park_rsrc, park_lane(tq), park_wg + BRP_PARK_SLOT(a >> 1), BR16_PARK_AUX_ST);
        }
#else
        {
            ep_u32x4 v;
            v[0] = (uint32_t)lo[a]; v[1] = (uint32_t)(lo[a] >> 32); v[2] = (uint32_t)hi[a]; v[3] = (uint32_t)(hi[a] >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(v, park_rsrc, park_lane(tq), park_wg + BRP_PARK_SLOT(a), BR16_PARK_AUX_ST);
        }
#endif
    };
    if (!idle_wave) {
        const int tq = brp_opaque_tid() & 255;
#pragma unroll
        for (int a = 0; a < 16; ++a) stage_park(a, tq);
    }
#ifdef BRP_ABL_SKEW
    // timing proxy (with BRP_ABL_NOBAR: no barrier is left in the loop): half 1 starts BRP_ABL_SKEW x 64 cycles late and the halves free-run
    if (hh) for (int i = 0; i < BRP_ABL_SKEW; i += 100) __builtin_amdgcn_s_sleep(100);
#endif
    for (uint32_t it = 0; it < A.iters; ++it) {
        const int t = mod_switch_1024(a_next);
        a_next = lwe[it + 1];
        const unsigned g_bytes = it * GGSW_BYTES;

        if (idle_wave) {
            // ---- a wavefront without polynomials (BRP_SKIP_IDLE_WAVES): the barriers of the others, its key rows, its share of the
            //      multiply-accumulate and of the products exchange -- nothing else.  Same sums in the same order as level_body below ----
            const int tq = brp_opaque_tid() & 255;
            double g2r[RT][2], g2i[RT][2], g4r[R], g4i[R];
#pragma unroll
            for (int r = 0; r < RT; ++r) { g2r[r][0] = g2r[r][1] = 0.0; g2i[r][0] = g2i[r][1] = 0.0; }
#pragma unroll
            for (int r = 0; r < R; ++r) { g4r[r] = 0.0; g4i[r] = 0.0; }
#pragma unroll 1
            for (int l = LEVELS - 1; l >= 0; --l) {
                const unsigned gl_bytes = g_bytes + (unsigned)l * (K1 * K1 * FHE_H * 16) + col_bytes;   // scalar
                double2 bm[K1][3];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int p = q / 3, j = q % 3;
                    const unsigned off = j < 2 ? gl_bytes + (unsigned)(p * K1 + j) * (FHE_H * 16)
                                               : gl_bytes - col_bytes + (unsigned)(p * K1 + K1 - 1) * (FHE_H * 16);
                    bm[p][j] = ep_key_load(bsk_rsrc, (unsigned)tq * 16u, off);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (l != LEVELS - 1) wg_barrier_lds_only();      // the others: "done reading the previous level's digits"
                wg_barrier_lds_only();                            // the digits of this level are visible
                const double *own = ldsh + 2 * tq, *oth = ldso + 2 * tq;
#pragma unroll
                for (int p = 0; p < K1; ++p) {
                    double2 d[RT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) d[r] = *reinterpret_cast<const double2 *>((r < R ? own : oth) + ((r % R) * K1 + p) * GROUP_TILE_DOUBLES);
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            g2r[r][j] = __builtin_fma(d[r].x, bm[p][j].x, g2r[r][j]);
                            g2r[r][j] = __builtin_fma(-d[r].y, bm[p][j].y, g2r[r][j]);
                            g2i[r][j] = __builtin_fma(d[r].x, bm[p][j].y, g2i[r][j]);
                            g2i[r][j] = __builtin_fma(d[r].y, bm[p][j].x, g2i[r][j]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        g4r[r] = __builtin_fma(d[r].x, bm[p][2].x, g4r[r]);
                        g4r[r] = __builtin_fma(-d[r].y, bm[p][2].y, g4r[r]);
                        g4i[r] = __builtin_fma(d[r].x, bm[p][2].y, g4i[r]);
                        g4i[r] = __builtin_fma(d[r].y, bm[p][2].x, g4i[r]);
                    }
                }
            }
            wg_barrier_lds_only();                                // everybody is done reading the last level's digits
            {
                double *own = ldsh + 2 * tq, *oth = ldso + 2 * tq;
                const int c0 = 2 * hh;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double2 v; v.x = g2r[r][j]; v.y = g2i[r][j];
                        *reinterpret_cast<double2 *>((r < R ? own : oth) + ((r % R) * K1 + c0 + j) * GROUP_TILE_DOUBLES) = v;
                    }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double2 v; v.x = g4r[r]; v.y = g4i[r];
                    *reinterpret_cast<double2 *>(own + (r * K1 + K1 - 1) * GROUP_TILE_DOUBLES) = v;
                }
            }
            wg_barrier_lds_only();                                // products in place
            continue;
        }
        // ---- d = acc * X^t - acc; first (least significant) digit ---------------------------------------------------------------
        uint32_t st_lo[16], st_hi[16];
        static_assert(BASE_LOG == 8 && LEVELS == 5, "the signed-byte decomposition state is written for five levels of eight bits");
        double xr[16], xi[16];
        double2 w0[8], w1[8];
        EP_STAMP(11);
        {
            const int tq = brp_opaque_tid() & 255;
            const int bq_ = tq & 15;
            uint64_t *stage = stage_of(tq);
            wave_lds_sync();
            fft_tw_load8(w0, tw, bq_, FHE_TW_STRIDE);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int u0 = 16 * a + bq_ - t, u1 = u0 + 256;
                const int s0 = u0 & 511, s1 = s0 ^ 256;
                const fhe_u32x2 v0 = *reinterpret_cast<const fhe_u32x2 *>(stage + s0), v1 = *reinterpret_cast<const fhe_u32x2 *>(stage + s1);
                const uint32_t m0 = (uint32_t)((u0 >> 9) & 1) - 1u, m1 = (uint32_t)((u1 >> 9) & 1) - 1u;
                constexpr uint64_t RND = 1ull << (64 - BASE_LOG * LEVELS - 1);
                static_assert(RND < (1ull << 31), "rounding constant must fit the low word");
                fhe_u32x2 x0w, x1w, k0, k1;
                x0w[0] = v0[0] ^ m0; x0w[1] = v0[1] ^ m0; x1w[0] = v1[0] ^ m1; x1w[1] = v1[1] ^ m1;
                // one 64-bit constant per coefficient carries the two's-complement "+1" (1 - ... = -m), the rounding 2^23 AND the
                // decomposition's offset (fft_dev.h decompose_offset<8, 5>: 0x80 in each of the five digit bytes): no sum overflows a word
                constexpr uint64_t OFF = decompose_offset<BASE_LOG, LEVELS>();
                static_assert((uint32_t)OFF + RND < (1ull << 32), "offset + rounding constant must fit the low word");
                k0[0] = (uint32_t)OFF + (uint32_t)RND - m0; k0[1] = (uint32_t)(OFF >> 32); k1[0] = (uint32_t)OFF + (uint32_t)RND - m1; k1[1] = (uint32_t)(OFF >> 32);
                const uint64_t x0 = (__builtin_bit_cast(uint64_t, x0w) + lo[a]) + __builtin_bit_cast(uint64_t, k0);
                const uint64_t x1 = (__builtin_bit_cast(uint64_t, x1w) + hi[a]) + __builtin_bit_cast(uint64_t, k1);
                xr[a] = (double)decompose8x5_first_z(x0, st_lo[a]);
                xi[a] = (double)decompose8x5_first_z(x1, st_hi[a]);
                if ((a & (EP_ROT_CHUNK - 1)) == EP_ROT_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_sync();
        }
        EP_STAMP(0);

        // sums: local ciphertext r' = 0..RT-1 is ciphertext (r' + R hh) mod RT of the unit, so r' < R are this half's own;
        // f2: output columns 2 hh, 2 hh + 1 of all RT; f4: column 4 of the own R
        double f2r[RT][2], f2i[RT][2], f4r[R], f4i[R];
#pragma unroll
        for (int r = 0; r < RT; ++r) { f2r[r][0] = f2r[r][1] = 0.0; f2i[r][0] = f2i[r][1] = 0.0; }
#pragma unroll
        for (int r = 0; r < R; ++r) { f4r[r] = 0.0; f4i[r] = 0.0; }

#if BRP_RESIDENT_HI
        uint64_t pkl[16];
#else
        ulonglong2 pk[16];
#endif
        auto level_body = [&](const int l, const bool tiles_busy, auto last) {
            const int tq = brp_opaque_tid() & 255;
            const int bq_ = tq & 15;
            double *tile = tile_of(tq);
            const unsigned gl_bytes = g_bytes + (unsigned)l * (K1 * K1 * FHE_H * 16) + col_bytes;   // scalar
            double2 bm[K1][3];                                    // [row][own column 0, own column 1, column 4]
            // entries [from, to) of this thread's NQ = 15 (row-major over [row][3])
            auto key_rows = [&](const int from, const int to) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q < from || q >= to) continue;
                    const int p = q / 3, j = q % 3;
                    // columns 2hh + j (j < 2): col_bytes is in gl_bytes; column 4: undo it
                    const unsigned off = j < 2 ? gl_bytes + (unsigned)(p * K1 + j) * (FHE_H * 16)
                                               : gl_bytes - col_bytes + (unsigned)(p * K1 + K1 - 1) * (FHE_H * 16);
#ifdef BR16_ABL_NOLOAD
                    bm[p][j] = make_double2((double)(tq + q), (double)(tq - q));
#else
                    bm[p][j] = ep_key_load(bsk_rsrc, (unsigned)tq * 16u, off);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#if !BRP_W1_LATE
            fft_tw_load8(w1, tw, 8 * FHE_TW_STRIDE + bq_, FHE_TW_STRIDE);
#endif
            __builtin_amdgcn_sched_barrier(0);
            dft16<false, true, BRP_CHUNK>(xr, xi);
            __builtin_amdgcn_sched_barrier(0);
            EP_STAMP(2);
            constexpr int NE = BRP_EARLY, NHOOK = 7, NT = BRP_TAIL;
            auto early = [&](const int h) { key_rows(NE * h / NHOOK, NE * (h + 1) / NHOOK); };
#ifndef BRP_ABL_NOBAR
            if (tiles_busy) wg_barrier_lds_only();                // every thread of BOTH halves is done reading the previous level's digits
#endif
            EP_STAMP(3);
            {
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(FFT_XPOSE_PRIO);
#endif
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
#if BRP_W1_LATE == 1
                    if (k1 == 0) { fft_tw_load8(w1, tw, 8 * FHE_TW_STRIDE + bq_, FHE_TW_STRIDE); __builtin_amdgcn_sched_barrier(0); }
#elif BRP_W1_LATE == 2
                    // the second half of the table column in two requests of four entries, each into registers the first half has just left
                    if (k1 == 3 || k1 == 7) {
                        const int e0 = k1 == 3 ? 0 : 4;
#pragma unroll
                        for (int e = e0; e < e0 + 4; ++e) w1[e] = tw[(8 + e) * FHE_TW_STRIDE + bq_];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
#ifdef BRP_ABL_FEWCMUL
                    if (k1 % 3 != 0 || k1 == 0)      // timing proxy: 5 of the 16 twiddle multiplies (20 of a transform's 404 f64 instructions) left out
#endif
                    {
                        if (k1 < 8) cmul(xr[k1], xi[k1], w0[k1].x, w0[k1].y); else cmul(xr[k1], xi[k1], w1[k1 - 8].x, w1[k1 - 8].y);
                    }
                    double2 v; v.x = xr[k1]; v.y = xi[k1];
#ifndef BRP_ABL_NOXSTORE
                    *reinterpret_cast<double2 *>(tile + 2 * (k1 * 17 + bq_)) = v;
#else
                    asm volatile("" :: "v"(v.x), "v"(v.y));      // timing proxy: the values are computed, the store is not issued
#endif
                    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    if (NE && k1 == 7) { __builtin_amdgcn_sched_barrier(0); early(0); }
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(1); }
                wave_lds_sync();
#ifndef BRP_ABL_NOXREAD
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = fft_reg(q);
                    double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
                    xr[c] = v.x; xi[c] = v.y;
                }
#endif
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(2); }
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                dft16<false, false, BRP_CHUNK>(xr, xi, [&](const int stage) { if (NE) { __builtin_amdgcn_sched_barrier(0); early(3 + stage); } },
                                    [&](const int stage, const int c0) {
                                        if (stage != 3) return;
                                        if (c0 == 0) wave_lds_sync();
#pragma unroll
                                        for (int j = 0; j < BRP_CHUNK; ++j) {
#pragma unroll
                                            for (int h = 0; h < 2; ++h) {
                                                const int k2 = c0 + j + 8 * h;
                                                double2 v; v.x = xr[fft_reg(k2)]; v.y = xi[fft_reg(k2)];
#ifndef BRP_ABL_NODSTORE
                                                *reinterpret_cast<double2 *>(tile + 2 * (bq_ + 16 * k2)) = v;
#else
                                                asm volatile("" :: "v"(v.x), "v"(v.y));
#endif
                                            }
                                        }
                                        __builtin_amdgcn_sched_barrier(0);
                                        {
                                            constexpr int NL = NQ - NE - NT, PARTS = 8 / BRP_CHUNK;
                                            const int part = c0 / BRP_CHUNK;
                                            key_rows(NE + NL * part / PARTS, NE + NL * (part + 1) / PARTS);
                                        }
                                    });
            }
            EP_STAMP(4);
            __builtin_amdgcn_sched_barrier(0);
            EP_STAMP(5);
#ifndef BRP_ABL_NOBAR
            wg_barrier_lds_only();                                // the digits of all 2R ciphertexts are visible; key loads stay in flight
#endif
            EP_STAMP(6);
            // ---- multiply-accumulate: thread (hh, tq) owns Fourier point tq; digits of row p for the RT ciphertexts, one row ahead ----
            // local ciphertext r': r' < R in this half's tiles, r' >= R in the other half's (tile (r' mod R) * K1 + p of that half)
#if BRP_MAC_PRIO
            __builtin_amdgcn_s_setprio(BRP_MAC_PRIO);
#endif
            const double *own = ldsh + 2 * tq, *oth = ldso + 2 * tq;
            double2 dn[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) dn[r] = *reinterpret_cast<const double2 *>((r < R ? own : oth) + ((r % R) * K1) * GROUP_TILE_DOUBLES);
#pragma unroll
            for (int p = 0; p < K1; ++p) {
                double2 d[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) d[r] = dn[r];
                if (p + 1 < K1) {
#pragma unroll
                    for (int r = 0; r < RT; ++r) dn[r] = *reinterpret_cast<const double2 *>((r < R ? own : oth) + ((r % R) * K1 + p + 1) * GROUP_TILE_DOUBLES);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef BR16_ABL_NOMAC
                if (p == 0)
#endif
                {
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f2r[r][j] = __builtin_fma(d[r].x, bm[p][j].x, f2r[r][j]);
                            f2r[r][j] = __builtin_fma(-d[r].y, bm[p][j].y, f2r[r][j]);
                            f2i[r][j] = __builtin_fma(d[r].x, bm[p][j].y, f2i[r][j]);
                            f2i[r][j] = __builtin_fma(d[r].y, bm[p][j].x, f2i[r][j]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f4r[r] = __builtin_fma(d[r].x, bm[p][2].x, f4r[r]);
                        f4r[r] = __builtin_fma(-d[r].y, bm[p][2].y, f4r[r]);
                        f4i[r] = __builtin_fma(d[r].x, bm[p][2].y, f4i[r]);
                        f4i[r] = __builtin_fma(d[r].y, bm[p][2].x, f4i[r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (NT && p == 0) key_rows(NQ - NT, NQ);
                if constexpr (decltype(last)::value) {
                    // parked accumulator back: lands during the products exchange and the inverse transform
#if BRP_RESIDENT_HI
#pragma unroll
                    for (int j = 8 * p / K1; j < 8 * (p + 1) / K1; ++j) {
#ifdef BR16_ABL_NOPARK
                        pkl[2 * j] = 0; pkl[2 * j + 1] = 0;
#else
                        const ep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(park_rsrc, park_lane(tq), park_wg + BRP_PARK_SLOT(j), BR16_PARK_AUX_LD);
                        pkl[2 * j] = ((unsigned long long)v[1] << 32) | v[0];
                        pkl[2 * j + 1] = ((unsigned long long)v[3] << 32) | v[2];
#endif
                    }
#else
#pragma unroll
                    for (int a = 16 * p / K1; a < 16 * (p + 1) / K1; ++a) {
                        const ep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(park_rsrc, park_lane(tq), park_wg + BRP_PARK_SLOT(a), BR16_PARK_AUX_LD);
                        pk[a].x = ((unsigned long long)v[1] << 32) | v[0];
                        pk[a].y = ((unsigned long long)v[3] << 32) | v[2];
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if BRP_MAC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            EP_STAMP(7);
        };

#pragma unroll 1
        for (int l = LEVELS - 1; l >= 1; --l) {
            level_body(l, l != LEVELS - 1, std::false_type{});
            {
                const int tq = brp_opaque_tid() & 255;
                fft_tw_load8(w0, tw, tq & 15, FHE_TW_STRIDE);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef BRP_ABL_NOPEEL
#pragma unroll
            for (int a = 0; a < 16; ++a) { xr[a] = (double)(int)st_lo[a]; xi[a] = (double)(int)st_hi[a]; }
#else
            // the digits of level l - 1: signed byte LEVELS - 1 - l of the state (one bit-field extract at a wave-uniform offset each)
            const unsigned dbit = 8u * (unsigned)(LEVELS - 1 - l);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                xr[a] = (double)decompose8x5_at(st_lo[a], dbit);
                xi[a] = (double)decompose8x5_at(st_hi[a], dbit);
            }
#endif
            EP_STAMP(1);
        }
        level_body(0, LEVELS > 1, std::true_type{});

        const int tq = brp_opaque_tid() & 255;
        const int bq_ = tq & 15;
        double *tile = tile_of(tq);
        // ---- products back to the owning groups: column c of local ciphertext r' goes to tile (r' mod R) * K1 + c of its half ------
#if !(defined(BRP_ABL_NOBAR) && defined(BRP_ABL_SKEW))
        wg_barrier_lds_only();
#endif
        {
            double *own = ldsh + 2 * tq, *oth = ldso + 2 * tq;
            const int c0 = 2 * hh;                                 // scalar
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double2 v; v.x = f2r[r][j]; v.y = f2i[r][j];
                    *reinterpret_cast<double2 *>((r < R ? own : oth) + ((r % R) * K1 + c0 + j) * GROUP_TILE_DOUBLES) = v;
                }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double2 v; v.x = f4r[r]; v.y = f4i[r];
                *reinterpret_cast<double2 *>(own + (r * K1 + K1 - 1) * GROUP_TILE_DOUBLES) = v;
            }
        }
#if !(defined(BRP_ABL_NOBAR) && defined(BRP_ABL_SKEW))
        wg_barrier_lds_only();
#endif
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ + 16 * k2));
            xr[k2] = v.x; xi[k2] = v.y;
        }
        fft_inv_table(w0, w1, tw, bq_);
        wave_lds_sync();
        EP_STAMP(8);
        dft16<true, false, BRP_CHUNK>(xr, xi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c >= 1) { if (c < 8) cmulc(xr[c], xi[c], w0[c].x, w0[c].y); else cmulc(xr[c], xi[c], w1[c - 8].x, w1[c - 8].y); }
            double2 v; v.x = xr[c]; v.y = xi[c];
            *reinterpret_cast<double2 *>(tile + 2 * (c * 17 + bq_)) = v;
            if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = fft_reg(q);
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
            xr[c] = v.x; xi[c] = v.y;
        }
        if (home_wave) {
            // a home wavefront: the old accumulator (the lane's own coefficients) comes from its LDS home
            const uint64_t *home = stage_of(tq);
#pragma unroll
#if BRP_RESIDENT_HI
            for (int a = 0; a < 16; ++a) pkl[a] = home[16 * a + bq_];
#else
            for (int a = 0; a < 16; ++a) { pk[a].x = home[16 * a + bq_]; pk[a].y = home[256 + 16 * a + bq_]; }
#endif
        }
        dft16<true, false, BRP_CHUNK>(xr, xi);
        // conj psi^(16a) with the back-conversion's 2^-72 folded into the constants (exact: a power of two commutes with the roundings)
        xr[0] *= 0x1p-72; xi[0] *= 0x1p-72;
#pragma unroll
        for (int a = 1; a < 16; ++a) cmulc(xr[a], xi[a], FHE_PSI16_RE[a] * 0x1p-72, FHE_PSI16_IM[a] * 0x1p-72);
        EP_STAMP(9);
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#if BRP_RESIDENT_HI
            lo[a] = torus_acc_scaled(pkl[a], -xr[a]);
            hi[a] = torus_acc_scaled(hi[a], -xi[a]);
#else
            lo[a] = torus_acc_scaled(pk[a].x, -xr[a]);
            hi[a] = torus_acc_scaled(pk[a].y, -xi[a]);
#endif
            stage_park(a, tq);
            if ((a & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 8 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- the claimed slot goes back: only after every wavefront's parking stores (the last iteration's are dead, but in flight) have
    //      been acknowledged, so that the next owner's stores cannot be overtaken by them ----------------------------------------
    if (A.park_owner) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && park_slot < BRP_PARK_SLOTS) __hip_atomic_store(A.park_owner + park_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- sample extract coefficient 0 ---------------------------------------------------------------
    {
        const int te = brp_opaque_tid() & 255;
        const int ge = te >> 4, be = te & 15;
        const bool owner_e = ge < R * K1;
        const int re = owner_e ? ge / K1 : R - 1, pe = owner_e ? ge % K1 : K1 - 1;
        const uint64_t inst_e = inst0 + (uint64_t)hh * R + re;
        if (owner_e && inst_e < A.count) {
            const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
            uint64_t *o = A.out + inst_e * (big + 1);
            if (pe < K1 - 1) {
                uint64_t *om = o + (uint64_t)pe * FHE_N;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    int j0 = 16 * a + be, j1 = j0 + 256;
                    if (j0 == 0) om[0] = (uint64_t)0 - lo[a]; else om[FHE_N - j0] = lo[a];
                    om[FHE_N - j1] = hi[a];
                }
            } else if (be == 0) {
                o[big] = A.post_add - lo[0];
            }
        }
    }
}

// Workgroups 0 .. units_main-1 carry 2R = 6 ciphertexts, the rest 2R2 = 4 (R2 = 0: none): 16,384 bits = 2,560 x 6 + 256 x 4 = eleven
// whole generations of one workgroup per CU.
template <int K1, int LEVELS, int BASE_LOG, int R, int R2>
__global__ __launch_bounds__(BRP_THREADS, 1) void blind_rotate_pair_kernel(const ExtProdArgs A)
{
    __shared__ __attribute__((aligned(16))) double lds_all[BRP_LDS_DOUBLES(R) + BRP_LDS_EXTRA_DOUBLES];
    static_assert((BRP_LDS_DOUBLES(R) + BRP_LDS_EXTRA_DOUBLES) * 8 <= 163840, "one workgroup must fit the 160 KB of a CU");
    unsigned *slot_word = reinterpret_cast<unsigned *>(lds_all + BRP_LDS_DOUBLES(R));
    if constexpr (R2 > 0) {
        if (blockIdx.x >= A.units_main) {       // scalar branch
            blind_rotate_pair_unit<K1, LEVELS, BASE_LOG, R2>(A, lds_all, slot_word, (uint64_t)A.units_main * (2 * R) + (uint64_t)(blockIdx.x - A.units_main) * (2 * R2), blockIdx.x);
            return;
        }
    }
    blind_rotate_pair_unit<K1, LEVELS, BASE_LOG, R>(A, lds_all, slot_word, (uint64_t)blockIdx.x * (2 * R), blockIdx.x);
}
