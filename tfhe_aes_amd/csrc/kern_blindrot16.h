// kern_blindrot16.h -- K2, blind rotation of the circuit-bootstrap PBS + sample extract,
// throughput form for large batches (the same arithmetic and lane mapping as extprod_rotate_kernel in
// kern_extprod.h, which stays the form of K5; bit-identical results).
//
// What differs from kern_extprod.h, and why (measured with per-phase s_memtime stamps and rocprofv3 counters, profiles/r02_*, r03_*):
//  * the accumulator (64 VGPRs per lane) is dead weight between the rotation at the top of an iteration and the
//    accumulate at its end; it is PARKED: stored once per iteration with coalesced 16-byte buffer stores into a per-workgroup
//    slab, reloaded (non-temporal: its last use) into registers that are free by then, a whole products exchange + inverse
//    transform ahead of its use.  It is kept NEGATED, which turns the rotation's negate / select / subtract into two xor and
//    two 64-bit additions per coefficient (see the accumulator init).
//  * with two waves per SIMD nothing hides an LDS round trip but the wave's own instruction stream, and round 2's counters show the
//    waves waiting to ISSUE LDS instructions 19 % of their time.  The transform (fft_dev.h, canonical form v2) therefore reads
//    ONE table, 16 entries per lane and transform, in two batches issued a whole decomposition step / first pass ahead of their
//    use; its first pass has lane-independent twiddles (compile-time constants in SGPRs) and the twist costs no pass of its own.
//  * a burst of 25 key loads blocks the in-order wave for as long as the L1 takes to accept them (~150 cycles each with one
//    workgroup on the CU).  15 of the 25 GGSW entries of a level are therefore requested one or two at a time BETWEEN the
//    instructions of the transpose and of the second pass, into the registers the table entries have just left; the other 10
//    right after the digit stores, into the registers of the transform working set, before the exchange barrier.
//  * key rows and parking slots are addressed as (scalar base) + (16 * lane) through raw buffer instructions: no 64-bit vector
//    address arithmetic; LDS addresses are (one base per lane) + constants, recomputed per phase from an opaque lane index so the
//    compiler does not keep dozens of them live across the 669-iteration loop (lane roles for the epilogue included: that alone
//    freed 19 VGPRs).
#pragma once
#include "fft_dev.h"
#include "kern_extprod.h"

#ifndef BR16_MAC_PRIO
#define BR16_MAC_PRIO 0
#endif
#ifndef BR16_EARLY
#define BR16_EARLY 15      /* GGSW entries (of 25 per level; scaled to K1*K1) requested between the instructions of the transform's
                              second half.  Measured at 16,384 bits: 0 -> 267 ms, 12 -> 267, 13 -> 255, 14 -> 253, 15 -> 251.5, 16 -> 263,
                              20 -> 268, 25 -> 297 (spills past 18); some of them ahead of the tiles-free barrier, or the late ones
                              between the digit stores: no gain. */
#endif
#ifndef BR16_XPOSE_IN_TWIDDLE
#define BR16_XPOSE_IN_TWIDDLE 1    /* with BR16_STORE_IN_PASS2: 239 -> 231 ms per 16,384-bit launch; either one alone: no change */
#endif
#ifndef BR16_STAGE_AT_END
#define BR16_STAGE_AT_END 1
#endif
#ifndef BR16_HEAD
#define BR16_HEAD 0        /* GGSW entries (of 25 per level) requested between the stages of pass 1 */
#endif
#ifndef BR16_LATE_IN_PASS2
#define BR16_LATE_IN_PASS2 1
#endif
#ifndef BR16_READ_IN_PASS2
#define BR16_READ_IN_PASS2 1
#endif
#ifndef BR16_STORE_IN_PASS2
#define BR16_STORE_IN_PASS2 1
#endif
#ifndef BR16_PARK_AUX_ST
#define BR16_PARK_AUX_ST 0 /* cache policy bits of the parking stores (1 = sc0, 2 = nt, 16 = sc1; measured: see DESIGN.md) */
#endif
#ifndef BR16_PARK_AUX_LD
#define BR16_PARK_AUX_LD 2 /* ... and of the parking loads.  nt: the reload is the line's last use, it should not displace GGSW rows in L2.
                              Measured per 16,384-bit launch (rocprofv3 FETCH_SIZE x 2, same box): loads default 775 GB / 239 ms, nt 527 GB /
                              235 ms; stores sc1 or sc0+sc1 on top: no further change; nt STORES: 242-247 ms (slower), whatever the loads do */
#endif
#define BR16_PARK_WORDS_PER_WG (16 * EP_THREADS * 2)       /* 16 chunks of 16 bytes per thread (lo[a], hi[a]): 64 KB per workgroup */
#ifndef BR16_PARK_OWNERS_ONLY
#define BR16_PARK_OWNERS_ONLY 1   /* lane groups that own no polynomial (group 15 of a three-ciphertext unit, groups 10-15 of a two-ciphertext
                                     one) park nothing: their buffer offset is out of range, so the store is dropped and the load returns 0
                                     (round 3 stored and reloaded their 4 KB per group 669 times for nothing: 45 GB per launch) */
#endif
#ifndef BR16_W3_LDS_HOME
#define BR16_W3_LDS_HOME 1        /* three-ciphertext units: the accumulators of wavefront 3 (groups 12-14) LIVE in LDS instead of the parking
                                     slab -- see blind_rotate16_unit.  Needs 81,920 B of LDS per workgroup (two of them = all 160 KB of a CU);
                                     the launcher falls back to the parked form if the runtime does not place two such workgroups on a CU */
#endif
#ifndef BR16_RESIDENT_HI
#define BR16_RESIDENT_HI 0        /* 1: the upper halves hi[] of the (negated) accumulator stay in registers for the whole rotation; only lo[]
                                     is parked (two coefficients per 16-byte chunk: 8 stores + 8 loads per lane and iteration instead of 16 + 16).
                                     Costs 32 VGPRs, i.e. 8 of the 15 GGSW entries that BR16_EARLY keeps in flight across the transform */
#endif
#ifndef BR16_MAC_TAIL
#define BR16_MAC_TAIL (BR16_RESIDENT_HI ? 5 : 0)   /* GGSW entries (of 25 per level) requested only after the multiply-accumulate has used row 0 */
#endif
#ifndef BR16_W1_LATE
#define BR16_W1_LATE BR16_RESIDENT_HI
#endif
#define BR16_HOME_LDS_DOUBLES (2 * FHE_TW_ENTRIES + (EP_GROUPS - 1) * GROUP_TILE_DOUBLES + 3 * FHE_N)   /* table + 15 tiles + 3 accumulators = 81,920 B */

__device__ __forceinline__ int br16_opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

#ifndef BR16_PAD_DOUBLES
#define BR16_PAD_DOUBLES 0     /* developer ablation: extra LDS so that only one workgroup fits a CU */
#endif
// One unit of work = R ciphertexts starting at `inst0`, one workgroup (the body of the kernel below).
//
// HOME (three-ciphertext units, R * K1 = 15): group 15 owns nothing and has always computed a duplicate of group 14 (same ciphertext,
// same polynomial).  Here it is an exact MIRROR of it: same tile, same values, same wavefront, so every LDS write of a group-15 lane
// hits the address its group-14 twin writes in the same instruction with the same value.  That frees tile 15, and tile 15 + the LDS two
// workgroups left unused on a CU (2 x 7,936 B) is exactly three accumulators (12,288 B): the accumulators of wavefront 3 (groups
// 12, 13, 14) live THERE for the whole rotation -- the rotation reads them in place, the new value overwrites them in place, and
// wavefront 3 has no parking traffic at all (a quarter of the slab's 64 KB per workgroup and iteration).
template <int K1, int LEVELS, int BASE_LOG, int R, bool HOME = false>
__device__ __forceinline__ void blind_rotate16_unit(const ExtProdArgs &A, double *lds_all, const uint64_t inst0)
{
    static_assert(R * K1 <= EP_GROUPS, "too many polynomials for 16 lane groups");
    static_assert(!HOME || R * K1 == EP_GROUPS - 1, "the LDS home needs exactly one idle lane group (its tile is the home's first third)");
    constexpr int LAST_G = R * K1 - 1, HOME_G0 = 12;          // last owner group; first group of wavefront 3
    // the twiddle table first: its addresses then fit the 16-bit offset field of the LDS instructions
    double2 *tw = reinterpret_cast<double2 *>(lds_all);
    double *lds = lds_all + 2 * FHE_TW_ENTRIES;                   // the 16 group tiles

    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    // wave-uniform: this wavefront's accumulators live in LDS (scalar branches below)
    const bool home_wave = HOME && __builtin_amdgcn_readfirstlane(tid) >= 16 * HOME_G0;
    // the tile a lane's group transposes through / exchanges digits in (HOME: group 15 shares group 14's, see above)
    auto tile_of = [&](const int tq) -> double * {
        int gq = tq >> 4;
        if (HOME) gq = gq < LAST_G ? gq : LAST_G;
        return lds + gq * GROUP_TILE_DOUBLES;
    };
    // where the rotation reads the group's (negated) accumulator: its tile (copied there by stage_park), or its LDS home
    auto stage_of = [&](const int tq) -> uint64_t * {
        int gq = tq >> 4;
        if (HOME) {
            gq = gq < LAST_G ? gq : LAST_G;
            const int tile_words = gq * GROUP_TILE_DOUBLES, home_words = (EP_GROUPS - 1) * GROUP_TILE_DOUBLES + (gq - HOME_G0) * FHE_N;
            return reinterpret_cast<uint64_t *>(lds) + (gq >= HOME_G0 ? home_words : tile_words);
        }
        return reinterpret_cast<uint64_t *>(lds + gq * GROUP_TILE_DOUBLES);
    };
    // byte offset of a lane in a parking chunk; lanes that park nothing get an out-of-range offset (raw buffer: dropped / zero)
    auto park_lane = [&](const int tq) -> unsigned {
        if (HOME) return (tq >> 4) < HOME_G0 ? (unsigned)tq * 16u : 0x80000000u;              // wavefront 3 parks nothing
        if (BR16_PARK_OWNERS_ONLY && R * K1 < EP_GROUPS) return (tq >> 4) < R * K1 ? (unsigned)tq * 16u : 0x80000000u;
        return (unsigned)tq * 16u;
    };

    ep_load_table(tw, A.tw);

    uint64_t inst = inst0 + r_own;
    const bool valid = inst < A.count;
    if (!valid) inst = A.count - 1;
    const uint64_t *lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);

    // ---- accumulator init (coefficients 16a+b and 256+16a+b in lane b) ------------------------------------------
    // The registers, the LDS tile and the parking slab hold the NEGATED accumulator (nacc = -acc): the rotation then needs
    //   d = acc * X^t - acc = (+-)(-nacc[src]) + nacc[j],
    // i.e. a conditional two's complement (xor with a mask; its "+1" rides on the decomposition's rounding constant) and ONE
    // 64-bit addition per coefficient, instead of a negate, a select and a subtraction with their carry chains.
    uint64_t lo[16], hi[16];
    {
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            int j0 = 16 * a + b, j1 = j0 + 256;
            int e0 = ((j0 - t) & 511) + t, e1 = ((j1 - t) & 511) + t;
            uint64_t v0 = ((e0 >> 9) & 1) ? A.tv_const : (uint64_t)0 - A.tv_const;      // negated (see above)
            uint64_t v1 = ((e1 >> 9) & 1) ? A.tv_const : (uint64_t)0 - A.tv_const;
            lo[a] = (p_own == K1 - 1) ? v0 : 0;
            hi[a] = (p_own == K1 - 1) ? v1 : 0;
        }
    }
    // the parking slab as one raw buffer: (scalar: workgroup slab + chunk) + (16 * lane) -- no vector address arithmetic
    const __amdgpu_buffer_rsrc_t park_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.park, 0, (int)A.park_bytes, 0x00020000);
    const unsigned park_wg = blockIdx.x * (unsigned)(BR16_PARK_WORDS_PER_WG * 8);                       // wave-uniform
#define BR16_PARK_SLOT(a) ((unsigned)(a) * (EP_THREADS * 16))
    __syncthreads();   // tables visible

    constexpr unsigned GGSW_BYTES = LEVELS * K1 * K1 * FHE_H * 16;   // one GGSW of the Fourier BSK
    // the whole Fourier BSK as one raw buffer (< 2^31 bytes for every supported parameter set: checked by the launcher)
    const __amdgpu_buffer_rsrc_t bsk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(A.ggsw), 0, (int)(A.iters * GGSW_BYTES), 0x00020000);
    uint64_t a_next = lwe[0];

#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    // coefficient pair a of the (negated) accumulator -> the group's LDS tile (for the next rotation) and the parking slab
    auto stage_park = [&](const int a, const int tq) {
        uint64_t *stage = stage_of(tq);
        stage[16 * a + (tq & 15)] = lo[a];
        stage[256 + 16 * a + (tq & 15)] = hi[a];
#ifndef BR16_ABL_NOPARK
#if BR16_RESIDENT_HI
        if (a & 1) {                                             // compile-time: a is an unrolled loop index
            ep_u32x4 v;
            v[0] = (uint32_t)lo[a - 1]; v[1] = (uint32_t)(lo[a - 1] >> 32); v[2] = (uint32_t)lo[a]; v[3] = (uint32_t)(lo[a] >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(v, park_rsrc, park_lane(tq), park_wg + BR16_PARK_SLOT(a >> 1), BR16_PARK_AUX_ST);
        }
#else
        {                                                        // (wavefront 3 of a HOME unit: out of range, it has just written its home)
            ep_u32x4 v;
            v[0] = (uint32_t)lo[a]; v[1] = (uint32_t)(lo[a] >> 32); v[2] = (uint32_t)hi[a]; v[3] = (uint32_t)(hi[a] >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(v, park_rsrc, park_lane(tq), park_wg + BR16_PARK_SLOT(a), BR16_PARK_AUX_ST);
        }
#endif
#endif
    };
#if BR16_STAGE_AT_END
    {
        const int tq = br16_opaque_tid();
#pragma unroll
        for (int a = 0; a < 16; ++a) stage_park(a, tq);
    }
#endif
    for (uint32_t it = 0; it < A.iters; ++it) {
        const int t = mod_switch_1024(a_next);
        a_next = lwe[it + 1];                                    // one iteration ahead (the last one reads the body: unused)
#ifdef BR16_ABL_SAMEKEY
        const unsigned g_bytes = (it & 1) * GGSW_BYTES;          // developer ablation (wrong results): two L2-resident GGSWs = a 100 % L2 hit rate
#else
        const unsigned g_bytes = it * GGSW_BYTES;                // wave-uniform byte offset of this iteration's GGSW
#endif

        // ---- accumulator -> tile and parking slab; d = acc * X^t - acc; first (least significant) digit -----------
        uint32_t st_lo[16], st_hi[16];
        double xr[16], xi[16];
        double2 w0[8], w1[8];                                     // twiddle buffers
        EP_STAMP(11);
        {
            const int tq = br16_opaque_tid();
            const int bq_ = tq & 15;
            uint64_t *stage = stage_of(tq);
#if !BR16_STAGE_AT_END
#pragma unroll
            for (int a = 0; a < 16; ++a) stage_park(a, tq);
#endif
            wave_lds_sync();
            fft_tw_load8(w0, tw, bq_, FHE_TW_STRIDE);             // first half of the lane's table column (T[0..7][b]): lands during the rotation
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                // coefficient j of acc * X^t is acc[(j - t) mod 512], negated iff bit 9 of (j - t) is set (t < 1024)
                const int u0 = 16 * a + bq_ - t, u1 = u0 + 256;
                const int s0 = u0 & 511, s1 = s0 ^ 256;
                const fhe_u32x2 v0 = *reinterpret_cast<const fhe_u32x2 *>(stage + s0), v1 = *reinterpret_cast<const fhe_u32x2 *>(stage + s1);   // -acc[src]
                // mask = ~0: no wrap, take +acc[src] = ~v + 1; mask = 0: wrapped, take -acc[src] = v.  The "+1" joins the rounding
                // constant 2^(R-1) of the decomposition (one 32-bit subtraction), so each coefficient costs two xor and two 64-bit adds.
                const uint32_t m0 = (uint32_t)((u0 >> 9) & 1) - 1u, m1 = (uint32_t)((u1 >> 9) & 1) - 1u;
                constexpr uint64_t RND = 1ull << (64 - BASE_LOG * LEVELS - 1);
                static_assert(RND < (1ull << 31), "rounding constant must fit the low word");
                fhe_u32x2 w0, w1, k0, k1;
                w0[0] = v0[0] ^ m0; w0[1] = v0[1] ^ m0; w1[0] = v1[0] ^ m1; w1[1] = v1[1] ^ m1;
                k0[0] = (uint32_t)RND - m0; k0[1] = 0; k1[0] = (uint32_t)RND - m1; k1[1] = 0;
                const uint64_t x0 = (__builtin_bit_cast(uint64_t, w0) + lo[a]) + __builtin_bit_cast(uint64_t, k0);   // d + rounding constant
                const uint64_t x1 = (__builtin_bit_cast(uint64_t, w1) + hi[a]) + __builtin_bit_cast(uint64_t, k1);
                xr[a] = (double)decompose_first_rounded<BASE_LOG, LEVELS>(x0, st_lo[a]);
                xi[a] = (double)decompose_first_rounded<BASE_LOG, LEVELS>(x1, st_hi[a]);
                if ((a & (EP_ROT_CHUNK - 1)) == EP_ROT_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_sync();
        }
        EP_STAMP(0);

        double fr[R][K1], fi[R][K1];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) { fr[r][c] = 0.0; fi[r][c] = 0.0; }

        // One decomposition level.  On entry w0 / w1 hold (or are about to receive) the lane's table column, and xr/xi the digits.
        // `last`: the most significant level (the last one of the iteration): its multiply-accumulate also requests the parked
        // accumulator, a few chunks per row, instead of a burst of 16 loads behind it
#if BR16_RESIDENT_HI
        uint64_t pkl[16];                                         // the parked half (lo[]) on its way back
#else
        ulonglong2 pk[16];
#endif
        auto level_body = [&](const int l, const bool tiles_busy, auto last) {
            const int tq = br16_opaque_tid();
            const int bq_ = tq & 15;
            double *tile = tile_of(tq);
            const unsigned gl_bytes = g_bytes + (unsigned)l * (K1 * K1 * FHE_H * 16);   // scalar; the lane adds 16 * point
            double2 bm[K1][K1];
            // GGSW entries [from, to) of this level (row-major: the multiply-accumulate consumes them in this order)
            auto key_rows = [&](const int from, const int to) {
#pragma unroll
                for (int q = 0; q < K1 * K1; ++q) {
                    if (q < from || q >= to) continue;
#ifdef BR16_ABL_HALFKEY
                    if (q % K1 >= 3) continue;                    // developer ablation (wrong results): 60 % of the key bytes, same arithmetic
#endif
#ifdef BR16_ABL_NOLOAD
                    bm[q / K1][q % K1] = make_double2((double)(tq + q), (double)(tq - q));
#else
                    bm[q / K1][q % K1] = ep_key_load(bsk_rsrc, (unsigned)tq * 16u, gl_bytes + (unsigned)q * (FHE_H * 16));
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- forward transform (fft_dev.h): pass 1 (frequency offset 1/4, constants only), twiddle by the table column read a
            //      whole decomposition step ago, transpose, pass 2 ---------------------------------------------------------------
#if !BR16_W1_LATE
            fft_tw_load8(w1, tw, 8 * FHE_TW_STRIDE + bq_, FHE_TW_STRIDE);   // second half of the column (T[8..15][b]): lands during pass 1
#endif
            __builtin_amdgcn_sched_barrier(0);
            // pass 1 is pure vector work (192 fused operations, no table): the first BR16_HEAD GGSW entries are requested between its
            // stages, as many as the register file has room for while the table column and the working set are both live
            constexpr int NH = BR16_HEAD * K1 * K1 / 25;
#ifndef BR16_ABL_NOFFT
            dft16<false, true>(xr, xi, [&](const int stage) { if (NH) { __builtin_amdgcn_sched_barrier(0); key_rows(NH * stage / 4, NH * (stage + 1) / 4); } });
#else
            key_rows(0, NH);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if !BR16_XPOSE_IN_TWIDDLE
            fft_tw_mul<false, 8>(xr, xi, w0);
            fft_tw_mul<false, 8>(xr + 8, xi + 8, w1);
#endif
            EP_STAMP(2);
            // The first BR16_EARLY entries are requested a few at a time BETWEEN the instructions of the transpose and of the
            // second DFT16, into the registers the twiddle buffers have just left: a burst of 25 loads blocks the in-order
            // wave for as long as the L1 takes to accept them (~150 cycles each with one workgroup per CU); spaced out,
            // the same acceptance time passes under the wave's own LDS and vector work.
            constexpr int NE = BR16_EARLY * K1 * K1 / 25 > NH ? BR16_EARLY * K1 * K1 / 25 : NH, NHOOK = 7;
            constexpr int NT = BR16_MAC_TAIL * K1 * K1 / 25;    // the last NT entries (of the last rows) are requested from inside the multiply-accumulate
            auto early = [&](const int h) { key_rows(NH + (NE - NH) * h / NHOOK, NH + (NE - NH) * (h + 1) / NHOOK); };
            if (tiles_busy) wg_barrier_lds_only();                // every thread is done reading the previous level's digits
            EP_STAMP(3);
#if defined(BR16_ABL_NOFFT)
            group_transpose(xr, xi, tile, bq_);
            key_rows(0, NE);
#elif defined(BR16_ABL_NOXPOSE)
            dft16<false, false>(xr, xi);
            key_rows(0, NE);
#else
            {
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(FFT_XPOSE_PRIO);
#endif
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
#if BR16_W1_LATE
                    // the second half of the table column is requested only now (it lands during the first eight multiplies): while pass 1
                    // runs the registers hold one half of the column, not both
                    if (k1 == 0) { fft_tw_load8(w1, tw, 8 * FHE_TW_STRIDE + bq_, FHE_TW_STRIDE); __builtin_amdgcn_sched_barrier(0); }
#endif
#if BR16_XPOSE_IN_TWIDDLE
                    // each value leaves for the transpose tile as soon as its twiddle multiply is done: 16 stores spread over 64
                    // vector instructions instead of a burst
                    if (k1 < 8) cmul(xr[k1], xi[k1], w0[k1].x, w0[k1].y); else cmul(xr[k1], xi[k1], w1[k1 - 8].x, w1[k1 - 8].y);
#endif
                    double2 v; v.x = xr[k1]; v.y = xi[k1];
                    *reinterpret_cast<double2 *>(tile + 2 * (k1 * 17 + bq_)) = v;
#if BR16_XPOSE_IN_TWIDDLE
                    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
#endif
                    if (NE && k1 == 7) { __builtin_amdgcn_sched_barrier(0); early(0); }
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(1); }
                wave_lds_sync();
#if BR16_READ_IN_PASS2 && BR16_STORE_IN_PASS2
                // transposed reads in the order the first butterfly stage consumes them (registers fft_reg(0), fft_reg(1), ...), and
                // NO wait behind them: the butterflies start as the pairs arrive.  The group's reads must all have been issued and
                // returned before its first digit store reuses the tile: that wait sits in front of that store (stage 3), by when it is free.
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = fft_reg(q);
                    double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
                    xr[c] = v.x; xi[c] = v.y;
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(2); }
#else
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
                    xr[c] = v.x; xi[c] = v.y;
                }
                if (NE) { __builtin_amdgcn_sched_barrier(0); early(2); }
                wave_lds_sync();
#endif
#if FFT_XPOSE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
#if BR16_STORE_IN_PASS2
                // the transformed digits leave for the tile as the last butterfly stage produces them (outputs k and k + 8 of butterfly
                // k), instead of as a burst of 16 stores behind the transform: the LDS queue is what the waves of a CU wait for most
                dft16<false, false>(xr, xi, [&](const int stage) { if (NE) { __builtin_amdgcn_sched_barrier(0); early(3 + stage); } },
                                    [&](const int stage, const int c0) {
                                        if (stage != 3) return;
#if BR16_READ_IN_PASS2
                                        if (c0 == 0) wave_lds_sync();      // every lane of the group has its transposed values (see the reads)
#endif
#pragma unroll
                                        for (int j = 0; j < FFT_CHUNK; ++j) {
#pragma unroll
                                            for (int h = 0; h < 2; ++h) {
                                                const int k2 = c0 + j + 8 * h;
                                                double2 v; v.x = xr[fft_reg(k2)]; v.y = xi[fft_reg(k2)];
                                                *reinterpret_cast<double2 *>(tile + 2 * (bq_ + 16 * k2)) = v;
                                            }
                                        }
                                        __builtin_amdgcn_sched_barrier(0);
#if BR16_LATE_IN_PASS2
                                        // ... and the registers of the values just stored take the next share of the remaining GGSW entries
                                        {
                                            constexpr int NL = K1 * K1 - NE - NT, PARTS = 8 / FFT_CHUNK;
                                            const int part = c0 / FFT_CHUNK;
                                            key_rows(NE + NL * part / PARTS, NE + NL * (part + 1) / PARTS);
                                        }
#endif
                                    });
#else
                dft16<false, false>(xr, xi, [&](const int stage) { if (NE) { __builtin_amdgcn_sched_barrier(0); early(3 + stage); } });
#endif
            }
#endif
            EP_STAMP(4);
            // store the transformed digits, then request the remaining GGSW entries of this level into the registers the
            // working set has just left
#if !BR16_STORE_IN_PASS2 || defined(BR16_ABL_NOFFT) || defined(BR16_ABL_NOXPOSE)
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (bq_ + 16 * k2)) = v;
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#if !(BR16_LATE_IN_PASS2 && BR16_STORE_IN_PASS2) || defined(BR16_ABL_NOFFT) || defined(BR16_ABL_NOXPOSE)
            key_rows(NE, K1 * K1 - NT);
#endif
            EP_STAMP(5);
            wg_barrier_lds_only();                                // digits of all groups visible; key loads stay in flight
            EP_STAMP(6);
            // ---- multiply-accumulate role: thread tq owns Fourier point tq; digits are read one row ahead ---------------
#if BR16_MAC_PRIO
            __builtin_amdgcn_s_setprio(BR16_MAC_PRIO);
#endif
            double2 dn[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dn[r] = *reinterpret_cast<const double2 *>(lds + (r * K1) * GROUP_TILE_DOUBLES + 2 * tq);
#pragma unroll
            for (int p = 0; p < K1; ++p) {
                double2 d[R];
#pragma unroll
                for (int r = 0; r < R; ++r) d[r] = dn[r];
                if (p + 1 < K1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dn[r] = *reinterpret_cast<const double2 *>(lds + (r * K1 + p + 1) * GROUP_TILE_DOUBLES + 2 * tq);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef BR16_ABL_NOMAC
                if (p == 0)
#endif
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int c = 0; c < K1; ++c) {
#ifdef BR16_ABL_HALFKEY
                        const double2 kq = bm[p][c >= 3 ? c - 3 : c];
#else
                        const double2 kq = bm[p][c];
#endif
                        fr[r][c] = __builtin_fma(d[r].x, kq.x, fr[r][c]);
                        fr[r][c] = __builtin_fma(-d[r].y, kq.y, fr[r][c]);
                        fi[r][c] = __builtin_fma(d[r].x, kq.y, fi[r][c]);
                        fi[r][c] = __builtin_fma(d[r].y, kq.x, fi[r][c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (NT && p == 0) key_rows(K1 * K1 - NT, K1 * K1);      // into the registers row 0 has just left
                if constexpr (decltype(last)::value) {
                    // parked accumulator back: lands during the products exchange and the inverse transform
#if BR16_RESIDENT_HI
#pragma unroll
                    for (int j = 8 * p / K1; j < 8 * (p + 1) / K1; ++j) {
#ifdef BR16_ABL_NOPARK
                        pkl[2 * j] = 0; pkl[2 * j + 1] = 0;
#else
                        const ep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(park_rsrc, park_lane(tq), park_wg + BR16_PARK_SLOT(j), BR16_PARK_AUX_LD);
                        pkl[2 * j] = ((unsigned long long)v[1] << 32) | v[0];
                        pkl[2 * j + 1] = ((unsigned long long)v[3] << 32) | v[2];
#endif
                    }
#elif defined(BR16_ABL_NOPARK)
#pragma unroll
                    for (int a = 16 * p / K1; a < 16 * (p + 1) / K1; ++a) { pk[a].x = 0; pk[a].y = 0; }
#else
                    {
#pragma unroll
                        for (int a = 16 * p / K1; a < 16 * (p + 1) / K1; ++a) {
                            const ep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(park_rsrc, park_lane(tq), park_wg + BR16_PARK_SLOT(a), BR16_PARK_AUX_LD);
                            pk[a].x = ((unsigned long long)v[1] << 32) | v[0];
                            pk[a].y = ((unsigned long long)v[3] << 32) | v[2];
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if BR16_MAC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            EP_STAMP(7);
        };

        // levels L-1 .. 1 rolled (transform + multiply-accumulate, then the next level's digits), the last level on its own
#pragma unroll 1
        for (int l = LEVELS - 1; l >= 1; --l) {
            level_body(l, l != LEVELS - 1, std::false_type{});
            {
                const int tq = br16_opaque_tid();
                fft_tw_load8(w0, tw, tq & 15, FHE_TW_STRIDE);     // lands during the decomposition step
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                xr[a] = (double)decompose_next<BASE_LOG>(st_lo[a]);
                xi[a] = (double)decompose_next<BASE_LOG>(st_hi[a]);
            }
            EP_STAMP(1);
        }
        level_body(0, LEVELS > 1, std::true_type{});

        const int tq = br16_opaque_tid();
        const int bq_ = tq & 15;
        double *tile = tile_of(tq);
        // ---- products back to the owning groups, inverse transform, accumulate --------------------------------------
        wg_barrier_lds_only();       // every thread is done reading the last level's digits from the tiles
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) {
                double2 v; v.x = fr[r][c]; v.y = fi[r][c];
                *reinterpret_cast<double2 *>(lds + (r * K1 + c) * GROUP_TILE_DOUBLES + 2 * tq) = v;
            }
        wg_barrier_lds_only();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ + 16 * k2));
            xr[k2] = v.x; xi[k2] = v.y;
        }
        fft_inv_table(w0, w1, tw, bq_);                            // the table row of this lane
        wave_lds_sync();
        EP_STAMP(8);
        // inverse transform (fft_dev.h's nega_inv, the table row read a pass ahead)
        dft16<true, false>(xr, xi);
        __builtin_amdgcn_sched_barrier(0);
#if BR16_XPOSE_IN_TWIDDLE
        // as in the forward transform: every value leaves for the transpose tile as soon as its twiddle multiply is done
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
            const int c = fft_reg(q);                                 // in the order the first butterfly stage consumes them
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (bq_ * 17 + c));
            xr[c] = v.x; xi[c] = v.y;
        }
#else
        fft_tw_mul<true, 8>(xr, xi, w0, 1);
        fft_tw_mul<true, 8>(xr + 8, xi + 8, w1);
        __builtin_amdgcn_sched_barrier(0);
        group_transpose(xr, xi, tile, bq_);
#endif
        if (home_wave) {
            // wavefront 3 of a HOME unit: the old accumulator comes from its LDS home (the lane's own coefficients: written by this
            // lane, no synchronisation needed), landing during the transform's last pass
            const uint64_t *home = stage_of(tq);
#pragma unroll
#if BR16_RESIDENT_HI
            for (int a = 0; a < 16; ++a) pkl[a] = home[16 * a + bq_];
#else
            for (int a = 0; a < 16; ++a) { pk[a].x = home[16 * a + bq_]; pk[a].y = home[256 + 16 * a + bq_]; }
#endif
        }
        dft16<true, false>(xr, xi);
#pragma unroll
        for (int a = 1; a < 16; ++a) cmulc(xr[a], xi[a], FHE_PSI16_RE[a], FHE_PSI16_IM[a]);
        EP_STAMP(9);
#if BR16_STAGE_AT_END
        wave_lds_sync();     // the inverse transform's transposed reads of this tile are complete in every lane of the group
#endif
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#if BR16_RESIDENT_HI
            lo[a] = torus_acc(pkl[a], -xr[a]);                // negated accumulator: -(acc + r) = -acc + (-r)
            hi[a] = torus_acc(hi[a], -xi[a]);
#else
            lo[a] = torus_acc(pk[a].x, -xr[a]);               // negated accumulator: -(acc + r) = -acc + (-r)
            hi[a] = torus_acc(pk[a].y, -xi[a]);
#endif
#if BR16_STAGE_AT_END
            // ... and leaves for the tile and the parking slab at once: 16 LDS + 16 memory stores spread over the conversion's
            // vector work instead of a burst at the top of the next iteration (the last iteration's copies are never read)
            stage_park(a, tq);
            if ((a & 1) == 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 4 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- sample extract coefficient 0 ---------------------------------------------------------------
    // (lane roles recomputed from the opaque lane index: nothing of them stays live across the 669-iteration loop)
    {
        const int te = br16_opaque_tid();
        const int ge = te >> 4, be = te & 15;
        const bool owner_e = ge < R * K1;
        const int re = owner_e ? ge / K1 : R - 1, pe = owner_e ? ge % K1 : K1 - 1;
        const uint64_t inst_e = inst0 + re;
        if (owner_e && inst_e < A.count) {
            const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
            uint64_t *o = A.out + inst_e * (big + 1);
            if (pe < K1 - 1) {
                uint64_t *om = o + (uint64_t)pe * FHE_N;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    int j0 = 16 * a + be, j1 = j0 + 256;
                    if (j0 == 0) om[0] = (uint64_t)0 - lo[a]; else om[FHE_N - j0] = lo[a];      // lo/hi hold -acc
                    om[FHE_N - j1] = hi[a];
                }
            } else if (be == 0) {
                o[big] = A.post_add - lo[0];
            }
        }
    }
}

// The launch: workgroups 0 .. units_main-1 carry R ciphertexts each, the rest R2 (R2 = 0: none).  With more units than the chip
// has slots (two per CU) the launcher picks both counts so that the total is a whole number of generations and covers the batch
// exactly: 16,384 bits = 5,120 x 3 + 512 x 2 = 11 full generations, instead of 5,462 x 3 whose eleventh generation leaves a third of
// the CUs idle for the length of a full one.  Workgroups are dispatched in index order, so the smaller units form the last generation.
// HOME: the R-ciphertext units keep wavefront 3's accumulators in LDS (blind_rotate16_unit); the R2 units are the parked form.
template <int K1, int LEVELS, int BASE_LOG, int R, int R2 = 0, bool HOME = false>
__global__ __launch_bounds__(EP_THREADS, 2) void blind_rotate16_kernel(const ExtProdArgs A)
{
    constexpr int LDS_DOUBLES = HOME ? BR16_HOME_LDS_DOUBLES : EP_LDS_DOUBLES + BR16_PAD_DOUBLES;
    static_assert(LDS_DOUBLES >= EP_LDS_DOUBLES && (BR16_PAD_DOUBLES > 0 || LDS_DOUBLES * 8 <= 81920), "two workgroups must fit the 160 KB of a CU");
    __shared__ __attribute__((aligned(16))) double lds_all[LDS_DOUBLES];
    if constexpr (R2 > 0) {
        if (blockIdx.x >= A.units_main) {       // scalar branch
            blind_rotate16_unit<K1, LEVELS, BASE_LOG, R2, false>(A, lds_all, (uint64_t)A.units_main * R + (uint64_t)(blockIdx.x - A.units_main) * R2);
            return;
        }
    }
    blind_rotate16_unit<K1, LEVELS, BASE_LOG, R, HOME>(A, lds_all, (uint64_t)blockIdx.x * R);
}
