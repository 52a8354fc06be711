// kern_extprod.h -- the external-product kernels:
//   K2  blind rotation of the circuit-bootstrap PBS + sample extract   
//   K5  vertical packing = CMUX blind rotation over the LUT + sample extract (8 a15)
//   K4  torus polynomials -> Fourier domain (BSK upload, ggsw.fill_with_forward_fourier, 8 a14)
//
// One 256-thread workgroup = 16 lane-groups of 16.  Group g < R*K1 owns polynomial p = g % K1 of
// ciphertext r = g / K1 of the workgroup's R ciphertexts: its 512 accumulator coefficients live
// in that group's registers for the whole rotation (coefficients 16a+b and 256+16a+b in lane b).
// Per iteration and per decomposition level the groups transform their digit polynomials
// (fft_dev.h) into the LDS tile ring, then ALL 256 threads switch roles: thread t owns Fourier
// point t and multiplies the R x K1 transformed digits with the K1 x K1 GGSW entries streamed from
// HBM/L2, so each 16-byte key element fetched is used by R ciphertexts.  After the last level the
// products go back through LDS to the owning groups for the inverse transform.
#pragma once
#include "fft_dev.h"

#define EP_THREADS 256
#ifndef EP_EARLY_LOAD
#define EP_EARLY_LOAD 2   /* GGSW rows started before the pre-multiply barrier (measured -3.5 %); 0 = none */
#endif
#ifndef EP_MAC_PRIO
#define EP_MAC_PRIO 1   /* wave priority while the load-latency-bound multiply-accumulate runs (measured -1.7 %) */
#endif
#ifndef EP_ROT_CHUNK
#define EP_ROT_CHUNK 4
#endif
#ifdef ABL_MAC_NOLOAD   /* developer ablation: no key traffic */
#define EP_LOADB(expr, p, c) make_double2((double)(tid + (c) + (p)), (double)(tid - (c)))
#else
#define EP_LOADB(expr, p, c) (expr)
#endif
#ifndef EP_PREFETCH
#define EP_PREFETCH 2
#endif
#define EP_GROUPS 16
#define EP_LDS_DOUBLES (EP_GROUPS * GROUP_TILE_DOUBLES + 2 * FHE_TW_ENTRIES)

// the twiddle table into LDS (272 entries, any workgroup size >= 64)
__device__ __forceinline__ void ep_load_table(double2 *tw_lds, const double2 *tw_g)
{
    for (int i = threadIdx.x; i < FHE_TW_ENTRIES; i += blockDim.x) tw_lds[i] = tw_g[i];
}

struct ExtProdArgs {
    // common
    const double2 *ggsw;        // PBS: BSK Fourier [n][L][K1][K1][256]; VP: [n_inputs][bits][L][K1][K1][256]
    const double2 *tw;          // [FHE_TW_ENTRIES] tw[17 k1 + b] = psi^(b (4 k1 + 1)), the transform's one table (fft_dev.h)
    uint64_t *out;              // PBS: [m][big+1]; VP: [n_inputs][n_luts][bits][big+1]
    uint64_t count;             // PBS: ciphertexts m; VP: n_inputs * n_luts * bits instances
    uint32_t iters;             // PBS: n; VP: bits
    // PBS mode
    const uint64_t *lwe_in;     // [m][n+1]
    uint64_t tv_const;          // every coefficient of the test vector
    uint64_t body_shift;        // added to the input body before the modulus switch (2^62)
    uint64_t post_add;          // added to the output body
    // VP mode
    const uint64_t *luts;       // [n_sets][n_luts][bits][512]
    uint32_t n_luts;
    uint32_t lut_per_input;
    uint32_t inst_per_input;    // n_luts * bits
    uint32_t wg_per_input;      // ceil(inst_per_input / R)
    uint32_t ggsw_per_input;    // VP: GGSWs per input in `ggsw` (= input bits; `iters` of them, the low bits, drive the rotation)
    uint64_t lut_words;         // VP: words per (LUT, output bit) in `luts` (512, or 2^bits when a CMUX tree ran first)
    const uint64_t *glwe_in;    // VP: non-null: the accumulator starts from this GLWE [instance][K1][512] (root of the CMUX tree)
    uint64_t *park;             // kern_blindrot16.h: accumulator parking space, 64 KB per workgroup
    uint64_t park_bytes;        //   its size (< 2^31: one raw buffer)
    uint32_t *park_owner;       // kern_blindrot_pair.h: owner words of the shared parking slots (0 = free), or null: one private slot per workgroup
    uint32_t units_main;        // kern_blindrot16.h: workgroups below this index carry R ciphertexts, the others R2
#ifdef EP_STAMPS
    unsigned long long *stamps; // developer build: per-wave cycles per phase [grid][4 waves][EP_NPH]
#endif
};

// One 16-byte key element through a raw buffer load: the address is (buffer base, scalar) + (row offset, scalar) +
// (16 * point, the only vector part), so a key fetch costs no vector address arithmetic at all.
typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
#ifndef EP_KEY_AUX
#define EP_KEY_AUX 0       /* cache policy bits of the key loads (1 = sc0, 2 = nt, 16 = sc1) */
#endif
__device__ __forceinline__ double2 ep_key_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_bytes, unsigned row_bytes)
{
    ep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, row_bytes, EP_KEY_AUX);
    double2 d;
    d.x = __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0]));
    d.y = __longlong_as_double((long long)(((unsigned long long)v[3] << 32) | v[2]));
    return d;
}

#ifdef EP_STAMPS
#define EP_NPH 12
#define EP_STAMP(ph) do { unsigned long long t__ = __builtin_readcyclecounter(); ph_cyc[ph] += t__ - t_last; t_last = t__; } while (0)
#else
#ifndef EP_FENCE_MASK
#define EP_FENCE_MASK 0        /* developer knob: bit ph set = the wave drains its LDS/scalar-memory counter at phase boundary ph (what a stamp does) */
#endif
#define EP_STAMP(ph) do { if ((EP_FENCE_MASK >> (ph)) & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#endif

template <int K1, int LEVELS, int BASE_LOG, int R, bool VP>
#ifndef EP_MIN_WAVES
#define EP_MIN_WAVES 2
#endif
__global__ __launch_bounds__(EP_THREADS, EP_MIN_WAVES) void extprod_rotate_kernel(const ExtProdArgs A)
{
    static_assert(R * K1 <= EP_GROUPS, "too many polynomials for 16 lane groups");
    __shared__ __attribute__((aligned(16))) double lds[EP_LDS_DOUBLES];
    double2 *tw = reinterpret_cast<double2 *>(lds + EP_GROUPS * GROUP_TILE_DOUBLES);

    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    double *tile = lds + g * GROUP_TILE_DOUBLES;

    ep_load_table(tw, A.tw);

    // ---- which ciphertext / instance does this group work on -------------------------------
    uint64_t inst;            // PBS: ciphertext index; VP: global instance index
    uint64_t input = 0;       // VP: which radix input (selects the GGSW list)
    bool valid;
    if (!VP) {
        inst = (uint64_t)blockIdx.x * R + r_own;
        valid = inst < A.count;
        if (!valid) inst = A.count - 1;
    } else {
        input = blockIdx.x / A.wg_per_input;
        uint32_t local = (blockIdx.x % A.wg_per_input) * R + r_own;
        valid = local < A.inst_per_input;
        if (!valid) local = A.inst_per_input - 1;
        inst = input * A.inst_per_input + local;
    }

    // ---- accumulator init --------------------------------------------------------------------
    uint64_t lo[16], hi[16];
    const uint64_t *lwe = nullptr;
    if (!VP) {
        lwe = A.lwe_in + inst * (uint64_t)(A.iters + 1);
        const int bt = mod_switch_1024(lwe[A.iters] + A.body_shift);
        const int t = (1024 - bt) & 1023;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            int j0 = 16 * a + b, j1 = j0 + 256;
            int e0 = ((j0 - t) & 511) + t, e1 = ((j1 - t) & 511) + t;
            uint64_t v0 = ((e0 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            uint64_t v1 = ((e1 >> 9) & 1) ? (uint64_t)0 - A.tv_const : A.tv_const;
            lo[a] = (p_own == K1 - 1) ? v0 : 0;
            hi[a] = (p_own == K1 - 1) ? v1 : 0;
        }
    } else {
        if (A.glwe_in) {
            const uint64_t *src = A.glwe_in + (inst * K1 + p_own) * FHE_N;
#pragma unroll
            for (int a = 0; a < 16; ++a) { lo[a] = src[16 * a + b]; hi[a] = src[256 + 16 * a + b]; }
        } else {
            uint64_t local = inst % A.inst_per_input;
            uint64_t set = A.lut_per_input ? input : 0;
            const uint64_t *lut = A.luts + (set * A.inst_per_input + local) * A.lut_words;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                lo[a] = (p_own == K1 - 1) ? lut[16 * a + b] : 0;
                hi[a] = (p_own == K1 - 1) ? lut[256 + 16 * a + b] : 0;
            }
        }
    }
    __syncthreads();   // tables visible
#ifdef EP_STAMPS
    unsigned long long ph_cyc[EP_NPH];
    for (int i = 0; i < EP_NPH; ++i) ph_cyc[i] = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif

    const size_t ggsw_stride = (size_t)LEVELS * K1 * K1 * FHE_H;   // double2 elements per GGSW

    for (uint32_t it = 0; it < A.iters; ++it) {
        int t;
        const double2 *G;
        if (!VP) {
            t = mod_switch_1024(lwe[it]);
            G = A.ggsw + (size_t)it * ggsw_stride;
        } else {
            t = (1024 - (1 << it)) & 1023;
            G = A.ggsw + ((size_t)input * A.ggsw_per_input + it) * ggsw_stride;
        }

        // ---- d = acc * X^t - acc, first decomposition level -----------------------------------
        EP_STAMP(11);
        uint64_t *stage = reinterpret_cast<uint64_t *>(tile);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            stage[16 * a + b] = lo[a];
            stage[256 + 16 * a + b] = hi[a];
        }
        wave_lds_sync();
        uint32_t st_lo[16], st_hi[16];
        double xr[16], xi[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            int j0 = 16 * a + b, j1 = j0 + 256;
            int s0 = (j0 - t) & 511, s1 = (j1 - t) & 511;
            uint64_t v0 = stage[s0], v1 = stage[s1];
            if (((s0 + t) >> 9) & 1) v0 = (uint64_t)0 - v0;
            if (((s1 + t) >> 9) & 1) v1 = (uint64_t)0 - v1;
            v0 -= lo[a]; v1 -= hi[a];
            xr[a] = (double)decompose_first<BASE_LOG, LEVELS>(v0, st_lo[a]);
            xi[a] = (double)decompose_first<BASE_LOG, LEVELS>(v1, st_hi[a]);
            // bound the number of rotated coefficients in flight (each is 2 VGPRs on top of acc, state and digits)
            if ((a & (EP_ROT_CHUNK - 1)) == EP_ROT_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);
        }
        wave_lds_sync();
        EP_STAMP(0);

        double fr[R][K1], fi[R][K1];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) { fr[r][c] = 0.0; fi[r][c] = 0.0; }

        // One decomposition level: transform the digit polynomials, exchange through LDS, multiply-accumulate.
        // (A lambda called once for the first level and once inside the loop, so that xr/xi are provably dead
        // during the multiply-accumulate of every level: no loop-carried copy survives the level.)
        auto level_body = [&](const int l, const bool tiles_busy) {
#ifndef EP_LATE_BARRIER
            if (tiles_busy) __syncthreads();
#endif
            EP_STAMP(3);
            // first half of the transform needs no tile; the barrier that frees the tiles (other threads
            // may still be reading the previous level's digits) sits as late as possible
#ifndef ABL_NO_FFT
            {
                double2 w0[8], w1[8];
                fft_fwd_table(w0, w1, tw, b);
                nega_fwd_head(xr, xi, w0, w1);
            }
#endif
#ifdef EP_LATE_BARRIER
            if (tiles_busy) __syncthreads();
#endif
            EP_STAMP(2);
#ifndef ABL_NO_FFT
            nega_fwd_tail(xr, xi, tile, b);
#endif
            EP_STAMP(4);
            const double2 *Gl = G + (size_t)l * K1 * K1 * FHE_H + tid;
#if EP_EARLY_LOAD
            // Store a few transformed digits, start a GGSW row into the registers that just died, repeat: the
            // first EARLY rows are in flight across the remaining stores and the barrier (which therefore
            // must not drain vmcnt).
            constexpr int PF0 = (K1 < EP_PREFETCH) ? K1 : EP_PREFETCH;
            constexpr int EARLY = (EP_EARLY_LOAD < PF0) ? EP_EARLY_LOAD : PF0;
            constexpr int CHUNK = (K1 * 2 * 2 + 3) / 4;          // digits whose registers hold one row of K1 double2
            double2 bq[PF0][K1];
#pragma unroll
            for (int e = 0; e < EARLY; ++e) {
#pragma unroll
                for (int k2 = e * CHUNK; k2 < (e + 1) * CHUNK && k2 < 16; ++k2) {
                    double2 v; v.x = xr[k2]; v.y = xi[k2];
                    *reinterpret_cast<double2 *>(tile + 2 * (b + 16 * k2)) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < K1; ++c) bq[e][c] = Gl[(size_t)(e * K1 + c) * FHE_H];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k2 = (EARLY * CHUNK < 16 ? EARLY * CHUNK : 16); k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (b + 16 * k2)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
            EP_STAMP(5);
            wg_barrier_lds_only();
            EP_STAMP(6);
#else
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                double2 v; v.x = xr[k2]; v.y = xi[k2];
                *reinterpret_cast<double2 *>(tile + 2 * (b + 16 * k2)) = v;
            }
            __syncthreads();
#endif
            // ---- multiply-accumulate role: thread tid owns Fourier point tid ------------------
#ifndef ABL_NO_MAC
            // The K1 x K1 GGSW entries of this level stream from L2; without software pipelining every
            // row costs one exposed round trip (measured: 71 of 341 ms).  xr/xi are dead here, so PF rows
            // are kept in flight in their registers.
#if EP_MAC_PRIO
            __builtin_amdgcn_s_setprio(EP_MAC_PRIO);
#endif
            constexpr int PF = (K1 < EP_PREFETCH) ? K1 : EP_PREFETCH;
#if !EP_EARLY_LOAD
            double2 bq[PF][K1];
            constexpr int P_FIRST = 0;
#else
            constexpr int P_FIRST = EARLY;
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = P_FIRST; p < PF; ++p)
#pragma unroll
                for (int c = 0; c < K1; ++c) bq[p][c] = EP_LOADB(Gl[(size_t)(p * K1 + c) * FHE_H], p, c);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < K1; ++p) {
                double2 bv[K1];
#pragma unroll
                for (int c = 0; c < K1; ++c) bv[c] = bq[p % PF][c];
                if (p + PF < K1) {
#pragma unroll
                    for (int c = 0; c < K1; ++c) bq[p % PF][c] = EP_LOADB(Gl[(size_t)((p + PF) * K1 + c) * FHE_H], p, c);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < R; ++r) {
#ifdef ABL_MAC_NOLDS
                    double2 d; d.x = (double)(tid + r); d.y = (double)(p - tid);
#else
                    double2 d = *reinterpret_cast<const double2 *>(lds + (r * K1 + p) * GROUP_TILE_DOUBLES + 2 * tid);
#endif
#pragma unroll
                    for (int c = 0; c < K1; ++c) {
                        fr[r][c] = __builtin_fma(d.x, bv[c].x, fr[r][c]);
                        fr[r][c] = __builtin_fma(-d.y, bv[c].y, fr[r][c]);
                        fi[r][c] = __builtin_fma(d.x, bv[c].y, fi[r][c]);
                        fi[r][c] = __builtin_fma(d.y, bv[c].x, fi[r][c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#if EP_MAC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#endif
            EP_STAMP(7);
        };
        level_body(LEVELS - 1, false);
#pragma unroll 1
        for (int l = LEVELS - 2; l >= 0; --l) {
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                xr[a] = (double)decompose_next<BASE_LOG>(st_lo[a]);
                xi[a] = (double)decompose_next<BASE_LOG>(st_hi[a]);
            }
            EP_STAMP(1);
            level_body(l, true);
        }

        // ---- products back to the owning groups, inverse transform, accumulate ----------------
        __syncthreads();             // every thread is done reading the last level's digits from the tiles
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) {
                double2 v; v.x = fr[r][c]; v.y = fi[r][c];
                *reinterpret_cast<double2 *>(lds + (r * K1 + c) * GROUP_TILE_DOUBLES + 2 * tid) = v;
            }
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (b + 16 * k2));
            xr[k2] = v.x; xi[k2] = v.y;
        }
        wave_lds_sync();
        EP_STAMP(8);
#ifndef ABL_NO_FFT
        nega_inv(xr, xi, tw, tile, b);
#endif
        EP_STAMP(9);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            lo[a] += torus_from_double(xr[a]);
            hi[a] += torus_from_double(xi[a]);
        }
        EP_STAMP(10);
    }
#ifdef EP_STAMPS
    if (A.stamps && (tid & 63) == 0)
        for (int i = 0; i < EP_NPH; ++i) A.stamps[((size_t)blockIdx.x * 4 + (tid >> 6)) * EP_NPH + i] = ph_cyc[i];
#endif

    // ---- sample extract coefficient 0 ---------------------------------------------
    if (owner && valid) {
        const uint64_t big = (uint64_t)(K1 - 1) * FHE_N;
        uint64_t *o = A.out + inst * (big + 1);
        if (p_own < K1 - 1) {
            uint64_t *om = o + (uint64_t)p_own * FHE_N;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                int j0 = 16 * a + b, j1 = j0 + 256;
                if (j0 == 0) om[0] = lo[a]; else om[FHE_N - j0] = (uint64_t)0 - lo[a];
                om[FHE_N - j1] = (uint64_t)0 - hi[a];
            }
        } else if (b == 0) {
            o[big] = lo[0] + (VP ? 0 : A.post_add);
        }
    }
}

// ---- CMUX tree of vertical_packing for inputs wider than log2(N) = 9 bits (upstream vertical_packing, called
//      at many_wopbs.rs:277) ----------------------------------------------------------------------------------------------
// A LUT of 2^bits entries per output bit is P = 2^(bits-9) polynomials; input bits 9..bits-1 select one of them through a
// binary tree of CMUXes (bit 9 at the leaves), then the blind rotation over bits 0..8 selects the coefficient.  One launch per
// tree level: job = (instance, node) computes  out = ct0 + GGSW_bit (x) (ct1 - ct0)  with ct0 / ct1 = children 2*node, 2*node+1
// (leaf level: trivial GLWEs of LUT polynomials).  Same lane mapping and arithmetic as one iteration of the kernel above with
// one decomposition level.  Not on the AES path (8- and 9-bit inputs only); it completes many_wopbs_without_padding.
struct CmuxArgs {
    const double2 *ggsw;        // Fourier GGSWs [n_inputs][bits][K1][K1][256]
    const double2 *tw;          // the transform's table (see ExtProdArgs)
    const uint64_t *luts;       // leaf level: [n_sets][inst_per_input][lut_words]; else null
    const uint64_t *in;         // inner levels: GLWE [instances][2 * nodes_out][K1][512]
    uint64_t *out;              // GLWE [instances][nodes_out][K1][512]
    uint32_t bits, bit;         // GGSWs per input; which of them drives this level
    uint32_t nodes_out;         // nodes per instance after this level
    uint32_t inst_per_input;    // n_luts * bits
    uint32_t lut_per_input;
    uint32_t wg_per_input;      // ceil(inst_per_input * nodes_out / R)
    uint64_t lut_words;
};

template <int K1, int BASE_LOG, int R>
__global__ __launch_bounds__(EP_THREADS, 2) void cmux_level_kernel(const CmuxArgs A)
{
    static_assert(R * K1 <= EP_GROUPS, "too many polynomials for 16 lane groups");
    __shared__ __attribute__((aligned(16))) double lds[EP_LDS_DOUBLES];
    double2 *tw = reinterpret_cast<double2 *>(lds + EP_GROUPS * GROUP_TILE_DOUBLES);
    const int tid = threadIdx.x;
    const int g = tid >> 4, b = tid & 15;
    const bool owner = g < R * K1;
    const int r_own = owner ? g / K1 : R - 1;
    const int p_own = owner ? g % K1 : K1 - 1;
    double *tile = lds + g * GROUP_TILE_DOUBLES;
    ep_load_table(tw, A.tw);

    const uint64_t input = blockIdx.x / A.wg_per_input;
    const uint32_t jobs_per_input = A.inst_per_input * A.nodes_out;
    uint32_t local = (blockIdx.x % A.wg_per_input) * R + r_own;
    const bool valid = owner && local < jobs_per_input;
    if (local >= jobs_per_input) local = jobs_per_input - 1;
    const uint32_t instl = local / A.nodes_out, node = local % A.nodes_out;
    const uint64_t inst = input * A.inst_per_input + instl;

    uint64_t c0lo[16], c0hi[16];
    double xr[16], xi[16];
    {
        const uint64_t *p0, *p1;
        bool zero = false;
        if (A.luts) {
            const uint64_t set = A.lut_per_input ? input : 0;
            const uint64_t *lut = A.luts + (set * A.inst_per_input + instl) * A.lut_words;
            p0 = lut + (uint64_t)(2 * node) * FHE_N;
            p1 = p0 + FHE_N;
            zero = p_own != K1 - 1;                          // trivial GLWE: only the body polynomial is non-zero
        } else {
            p0 = A.in + ((inst * (2 * A.nodes_out) + 2 * node) * K1 + p_own) * FHE_N;
            p1 = p0 + (uint64_t)K1 * FHE_N;
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const uint64_t a0 = zero ? 0 : p0[16 * a + b], a1 = zero ? 0 : p0[256 + 16 * a + b];
            const uint64_t b0 = zero ? 0 : p1[16 * a + b], b1 = zero ? 0 : p1[256 + 16 * a + b];
            c0lo[a] = a0; c0hi[a] = a1;
            uint32_t st;
            xr[a] = (double)decompose_first<BASE_LOG, 1>(b0 - a0, st);
            xi[a] = (double)decompose_first<BASE_LOG, 1>(b1 - a1, st);
        }
    }
    __syncthreads();   // tables visible
    nega_fwd(xr, xi, tw, tile, b);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        double2 v; v.x = xr[k2]; v.y = xi[k2];
        *reinterpret_cast<double2 *>(tile + 2 * (b + 16 * k2)) = v;
    }
    __syncthreads();
    // multiply-accumulate role: thread tid owns Fourier point tid
    double fr[R][K1], fi[R][K1];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) { fr[r][c] = 0.0; fi[r][c] = 0.0; }
    const double2 *G = A.ggsw + ((size_t)input * A.bits + A.bit) * (size_t)(K1 * K1 * FHE_H) + tid;
#pragma unroll
    for (int p = 0; p < K1; ++p) {
        double2 bv[K1];
#pragma unroll
        for (int c = 0; c < K1; ++c) bv[c] = G[(size_t)(p * K1 + c) * FHE_H];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 d = *reinterpret_cast<const double2 *>(lds + (r * K1 + p) * GROUP_TILE_DOUBLES + 2 * tid);
#pragma unroll
            for (int c = 0; c < K1; ++c) {
                fr[r][c] = __builtin_fma(d.x, bv[c].x, fr[r][c]);
                fr[r][c] = __builtin_fma(-d.y, bv[c].y, fr[r][c]);
                fi[r][c] = __builtin_fma(d.x, bv[c].y, fi[r][c]);
                fi[r][c] = __builtin_fma(d.y, bv[c].x, fi[r][c]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) {
            double2 v; v.x = fr[r][c]; v.y = fi[r][c];
            *reinterpret_cast<double2 *>(lds + (r * K1 + c) * GROUP_TILE_DOUBLES + 2 * tid) = v;
        }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        double2 v = *reinterpret_cast<const double2 *>(tile + 2 * (b + 16 * k2));
        xr[k2] = v.x; xi[k2] = v.y;
    }
    wave_lds_sync();
    nega_inv(xr, xi, tw, tile, b);
    if (valid) {
        uint64_t *o = A.out + ((inst * A.nodes_out + node) * K1 + p_own) * FHE_N;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            o[16 * a + b] = c0lo[a] + torus_from_double(xr[a]);
            o[256 + 16 * a + b] = c0hi[a] + torus_from_double(xi[a]);
        }
    }
}

// K4: `polys` torus polynomials (natural u64[512]) -> Fourier double2[256], one polynomial per lane group
__global__ __launch_bounds__(EP_THREADS) void forward_fourier_kernel(const uint64_t *in, double2 *out, uint64_t polys, const double2 *tw_g)
{
    __shared__ __attribute__((aligned(16))) double lds[EP_LDS_DOUBLES];
    double2 *tw = reinterpret_cast<double2 *>(lds + EP_GROUPS * GROUP_TILE_DOUBLES);
    const int tid = threadIdx.x, g = tid >> 4, b = tid & 15;
    double *tile = lds + g * GROUP_TILE_DOUBLES;
    ep_load_table(tw, tw_g);
    __syncthreads();
    for (uint64_t base = (uint64_t)blockIdx.x * EP_GROUPS; base < polys; base += (uint64_t)gridDim.x * EP_GROUPS) {
        uint64_t q = base + g;
        bool valid = q < polys;
        if (!valid) q = polys - 1;
        const uint64_t *p = in + q * FHE_N;
        double xr[16], xi[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            xr[a] = double_from_torus(p[16 * a + b]);
            xi[a] = double_from_torus(p[256 + 16 * a + b]);
        }
        nega_fwd(xr, xi, tw, tile, b);
        if (valid) {
            double2 *o = out + q * FHE_H;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) { double2 v; v.x = xr[k2]; v.y = xi[k2]; o[b + 16 * k2] = v; }
        }
    }
}
