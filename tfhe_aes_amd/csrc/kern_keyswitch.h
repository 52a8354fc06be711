// kern_keyswitch.h -- the two integer key-switching stages:
//   K1  LWE keyswitch big -> small ("bit extraction")
//   K3  private functional packing keyswitch LWE -> GLWE, k+1 keys (8 a13)
// Both are  out[m][o] = (body) - sum_{i,l} digit_l(in[m][i]) * KEY[i][l][o]   in uint64 wrapping arithmetic.
// Over a batch of M ciphertexts this is an exact integer matrix product
//       D[M x Q] * KEY[Q x ncols]  (mod 2^64),   Q = inputs x levels  (6,147 x 12,800 per bit for K3),
// and at the batch sizes of the path (M = 16,384 per AES round) it is arithmetic-bound, not HBM-bound:
// the v_mad_u64_u32 form measured 8.9 T MAC/s (58 % of the half-rate integer-multiply issue limit).
// The kernels below slice it exactly onto the i8 matrix cores instead:
//   KEY = sum_j kb_j * 2^(8j)  with balanced bytes kb_j in [-128,127]   (precomputed at key upload)
//   d   = dlo + 256*dhi        with dlo in [-128,127], dhi small        (digits_kernel, once per launch)
//   d*KEY = sum_{j,k} 2^(8(j+k)) * (d_k * kb_j), j+k <= 7: 15 (K3) or 8 (K1) int8 products per u64 product,
// each accumulated exactly in int32 by v_mfma_i32_16x16x64_i8 (|sum| <= Q*128*128 < 2^31) and recombined
// with shifts in uint64.  Bit-exact against the oracle by construction (integer arithmetic, any order).
//
// Operands are stored in HBM directly in MFMA fragment order, so every lane fetches its 16 operand bytes
// with one coalesced 16-byte load (1 KB per wave per fragment) and nothing is staged through LDS:
//   A (digits): [ct tile of 16][kstep][plane][lane 64][16 B]   lane l: ciphertext 16T + l%16, rows 64*kstep + 16*(l/16) + e
//   B (key)   : [z][kstep][column tile of 16][byte plane j][lane 64][16 B]   lane l: column 16C + l%16, same rows
// One wave owns a 64-ciphertext x 16-column output tile (4 x 8 int32x4 accumulators); the 4 waves of a
// workgroup take 4 adjacent column tiles of the same 64 ciphertexts, so their A loads coincide in L1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_dev.h"

#define KS_THREADS 256
#define KS_CT_TILE 64          /* ciphertexts per workgroup */
#define KS_KSTEP 64            /* K rows per MFMA */

typedef int ks_int4 __attribute__((ext_vector_type(4)));

// With FHEAES_KS_DECLS_ONLY only the constants and the argument block are visible (engine.hip in the two-unit product build, where the
// kernels of this file are compiled in keyswitch_tu.hip under other scheduler flags: csrc/ks_launch.h).
struct KeyswitchArgs {
    const int8_t *afrag;       // [ct tiles][ksteps][PLANES][64][16]
    const int8_t *bfrag;       // [z][ksteps][coltiles][8][64][16]
    uint32_t ksteps, coltiles;
    const uint64_t *in;        // original LWE words (for the body term of K1)
    uint64_t in_stride;
    int32_t body_index;        // >= 0: in[m][body_index] is added to column body_col; < 0: none
    uint32_t body_col;
    uint32_t ncols;
    uint64_t *out;             // column o of key z at out[m*out_stride + z*out_z_stride + o]
    uint64_t out_stride, out_z_stride;
    uint64_t m;
};

// ---- LDS-tiled form: 512 threads = 8 waves own a 128-ciphertext x 64-column output tile -------------------
// The one-wave-one-tile kernel above is bound by the L2 -> CU operand stream (40 KB per 240 MFMAs: measured
// 36 % of the int8 MFMA rate).  Here the 8 digit fragments and the 32 key fragments of a K step are brought in
// ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPR staging; the
// fragments are already stored in MFMA lane order, so the LDS image is the HBM image) and read by every wave
// that needs them: 48 KB per 480 MFMAs.  Double buffered: the loads of step ks+1 fly during the MFMAs of ks.
#ifndef KSL_CT_TILES
#define KSL_CT_TILES 8          /* 8: 128 ciphertexts per 512-thread workgroup, one workgroup per CU (96 KB of LDS); 4: 64 ciphertexts per 256-thread
                                   workgroup, two independent workgroups per CU (80 KB each: no common barrier, the key fragments come in twice) */
#endif
#define KSL_THREADS (64 * KSL_CT_TILES)
#define KSL_COL_TILES 4         /* 64 columns */

#ifndef FHEAES_KS_DECLS_ONLY
// ---- key bytes: u64 KEY[z][rows][ncols] -> balanced int8 planes in B-fragment order (run once at upload) ----
__global__ __launch_bounds__(256) void keybytes_kernel(const uint64_t *key, uint64_t key_z_stride, uint32_t rows, uint32_t ncols,
                                                       uint32_t ksteps, uint32_t coltiles, int8_t *frag)
{
    // one thread per (kstep, coltile, lane); key index z = blockIdx.y
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = gid & 63;
    const uint64_t t = gid >> 6;
    const uint32_t C = t % coltiles;
    const uint32_t ks = (uint32_t)(t / coltiles);
    const uint32_t z = blockIdx.y;
    if (ks >= ksteps) return;
    const uint32_t col = C * 16 + (lane & 15);
    const uint32_t row0 = ks * KS_KSTEP + (lane >> 4) * 16;
    int8_t b[8][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        uint64_t x = 0;
        if (col < ncols && row0 + e < rows) x = key[(uint64_t)z * key_z_stride + (uint64_t)(row0 + e) * ncols + col];
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t v = (uint32_t)((x >> (8 * j)) & 0xFF) + carry;      // 0..256
            carry = v >= 128 ? 1u : 0u;
            b[j][e] = (int8_t)(v & 0xFF);                                 // v - 256*carry as a signed byte
        }
    }
    int8_t *o = frag + ((((uint64_t)z * ksteps + ks) * coltiles + C) * 8) * 1024 + (uint64_t)lane * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ks_int4 w;
        memcpy(&w, b[j], 16);
        *reinterpret_cast<ks_int4 *>(o + (uint64_t)j * 1024) = w;
    }
}

// ---- digits: LWE words -> int8 digit planes in A-fragment order ------------------------------------------
template <int BASE_LOG, int LEVELS, int PLANES>
__global__ __launch_bounds__(256) void digits_kernel(const uint64_t *in, uint64_t in_stride, uint32_t n_in, uint64_t m,
                                                     uint32_t ksteps, int8_t *frag)
{
    // one thread per (ct tile T, kstep, lane): produces rows q0 .. q0+15 of ciphertext 16T + lane%16
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = gid & 63;
    const uint64_t t = gid >> 6;
    const uint32_t ks = t % ksteps;
    const uint64_t T = t / ksteps;
    const uint64_t ct = T * 16 + (lane & 15);
    const uint32_t q0 = ks * KS_KSTEP + (lane >> 4) * 16;
    int8_t lo[16], hi[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { lo[e] = 0; hi[e] = 0; }
    if (ct < m) {
        const uint32_t i0 = q0 / LEVELS;
        constexpr int NI = (16 + LEVELS - 1) / LEVELS + 1;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const uint32_t i = i0 + ii;
            if (i >= n_in) break;
            int d[LEVELS];
            decompose_all<BASE_LOG, LEVELS>(in[ct * in_stride + i], d);
#pragma unroll
            for (int l = 0; l < LEVELS; ++l) {
                const int e = (int)(i * LEVELS + l) - (int)q0;
                if (e >= 0 && e < 16) {
                    const int dl = ((d[l] + 128) & 255) - 128;
#pragma unroll
                    for (int ee = 0; ee < 16; ++ee) if (ee == e) { lo[ee] = (int8_t)dl; hi[ee] = (int8_t)((d[l] - dl) >> 8); }
                }
            }
        }
    }
    int8_t *o = frag + ((T * ksteps + ks) * PLANES) * 1024 + (uint64_t)lane * 16;
    ks_int4 w;
    memcpy(&w, lo, 16);
    *reinterpret_cast<ks_int4 *>(o) = w;
    if (PLANES == 2) {
        memcpy(&w, hi, 16);
        *reinterpret_cast<ks_int4 *>(o + 1024) = w;
    }
}

template <int PLANES>
__global__ __launch_bounds__(KS_THREADS, 2) void keyswitch_mfma_kernel(const KeyswitchArgs A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t C = blockIdx.x * 4 + wave;                 // this wave's column tile
    if (C >= A.coltiles) return;                              // whole wave exits together (no barriers in this kernel)
    const uint64_t T0 = (uint64_t)blockIdx.y * (KS_CT_TILE / 16);
    const uint32_t z = blockIdx.z;

    ks_int4 acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[t][s] = (ks_int4){0, 0, 0, 0};

    const int8_t *ap = A.afrag + (T0 * A.ksteps * PLANES) * 1024 + (uint64_t)lane * 16;
    const int8_t *bp = A.bfrag + (((uint64_t)z * A.ksteps * A.coltiles + C) * 8) * 1024 + (uint64_t)lane * 16;
    const uint64_t a_tile_stride = (uint64_t)A.ksteps * PLANES * 1024;      // between ciphertext tiles
    const uint64_t b_step_stride = (uint64_t)A.coltiles * 8 * 1024;         // between k steps

    for (uint32_t ks = 0; ks < A.ksteps; ++ks) {
        ks_int4 a[4][PLANES], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const ks_int4 *>(bp + (uint64_t)ks * b_step_stride + (uint64_t)s * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
                a[t][pl] = *reinterpret_cast<const ks_int4 *>(ap + (uint64_t)t * a_tile_stride + ((uint64_t)ks * PLANES + pl) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][0], b[s], acc[t][s], 0, 0, 0);
            if (PLANES == 2 && s >= 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][1], b[s - 1], acc[t][s], 0, 0, 0);
            }
        }
    }

    // D layout of v_mfma_*_16x16: lane l holds column l%16, rows 4*(l/16) + r, r = 0..3
    const uint32_t col = C * 16 + (lane & 15);
    if (col >= A.ncols) return;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t ct = (T0 + t) * 16 + (uint64_t)(lane >> 4) * 4 + r;
            if (ct >= A.m) continue;
            uint64_t v = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) v += (uint64_t)(int64_t)acc[t][s][r] << (8 * s);
            v = (uint64_t)0 - v;
            if (A.body_index >= 0 && col == A.body_col) v += A.in[ct * A.in_stride + (uint32_t)A.body_index];
            A.out[ct * A.out_stride + (uint64_t)z * A.out_z_stride + col] = v;
        }
}


template <int PLANES>
__global__ __launch_bounds__(KSL_THREADS, 2) void keyswitch_mfma_lds_kernel(const KeyswitchArgs A)
{
    constexpr int A_FRAGS = KSL_CT_TILES * PLANES;
    constexpr int B_FRAGS = KSL_COL_TILES * 8;
    constexpr int FRAGS = A_FRAGS + B_FRAGS;                  // 1 KB each
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * FRAGS * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = wave & 3, wt = wave >> 2;                  // column tile / ciphertext half of this wave
    const uint32_t C0 = blockIdx.x * KSL_COL_TILES;
    const uint64_t T0 = (uint64_t)blockIdx.y * KSL_CT_TILES;
    const uint32_t z = blockIdx.z;
    const uint64_t ct_tiles_total = (A.m + 15) / 16;

    ks_int4 acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[t][s] = (ks_int4){0, 0, 0, 0};

    // fragment f of a K step: f < A_FRAGS: digits (ct tile f / PLANES, plane f % PLANES); else key (col tile, byte plane)
    auto issue = [&](uint32_t ks, int buf) {
        constexpr int WAVES = KSL_THREADS / 64;
#pragma unroll
        for (int i = 0; i < (FRAGS + WAVES - 1) / WAVES; ++i) {
            const int f = wave + WAVES * i;                    // wave-uniform
            if (f < FRAGS) {
                const int8_t *src;
                if (f < A_FRAGS) {
                    uint64_t T = T0 + f / PLANES;
                    if (T >= ct_tiles_total) T = ct_tiles_total - 1;          // clamp: rows beyond m are never stored
                    src = A.afrag + ((T * A.ksteps + ks) * PLANES + (f % PLANES)) * 1024;
                } else {
                    const int g = f - A_FRAGS;
                    uint32_t C = C0 + g / 8;
                    if (C >= A.coltiles) C = A.coltiles - 1;
                    src = A.bfrag + ((((uint64_t)z * A.ksteps + ks) * A.coltiles + C) * 8 + (g % 8)) * 1024;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + ((size_t)buf * FRAGS + f) * 1024), 16, 0, 0);
            }
        }
    };

    issue(0, 0);
    for (uint32_t ks = 0; ks < A.ksteps; ++ks) {
        const int buf = ks & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's fragments of step ks have landed
        __syncthreads();                                      // everybody's have, and step ks-1 is fully consumed
        if (ks + 1 < A.ksteps) issue(ks + 1, buf ^ 1);
        const int8_t *base = lds + (size_t)buf * FRAGS * 1024 + lane * 16;
        ks_int4 a[4][PLANES], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const ks_int4 *>(base + (A_FRAGS + wc * 8 + s) * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) a[t][pl] = *reinterpret_cast<const ks_int4 *>(base + ((wt * 4 + t) * PLANES + pl) * 1024);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][0], b[s], acc[t][s], 0, 0, 0);
            if (PLANES == 2 && s >= 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][1], b[s - 1], acc[t][s], 0, 0, 0);
            }
        }
    }

    const uint32_t C = C0 + wc;
    const uint32_t col = C * 16 + (lane & 15);
    if (C >= A.coltiles || col >= A.ncols) return;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t ct = (T0 + wt * 4 + t) * 16 + (uint64_t)(lane >> 4) * 4 + r;
            if (ct >= A.m) continue;
            uint64_t v = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) v += (uint64_t)(int64_t)acc[t][s][r] << (8 * s);
            v = (uint64_t)0 - v;
            if (A.body_index >= 0 && col == A.body_col) v += A.in[ct * A.in_stride + (uint32_t)A.body_index];
            A.out[ct * A.out_stride + (uint64_t)z * A.out_z_stride + col] = v;
        }
}
#endif  // FHEAES_KS_DECLS_ONLY
