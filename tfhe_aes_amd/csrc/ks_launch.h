// ks_launch.h -- the launches of the key-switching kernels (kern_keyswitch.h), as functions.
//
// Why functions: the product library is built from TWO translation units (tfhe_aes_amd/_build.py).  The blind rotation gains 1.6 %
// (200.5 -> 197.3 ms per 16,384-bit launch, same box, same words) when LLVM's post-register-allocation scheduler is switched off
// (-mllvm -enable-post-misched=false: it re-orders what kern_blindrot_pair.h has placed by hand), the packing key switch LOSES 4.6 %
// under the same flag (15.9 -> 16.6 ms: there it is the scheduler that interleaves the LDS fragment reads with the matrix
// instructions).  A compiler flag is per translation unit, so:
//   engine.hip        -DFHEAES_SPLIT_KS, post-RA scheduler off: everything but the key-switching kernels; it sees their
//                     constants and argument block only (FHEAES_KS_DECLS_ONLY) and calls the functions below;
//   keyswitch_tu.hip  the key-switching kernels and the definitions of these functions, default scheduler.
// Without -DFHEAES_SPLIT_KS (developer tools that compile engine.hip alone: tools/ablate_*.py, the ISA listings) the functions are
// defined right here and engine.hip is a complete library by itself.
#pragma once
#ifdef FHEAES_SPLIT_KS
#ifndef FHEAES_KS_TU
#define FHEAES_KS_DECLS_ONLY
#endif
#endif
#include "kern_keyswitch.h"

#if defined(FHEAES_SPLIT_KS) && !defined(FHEAES_KS_TU)
#define FHEAES_KS_LAUNCH_DECL
#else
#define FHEAES_KS_LAUNCH_DEFINE
#ifdef FHEAES_SPLIT_KS
#define FHEAES_KS_LAUNCH_LINKAGE
#else
#define FHEAES_KS_LAUNCH_LINKAGE static inline
#endif
#endif

#ifdef FHEAES_KS_LAUNCH_DECL
void ks_launch_digits_k1(dim3 grid, hipStream_t s, const uint64_t *in, uint64_t in_stride, uint32_t n_in, uint64_t m, uint32_t ksteps, int8_t *frag);
void ks_launch_digits_k3(dim3 grid, hipStream_t s, const uint64_t *in, uint64_t in_stride, uint32_t n_in, uint64_t m, uint32_t ksteps, int8_t *frag);
void ks_launch_mfma(int planes, dim3 grid, hipStream_t s, const KeyswitchArgs &a);
void ks_launch_mfma_lds(int planes, dim3 grid, hipStream_t s, const KeyswitchArgs &a);
void ks_launch_keybytes(dim3 grid, hipStream_t s, const uint64_t *key, uint64_t key_z_stride, uint32_t rows, uint32_t ncols, uint32_t ksteps,
                        uint32_t coltiles, int8_t *frag);
#endif

#ifdef FHEAES_KS_LAUNCH_DEFINE
// digits_kernel<2, 6, 1>: KS gadget (2^2, 6 levels), one digit plane
FHEAES_KS_LAUNCH_LINKAGE void ks_launch_digits_k1(dim3 grid, hipStream_t s, const uint64_t *in, uint64_t in_stride, uint32_t n_in, uint64_t m, uint32_t ksteps, int8_t *frag)
{
    hipLaunchKernelGGL((digits_kernel<2, 6, 1>), grid, dim3(256), 0, s, in, in_stride, n_in, m, ksteps, frag);
}
// digits_kernel<12, 3, 2>: PFKS gadget (2^12, 3 levels), two digit planes
FHEAES_KS_LAUNCH_LINKAGE void ks_launch_digits_k3(dim3 grid, hipStream_t s, const uint64_t *in, uint64_t in_stride, uint32_t n_in, uint64_t m, uint32_t ksteps, int8_t *frag)
{
    hipLaunchKernelGGL((digits_kernel<12, 3, 2>), grid, dim3(256), 0, s, in, in_stride, n_in, m, ksteps, frag);
}
FHEAES_KS_LAUNCH_LINKAGE void ks_launch_mfma(int planes, dim3 grid, hipStream_t s, const KeyswitchArgs &a)
{
    if (planes == 1) hipLaunchKernelGGL((keyswitch_mfma_kernel<1>), grid, dim3(KS_THREADS), 0, s, a);
    else hipLaunchKernelGGL((keyswitch_mfma_kernel<2>), grid, dim3(KS_THREADS), 0, s, a);
}
FHEAES_KS_LAUNCH_LINKAGE void ks_launch_mfma_lds(int planes, dim3 grid, hipStream_t s, const KeyswitchArgs &a)
{
    if (planes == 1) hipLaunchKernelGGL((keyswitch_mfma_lds_kernel<1>), grid, dim3(KSL_THREADS), 0, s, a);
    else hipLaunchKernelGGL((keyswitch_mfma_lds_kernel<2>), grid, dim3(KSL_THREADS), 0, s, a);
}
FHEAES_KS_LAUNCH_LINKAGE void ks_launch_keybytes(dim3 grid, hipStream_t s, const uint64_t *key, uint64_t key_z_stride, uint32_t rows, uint32_t ncols, uint32_t ksteps,
                                          uint32_t coltiles, int8_t *frag)
{
    hipLaunchKernelGGL(keybytes_kernel, grid, dim3(256), 0, s, key, key_z_stride, rows, ncols, ksteps, coltiles, frag);
}
#endif
