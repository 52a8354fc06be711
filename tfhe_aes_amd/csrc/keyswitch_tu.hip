// keyswitch_tu.hip -- second translation unit of the product library: the key-switching kernels (K1, K3, the key-byte conversion) and
// their launch functions, compiled with LLVM's default schedulers; everything else is engine.hip, compiled with the post-RA scheduler
// off.  Why two units: csrc/ks_launch.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>

#include "knobs.h"
#ifndef FHEAES_SPLIT_KS
#error "keyswitch_tu.hip is the second unit of the split build: compile it with -DFHEAES_SPLIT_KS (tfhe_aes_amd/_build.py)"
#endif
#define FHEAES_KS_TU
#include "ks_launch.h"
