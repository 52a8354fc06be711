// knobs.h -- fence around the developer knobs of the kernels (included first by engine.hip).
//
// kern_*.h / fft_dev.h / engine.hip carry `#ifndef KNOB / #define KNOB default` tuning knobs and `#ifdef ABLATION` switches that
// tools/ablate_*.py set with -D to time variants.  Several of them are WRONG-RESULT modes (phases compiled out for a timing
// proxy).  A product build must not be reachable by a stray -D: every knob named here is an #error unless the translation unit is
// compiled with -DFHEAES_DEV_BUILD, and fheaes_version() then says "dev" (tfhe_aes_amd/_native.py refuses such a library unless
// asked).  tests/test_cabi_cpu.py checks that this list names every knob the sources test and that _build.engine_flags() sets none.
#pragma once

#define FHEAES_KNOB_LIST(X) \
    /* wrong-result ablations and instrumented builds */ \
    X(EP_STAMPS) X(BR16_ABL_SAMEKEY) X(BR16_ABL_NOLOAD) X(BR16_ABL_NOMAC) X(BR16_ABL_NOFFT) X(BR16_ABL_NOXPOSE) X(BR16_ABL_NOPARK) X(BR16_ABL_HALFKEY) X(BRP_ABL_NOXSTORE) X(BRP_ABL_NODSTORE) X(BRP_ABL_NOBAR) X(BRP_ABL_NOPEEL) X(BRP_ABL_NOXREAD) X(BRP_ABL_SKEW) X(BRP_ABL_FEWCMUL) \
    X(BL_ABL_SAMEKEY) X(ABL_MAC_NOLOAD) X(ABL_NO_FFT) X(ABL_NO_MAC) X(ABL_MAC_NOLDS) X(BR16_PAD_DOUBLES) \
    /* equivalent-result tuning knobs (defaults in the headers are the measured best) */ \
    X(BR16_MAC_PRIO) X(BR16_EARLY) X(BR16_XPOSE_IN_TWIDDLE) X(BR16_STAGE_AT_END) X(BR16_HEAD) X(BR16_LATE_IN_PASS2) \
    X(BR16_READ_IN_PASS2) X(BR16_STORE_IN_PASS2) X(BR16_PARK_AUX_ST) X(BR16_PARK_AUX_LD) X(BR16_PARK_OWNERS_ONLY) X(BR16_W3_LDS_HOME) X(BR16_RESIDENT_HI) X(BR16_W1_LATE) X(BR16_MAC_TAIL) \
    X(BL_L2_PREFETCH) X(BL_ROWS_AHEAD) X(EP_EARLY_LOAD) X(EP_MAC_PRIO) X(EP_ROT_CHUNK) X(EP_PREFETCH) X(EP_KEY_AUX) X(EP_FENCE_MASK) X(EP_MIN_WAVES) \
    X(EP_LATE_BARRIER) X(FFT_XPOSE_PRIO) X(FFT_CHUNK) X(FFT_CHUNK_BARRIERS) X(FHE_TORUS_CONV_OLD) \
    X(LATENCY_BATCH_BITS) X(PBS_BALANCE) X(PBS_SMALL_R2) X(KS_LDS) X(KSL_SPLIT4) X(KSL_CT_TILES) X(KS1_LDS) X(K2_PAIR) X(K2_PAIR_MIN_BITS) X(K2_PAIR_TAIL4) X(BRP_EARLY) X(BRP_TAIL) X(BRP_RESIDENT_HI) X(BRP_W1_LATE) X(BRP_MAC_PRIO) X(BRP_CHUNK) X(BRP_SKIP_IDLE_WAVES)

#ifndef FHEAES_DEV_BUILD
#if defined(EP_STAMPS) || defined(BR16_ABL_SAMEKEY) || defined(BR16_ABL_NOLOAD) || defined(BR16_ABL_NOMAC) || defined(BR16_ABL_NOFFT) || \
    defined(BR16_ABL_NOXPOSE) || defined(BR16_ABL_NOPARK) || defined(BR16_ABL_HALFKEY) || defined(BRP_ABL_NOXSTORE) || defined(BRP_ABL_NODSTORE) || defined(BRP_ABL_NOBAR) || defined(BRP_ABL_NOPEEL) || defined(BRP_ABL_NOXREAD) || defined(BRP_ABL_SKEW) || defined(BRP_ABL_FEWCMUL) || defined(BL_ABL_SAMEKEY) || defined(ABL_MAC_NOLOAD) || defined(ABL_NO_FFT) || \
    defined(ABL_NO_MAC) || defined(ABL_MAC_NOLDS) || defined(BR16_PAD_DOUBLES) || \
    defined(BR16_MAC_PRIO) || defined(BR16_EARLY) || defined(BR16_XPOSE_IN_TWIDDLE) || defined(BR16_STAGE_AT_END) || defined(BR16_HEAD) || \
    defined(BR16_LATE_IN_PASS2) || defined(BR16_READ_IN_PASS2) || defined(BR16_STORE_IN_PASS2) || defined(BR16_PARK_AUX_ST) || \
    defined(BR16_PARK_AUX_LD) || defined(BR16_PARK_OWNERS_ONLY) || defined(BR16_W3_LDS_HOME) || defined(BR16_RESIDENT_HI) || defined(BR16_W1_LATE) || defined(BR16_MAC_TAIL) || defined(BL_L2_PREFETCH) || defined(BL_ROWS_AHEAD) || \
    defined(EP_EARLY_LOAD) || defined(EP_MAC_PRIO) || \
    defined(EP_ROT_CHUNK) || defined(EP_PREFETCH) || defined(EP_KEY_AUX) || defined(EP_FENCE_MASK) || defined(EP_MIN_WAVES) || \
    defined(EP_LATE_BARRIER) || defined(FFT_XPOSE_PRIO) || defined(FFT_CHUNK) || defined(FFT_CHUNK_BARRIERS) || defined(FHE_TORUS_CONV_OLD) || \
    defined(LATENCY_BATCH_BITS) || defined(PBS_BALANCE) || defined(PBS_SMALL_R2) || defined(KS_LDS) || defined(KSL_SPLIT4) || defined(KSL_CT_TILES) || defined(KS1_LDS) || defined(K2_PAIR) || defined(K2_PAIR_MIN_BITS) || defined(K2_PAIR_TAIL4) || \
    defined(BRP_EARLY) || defined(BRP_TAIL) || defined(BRP_RESIDENT_HI) || defined(BRP_W1_LATE) || defined(BRP_MAC_PRIO) || defined(BRP_CHUNK) || defined(BRP_SKIP_IDLE_WAVES)
#error "a developer knob of the kernels is defined on the command line: product builds take no knobs (add -DFHEAES_DEV_BUILD for a developer build; see csrc/knobs.h)"
#endif
#define FHEAES_BUILD_KIND ""
#else
#define FHEAES_BUILD_KIND " dev"
#endif
