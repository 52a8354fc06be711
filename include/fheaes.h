/*
 * fheaes.h -- C ABI of the MI355X-native FHE-AES engine (libfheaes.so).
 *
 * Drop-in boundary for the WoPBS S-Box hot path of rostin79s/TFHE-AES.  Every entry
 * point names the reference interface it replaces (paths relative to the reference
 * repository).  The reference is a pure-Rust crate with no FFI of its own; these are
 * the symbols a Rust `extern "C"` shim would bind.
 *
 * Conventions
 *   - all ciphertext / key words are uint64_t (torus q = 2^64), little endian, contiguous;
 *   - LWE ciphertext  = [a_0 .. a_{d-1}, b]                      (d+1 words);
 *   - GLWE ciphertext = [A_0 | .. | A_{k-1} | B], each N words   ((k+1)N words);
 *   - an AES byte     = 8 LWE ciphertexts under the big key (d = kN), block j = bit j (LSB first),
 *                       message bit at the MSB (delta = 2^63, no padding)   (client.rs:123-138);
 *   - an AES state    = 16 bytes, index = 4*col + row, byte 0 = MSB of the u128;
 *   - every function returns FHEAES_OK (0) or a negative error code and never unwinds;
 *     fheaes_last_error() gives the message.  Shape / parameter mismatches are errors;
 *   - `memspace` says where the data pointers of that call live: FHEAES_HOST (the engine
 *     stages them through HBM) or FHEAES_DEVICE (HBM pointers, work is enqueued on the
 *     context's stream and NOT synchronised: call fheaes_synchronize()).
 *   - per-call pointers are borrowed for the duration of the call; keys are copied into
 *     HBM by fheaes_upload_keys() and owned by the context (server.rs:32-35 takes keys by value).
 */
#ifndef FHEAES_H
#define FHEAES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHEAES_OK 0
#define FHEAES_ERR_INVALID -1  /* bad argument / shape / parameter set */
#define FHEAES_ERR_NOKEYS -2   /* evaluation before fheaes_upload_keys */
#define FHEAES_ERR_DEVICE -3   /* HIP runtime error (message in last_error) */
#define FHEAES_ERR_NOMEM -4

#define FHEAES_HOST 0
#define FHEAES_DEVICE 1

/* WopbsParameters of the reference (client.rs:31-57), minus the noise fields that only
 * key generation needs.  polynomial_size must be 512. */
typedef struct fheaes_params {
    uint32_t lwe_dimension;   /* n   = 669 */
    uint32_t glwe_dimension;  /* k   = 4   */
    uint32_t polynomial_size; /* N   = 512 */
    uint32_t pbs_base_log;    /* 8  */
    uint32_t pbs_level;       /* 5  */
    uint32_t ks_base_log;     /* 2  */
    uint32_t ks_level;        /* 6  */
    uint32_t pfks_base_log;   /* 12 */
    uint32_t pfks_level;      /* 3  */
    uint32_t cbs_base_log;    /* 15 */
    uint32_t cbs_level;       /* 1  */
} fheaes_params;

typedef struct fheaes_ctx fheaes_ctx;

/* ---- lifetime ------------------------------------------------------------------ */
/* Server::new (server.rs:32): create an engine on HIP device `device`. */
int fheaes_create(const fheaes_params *params, int device, fheaes_ctx **out);
void fheaes_destroy(fheaes_ctx *ctx);
/* message of the last failing call on this context (ctx == NULL: last fheaes_create failure).  The pointer is a per-thread copy,
 * valid until the same thread's next call into the library (a context may be shared between threads). */
const char *fheaes_last_error(const fheaes_ctx *ctx);

/* ---- keys ---------------------------------------------------------------------- */
/* word counts of the three evaluation keys for this parameter set */
#define FHEAES_KEY_KSK 0    /* [kN][ks_level][n+1]                 pbs_server_key.key_switching_key (many_wopbs.rs:168) */
#define FHEAES_KEY_BSK 1    /* [n][pbs_level][k+1][k+1][N]  STANDARD domain  wopbs_server_key.bootstrapping_key (many_wopbs.rs:34-35) */
#define FHEAES_KEY_PFPKSK 2 /* [k+1][kN+1][pfks_level][(k+1)N]     cbs_pfpksk (many_wopbs.rs:76) */
size_t fheaes_key_words(const fheaes_ctx *ctx, int which);

/* Copies the keys into HBM and converts the BSK to the engine's Fourier layout.
 * Level index 0 is the most significant level (weight 2^(64-base_log)). */
int fheaes_upload_keys(fheaes_ctx *ctx, const uint64_t *ksk, const uint64_t *bsk, const uint64_t *pfpksk, int memspace);

/* The same keys as (public mask key, bodies): every mask word of the three keys is the output of a public, counter-based
 * ChaCha20 stream -- mask word j of key ciphertext q of key t (t = 3 KSK, 4 BSK, 5 PFPKSK) is 64-bit word j % 8 of the
 * RFC 8439 block j / 8 under (mask_key, nonce = (t, q)) (csrc/client.c) -- so only the bodies travel: KSK [kN][ks_level]
 * words, BSK [n][pbs_level][k+1][N], PFPKSK [k+1][kN+1][pfks_level][N]: 0.19 GB instead of 1.04 GB at PARAM_OPT, and the masks
 * are regenerated on the GPU.  `mask_key` is a HOST array of 8 uint32 (256 bits) whatever `memspace` says about the bodies.
 * The counterpart in the reference's world are tfhe-rs' Seeded* key containers, which client.rs:106-107 does not use (it builds
 * full keys in memory and hands them over by value, server.rs:32); device keys are bit-identical to fheaes_upload_keys of the
 * expanded keys. */
size_t fheaes_key_body_words(const fheaes_ctx *ctx, int which);
int fheaes_upload_keys_seeded(fheaes_ctx *ctx, const uint32_t *mask_key, const uint64_t *ksk_body, const uint64_t *bsk_body,
                              const uint64_t *pfpksk_body, int memspace);

/* Several engines, one upload.  The reference is ONE process that fans the CTR blocks out over rayon worker threads which share
 * `&Server` (main.rs:55-64, server.rs:32-35): a drop-in that wants G GPUs (or several concurrent streams on one GPU) creates G
 * contexts, uploads the keys into the first and clones the CONVERTED key images (1.04 GB) into the others, device to device:
 * hipMemcpyPeerAsync over xGMI between GPUs, an HBM copy inside one.  Block i then goes to context i * G / n_blocks, one host
 * thread per context (contexts are independent: own stream, own workspace; calls on ONE context are serialised by its lock).
 * Both contexts must have been created with the same parameter set. */
int fheaes_clone_keys(fheaes_ctx *dst, fheaes_ctx *src);
/* How the last fheaes_clone_keys INTO `ctx` moved the key images: `path` = FHEAES_CLONE_NONE (no clone yet), _SAME_DEVICE (HBM copy),
 * _PEER (hipDeviceCanAccessPeer said yes and peer access is enabled: hipMemcpyPeerAsync is a direct xGMI transfer) or _STAGED (no
 * peer access between the two devices: the runtime stages the copy through host memory); `bytes` moved and wall `seconds` of the
 * copies.  Any of the three out-pointers may be NULL.  (The cross-device paths have not run on hardware yet: a builder's box has
 * one GPU.  The reference has no counterpart: its rayon workers share one `&Server` in host memory, main.rs:55-64.) */
#define FHEAES_CLONE_NONE 0
#define FHEAES_CLONE_SAME_DEVICE 1
#define FHEAES_CLONE_PEER 2
#define FHEAES_CLONE_STAGED 3
int fheaes_clone_info(fheaes_ctx *ctx, int *path, uint64_t *bytes, double *seconds);

/* ---- noise guard ---------------------------------------------------------------- */
/* The reference builds tfhe-rs with `noise-asserts` (Cargo.toml:7) under MaxNoiseLevel::new(5) (client.rs:92): a sum of more than
 * five nominal-noise ciphertexts between two bootstraps panics (many_wopbs.rs:101-108 resets every WoPBS output to NOMINAL).
 * What the engine has is a STATIC SCHEDULE ASSERTION, not runtime noise tracking: each linear layer of the engine's own AES schedule
 * declares how many WoPBS outputs it sums per output word (MixColumns + AddRoundKey = 4 + 1, the key-expansion sums = 2); a layer
 * whose gather table would sum more than the limit is refused with FHEAES_ERR_INVALID, and the largest count any call on this context
 * has declared can be read back.  Ciphertext words carry no noise metadata: a state that a caller has already summed before passing
 * it in counts as nominal here.  Callers that add ciphertext words themselves (the stage-level entry points hand out raw uint64
 * words) keep their own count, as users of tfhe-rs' `unchecked_*` do. */
#define FHEAES_MAX_NOISE_LEVEL 5
int fheaes_noise_level_seen(fheaes_ctx *ctx, uint32_t *max_seen, uint32_t *limit);

/* ---- stream / sync / workspace ------------------------------------------------- */
int fheaes_set_stream(fheaes_ctx *ctx, void *hip_stream); /* NULL: the context's own stream */
int fheaes_synchronize(fheaes_ctx *ctx);
/* pre-size the device workspace for batches of up to `max_bits` one-bit inputs in flight */
int fheaes_reserve(fheaes_ctx *ctx, uint64_t max_bits);

/* ---- the hot path, stage by stage (what many_wopbs.rs calls into tfhe 0.11.2) --- */
/* K1  shortint::wopbs::WopbsKey::extract_bits_assign (many_wopbs.rs:194): one LWE keyswitch
 *     big -> small per bit.  in [m][kN+1] -> out [m][n+1]. */
int fheaes_keyswitch_batch(fheaes_ctx *ctx, const uint64_t *lwe_in, uint64_t m, uint64_t *lwe_out, int memspace);
/* K2  the PBS inside circuit_bootstrap_boolean (many_wopbs.rs:253), CBS level `level` (1-based):
 *     in [m][n+1] -> out [m][kN+1] = LWE of bit * 2^(64 - cbs_base_log*level). */
int fheaes_cbs_pbs_batch(fheaes_ctx *ctx, const uint64_t *lwe_small, uint64_t m, uint32_t level, uint64_t *lwe_out, int memspace);
/* K3  the k+1 private functional packing keyswitches of circuit_bootstrap_boolean:
 *     in [m][kN+1] -> out [m][k+1][(k+1)N]  (one GGSW level, standard domain). */
int fheaes_pfpks_batch(fheaes_ctx *ctx, const uint64_t *lwe_in, uint64_t m, uint64_t *ggsw_rows_out, int memspace);
/* K4  ggsw.fill_with_forward_fourier (many_wopbs.rs:263): `polys` torus polynomials -> Fourier,
 *     out [polys][256][2] doubles (natural order, re/im interleaved). */
int fheaes_forward_fourier_batch(fheaes_ctx *ctx, const uint64_t *polys_in, uint64_t polys, double *fourier_out, int memspace);
/* K5  vertical_packing (many_wopbs.rs:277).  ggsw_fourier [n_inputs][bits][cbs_level][k+1][k+1][256][2];
 *     luts [n_sets][n_luts][bits][W] with n_sets = lut_per_input ? n_inputs : 1 and W = max(2^bits, N) words per
 *     (LUT, output bit) as gen_lut.rs:19-23 sizes them; out [n_inputs][n_luts][bits][kN+1].
 *     bits <= 9 (all the AES path uses): one LUT polynomial per output bit, blind rotation only.  9 < bits <= 16: the
 *     2^(bits-9) polynomials go through the CMUX tree over input bits 9..bits-1 first, then the rotation over bits 0..8
 *     ("parity unpinned" for bits > 9: the reference never calls many_wopbs_without_padding wider than 9 bits and holds no
 *     fixture for it; the split follows upstream vertical_packing and is checked against this repo's oracle only). */
int fheaes_vertical_packing_batch(fheaes_ctx *ctx, const double *ggsw_fourier, uint64_t n_inputs, uint32_t bits,
                                  const uint64_t *luts, uint32_t n_luts, int lut_per_input, uint64_t *lwe_out, int memspace);

/* ---- the plugin API of the path ------------------------------------------------ */
/* many_wopbs_without_padding (many_wopbs.rs:31), batched over radix inputs.
 *   lwe_in [n_inputs][bits][kN+1], bits in {1..16}; luts as above; out [n_inputs][n_luts][bits][kN+1]. */
int fheaes_wopbs_batch(fheaes_ctx *ctx, const uint64_t *lwe_in, uint64_t n_inputs, uint32_t bits,
                       const uint64_t *luts, uint32_t n_luts, int lut_per_input, uint64_t *lwe_out, int memspace);
/* gen_lut (gen_lut.rs:9) for message_modulus 2, carry_modulus 1: f_table[2^nb_block] -> out [nb_block][max(2^nb_block, N)]
 * (host only), nb_block in 1..16. */
int fheaes_gen_lut(uint32_t nb_block, const uint64_t *f_table, uint64_t *lut_out);
/* sbox (sbox.rs:46), in place over n_bytes bytes: bytes [n_bytes][8][kN+1]; inv = 0 SBOX, 1 INV_SBOX. */
int fheaes_sbox(fheaes_ctx *ctx, uint64_t *bytes, uint64_t n_bytes, int inv, int memspace);
/* many_sbox (sbox.rs:68): out [n_bytes][L][8][kN+1], L = 3 {S,2S,3S} (inv=0) or 4 {9x,11x,13x,14x} (inv=1). */
int fheaes_many_sbox(fheaes_ctx *ctx, const uint64_t *bytes, uint64_t n_bytes, int inv, uint64_t *out, int memspace);

/* ---- Server API (server.rs) ---------------------------------------------------- */
/* Server::aes_key_expansion (server.rs:107): key [16][8][kN+1] -> round_keys [11][16][8][kN+1]. */
int fheaes_aes_key_expansion(fheaes_ctx *ctx, const uint64_t *key, uint64_t *round_keys, int memspace);
/* Server::aes_encrypt (server.rs:39), batched: state [n_blocks][16][8][kN+1] in place, one set of round keys. */
int fheaes_aes_encrypt(fheaes_ctx *ctx, const uint64_t *round_keys, uint64_t *state, uint64_t n_blocks, int memspace);
/* Server::aes_decrypt (server.rs:67), batched. */
int fheaes_aes_decrypt(fheaes_ctx *ctx, const uint64_t *round_keys, uint64_t *state, uint64_t n_blocks, int memspace);
/* Server::add_scalar (server.rs:172), batched: state[b] += counters[b] (u128 as {hi, lo}, host array
 * of 2*n_blocks words regardless of memspace).  The first-byte carry uses counter & 0xFF (the
 * reference's server.rs:182 is wrong for counters >= 256). */
int fheaes_add_scalar(fheaes_ctx *ctx, uint64_t *state, uint64_t n_blocks, const uint64_t *counters_hi_lo, int memspace);

/* ---- measurement --------------------------------------------------------------- */
#define FHEAES_STAGE_KEYSWITCH 0
#define FHEAES_STAGE_BLIND_ROTATE 1
#define FHEAES_STAGE_PFPKS 2
#define FHEAES_STAGE_GGSW_FFT 3
#define FHEAES_STAGE_VERTICAL_PACKING 4
#define FHEAES_STAGE_LINEAR 5
#define FHEAES_STAGE_COUNT 6
/* When enabled every kernel launch is bracketed by HIP events on the launch stream. */
int fheaes_profile_enable(fheaes_ctx *ctx, int on);
int fheaes_profile_reset(fheaes_ctx *ctx);
/* synchronises, then returns accumulated kernel time, launches and units (bits or polys) of a stage */
int fheaes_profile_read(fheaes_ctx *ctx, int stage, double *total_ms, uint64_t *launches, uint64_t *units);

/* ---- introspection (parity tests) ---------------------------------------------- */
/* psi^j = exp(i*pi*j/512), j < 512, re/im interleaved: the twiddle table the kernels use */
int fheaes_get_twiddles(double *psi_out);
/* Fourier image of GGSW `i` of the uploaded BSK: out [pbs_level][k+1][k+1][256][2] */
int fheaes_read_bsk_fourier(fheaes_ctx *ctx, uint32_t i, double *out);
/* How a blind-rotation launch of `m` bits is cut into workgroups on a device with `cu_count` compute units at GLWE dimension k
 * (host logic only, no GPU needed, DEVICE-INDEPENDENT: the plan a device takes when every kernel form can be placed on it).
 * `form` 0 = latency form (kern_blindrot_latency.h: one ciphertext per 512-thread workgroup, m <= 256), 1 = 16-form
 * (kern_blindrot16.h: 256-thread workgroups of 3 / 2 ciphertexts, two per CU; 257..768 bits, and every batch at k = 1),
 * 2 = paired form (kern_blindrot_pair.h: ONE 512-thread workgroup per CU carrying 6 / 4 ciphertexts; k = 4, m > 768).
 * `units_main` workgroups of `r_main` ciphertexts are followed by `units_tail` of `r_tail`.  With more workgroups than the device has
 * slots (form 1: two per CU, form 2: one per CU) the counts make the launch a whole number of generations that covers the batch exactly. */
int fheaes_k2_launch_plan(uint64_t m, uint32_t cu_count, uint32_t k, int *form, uint64_t *units_main, uint32_t *r_main,
                          uint64_t *units_tail, uint32_t *r_tail);
/* The same for a CONTEXT: the form and the kernel this context really launches for a batch of `m` bits on its device, after the
 * occupancy fallbacks (the paired kernel needs 159,504 B of LDS per workgroup: where the runtime cannot place one on a CU every batch
 * takes form 1; the 16-form's LDS-home variant needs two workgroups of 81,920 B per CU, else its parked variant runs).  `kernel`
 * (may be NULL) receives the kernel's name, e.g. "blind_rotate_pair_kernel<5,5,8,3,2>".  Measurements must be labelled from this call. */
int fheaes_k2_context_plan(fheaes_ctx *ctx, uint64_t m, int *form, uint64_t *units_main, uint32_t *r_main, uint64_t *units_tail,
                           uint32_t *r_tail, char *kernel, size_t kernel_cap);
/* Where the paired blind-rotation kernel parks the half of its accumulators that does not fit a CU's registers and LDS: `claimed` = 1
 * (default): 64 KB slots of a shared pool, 128 per XCC, claimed with one compare-and-swap when a workgroup starts and released when it
 * ends, so that the slots the resident workgroups use stay in the caches; 0: one private slot per workgroup of the launch (64 KB x
 * the grid).  Same words either way (tests/test_gpu_fullsize.py).  The name `fheaes_k2_context_plan` reports carries the setting
 * (" parking=claimed" / " parking=private").  Ownership of a slot is recorded in memory and never inferred from the compute unit a
 * workgroup runs on: a queue preempted mid-kernel resumes its workgroups on other compute units (round 5's defect). */
int fheaes_k2_set_parking(fheaes_ctx *ctx, int claimed);
const char *fheaes_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FHEAES_H */
