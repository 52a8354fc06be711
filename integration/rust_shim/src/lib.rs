//! Rust binding of include/fheaes.h and a `GpuServer` with the reference's method names
//! (src/server/server.rs:32-274 of rostin79s/TFHE-AES).  UNCOMPILED sources (no Rust toolchain in the build
//! image: never built, never run); the tfhe-rs accessor names follow tfhe 0.11.2 as used by the reference
//! (many_wopbs.rs:34-35, :41, :76, :99-111, :168) and, where the reference does not show them, recall of the
//! crate (marked UPSTREAM-RECALL).
//!
//! Which reference function each method replaces:
//!   GpuServer::new                         Server::new                     src/server/server.rs:32
//!   GpuServer::sbox                        sbox                            src/server/sbox/sbox.rs:46   (callers server.rs:60,77,101)
//!   GpuServer::many_sbox                   many_sbox                       src/server/sbox/sbox.rs:68   (callers server.rs:48,89)
//!   GpuServer::many_wopbs_without_padding  many_wopbs_without_padding      src/server/sbox/many_wopbs.rs:31
//!   GpuServer::fhe_sub_word                fhe_sub_word                    src/server/key_expansion/key_expansion_utils.rs:24-28
//!   GpuServer::aes_key_expansion           Server::aes_key_expansion       src/server/server.rs:107
//!   GpuServer::aes_encrypt                 Server::aes_encrypt             src/server/server.rs:39   (batched over CTR blocks)
//!   GpuServer::aes_decrypt                 Server::aes_decrypt             src/server/server.rs:67   (batched)
//!   GpuServer::add_scalar                  Server::add_scalar              src/server/server.rs:172  (batched; carry defect of :182 fixed)
//!   GpuServer::clone_on                    (Server is shared by reference between rayon threads, main.rs:55-64; a GPU context
//!                                           is cloned instead: one PCIe key upload, device-to-device copies, fheaes_clone_keys)
//!   GpuServerGroup::{new, aes_encrypt, aes_decrypt, add_scalar}
//!                                          the rayon loop over CTR blocks of src/main.rs:55-64, one context per GPU
//!   fourier_bsk_to_standard                (no counterpart: the reference only holds the Fourier BSK, many_wopbs.rs:34-35)
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

use tfhe::core_crypto::fft_impl::fft64::crypto::bootstrap::FourierLweBootstrapKeyOwned; // the type behind many_wopbs.rs:34-35
use tfhe::core_crypto::fft_impl::fft64::math::fft::Fft;                                   // many_wopbs.rs:64
use tfhe::core_crypto::prelude::*;
use tfhe::integer::ciphertext::BaseRadixCiphertext;
use tfhe::integer::IntegerCiphertext;
use tfhe::shortint::parameters::{Degree, NoiseLevel};
use tfhe::shortint::{Ciphertext, WopbsParameters};

// ------------------------------------------------------------------------------------------------ FFI
#[repr(C)]
#[derive(Clone, Copy)]
pub struct fheaes_params {
    pub lwe_dimension: u32,
    pub glwe_dimension: u32,
    pub polynomial_size: u32,
    pub pbs_base_log: u32,
    pub pbs_level: u32,
    pub ks_base_log: u32,
    pub ks_level: u32,
    pub pfks_base_log: u32,
    pub pfks_level: u32,
    pub cbs_base_log: u32,
    pub cbs_level: u32,
}
#[repr(C)]
pub struct fheaes_ctx {
    _private: [u8; 0],
}
pub const FHEAES_HOST: c_int = 0;

extern "C" {
    pub fn fheaes_create(p: *const fheaes_params, device: c_int, out: *mut *mut fheaes_ctx) -> c_int;
    pub fn fheaes_destroy(ctx: *mut fheaes_ctx);
    pub fn fheaes_last_error(ctx: *const fheaes_ctx) -> *const c_char;
    pub fn fheaes_upload_keys(ctx: *mut fheaes_ctx, ksk: *const u64, bsk: *const u64, pfpksk: *const u64, memspace: c_int) -> c_int;
    pub fn fheaes_clone_keys(dst: *mut fheaes_ctx, src: *mut fheaes_ctx) -> c_int;
    pub fn fheaes_clone_info(ctx: *mut fheaes_ctx, path: *mut c_int, bytes: *mut u64, seconds: *mut f64) -> c_int;
    pub fn fheaes_noise_level_seen(ctx: *mut fheaes_ctx, max_seen: *mut u32, limit: *mut u32) -> c_int;
    pub fn fheaes_synchronize(ctx: *mut fheaes_ctx) -> c_int;
    /// where the paired blind rotation parks half of its accumulators: 1 = slots claimed from a shared pool (default), 0 = one private
    /// slot per workgroup; same words either way (a maintainer only needs it to rule the pool out when chasing a wrong result)
    pub fn fheaes_k2_set_parking(ctx: *mut fheaes_ctx, claimed: c_int) -> c_int;
    pub fn fheaes_wopbs_batch(ctx: *mut fheaes_ctx, lwe_in: *const u64, n_inputs: u64, bits: u32, luts: *const u64,
                              n_luts: u32, lut_per_input: c_int, lwe_out: *mut u64, memspace: c_int) -> c_int;
    pub fn fheaes_sbox(ctx: *mut fheaes_ctx, bytes: *mut u64, n_bytes: u64, inv: c_int, memspace: c_int) -> c_int;
    pub fn fheaes_many_sbox(ctx: *mut fheaes_ctx, bytes: *const u64, n_bytes: u64, inv: c_int, out: *mut u64, memspace: c_int) -> c_int;
    pub fn fheaes_aes_key_expansion(ctx: *mut fheaes_ctx, key: *const u64, round_keys: *mut u64, memspace: c_int) -> c_int;
    pub fn fheaes_aes_encrypt(ctx: *mut fheaes_ctx, round_keys: *const u64, state: *mut u64, n_blocks: u64, memspace: c_int) -> c_int;
    pub fn fheaes_aes_decrypt(ctx: *mut fheaes_ctx, round_keys: *const u64, state: *mut u64, n_blocks: u64, memspace: c_int) -> c_int;
    pub fn fheaes_add_scalar(ctx: *mut fheaes_ctx, state: *mut u64, n_blocks: u64, counters_hi_lo: *const u64, memspace: c_int) -> c_int;
}

// ------------------------------------------------------------------------------------------------ helpers
type Radix = BaseRadixCiphertext<Ciphertext>;

/// tfhe-rs keyswitch keys store, inside every input block, the LEAST significant level first; the engine
/// wants level index 0 = most significant (include/fheaes.h).  `block` = levels * row_words.
fn reverse_levels(src: &[u64], levels: usize, row_words: usize) -> Vec<u64> {
    let block = levels * row_words;
    let mut out = vec![0u64; src.len()];
    for (b_in, b_out) in src.chunks_exact(block).zip(out.chunks_exact_mut(block)) {
        for l in 0..levels {
            b_out[l * row_words..(l + 1) * row_words].copy_from_slice(&b_in[(levels - 1 - l) * row_words..(levels - l) * row_words]);
        }
    }
    out
}

/// one AES byte (radix of 8 one-bit blocks, block j = bit j) -> 8 * lwe_size words
fn flatten_radix(ct: &Radix, out: &mut Vec<u64>) {
    for b in ct.blocks() {
        out.extend_from_slice(b.ct.as_ref());
    }
}

/// many_wopbs.rs:87-115: re-wrap `bits` LWE ciphertexts as shortint blocks with the metadata of `like`
fn rewrap(words: &[u64], like: &Radix) -> Radix {
    let lwe_words = like.blocks()[0].ct.lwe_size().0;
    let blocks = like
        .blocks()
        .iter()
        .zip(words.chunks_exact(lwe_words))
        .map(|(b, w)| {
            Ciphertext::new(
                LweCiphertextOwned::from_container(w.to_vec(), b.ct.ciphertext_modulus()),
                Degree::new(b.message_modulus.0 - 1),
                NoiseLevel::NOMINAL,
                b.message_modulus,
                b.carry_modulus,
                b.pbs_order,
            )
        })
        .collect();
    Radix::from_blocks(blocks)
}

/// A whole AES state / key (16 bytes, index = 4*col + row, client.rs:126-129) -> 16 * 8 * lwe_size words
fn flatten_state(state: &[Radix], out: &mut Vec<u64>) {
    for b in state {
        flatten_radix(b, out);
    }
}

/// write `words` ([16][8][lwe_size]) back into the 16 radix bytes of `state` (metadata as many_wopbs.rs:87-115)
fn rewrap_state(words: &[u64], state: &mut [Radix]) {
    let byte_words = words.len() / state.len();
    for (b, w) in state.iter_mut().zip(words.chunks_exact(byte_words)) {
        *b = rewrap(w, b);
    }
}

/// The engine wants the bootstrapping key in the STANDARD domain ([n][level][k+1][k+1][N] torus words, level 0 = most
/// significant) and re-transforms it with its own FFT; the reference only holds tfhe-fft's Fourier image
/// (`wopbs_key.wopbs_server_key.bootstrapping_key`, many_wopbs.rs:34-35).  Two ways to get the standard words:
///   (a) preferred: keep the `LweBootstrapKeyOwned<u64>` that key generation produces before converting it -- a small change
///       to `Client::new` (client.rs:106-107: build the ServerKey / WopbsKey from core_crypto pieces instead of
///       `gen_keys_radix` + `WopbsKey::new_wopbs_key_only_for_wopbs`, which drop it);
///   (b) this function: run tfhe-fft backwards over every polynomial of the Fourier key.  The round trip costs at most the
///       f64 rounding of one forward + one backward transform per coefficient (~2^-50 relative, far below the key's
///       noise of 2^-52 * 2^64 * sigma_glwe).
/// UPSTREAM-RECALL: `FourierLweBootstrapKey::as_view().data()` is `[c64]` with the polynomials of every GGSW laid out
/// [input bit][level][row][column][N/2]; `FftView::backward_as_torus` adds nothing and writes the rounded torus values.
/// Both the level order inside a GGSW and the bit-reversed point order of tfhe-fft are internal to tfhe-rs, which is
/// exactly why the inverse is done with tfhe-rs' own Fft here and not re-implemented.
pub fn fourier_bsk_to_standard(fbsk: &FourierLweBootstrapKeyOwned, level_most_significant_first: bool) -> Vec<u64> {
    let n = fbsk.input_lwe_dimension().0;
    let glwe_size = fbsk.glwe_size().0;
    let poly = fbsk.polynomial_size();
    let levels = fbsk.decomposition_level_count().0;
    let fft = Fft::new(poly);
    let fft = fft.as_view();
    let mut mem = dyn_stack::GlobalPodBuffer::new(fft.backward_scratch().unwrap());
    let mut stack = dyn_stack::PodStack::new(&mut mem);
    let half = poly.0 / 2;
    let polys_per_ggsw = levels * glwe_size * glwe_size;
    let mut out = vec![0u64; n * polys_per_ggsw * poly.0];
    let view = fbsk.as_view();
    let data = view.data();                                               // &[c64]
    for i in 0..n {
        for l in 0..levels {
            // the engine's level index 0 is the most significant level; flip if tfhe-rs stores the least significant first
            let l_src = if level_most_significant_first { l } else { levels - 1 - l };
            for rc in 0..glwe_size * glwe_size {
                let src = ((i * levels + l_src) * glwe_size * glwe_size + rc) * half;
                let dst = ((i * levels + l) * glwe_size * glwe_size + rc) * poly.0;
                let fourier = FourierPolynomial { data: &data[src..src + half] };
                let mut torus = Polynomial::from_container(&mut out[dst..dst + poly.0]);
                fft.backward_as_torus(torus.as_mut_view(), fourier, stack.rb_mut());
            }
        }
    }
    out
}

// ------------------------------------------------------------------------------------------------ GpuServer
pub struct GpuServer {
    ctx: *mut fheaes_ctx,
    params: fheaes_params,
}
// one context = one HIP stream + one workspace; the engine serialises calls on a context (its own lock), so `&GpuServer` may
// be shared between threads exactly like the reference's `&Server` (main.rs:55-61) -- safe, but serialised: use one context
// per thread (GpuServerGroup) for concurrency
unsafe impl Send for GpuServer {}
unsafe impl Sync for GpuServer {}

impl GpuServer {
    /// Server::new (server.rs:32) from the reference's own key object alone: the Fourier BSK is taken back to the standard
    /// domain with tfhe-rs' Fft (fourier_bsk_to_standard, route (b)).  UPSTREAM-RECALL: the Classic arm of
    /// ShortintBootstrappingKey holds `bsk: FourierLweBootstrapKeyOwned` (the match of many_wopbs.rs:69-82).
    pub fn from_wopbs_key(wopbs_key_short: &tfhe::shortint::wopbs::WopbsKey, device: i32) -> Self {
        use tfhe::shortint::server_key::ShortintBootstrappingKey;
        let std_words = match &wopbs_key_short.wopbs_server_key.bootstrapping_key {
            ShortintBootstrappingKey::Classic(fbsk) => fourier_bsk_to_standard(fbsk, true),
            ShortintBootstrappingKey::MultiBit { .. } => panic!("multi-bit bootstrapping keys are not on this path (many_wopbs.rs:83-84 is a no-op)"),
        };
        let p = wopbs_key_short.param;
        let std_bsk = LweBootstrapKeyOwned::from_container(std_words, p.glwe_dimension.to_glwe_size(), p.polynomial_size, p.pbs_base_log,
                                                           p.pbs_level, wopbs_key_short.wopbs_server_key.ciphertext_modulus);
        Self::new(wopbs_key_short, &std_bsk, device)
    }

    /// Server::new (server.rs:32).  `std_bsk` is the standard-domain bootstrapping key (route (a) of
    /// fourier_bsk_to_standard: kept from key generation); the engine re-transforms it on upload.
    pub fn new(wopbs_key_short: &tfhe::shortint::wopbs::WopbsKey, std_bsk: &LweBootstrapKeyOwned<u64>, device: i32) -> Self {
        let p: WopbsParameters = wopbs_key_short.param;
        let params = fheaes_params {
            lwe_dimension: p.lwe_dimension.0 as u32,
            glwe_dimension: p.glwe_dimension.0 as u32,
            polynomial_size: p.polynomial_size.0 as u32,
            pbs_base_log: p.pbs_base_log.0 as u32,
            pbs_level: p.pbs_level.0 as u32,
            ks_base_log: p.ks_base_log.0 as u32,
            ks_level: p.ks_level.0 as u32,
            pfks_base_log: p.pfks_base_log.0 as u32,
            pfks_level: p.pfks_level.0 as u32,
            cbs_base_log: p.cbs_base_log.0 as u32,
            cbs_level: p.cbs_level.0 as u32,
        };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { fheaes_create(&params, device, &mut ctx) };
        assert!(rc == 0, "fheaes_create failed: {}", last_error(std::ptr::null()));
        let ksk = &wopbs_key_short.pbs_server_key.key_switching_key; // many_wopbs.rs:168
        let ksk_flat = reverse_levels(ksk.as_ref(), p.ks_level.0, p.lwe_dimension.0 + 1);
        let glwe_words = (p.glwe_dimension.0 + 1) * p.polynomial_size.0;
        let pf_flat = reverse_levels(wopbs_key_short.cbs_pfpksk.as_ref(), p.pfks_level.0, glwe_words); // many_wopbs.rs:76
        let rc = unsafe { fheaes_upload_keys(ctx, ksk_flat.as_ptr(), std_bsk.as_ref().as_ptr(), pf_flat.as_ptr(), FHEAES_HOST) };
        let s = GpuServer { ctx, params };
        assert!(rc == 0, "fheaes_upload_keys: {}", s.last_error());
        s
    }

    /// A second engine with the SAME keys on HIP device `device` (the same GPU or another one): the converted key images
    /// (1.04 GB) are copied device to device -- hipMemcpyPeerAsync over xGMI between GPUs -- instead of being uploaded
    /// and converted again.  Replaces sharing `&Server` between rayon workers (main.rs:55-64, server.rs:32-35).
    pub fn clone_on(&self, device: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { fheaes_create(&self.params, device, &mut ctx) };
        assert!(rc == 0, "fheaes_create failed: {}", last_error(std::ptr::null()));
        let s = GpuServer { ctx, params: self.params };
        let rc = unsafe { fheaes_clone_keys(s.ctx, self.ctx) };
        assert!(rc == 0, "fheaes_clone_keys: {}", s.last_error());
        s
    }

    /// How the keys of this (cloned) context got here: (path, bytes, seconds); path 1 = copy inside one GPU's HBM, 2 = direct xGMI
    /// peer copy, 3 = staged through host memory because the two GPUs have no peer access (include/fheaes.h: FHEAES_CLONE_*).
    pub fn clone_info(&self) -> (i32, u64, f64) {
        let (mut path, mut bytes, mut secs) = (0 as c_int, 0u64, 0f64);
        let rc = unsafe { fheaes_clone_info(self.ctx, &mut path, &mut bytes, &mut secs) };
        assert!(rc == 0, "{}", self.last_error());
        (path as i32, bytes, secs)
    }

    /// The engine's counterpart of tfhe-rs' `noise-asserts` (Cargo.toml:7, MaxNoiseLevel::new(5) at client.rs:92): the highest number
    /// of nominal-noise ciphertexts any linear layer of this context has summed between two bootstraps, and the limit.
    pub fn noise_level_seen(&self) -> (u32, u32) {
        let (mut seen, mut limit) = (0u32, 0u32);
        let rc = unsafe { fheaes_noise_level_seen(self.ctx, &mut seen, &mut limit) };
        assert!(rc == 0, "{}", self.last_error());
        (seen, limit)
    }

    pub fn synchronize(&self) {
        let rc = unsafe { fheaes_synchronize(self.ctx) };
        assert!(rc == 0, "{}", self.last_error());
    }

    pub fn last_error(&self) -> String {
        last_error(self.ctx)
    }

    /// many_wopbs_without_padding (many_wopbs.rs:31)
    pub fn many_wopbs_without_padding(&self, ct_in: &Radix, luts: &[tfhe::integer::wopbs::IntegerWopbsLUT]) -> Vec<Radix> {
        let bits = ct_in.blocks().len();
        let lwe_words = ct_in.blocks()[0].ct.lwe_size().0;
        let mut flat_in = Vec::with_capacity(bits * lwe_words);
        flatten_radix(ct_in, &mut flat_in);
        let mut flat_luts = Vec::new();
        for l in luts {
            flat_luts.extend_from_slice(l.as_ref().lut().as_ref()); // many_wopbs.rs:41
        }
        let mut out = vec![0u64; luts.len() * bits * lwe_words];
        let rc = unsafe {
            fheaes_wopbs_batch(self.ctx, flat_in.as_ptr(), 1, bits as u32, flat_luts.as_ptr(), luts.len() as u32, 0, out.as_mut_ptr(), FHEAES_HOST)
        };
        assert!(rc == 0, "{}", self.last_error()); // the reference panics too (many_wopbs.rs:125,143)
        out.chunks_exact(bits * lwe_words).map(|w| rewrap(w, ct_in)).collect()
    }

    /// Server::aes_encrypt (server.rs:39) over a whole vector of CTR blocks in one call (replaces the rayon loop of main.rs:55-64)
    pub fn aes_encrypt(&self, round_keys: &[Vec<Radix>], states: &mut [Vec<Radix>]) {
        let mut rk = Vec::new();
        for r in round_keys {
            for b in r {
                flatten_radix(b, &mut rk);
            }
        }
        let mut st = Vec::new();
        for s in states.iter() {
            for b in s {
                flatten_radix(b, &mut st);
            }
        }
        let rc = unsafe { fheaes_aes_encrypt(self.ctx, rk.as_ptr(), st.as_mut_ptr(), states.len() as u64, FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        let byte_words = st.len() / (states.len() * 16);
        for (s, words) in states.iter_mut().zip(st.chunks_exact(16 * byte_words)) {
            for (b, w) in s.iter_mut().zip(words.chunks_exact(byte_words)) {
                *b = rewrap(w, b);
            }
        }
    }

    /// sbox (sbox.rs:46), in place on one byte: the engine holds the {SBOX} / {INV_SBOX} LUT sets, so neither `gen_lut`
    /// (sbox.rs:54-60) nor the integer WopbsKey argument is needed.  Callers: server.rs:60, :77, :101.
    pub fn sbox(&self, ct_in: &mut Radix, inv: bool) {
        let mut flat = Vec::new();
        flatten_radix(ct_in, &mut flat);
        let rc = unsafe { fheaes_sbox(self.ctx, flat.as_mut_ptr(), 1, inv as c_int, FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        *ct_in = rewrap(&flat, ct_in);
    }

    /// The same over a whole vector of bytes in ONE launch sequence (16 * n_blocks bytes of an AES round): what the
    /// loops of server.rs:59-61 / :76-78 / :100-102 should call instead of 16 separate `sbox`.
    pub fn sbox_many_bytes(&self, bytes: &mut [Radix], inv: bool) {
        let mut flat = Vec::new();
        flatten_state(bytes, &mut flat);
        let rc = unsafe { fheaes_sbox(self.ctx, flat.as_mut_ptr(), bytes.len() as u64, inv as c_int, FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        rewrap_state(&flat, bytes);
    }

    /// many_sbox (sbox.rs:68): one set of 8 circuit bootstraps, 3 LUTs {S, 2S, 3S} (inv = false, consumed positionally by
    /// mix_columns.rs:36-75) or 4 LUTs {9x, 11x, 13x, 14x} (inv = true, inv_mix_columns.rs:17-55).  Callers: server.rs:48, :89.
    pub fn many_sbox(&self, ct_in: &Radix, inv: bool) -> Vec<Radix> {
        let n_luts = if inv { 4 } else { 3 };
        let mut flat = Vec::new();
        flatten_radix(ct_in, &mut flat);
        let mut out = vec![0u64; n_luts * flat.len()];
        let rc = unsafe { fheaes_many_sbox(self.ctx, flat.as_ptr(), 1, inv as c_int, out.as_mut_ptr(), FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        out.chunks_exact(flat.len()).map(|w| rewrap(w, ct_in)).collect()
    }

    /// fhe_sub_word (key_expansion_utils.rs:24-28): SBOX on the 4 bytes of a key-schedule word, one call
    pub fn fhe_sub_word(&self, word: &mut [Radix]) {
        self.sbox_many_bytes(word, false);
    }

    /// Server::aes_key_expansion (server.rs:107): encrypted key [16 bytes] -> 11 round keys.  RCON enters as a trivial
    /// encoding inside the engine (the reference encrypts it with the public key, server.rs:138-143: same plaintext effect).
    pub fn aes_key_expansion(&self, key: &[Radix]) -> Vec<Vec<Radix>> {
        let mut flat = Vec::new();
        flatten_state(key, &mut flat);
        let mut rk = vec![0u64; 11 * flat.len()];
        let rc = unsafe { fheaes_aes_key_expansion(self.ctx, flat.as_ptr(), rk.as_mut_ptr(), FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        let byte_words = flat.len() / key.len();
        rk.chunks_exact(flat.len())
            .map(|round| round.chunks_exact(byte_words).zip(key.iter()).map(|(w, like)| rewrap(w, like)).collect())
            .collect()
    }

    /// Server::aes_decrypt (server.rs:67) over a whole vector of blocks in one call
    pub fn aes_decrypt(&self, round_keys: &[Vec<Radix>], states: &mut [Vec<Radix>]) {
        let mut rk = Vec::new();
        for r in round_keys {
            flatten_state(r, &mut rk);
        }
        let mut st = Vec::new();
        for s in states.iter() {
            flatten_state(s, &mut st);
        }
        let rc = unsafe { fheaes_aes_decrypt(self.ctx, rk.as_ptr(), st.as_mut_ptr(), states.len() as u64, FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        let state_words = st.len() / states.len();
        for (s, words) in states.iter_mut().zip(st.chunks_exact(state_words)) {
            rewrap_state(words, s);
        }
    }

    /// Server::add_scalar (server.rs:172) for a vector of blocks: states[b] += counters[b] (the CTR loop of main.rs:59-61
    /// calls it with the block index).  The engine derives the first-byte carry from `counter & 0xFF`; server.rs:182 uses
    /// the whole counter and is wrong for counters >= 256.
    pub fn add_scalar(&self, states: &mut [Vec<Radix>], counters: &[u128]) {
        assert_eq!(states.len(), counters.len());
        let mut st = Vec::new();
        for s in states.iter() {
            flatten_state(s, &mut st);
        }
        let hi_lo: Vec<u64> = counters.iter().flat_map(|c| [(c >> 64) as u64, *c as u64]).collect();
        let rc = unsafe { fheaes_add_scalar(self.ctx, st.as_mut_ptr(), states.len() as u64, hi_lo.as_ptr(), FHEAES_HOST) };
        assert!(rc == 0, "{}", self.last_error());
        let state_words = st.len() / states.len();
        for (s, words) in states.iter_mut().zip(st.chunks_exact(state_words)) {
            rewrap_state(words, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GpuServerGroup
/// The reference's CTR loop (src/main.rs:55-64) runs blocks in parallel with rayon over ONE `&Server`.  On GPUs the unit of
/// parallelism is a context: `GpuServerGroup` holds one `GpuServer` per device (keys uploaded once, cloned device to device),
/// gives block i to context i * G / n_blocks (contiguous shards, no data moves between contexts) and drives every context
/// from its own host thread.  (bench.py reaches the same split with one PROCESS per GPU and an RCCL broadcast of the seeded
/// keys; this is the in-process form a drop-in for the reference needs.)
pub struct GpuServerGroup {
    servers: Vec<GpuServer>,
}

impl GpuServerGroup {
    pub fn new(wopbs_key_short: &tfhe::shortint::wopbs::WopbsKey, std_bsk: &LweBootstrapKeyOwned<u64>, devices: &[i32]) -> Self {
        assert!(!devices.is_empty());
        let first = GpuServer::new(wopbs_key_short, std_bsk, devices[0]);
        let mut servers = Vec::with_capacity(devices.len());
        for &d in &devices[1..] {
            servers.push(first.clone_on(d));
        }
        servers.insert(0, first);
        GpuServerGroup { servers }
    }

    /// [start, end) of the blocks of context `i` (the same rule as tfhe_aes_amd/dist.py::shard_blocks)
    fn shard(n: usize, g: usize, i: usize) -> (usize, usize) {
        let (base, rem) = (n / g, n % g);
        let start = i * base + i.min(rem);
        (start, start + base + usize::from(i < rem))
    }

    fn fan_out<F>(&self, states: &mut [Vec<Radix>], f: F)
    where
        F: Fn(&GpuServer, &mut [Vec<Radix>], usize) + Sync,
    {
        let g = self.servers.len();
        let n = states.len();
        std::thread::scope(|scope| {
            let mut rest = states;
            for (i, srv) in self.servers.iter().enumerate() {
                let (lo, hi) = Self::shard(n, g, i);
                let (mine, tail) = rest.split_at_mut(hi - lo);
                rest = tail;
                let f = &f;
                scope.spawn(move || {
                    if !mine.is_empty() {
                        f(srv, mine, lo);
                    }
                });
            }
        });
    }

    /// Server::aes_encrypt (server.rs:39) over all CTR blocks, sharded over the contexts
    pub fn aes_encrypt(&self, round_keys: &[Vec<Radix>], states: &mut [Vec<Radix>]) {
        self.fan_out(states, |srv, shard, _| srv.aes_encrypt(round_keys, shard));
    }

    /// Server::aes_decrypt (server.rs:67)
    pub fn aes_decrypt(&self, round_keys: &[Vec<Radix>], states: &mut [Vec<Radix>]) {
        self.fan_out(states, |srv, shard, _| srv.aes_decrypt(round_keys, shard));
    }

    /// Server::add_scalar (server.rs:172): states[b] += counters[b]
    pub fn add_scalar(&self, states: &mut [Vec<Radix>], counters: &[u128]) {
        assert_eq!(states.len(), counters.len());
        self.fan_out(states, |srv, shard, lo| srv.add_scalar(shard, &counters[lo..lo + shard.len()]));
    }

    /// Server::aes_key_expansion (server.rs:107): a latency chain of 50 dependent steps, one context
    pub fn aes_key_expansion(&self, key: &[Radix]) -> Vec<Vec<Radix>> {
        self.servers[0].aes_key_expansion(key)
    }
}

impl Drop for GpuServer {
    fn drop(&mut self) {
        unsafe { fheaes_destroy(self.ctx) }
    }
}

fn last_error(ctx: *const fheaes_ctx) -> String {
    unsafe { CStr::from_ptr(fheaes_last_error(ctx)).to_string_lossy().into_owned() }
}
