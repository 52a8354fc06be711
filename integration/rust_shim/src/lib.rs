//! Rust binding of include/fheaes.h and a `GpuServer` with the reference's method names
//! (src/server/server.rs:32-274 of rostin79s/TFHE-AES).  UNCOMPILED sources (no Rust toolchain in the build
//! image); the tfhe-rs accessor names follow tfhe 0.11.2 as used by the reference
//! (many_wopbs.rs:34-35, :41, :76, :99-111, :168).
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

use tfhe::core_crypto::prelude::*;
use tfhe::integer::ciphertext::BaseRadixCiphertext;
use tfhe::integer::IntegerCiphertext;
use tfhe::shortint::parameters::{Degree, NoiseLevel};
use tfhe::shortint::{Ciphertext, WopbsParameters};

// ------------------------------------------------------------------------------------------------ FFI
#[repr(C)]
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

// ------------------------------------------------------------------------------------------------ GpuServer
pub struct GpuServer {
    ctx: *mut fheaes_ctx,
}
unsafe impl Send for GpuServer {}

impl GpuServer {
    /// Server::new (server.rs:32).  `std_bsk` is the standard-domain bootstrapping key of the ServerKey
    /// (keep it from key generation, or convert the Fourier key back with tfhe-rs); the engine re-transforms it.
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
        let s = GpuServer { ctx };
        assert!(rc == 0, "fheaes_upload_keys: {}", s.last_error());
        s
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
    // aes_decrypt, aes_key_expansion and add_scalar follow the same flatten -> call -> rewrap pattern
    // (fheaes_aes_decrypt, fheaes_aes_key_expansion, fheaes_add_scalar).
}

impl Drop for GpuServer {
    fn drop(&mut self) {
        unsafe { fheaes_destroy(self.ctx) }
    }
}

fn last_error(ctx: *const fheaes_ctx) -> String {
    unsafe { CStr::from_ptr(fheaes_last_error(ctx)).to_string_lossy().into_owned() }
}
