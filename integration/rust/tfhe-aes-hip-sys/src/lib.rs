//! Declarations of `include/tfhe_aes_hip.h`, one to one.  Not compiled in the build image (no Rust
//! toolchain there); kept in step with the header by `tests/test_abi.py`, which checks that every
//! function named here is exported by the library.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct tfa_params {
    pub lwe_dimension: u32,
    pub glwe_dimension: u32,
    pub polynomial_size: u32,
    pub pbs_base_log: u32,
    pub pbs_level: u32,
    pub ks_base_log: u32,
    pub ks_level: u32,
    pub lwe_noise_std: f64,
    pub glwe_noise_std: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct tfa_gate {
    pub op: u32,
    pub in0: u32,
    pub in1: u32,
    pub in2: u32,
    pub out: u32,
    /// bit k set: operand k is negated (ServerKey::not folded into the consumer)
    pub neg: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct tfa_stats {
    pub pbs_count: u64,
    pub ks_count: u64,
    pub gate_count: u64,
    pub stage_count: u64,
    pub pbs_kernel_ms: f64,
    pub ks_kernel_ms: f64,
    pub pbs_launches: u64,
    pub ks_launches: u64,
}

/// One timed launch (profiling on): kind 0 = the bootstraps of a stage, 1 = its keyswitches.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct tfa_launch_record {
    pub units: u32,
    pub kind: u32,
    pub mapping: u32,
    pub ms: f32,
}

#[repr(C)]
pub struct tfa_ctx {
    _private: [u8; 0],
}

pub const TFA_AND: u32 = 0;
pub const TFA_NAND: u32 = 1;
pub const TFA_OR: u32 = 2;
pub const TFA_NOR: u32 = 3;
pub const TFA_XOR: u32 = 4;
pub const TFA_XNOR: u32 = 5;
pub const TFA_MUX: u32 = 6;
pub const TFA_NOT: u32 = 7;
pub const TFA_COPY: u32 = 8;

pub const TFA_NOALLOC_MAX_GATES: u32 = 4096;
pub const TFA_NOALLOC_MAX_PBS: u32 = 4096;

pub const TFA_OK: c_int = 0;
pub const TFA_EINVAL: c_int = -1;
pub const TFA_EHIP: c_int = -2;
pub const TFA_ESTATE: c_int = -3;
pub const TFA_ENOMEM: c_int = -4;

extern "C" {
    pub fn tfa_default_params(p: *mut tfa_params);
    pub fn tfa_bsk_words(p: *const tfa_params) -> u64;
    pub fn tfa_ksk_words(p: *const tfa_params) -> u64;
    pub fn tfa_ctx_create(p: *const tfa_params, hip_device: c_int, out: *mut *mut tfa_ctx) -> c_int;
    pub fn tfa_ctx_destroy(ctx: *mut tfa_ctx) -> c_int;
    pub fn tfa_last_error(ctx: *mut tfa_ctx) -> *const c_char;
    pub fn tfa_keys_upload(ctx: *mut tfa_ctx, bsk: *const u32, ksk: *const u32) -> c_int;
    pub fn tfa_pool_reserve(ctx: *mut tfa_ctx, n_slots: u32) -> c_int;
    pub fn tfa_pool_write(ctx: *mut tfa_ctx, slots: *const u32, count: u32, lwe: *const u32) -> c_int;
    pub fn tfa_pool_read(ctx: *mut tfa_ctx, slots: *const u32, count: u32, lwe: *mut u32) -> c_int;
    pub fn tfa_pool_read_device(ctx: *mut tfa_ctx, slots: *const u32, negate: *const u8, count: u32, d_out: *mut c_void) -> c_int;
    pub fn tfa_pool_device_ptr(ctx: *mut tfa_ctx, dptr: *mut *mut c_void, stride_words: *mut u32, n_slots: *mut u32) -> c_int;
    pub fn tfa_stage_submit(ctx: *mut tfa_ctx, gates: *const tfa_gate, count: u32) -> c_int;
    /// `tfa_stage_submit` on the library's static device scratch: no allocation in the call, at most
    /// `TFA_NOALLOC_MAX_GATES` gates / `TFA_NOALLOC_MAX_PBS` bootstraps per call.
    pub fn tfa_stage_submit_noalloc(ctx: *mut tfa_ctx, gates: *const tfa_gate, count: u32) -> c_int;
    pub fn tfa_sync(ctx: *mut tfa_ctx) -> c_int;
    pub fn tfa_bootstrap_batch(ctx: *mut tfa_ctx, lwe_in: *const u32, count: u32, big_out: *mut u32) -> c_int;
    pub fn tfa_blind_rotate_batch(ctx: *mut tfa_ctx, lwe_in: *const u32, count: u32, steps: u32, acc_out: *mut u32) -> c_int;
    pub fn tfa_keyswitch_batch(ctx: *mut tfa_ctx, big_in: *const u32, count: u32, lwe_out: *mut u32) -> c_int;
    pub fn tfa_set_profiling(ctx: *mut tfa_ctx, on: c_int) -> c_int;
    pub fn tfa_get_stats(ctx: *mut tfa_ctx, out: *mut tfa_stats) -> c_int;
    pub fn tfa_reset_stats(ctx: *mut tfa_ctx) -> c_int;
    pub fn tfa_profile_launches(ctx: *mut tfa_ctx, out: *mut tfa_launch_record, capacity: u32, count: *mut u32) -> c_int;
    pub fn tfa_client_keygen(p: *const tfa_params, seed: u64, lwe_sk: *mut u32, glwe_sk: *mut u32, bsk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn tfa_client_encrypt(p: *const tfa_params, lwe_sk: *const u32, seed: u64, bits: *const u8, count: u32, lwe_out: *mut u32) -> c_int;
    pub fn tfa_client_decrypt(p: *const tfa_params, lwe_sk: *const u32, lwe: *const u32, count: u32, bits_out: *mut u8) -> c_int;
}
