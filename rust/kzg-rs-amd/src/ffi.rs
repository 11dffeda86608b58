//! The C ABI of `include/kzg_rs_amd.h`, and the mapping of its return codes onto `KzgError`.
use crate::enums::KzgError;
use alloc::string::{String, ToString};
use core::ffi::{c_char, c_int, CStr};

#[repr(C)]
pub struct RawSettings {
    _private: [u8; 0],
}

pub const KZG_OK: c_int = 0;
pub const KZG_BADARGS: c_int = 1;
pub const KZG_ERROR: c_int = 2;
pub const KZG_MALLOC: c_int = 3;
pub const KZG_INVALID_LENGTH: c_int = 4;
pub const KZG_BAD_SETUP: c_int = 5;

extern "C" {
    pub fn kzg_settings_load_trusted_setup(out: *mut *mut RawSettings, txt: *const c_char, len: usize) -> c_int;
    pub fn kzg_settings_load_trusted_setup_devices(out: *mut *mut RawSettings, txt: *const c_char, len: usize, devices: *const c_int, n_devices: usize) -> c_int;
    pub fn kzg_settings_from_tau_g2(out: *mut *mut RawSettings, tau_g2: *const u8) -> c_int;
    pub fn kzg_settings_from_tau_g2_devices(out: *mut *mut RawSettings, tau_g2: *const u8, devices: *const c_int, n_devices: usize) -> c_int;
    pub fn kzg_settings_devices(s: *const RawSettings, n_devices: *mut usize, devices_out: *mut c_int, cap: usize, exchange: *mut c_int) -> c_int;
    pub fn kzg_settings_free(s: *mut RawSettings);
    pub fn kzg_settings_note(s: *const RawSettings) -> *const c_char;
    pub fn kzg_settings_root_of_unity(s: *const RawSettings, i: usize, out: *mut u8) -> c_int;
    pub fn kzg_settings_g1_point(s: *const RawSettings, i: usize, out: *mut u8) -> c_int;
    pub fn kzg_settings_g2_point(s: *const RawSettings, i: usize, out: *mut u8) -> c_int;
    pub fn kzg_verify_kzg_proof(ok: *mut bool, commitment: *const u8, z: *const u8, y: *const u8, proof: *const u8, s: *const RawSettings) -> c_int;
    pub fn kzg_verify_kzg_proof_batch(ok: *mut bool, commitments: *const u8, zs: *const u8, ys: *const u8, proofs: *const u8, n: usize, s: *const RawSettings) -> c_int;
    pub fn kzg_verify_kzg_proofs(ok_out: *mut bool, err_out: *mut u8, commitments: *const u8, zs: *const u8, ys: *const u8, proofs: *const u8, n: usize, s: *const RawSettings) -> c_int;
    pub fn kzg_verify_blob_kzg_proof(ok: *mut bool, blob: *const u8, commitment: *const u8, proof: *const u8, s: *const RawSettings) -> c_int;
    pub fn kzg_verify_blob_kzg_proof_batch(ok: *mut bool, blobs: *const u8, commitments: *const u8, proofs: *const u8, n: usize, s: *const RawSettings) -> c_int;
    pub fn kzg_pairings_verify(ok: *mut bool, a1: *const u8, a2: *const u8, b1: *const u8, b2: *const u8, s: *const RawSettings) -> c_int;
    pub fn kzg_g1_msm(out: *mut u8, points48: *const u8, scalars: *const u8, n: usize, s: *const RawSettings) -> c_int;
    pub fn kzg_g1_msm_setup(out: *mut u8, scalars: *const u8, n: usize, s: *const RawSettings) -> c_int;
    pub fn kzg_last_error() -> *const c_char;
}

/// The thread-local message of the last failed call on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = kzg_last_error();
        if p.is_null() {
            return String::new();
        }
        CStr::from_ptr(p).to_string_lossy().to_string()
    }
}

/// What a successful constructor wants its caller to know about the handle ("" = nothing): fewer than 8 HIP hardware queues,
/// how a multi-device handle exchanges its partial sums and what its self-test found.
pub fn settings_note(h: &Handle) -> String {
    unsafe {
        let p = kzg_settings_note(h.0);
        if p.is_null() {
            return String::new();
        }
        CStr::from_ptr(p).to_string_lossy().to_string()
    }
}

/// `KzgRet` -> `Result`: KZG_OK is "the boolean is valid" (`Ok(true)` / `Ok(false)`), everything else one of the
/// reference's `Err(KzgError::...)` (include/kzg_rs_amd.h, "Conventions").
pub fn check(rc: c_int) -> Result<(), KzgError> {
    KzgError::from_ret(rc, last_error)
}

/// Owner of one `KzgSettings*` of the library.  The handle is immutable after creation and the library serves it to any number
/// of threads at once - the small calls of concurrent callers are coalesced into shared launches on pooled lanes, each caller
/// with its own verdict (csrc/capi_coalesce.hpp), large calls take the handle's own lock - so sharing it between threads is
/// sound AND scales: one `static` settings value for a thread pool of `verify_kzg_proof` callers is the intended use, as with the
/// reference's `&'static` slices (src/trusted_setup.rs:44-50,80-92).
pub struct Handle(pub *mut RawSettings);
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { kzg_settings_free(self.0) }
    }
}
