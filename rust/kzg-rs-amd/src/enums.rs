//! The error type of the crate: the five variants of kzg-rs (`src/enums.rs:6-18`), and how a return code of the C ABI
//! (`KzgRet`, include/kzg_rs_amd.h) becomes one of them.
use alloc::string::String;
use core::ffi::c_int;
use core::fmt;

#[derive(Debug, Clone)]
pub enum KzgError {
    /// An input is invalid: an undecodable or off-subgroup point, a field element that is not canonical (`KZG_BADARGS`).
    BadArgs(String),
    /// The library could not do its work: a HIP / RCCL failure, no usable gfx950 device, an allocation failure
    /// (`KZG_ERROR`, `KZG_MALLOC`).  The reference never returns it from the verification path.
    InternalError,
    /// A byte string of the wrong length (`from_slice`, mismatched batch vectors; `KZG_INVALID_LENGTH`).
    InvalidBytesLength(String),
    /// Kept for source compatibility (hex decoding lives in the callers' test harnesses).
    InvalidHexFormat(String),
    /// The trusted-setup text was rejected (`KZG_BAD_SETUP`).
    InvalidTrustedSetup(String),
}

impl KzgError {
    /// The text carried by the variant ("Internal error" for the one without).
    pub fn message(&self) -> &str {
        match self {
            KzgError::InternalError => "Internal error",
            KzgError::BadArgs(m) | KzgError::InvalidBytesLength(m) | KzgError::InvalidHexFormat(m) | KzgError::InvalidTrustedSetup(m) => m,
        }
    }

    /// `Err` for every return code but `KZG_OK` (0); `message` is the thread-local text of `kzg_last_error()`.
    pub(crate) fn from_ret(rc: c_int, message: impl FnOnce() -> String) -> Result<(), KzgError> {
        match rc {
            0 => Ok(()),
            1 => Err(KzgError::BadArgs(message())),
            4 => Err(KzgError::InvalidBytesLength(message())),
            5 => Err(KzgError::InvalidTrustedSetup(message())),
            _ => Err(KzgError::InternalError), // 2 KZG_ERROR, 3 KZG_MALLOC
        }
    }
}

impl fmt::Display for KzgError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.write_str(self.message())
    }
}

impl std::error::Error for KzgError {}
