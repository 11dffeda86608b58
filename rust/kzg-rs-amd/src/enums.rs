//! `KzgError`, variant for variant as kzg-rs `src/enums.rs:6-31`.
use alloc::string::String;
use core::fmt;

#[derive(Debug, Clone)]
pub enum KzgError {
    /// The supplied data is invalid in some way.
    BadArgs(String),
    /// Internal error - here: a HIP / RCCL failure, no usable gfx950 device, or an allocation failure in the library.
    InternalError,
    /// The provided bytes are of incorrect length.
    InvalidBytesLength(String),
    /// Error when converting from hex to bytes.
    InvalidHexFormat(String),
    /// The provided trusted setup params are invalid.
    InvalidTrustedSetup(String),
}

impl fmt::Display for KzgError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Self::BadArgs(s) | Self::InvalidBytesLength(s) | Self::InvalidHexFormat(s) | Self::InvalidTrustedSetup(s) => f.write_str(s),
            Self::InternalError => f.write_str("Internal error"),
        }
    }
}

impl std::error::Error for KzgError {}
