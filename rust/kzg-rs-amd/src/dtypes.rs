//! `Bytes32`, `Bytes48`, `Blob`: the byte newtypes of kzg-rs `src/dtypes.rs:7-57`, with the same optional derives.
//! `#[repr(transparent)]` is added so that a `Vec<Blob>` is provably one contiguous `n * 131072`-byte region - what
//! the library's batch entry point takes as is (no per-blob copy).
use crate::enums::KzgError;
use crate::{BYTES_PER_BLOB, BYTES_PER_FIELD_ELEMENT};
use alloc::{string::ToString, vec::Vec};
use bls12_381::Scalar;

macro_rules! byte_newtype {
    ($name:ident, $size:expr) => {
        #[cfg_attr(feature = "rkyv", derive(rkyv::Archive, rkyv::Serialize, rkyv::Deserialize))]
        #[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
        #[derive(Debug, Clone)]
        #[repr(transparent)]
        pub struct $name(#[cfg_attr(feature = "serde", serde(with = "serde_arrays"))] pub [u8; $size]);

        impl $name {
            pub fn from_slice(slice: &[u8]) -> Result<Self, KzgError> {
                let bytes: [u8; $size] = slice.try_into().map_err(|_| KzgError::InvalidBytesLength("Invalid slice length".to_string()))?;
                Ok($name(bytes))
            }

            pub fn as_slice(&self) -> &[u8] {
                &self.0
            }
        }

        impl From<$name> for [u8; $size] {
            fn from(value: $name) -> [u8; $size] {
                value.0
            }
        }
    };
}

byte_newtype!(Bytes32, 32);
byte_newtype!(Bytes48, 48);
byte_newtype!(Blob, BYTES_PER_BLOB);

impl Blob {
    /// kzg-rs `src/dtypes.rs:48-57`: the 4096 field elements of the blob (big-endian, each below r, else `BadArgs`).
    /// Host-side convenience only - the verifier reads the blob bytes on the GPU and performs the same canonical check.
    pub fn as_polynomial(&self) -> Result<Vec<Scalar>, KzgError> {
        self.0
            .chunks(BYTES_PER_FIELD_ELEMENT)
            .map(|be| {
                let mut le = [0u8; 32];
                le.copy_from_slice(be);
                le.reverse();
                Option::<Scalar>::from(Scalar::from_bytes(&le)).ok_or_else(|| KzgError::BadArgs("Failed to parse G1Affine from bytes".to_string()))
            })
            .collect()
    }
}
