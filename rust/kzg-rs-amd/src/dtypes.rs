//! The three byte newtypes of the API - `Bytes32` (a field element), `Bytes48` (a compressed G1 point), `Blob` (4096 field
//! elements) - with the constructors, conversions and optional `serde` / `rkyv` derives of kzg-rs `src/dtypes.rs:7-57`.
//! `#[repr(transparent)]` is added: a `Vec<Blob>` is then provably one contiguous `n * 131072`-byte region, which is what
//! the library's batch entry point takes as it lies in memory (no per-blob copy).
use crate::enums::KzgError;
use crate::{BYTES_PER_BLOB, BYTES_PER_FIELD_ELEMENT};
use alloc::{string::ToString, vec::Vec};
use bls12_381::Scalar;

macro_rules! byte_newtype {
    ($(#[$doc:meta])* $name:ident[$size:expr]) => {
        $(#[$doc])*
        #[repr(transparent)]
        #[derive(Debug, Clone)]
        #[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
        #[cfg_attr(feature = "rkyv", derive(rkyv::Archive, rkyv::Serialize, rkyv::Deserialize))]
        pub struct $name(#[cfg_attr(feature = "serde", serde(with = "serde_arrays"))] pub [u8; $size]);

        impl TryFrom<&[u8]> for $name {
            type Error = KzgError;
            fn try_from(bytes: &[u8]) -> Result<Self, KzgError> {
                match <[u8; $size]>::try_from(bytes) {
                    Ok(array) => Ok($name(array)),
                    Err(_) => Err(KzgError::InvalidBytesLength("Invalid slice length".to_string())),
                }
            }
        }

        impl AsRef<[u8]> for $name {
            fn as_ref(&self) -> &[u8] {
                &self.0[..]
            }
        }

        impl From<$name> for [u8; $size] {
            fn from(value: $name) -> [u8; $size] {
                value.0
            }
        }

        impl $name {
            /// `Err(InvalidBytesLength)` unless the slice has exactly the type's size.
            pub fn from_slice(slice: &[u8]) -> Result<Self, KzgError> {
                Self::try_from(slice)
            }

            pub fn as_slice(&self) -> &[u8] {
                self.as_ref()
            }
        }
    };
}

byte_newtype!(
    /// 32 big-endian bytes: a field element (z, y) on the wire.
    Bytes32[32]
);
byte_newtype!(
    /// 48 bytes: a compressed G1 point (commitment, proof).
    Bytes48[48]
);
byte_newtype!(
    /// 131 072 bytes: 4096 field elements of 32 big-endian bytes each.
    Blob[BYTES_PER_BLOB]
);

impl Blob {
    /// The 4096 field elements of the blob; `Err(BadArgs)` if one of them is not below r (kzg-rs `src/dtypes.rs:48-57`).
    /// Host-side convenience only - the verifier reads the blob bytes on the GPU and performs the same canonical check.
    pub fn as_polynomial(&self) -> Result<Vec<Scalar>, KzgError> {
        let mut out = Vec::with_capacity(self.0.len() / BYTES_PER_FIELD_ELEMENT);
        for element in self.0.chunks_exact(BYTES_PER_FIELD_ELEMENT) {
            let mut le = [0u8; BYTES_PER_FIELD_ELEMENT];
            for (dst, src) in le.iter_mut().zip(element.iter().rev()) {
                *dst = *src;
            }
            match Option::<Scalar>::from(Scalar::from_bytes(&le)) {
                Some(scalar) => out.push(scalar),
                None => return Err(KzgError::BadArgs("Failed to parse G1Affine from bytes".to_string())),
            }
        }
        Ok(out)
    }
}
