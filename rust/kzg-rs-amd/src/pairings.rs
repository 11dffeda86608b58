//! `pairings_verify` of kzg-rs `src/pairings.rs:5-9` on the GPU: e(a1, a2) == e(b1, b2) for arbitrary arguments.
use crate::ffi;
use crate::trusted_setup::KzgSettings;
use bls12_381::{G1Affine, G2Affine};

/// Verifies that the pairings of two G1 and two G2 points are equal.  Same signature as the reference; the device and
/// the pairing programs come from the process-wide default settings handle.  A library failure (no device) is `false`.
pub fn pairings_verify(a1: G1Affine, a2: G2Affine, b1: G1Affine, b2: G2Affine) -> bool {
    let settings = match KzgSettings::load_trusted_setup_file() {
        Ok(s) => s,
        Err(_) => return false,
    };
    let (a1, a2, b1, b2) = (a1.to_compressed(), a2.to_compressed(), b1.to_compressed(), b2.to_compressed());
    let mut ok = false;
    let rc = unsafe { ffi::kzg_pairings_verify(&mut ok, a1.as_ptr(), a2.as_ptr(), b1.as_ptr(), b2.as_ptr(), settings.raw()) };
    rc == ffi::KZG_OK && ok
}
