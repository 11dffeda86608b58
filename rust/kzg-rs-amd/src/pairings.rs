//! `pairings_verify` of kzg-rs `src/pairings.rs:5-9` on the GPU: e(a1, a2) == e(b1, b2) for arbitrary arguments.
use crate::enums::KzgError;
use crate::ffi;
use crate::trusted_setup::KzgSettings;
use bls12_381::{G1Affine, G2Affine};

/// The check as a `Result`: `Err` when the LIBRARY failed (no device, allocation failure) - an infrastructure failure
/// must never look like a pairing mismatch.  The device and the pairing programs come from the process-wide default
/// settings handle.
pub fn try_pairings_verify(a1: G1Affine, a2: G2Affine, b1: G1Affine, b2: G2Affine) -> Result<bool, KzgError> {
    let settings = KzgSettings::load_trusted_setup_file()?;
    let (a1, a2, b1, b2) = (a1.to_compressed(), a2.to_compressed(), b1.to_compressed(), b2.to_compressed());
    let mut ok = false;
    ffi::check(unsafe { ffi::kzg_pairings_verify(&mut ok, a1.as_ptr(), a2.as_ptr(), b1.as_ptr(), b2.as_ptr(), settings.raw()) })?;
    Ok(ok)
}

/// Verifies that the pairings of two G1 and two G2 points are equal.  Same signature as the reference, where the function
/// is total: its arguments are typed points and nothing in it can fail.  Here the GPU library can (no device, out of
/// memory); that is not a `false` - it PANICS, as the reference's own infrastructure failure does
/// (`EnvKzgSettings::get`, kzg-rs `src/trusted_setup.rs:85-86`: `expect("failed to load default trusted setup")`).
/// Callers that want to handle it use [`try_pairings_verify`].
pub fn pairings_verify(a1: G1Affine, a2: G2Affine, b1: G1Affine, b2: G2Affine) -> bool {
    try_pairings_verify(a1, a2, b1, b2).expect("kzg-rs-amd: the GPU library failed in pairings_verify")
}
