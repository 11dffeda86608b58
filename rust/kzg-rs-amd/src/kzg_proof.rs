//! `KzgProof` of kzg-rs `src/kzg_proof.rs:350-526`: the four verification functions with their signatures, early
//! returns and error variants; each forwards to one entry point of the library.
use crate::dtypes::{Blob, Bytes32, Bytes48};
use crate::enums::KzgError;
use crate::ffi;
use crate::trusted_setup::KzgSettings;
use alloc::{string::ToString, vec::Vec};
use bls12_381::{G1Affine, Scalar};

pub struct KzgProof {}

impl KzgProof {
    /// kzg-rs `src/kzg_proof.rs:353-397`.
    pub fn verify_kzg_proof(
        commitment_bytes: &Bytes48,
        z_bytes: &Bytes32,
        y_bytes: &Bytes32,
        proof_bytes: &Bytes48,
        kzg_settings: &KzgSettings,
    ) -> Result<bool, KzgError> {
        let mut ok = false;
        ffi::check(unsafe {
            ffi::kzg_verify_kzg_proof(&mut ok, commitment_bytes.0.as_ptr(), z_bytes.0.as_ptr(), y_bytes.0.as_ptr(), proof_bytes.0.as_ptr(), kzg_settings.raw())
        })?;
        Ok(ok)
    }

    /// kzg-rs `src/kzg_proof.rs:399-444`: typed inputs; they cross the C ABI as compressed points and big-endian
    /// canonical scalars.  Slices shorter than `commitments` index out of bounds in the reference: a panic here too.
    pub fn verify_kzg_proof_batch(
        commitments: &[G1Affine],
        zs: &[Scalar],
        ys: &[Scalar],
        proofs: &[G1Affine],
        kzg_settings: &KzgSettings,
    ) -> Result<bool, KzgError> {
        let n = commitments.len();
        let (zs, ys, proofs) = (&zs[..n], &ys[..n], &proofs[..n]);
        let be = |s: &Scalar| {
            let mut b = s.to_bytes();
            b.reverse();
            b
        };
        let c: Vec<u8> = commitments.iter().flat_map(|p| p.to_compressed()).collect();
        let p: Vec<u8> = proofs.iter().flat_map(|p| p.to_compressed()).collect();
        let z: Vec<u8> = zs.iter().flat_map(be).collect();
        let y: Vec<u8> = ys.iter().flat_map(be).collect();
        let mut ok = false;
        ffi::check(unsafe { ffi::kzg_verify_kzg_proof_batch(&mut ok, c.as_ptr(), z.as_ptr(), y.as_ptr(), p.as_ptr(), n, kzg_settings.raw()) })?;
        Ok(ok)
    }

    /// Not in kzg-rs: n independent `verify_kzg_proof` calls (`src/kzg_proof.rs:353-397`) through one library call, each
    /// proof with its own pairing and its own result - what a caller looping over `verify_kzg_proof` (the revm point
    /// evaluation precompile, one call per transaction) would write.  Entry i is what `verify_kzg_proof` returns for
    /// tuple i.  The four slices must have equal lengths.
    pub fn verify_kzg_proofs(
        commitment_bytes: &[Bytes48],
        zs: &[Bytes32],
        ys: &[Bytes32],
        proof_bytes: &[Bytes48],
        kzg_settings: &KzgSettings,
    ) -> Result<Vec<Result<bool, KzgError>>, KzgError> {
        let n = commitment_bytes.len();
        if zs.len() != n || ys.len() != n || proof_bytes.len() != n {
            return Err(KzgError::InvalidBytesLength("verify_kzg_proofs: slices of unequal length".to_string()));
        }
        let mut ok = vec![false; n];
        let mut err = vec![0u8; n];
        // (`Bytes32` / `Bytes48` are `repr(transparent)` byte arrays: the slices are handed over as they lie in memory)
        ffi::check(unsafe {
            ffi::kzg_verify_kzg_proofs(
                ok.as_mut_ptr(),
                err.as_mut_ptr(),
                commitment_bytes.as_ptr() as *const u8,
                zs.as_ptr() as *const u8,
                ys.as_ptr() as *const u8,
                proof_bytes.as_ptr() as *const u8,
                n,
                kzg_settings.raw(),
            )
        })?;
        Ok((0..n)
            .map(|i| if err[i] != 0 { Err(KzgError::BadArgs("Failed to parse G1Affine from bytes".to_string())) } else { Ok(ok[i]) })
            .collect())
    }

    /// kzg-rs `src/kzg_proof.rs:446-470`.
    pub fn verify_blob_kzg_proof(blob: Blob, commitment_bytes: &Bytes48, proof_bytes: &Bytes48, kzg_settings: &KzgSettings) -> Result<bool, KzgError> {
        let mut ok = false;
        ffi::check(unsafe { ffi::kzg_verify_blob_kzg_proof(&mut ok, blob.0.as_ptr(), commitment_bytes.0.as_ptr(), proof_bytes.0.as_ptr(), kzg_settings.raw()) })?;
        Ok(ok)
    }

    /// kzg-rs `src/kzg_proof.rs:472-525`, including the order of its early returns (empty -> `Ok(true)` and the
    /// single-blob shortcut come BEFORE the length checks, `:478-501`).  `Blob`, `Bytes48` are `repr(transparent)` byte
    /// arrays, so the three `Vec`s are handed over as they lie in memory.
    pub fn verify_blob_kzg_proof_batch(
        blobs: Vec<Blob>,
        commitments_bytes: Vec<Bytes48>,
        proofs_bytes: Vec<Bytes48>,
        kzg_settings: &KzgSettings,
    ) -> Result<bool, KzgError> {
        if blobs.is_empty() {
            return Ok(true);
        }
        if blobs.len() == 1 {
            return Self::verify_blob_kzg_proof(blobs[0].clone(), &commitments_bytes[0], &proofs_bytes[0], kzg_settings);
        }
        if blobs.len() != commitments_bytes.len() {
            return Err(KzgError::InvalidBytesLength("Invalid commitments length".to_string()));
        }
        if blobs.len() != proofs_bytes.len() {
            return Err(KzgError::InvalidBytesLength("Invalid proofs length".to_string()));
        }
        let mut ok = false;
        ffi::check(unsafe {
            ffi::kzg_verify_blob_kzg_proof_batch(
                &mut ok,
                blobs.as_ptr().cast::<u8>(),
                commitments_bytes.as_ptr().cast::<u8>(),
                proofs_bytes.as_ptr().cast::<u8>(),
                blobs.len(),
                kzg_settings.raw(),
            )
        })?;
        Ok(ok)
    }
}
