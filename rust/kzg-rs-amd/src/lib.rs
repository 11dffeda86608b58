//! `kzg-rs` 0.2.8's public API (succinctlabs/kzg-rs `src/lib.rs:12-18`) over `libkzg_rs_amd.so`.
//!
//! Same module layout and re-exports as the reference, so `use kzg_rs::{KzgProof, KzgSettings, Blob, Bytes32,
//! Bytes48, KzgError, pairings_verify}` and every `kzg_rs::consts::*` item resolve unchanged.  Verification runs on
//! the GPU; this crate parses nothing and multiplies nothing on the verification path.
//!
//! Differences a caller can observe (all listed in README.md): the crate needs `std` (a mutex-free FFI handle in an
//! `Arc`, the loader), it links a shared library, and `KzgSettings` carries one private field next to the three
//! public slices.
extern crate alloc;

pub mod consts;
pub mod dtypes;
pub mod enums;
pub mod ffi;
pub mod kzg_proof;
pub mod pairings;
pub mod trusted_setup;

pub use consts::*;
pub use dtypes::*;
pub use kzg_proof::KzgProof;
pub use pairings::{pairings_verify, try_pairings_verify};
pub use trusted_setup::*;

pub use enums::KzgError;
