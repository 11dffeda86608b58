//! `KzgSettings` / `EnvKzgSettings` of kzg-rs `src/trusted_setup.rs:12-98` over a device-side handle.
//!
//! The reference's `KzgSettings` is `Clone + PartialEq + Eq + Debug` with three `pub` `&'static` slices obtained by
//! transmuting build-time dumps.  Here the same three public slices exist (materialised once per process from the
//! library's accessors - callers that read `settings.g2_points[1]` or clone settings keep compiling), next to a private
//! `Arc` of the library handle that the verification functions use.
use crate::enums::KzgError;
use crate::ffi;
use crate::{NUM_G1_POINTS, NUM_G2_POINTS, NUM_ROOTS_OF_UNITY};
use alloc::{boxed::Box, string::ToString, sync::Arc, vec::Vec};
use bls12_381::{G1Affine, G2Affine, Scalar};
use core::ffi::c_int;
use core::hash::{Hash, Hasher};
use spin::Once;

/// The ceremony output the reference embeds (`src/trusted_setup.txt`, identical bytes).
static TRUSTED_SETUP_TXT: &str = include_str!("../../../kzg_rs_amd/data/trusted_setup.txt");

#[derive(Clone)]
pub struct KzgSettings {
    pub roots_of_unity: &'static [Scalar],
    pub g1_points: &'static [G1Affine],
    pub g2_points: &'static [G2Affine],
    handle: Arc<ffi::Handle>,
}

impl core::fmt::Debug for KzgSettings {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        f.debug_struct("KzgSettings")
            .field("roots_of_unity", &self.roots_of_unity.len())
            .field("g1_points", &self.g1_points.len())
            .field("g2_points", &self.g2_points.len())
            .finish()
    }
}

/// Value equality of the three tables, as the derived `PartialEq` of the reference compares its slices.
impl PartialEq for KzgSettings {
    fn eq(&self, other: &Self) -> bool {
        Arc::ptr_eq(&self.handle, &other.handle)
            || (self.roots_of_unity == other.roots_of_unity && self.g1_points == other.g1_points && self.g2_points == other.g2_points)
    }
}
impl Eq for KzgSettings {}

fn leak<T>(v: Vec<T>) -> &'static [T] {
    Box::leak(v.into_boxed_slice())
}

/// The three public tables read back from a library handle (decoded by the GPU once, re-compressed by the accessors).
fn tables(h: &ffi::Handle, n_g1: usize, n_g2: usize) -> Result<(&'static [Scalar], &'static [G1Affine], &'static [G2Affine]), KzgError> {
    let mut roots = Vec::with_capacity(NUM_ROOTS_OF_UNITY);
    for i in 0..NUM_ROOTS_OF_UNITY {
        let mut be = [0u8; 32];
        ffi::check(unsafe { ffi::kzg_settings_root_of_unity(h.0, i, be.as_mut_ptr()) })?;
        be.reverse();
        roots.push(Option::<Scalar>::from(Scalar::from_bytes(&be)).ok_or(KzgError::InternalError)?);
    }
    let mut g1 = Vec::with_capacity(n_g1);
    for i in 0..n_g1 {
        let mut b = [0u8; 48];
        ffi::check(unsafe { ffi::kzg_settings_g1_point(h.0, i, b.as_mut_ptr()) })?;
        g1.push(Option::<G1Affine>::from(G1Affine::from_compressed_unchecked(&b)).ok_or(KzgError::InternalError)?);
    }
    let mut g2 = Vec::with_capacity(n_g2);
    for i in 0..n_g2 {
        let mut b = [0u8; 96];
        ffi::check(unsafe { ffi::kzg_settings_g2_point(h.0, i, b.as_mut_ptr()) })?;
        g2.push(Option::<G2Affine>::from(G2Affine::from_compressed_unchecked(&b)).ok_or(KzgError::InternalError)?);
    }
    Ok((leak(roots), leak(g1), leak(g2)))
}

impl KzgSettings {
    /// kzg-rs `src/trusted_setup.rs:94-98`.  The handle lives on the calling thread's current HIP device, or on the
    /// devices named by `KZG_DEVICES` ("all" or "0,1,...") - then every `verify_blob_kzg_proof_batch` of at least 256
    /// blobs is sharded over them from this one process (include/kzg_rs_amd.h).
    pub fn load_trusted_setup_file() -> Result<Self, KzgError> {
        static DEFAULT: Once<Result<KzgSettings, KzgError>> = Once::new();
        DEFAULT
            .call_once(|| {
                let mut raw = core::ptr::null_mut();
                ffi::check(unsafe { ffi::kzg_settings_load_trusted_setup(&mut raw, TRUSTED_SETUP_TXT.as_ptr().cast(), TRUSTED_SETUP_TXT.len()) })?;
                let handle = ffi::Handle(raw);
                let (roots_of_unity, g1_points, g2_points) = tables(&handle, NUM_G1_POINTS, NUM_G2_POINTS)?;
                Ok(KzgSettings { roots_of_unity, g1_points, g2_points, handle: Arc::new(handle) })
            })
            .clone()
    }

    /// The same over an explicit device list (HIP ordinals; empty = every visible device): one handle, several GPUs.
    pub fn load_trusted_setup_file_on_devices(devices: &[i32]) -> Result<Self, KzgError> {
        let devs: Vec<c_int> = devices.iter().map(|&d| d as c_int).collect();
        let mut raw = core::ptr::null_mut();
        ffi::check(unsafe {
            ffi::kzg_settings_load_trusted_setup_devices(&mut raw, TRUSTED_SETUP_TXT.as_ptr().cast(), TRUSTED_SETUP_TXT.len(), devs.as_ptr(), devs.len())
        })?;
        let handle = ffi::Handle(raw);
        let (roots_of_unity, g1_points, g2_points) = tables(&handle, NUM_G1_POINTS, NUM_G2_POINTS)?;
        Ok(KzgSettings { roots_of_unity, g1_points, g2_points, handle: Arc::new(handle) })
    }

    /// Custom settings from the three tables a caller of the reference would put into the struct literal (the private
    /// handle field rules the literal out).  Verification reads only `g2_points[1]` = [tau]G2 and recomputes the roots.
    pub fn from_parts(roots_of_unity: &'static [Scalar], g1_points: &'static [G1Affine], g2_points: &'static [G2Affine]) -> Result<Self, KzgError> {
        let tau_g2 = g2_points.get(1).ok_or_else(|| KzgError::InvalidTrustedSetup("g2_points[1] missing".to_string()))?.to_compressed();
        let mut raw = core::ptr::null_mut();
        ffi::check(unsafe { ffi::kzg_settings_from_tau_g2(&mut raw, tau_g2.as_ptr()) })?;
        Ok(KzgSettings { roots_of_unity, g1_points, g2_points, handle: Arc::new(ffi::Handle(raw)) })
    }

    /// `sum_i scalars[i] * g1_points[i mod 4096]` - `G1Projective::msm_variable_base` (kzg-rs `src/kzg_proof.rs:419,429,430`) when its
    /// points are this setup's own Lagrange points: computed from the tables the library made when the setup was loaded, nothing
    /// decoded per call (`kzg_g1_msm_setup`; not a function of kzg-rs).  Scalars as big-endian bytes, any value below 2^256
    /// (reduced mod r like `Scalar::from_raw`); the sum comes back compressed.  Needs settings loaded from a trusted-setup file
    /// (`from_parts` handles carry no G1 section: `KzgError::BadArgs`).
    pub fn g1_msm_over_setup_points(&self, scalars: &[crate::dtypes::Bytes32]) -> Result<crate::dtypes::Bytes48, KzgError> {
        let mut out = [0u8; 48];
        // (`Bytes32` is #[repr(transparent)] over [u8; 32]: a slice of them is n * 32 contiguous bytes)
        ffi::check(unsafe { ffi::kzg_g1_msm_setup(out.as_mut_ptr(), scalars.as_ptr() as *const u8, scalars.len(), self.handle.0) })?;
        Ok(crate::dtypes::Bytes48(out))
    }

    /// The devices this handle runs on and how its partial sums travel: 0 one device, 1 host staging, 2 in-process RCCL.
    pub fn devices(&self) -> Result<(Vec<i32>, i32), KzgError> {
        let (mut n, mut ex, mut devs) = (0usize, 0 as c_int, [0 as c_int; 64]);
        ffi::check(unsafe { ffi::kzg_settings_devices(self.handle.0, &mut n, devs.as_mut_ptr(), 64, &mut ex) })?;
        Ok((devs[..n].iter().map(|&d| d as i32).collect(), ex as i32))
    }

    pub(crate) fn raw(&self) -> *const ffi::RawSettings {
        self.handle.0
    }
}

pub fn get_roots_of_unity() -> &'static [Scalar] {
    get_kzg_settings().roots_of_unity
}
pub fn get_g1_points() -> &'static [G1Affine] {
    get_kzg_settings().g1_points
}
pub fn get_g2_points() -> &'static [G2Affine] {
    get_kzg_settings().g2_points // 65 entries (the reference's slice claims 4096 over a 65-element dump: SURVEY quirk Q4)
}
pub fn get_kzg_settings() -> KzgSettings {
    KzgSettings::load_trusted_setup_file().expect("failed to load default trusted setup")
}

/// kzg-rs `src/trusted_setup.rs:52-92`: the default (mainnet) settings or a caller's own behind an `Arc`.  Two values are
/// equal when both are `Default` or both hold the SAME `Arc` (pointer identity, not table contents), and they hash
/// accordingly - what a revm-style cache keyed by settings relies on.
#[derive(Debug, Clone, Default, Eq)]
pub enum EnvKzgSettings {
    #[default]
    Default,
    Custom(Arc<KzgSettings>),
}

impl EnvKzgSettings {
    /// The address that identifies a custom value (`None` for `Default`).
    fn identity(&self) -> Option<*const KzgSettings> {
        match self {
            EnvKzgSettings::Default => None,
            EnvKzgSettings::Custom(arc) => Some(Arc::as_ptr(arc)),
        }
    }

    /// The settings to verify against; the default ones are loaded once per process, on first use.
    pub fn get(&self) -> &KzgSettings {
        static MAINNET: Once<KzgSettings> = Once::new();
        match self {
            EnvKzgSettings::Custom(arc) => arc,
            EnvKzgSettings::Default => MAINNET.call_once(|| KzgSettings::load_trusted_setup_file().expect("failed to load default trusted setup")),
        }
    }
}

impl PartialEq for EnvKzgSettings {
    fn eq(&self, other: &Self) -> bool {
        self.identity() == other.identity()
    }
}

impl Hash for EnvKzgSettings {
    fn hash<H: Hasher>(&self, state: &mut H) {
        core::mem::discriminant(self).hash(state);
        if let Some(p) = self.identity() {
            p.hash(state);
        }
    }
}
