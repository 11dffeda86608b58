//! Sizes, domain strings and field constants, value for value as kzg-rs `src/consts.rs:1-15,212-219`.
pub const BYTES_PER_G1_POINT: usize = 48;
pub const BYTES_PER_G2_POINT: usize = 96;
pub const BYTES_PER_FIELD_ELEMENT: usize = 32;
pub const NUM_G1_POINTS: usize = 4096;
pub const NUM_G2_POINTS: usize = 65;
pub const NUM_ROOTS_OF_UNITY: usize = 4096;
pub const NUM_FIELD_ELEMENTS_PER_BLOB: usize = 4096;
pub const BYTES_PER_BLOB: usize = NUM_FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT;
pub const BYTES_PER_COMMITMENT: usize = 48;
pub const BYTES_PER_PROOF: usize = 48;
pub const DOMAIN_STR_LENGTH: usize = 16;
pub const CHALLENGE_INPUT_SIZE: usize = DOMAIN_STR_LENGTH + 16 + BYTES_PER_BLOB + BYTES_PER_COMMITMENT;
pub const FIAT_SHAMIR_PROTOCOL_DOMAIN: &str = "FSBLOBVERIFY_V1_";
pub const RANDOM_CHALLENGE_KZG_BATCH_DOMAIN: &str = "RCKZGBATCH___V1_";

/// r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, little-endian 64-bit limbs.
pub const MODULUS: [u64; 4] = [
    0xffff_ffff_0000_0001,
    0x53bd_a402_fffe_5bfe,
    0x3339_d808_09a1_d805,
    0x73ed_a753_299d_7d48,
];

/// The primitive 4096th root of unity the evaluation domain is generated from (the reference's
/// `SCALE2_ROOT_OF_UNITY[12]`, little-endian limbs).  The reference exports the whole 32-entry table; entry k is
/// `PRIMITIVE_ROOT_OF_UNITY_4096 ^ (2^(12 - k))` for k <= 12 and is not read by any verification function.
pub const PRIMITIVE_ROOT_OF_UNITY_4096: [u64; 4] = [
    0xe206_da11_a5d3_6306,
    0x0ad1_347b_378f_bf96,
    0xfc3e_8acf_e0f8_245f,
    0x564c_0a11_a0f7_04f4,
];
