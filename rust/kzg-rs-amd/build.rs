// Link against libkzg_rs_amd.so.  KZG_RS_AMD_LIB_DIR = the directory that holds it (default: ../../kzg_rs_amd of this
// repository, where `python -m kzg_rs_amd.build` puts it).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("KZG_RS_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../kzg_rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kzg_rs_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=KZG_RS_AMD_LIB_DIR");
}
