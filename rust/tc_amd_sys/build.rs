// Links libtc_amd.so.  TC_AMD_LIB_DIR points at the directory that holds it (threshold_crypto_amd/ of the repository
// after `python -m threshold_crypto_amd.build`); the default is that directory relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("TC_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../threshold_crypto_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=tc_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=TC_AMD_LIB_DIR");
}
