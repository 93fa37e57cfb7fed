// Points the linker at libbzk.so.  BZK_LIB_DIR = the directory holding it (default: <repo>/bazuka_amd, where
// `make -C bazuka_amd/csrc` puts it).  UNVERIFIED BY COMPILATION (no rustc in the image that builds libbzk).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("BZK_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../bazuka_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bzk");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BZK_LIB_DIR");
}
