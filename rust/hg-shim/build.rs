// Links hyper-greco_amd/libhypergreco.so (built by `make -C hyper-greco_amd/csrc`, hipcc --offload-arch=gfx950).
// HG_ROOT = root of the hyper-greco-amd checkout (default: two directories above this crate).
use std::{env, path::PathBuf};

fn main() {
    let root = env::var("HG_ROOT")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../.."));
    let lib_dir = root.join("hyper-greco_amd");
    assert!(
        lib_dir.join("libhypergreco.so").exists(),
        "{}/libhypergreco.so is missing: run `make -C hyper-greco_amd/csrc` first (there is no CPU fallback)",
        lib_dir.display()
    );
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=hypergreco");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=HG_ROOT");
    println!("cargo:rerun-if-changed={}", root.join("include/hg.h").display());
    println!("cargo:rustc-env=HG_ROOT={}", root.display());
}
