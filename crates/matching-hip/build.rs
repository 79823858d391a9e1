//! Links libslideo_amd.so (built in-tree by `python -m slideo_amd.build`: hipcc --offload-arch=gfx950).
use std::{env, path::PathBuf};

fn main() {
    // SLIDEO_AMD_LIB_DIR: the directory holding libslideo_amd.so (slideo_amd/lib of the MI355X repository)
    let dir = env::var("SLIDEO_AMD_LIB_DIR")
        .map(PathBuf::from)
        .expect("set SLIDEO_AMD_LIB_DIR to the directory that holds libslideo_amd.so (slideo_amd/lib)");
    if !dir.join("libslideo_amd.so").exists() {
        panic!("{:?} holds no libslideo_amd.so: run `python -m slideo_amd.build` first", dir);
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=slideo_amd");
    // the library's own dependency (HIP runtime) is resolved through its RUNPATH; the binary needs to find the library
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=SLIDEO_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
