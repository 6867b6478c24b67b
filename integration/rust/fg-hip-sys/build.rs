// Where libfg_hip.so lives: FG_HIP_LIB_DIR (the directory holding the library built by `python -m flowgger_amd.build`),
// else the system linker path.  The HIP runtime itself is a dependency of libfg_hip.so, not of this crate.
fn main() {
    println!("cargo:rerun-if-env-changed=FG_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("FG_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=fg_hip");
}
