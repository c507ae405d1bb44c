//! build.rs for rocoder with the gfx950 engine: link `librocoder_hip.so`.
//! `ROCODER_HIP_DIR` = the directory that holds the library (`<graft checkout>/rocoder_amd` after
//! `python -c "import __graft_entry__ as g; g.build()"` or `make -C rocoder_amd/csrc`).
fn main() {
    let dir = std::env::var("ROCODER_HIP_DIR").unwrap_or_else(|_| "/opt/rocoder_hip/lib".to_string());
    println!("cargo:rerun-if-env-changed=ROCODER_HIP_DIR");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rocoder_hip");
    // so the binary finds the engine without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
