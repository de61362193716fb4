// Links libpcp_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'` into <repo>/pcp_amd/).
fn main() {
    let dir = std::env::var("PCP_HIP_LIB_DIR").expect("set PCP_HIP_LIB_DIR to the directory that holds libpcp_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=pcp_hip");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64"); // hipMalloc / hipFree / hipMemcpy of the device-resident store
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=PCP_HIP_LIB_DIR");
}
