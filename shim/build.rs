// Links libbpgpu.so (built by `make -C mpc_bulletproof_amd/csrc`, HIP for gfx950).  BPGPU_LIB_DIR = directory holding it.
fn main() {
    let dir = std::env::var("BPGPU_LIB_DIR").unwrap_or_else(|_| "../mpc_bulletproof_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bpgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BPGPU_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/bpgpu.h");
}
