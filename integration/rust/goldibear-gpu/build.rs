fn main() {
    // directory holding libgoldibear_gpu.so (built by `python -c "import __graft_entry__ as g; g.build()"`)
    let dir = std::env::var("GOLDIBEAR_GPU_LIB_DIR").expect("set GOLDIBEAR_GPU_LIB_DIR to the directory of libgoldibear_gpu.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=goldibear_gpu");
    println!("cargo:rerun-if-env-changed=GOLDIBEAR_GPU_LIB_DIR");
}
