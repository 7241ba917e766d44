// Points the linker at librt_amd.so. RT_AMD_LIB_DIR names the directory that holds it
// (raytracer_amd/ in the library's source tree); the same path is baked in as an rpath so the
// binary finds the library at run time without LD_LIBRARY_PATH.
fn main() {
    println!("cargo:rerun-if-env-changed=RT_AMD_LIB_DIR");
    let dir = std::env::var("RT_AMD_LIB_DIR")
        .expect("set RT_AMD_LIB_DIR to the directory containing librt_amd.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
