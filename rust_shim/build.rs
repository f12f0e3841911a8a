// Links libbirefnet_hip.so when the `hip` feature is on.  BIREFNET_HIP_LIB_DIR = directory holding the library
// (candle_birefnet_amd/ of this repository after `make -C candle_birefnet_amd/csrc`).
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("BIREFNET_HIP_LIB_DIR").expect("set BIREFNET_HIP_LIB_DIR to the directory of libbirefnet_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=birefnet_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        println!("cargo:rerun-if-env-changed=BIREFNET_HIP_LIB_DIR");
    }
}
