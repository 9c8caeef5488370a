// Links libtfhe_aes_hip.so, built by `make -C tfhe-aes_amd`.  TFHE_AES_HIP_DIR overrides the location.
fn main() {
    let dir = std::env::var("TFHE_AES_HIP_DIR").unwrap_or_else(|_| "../../../tfhe-aes_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tfhe_aes_hip");
    println!("cargo:rerun-if-env-changed=TFHE_AES_HIP_DIR");
}
