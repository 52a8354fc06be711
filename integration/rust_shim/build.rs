// Link against the engine built by `python __graft_entry__.py` (tfhe_aes_amd/libfheaes.so).
fn main() {
    let dir = std::env::var("FHEAES_LIB_DIR").unwrap_or_else(|_| "../../tfhe_aes_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fheaes");
    println!("cargo:rerun-if-env-changed=FHEAES_LIB_DIR");
}
