// Links libmsi.so (built by `make -C meilisearch_amd/csrc ARCH=gfx950`).
fn main() {
    if let Ok(dir) = std::env::var("MSI_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=msi");
    println!("cargo:rerun-if-env-changed=MSI_LIB_DIR");
}
