#!/bin/sh
# Applies the GPU shim to an ark-bulletproofs checkout (v4.1.x).  NEVER RUN in the build image (no cargo there).
#   shim/apply.sh <checkout> [<dir holding libarkbp_hip.so>]
# 1. golden vectors from the UNMODIFIED reference (pins this repository's CPU oracle):
#      cd <checkout> && ARKBP_GOLDEN=<repo>/tests/golden/r1cs_golden.json cargo test --release --test golden_vectors -- --nocapture
#    (ARKBP_GOLDEN_WRITE=1 overwrites the self-generated fixture with the reference's values)
# 2. the drop-in: `ark_bulletproofs::r1cs::gpu::{Prover, Verifier, batch_verify, msm}` over libarkbp_hip.so (feature "gpu")
set -e
SRC="$(cd "$(dirname "$0")" && pwd)"
DST="$1"
LIBDIR="${2:-$SRC/../ark_bulletproofs_amd}"
[ -f "$DST/src/r1cs/mod.rs" ] || { echo "usage: apply.sh <ark-bulletproofs checkout> [libdir]"; exit 1; }
cp "$SRC/tests/golden_vectors.rs" "$DST/tests/golden_vectors.rs"
cp "$SRC/src/lib.rs" "$DST/src/r1cs/gpu.rs"
cp "$SRC/src/ffi.rs" "$DST/src/r1cs/gpu_ffi.rs"
grep -q "pub mod gpu;" "$DST/src/r1cs/mod.rs" || printf '\n#[cfg(feature = "gpu")]\npub mod gpu;\n#[cfg(feature = "gpu")]\npub mod gpu_ffi;\n' >> "$DST/src/r1cs/mod.rs"
grep -q '^gpu = ' "$DST/Cargo.toml" || sed -i 's/^\[features\]$/[features]\ngpu = ["std", "yoloproofs"]/' "$DST/Cargo.toml"
grep -q 'serde_json' "$DST/Cargo.toml" || sed -i 's/^\[dev-dependencies\]$/[dev-dependencies]\nserde_json = "1"\nhex = "0.4"/' "$DST/Cargo.toml"
grep -q 'name = "golden_vectors"' "$DST/Cargo.toml" || printf '\n[[test]]\nname = "golden_vectors"\nrequired-features = ["yoloproofs"]\n' >> "$DST/Cargo.toml"
cat > "$DST/build.rs" <<BUILD
fn main() {
    if std::env::var("CARGO_FEATURE_GPU").is_ok() {
        let dir = std::env::var("ARKBP_LIB_DIR").unwrap_or_else(|_| "$LIBDIR".to_string());
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=arkbp_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
}
BUILD
echo "applied: cargo test --release --test golden_vectors   |   cargo build --release --features gpu"
