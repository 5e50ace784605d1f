//! hg-shim: the reference's API for the GKR-prove path, backed by `libhypergreco.so` (hand-written HIP kernels for
//! MI355X / gfx950) through the C ABI of `include/hg.h`.
//!
//! Two levels, as SURVEY.md §8(b) lists them:
//!   * `HipBfvEncrypt`  - whole-prover level: `BfvEncrypt::<Params, K>::{new, setup, prove, verify}`
//!                        [REF bfv-gkr/src/sk_encryption_circuit.rs:300-517], driven exactly like the body of
//!                        `generate_sk_enc_test!` [REF bfv-gkr/src/test.rs:19-44].
//!   * `HipLassoNode`   - plug-in level: `impl gkr::circuit::node::Node<F, E>` wrapping the reference's own `LassoNode`
//!                        [REF lasso/src/lasso.rs:38-140]; only `prove_claim_reduction` crosses the boundary.
//!
//! `tests/proof_dump.rs` is the pin SURVEY.md §8(c) asks for: it proves the same JSON witness with the reference CPU
//! prover and with the HIP prover, dumps both proofs and diffs them element by element (scripts/proof_diff.py names the
//! protocol element and the convention - C1..C4 / G1..G4 of DESIGN.md §2 - behind the first differing byte).
#![allow(clippy::missing_safety_doc)]

pub mod bfv;
pub mod ffi;
pub mod node;

pub use bfv::HipBfvEncrypt;
pub use node::{CountingTranscript, HipLassoNode};

use std::ffi::CStr;
use std::os::raw::c_int;

/// The C ABI reports failures as a negative status plus a message; the reference `unwrap()`s / panics at the same
/// places (release profile aborts, Cargo.toml:60), so the shim turns the status back into a panic.
pub(crate) fn check(rc: c_int, what: &str) {
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(ffi::hg_last_error()) }.to_string_lossy().into_owned();
        panic!("hypergreco: {what}: {msg}");
    }
}
