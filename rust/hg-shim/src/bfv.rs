//! Whole-prover level: drop-in for `BfvEncrypt::<Params, K>` in `generate_sk_enc_test!` [REF bfv-gkr/src/test.rs:31-44]
//!
//! ```ignore
//! let bfv  = HipBfvEncrypt::new(Params::N, K);              // BfvEncrypt::<Params, K>::new(K)     test.rs:31
//! let pk   = bfv.setup();                                   // bfv.setup::<F, E, Pcs>(rng)         test.rs:35-36
//! let proof = bfv.prove_json(&file_path);                   // bfv.prove::<F, E, Pcs>(&args, pk)   test.rs:37-38
//! bfv.verify_json(&file_path, &proof);                      // bfv.verify::<F, E, Pcs>(vk, ..)     test.rs:43-44
//! ```
//! The JSON file is parsed by the library (`hg_witness_from_json` = serde + `get_inputs` / `Poly::{new, new_padded,
//! new_shifted}` [REF sk_encryption_circuit.rs:365-415, poly.rs:12-44]); `prove_args` takes the already laid-out tables
//! instead (what `get_inputs` returns, as canonical u64 limbs).
use crate::{check, ffi::*};
use std::ffi::CString;
use std::ptr;

/// Which test family of the reference: `generate_sk_enc_test!("goldilocks", Goldilocks, GoldilocksExt2, ..)` or
/// `("bn254", Fr, Fr, ..)` [REF sk_encryption_circuit.rs:552-626]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Family {
    Goldilocks,
    Bn254,
}

pub struct HipBfvEncrypt {
    pub(crate) ctx: *mut HgCtx,
    pub(crate) pk: *mut HgPk,
    pub(crate) params: HgParams,
    pub timings: HgTimings,
}

// one context per device; calls are serialised by the caller (the reference test is #[serial])
unsafe impl Send for HipBfvEncrypt {}

impl HipBfvEncrypt {
    /// `BfvEncrypt::<Params, K>::new(K)` + `setup` (LassoPreprocessing::preprocess + configure, done once)
    /// [REF sk_encryption_circuit.rs:305-363]. `n` = `Params::N`, `k` = the const generic `K`.
    pub fn new(n: usize, k: usize) -> Self {
        unsafe {
            let mut params: HgParams = std::mem::zeroed();
            check(hg_params_builtin(n as u32, k as u32, &mut params), "hg_params_builtin");
            let ctx = hg_create(0);
            assert!(!ctx.is_null(), "hypergreco: no HIP device (the prover has no CPU fallback)");
            let mut pk = ptr::null_mut();
            check(hg_setup(ctx, &params, &mut pk), "hg_setup");
            // the caller proves ONCE per witness [REF test.rs:37-38]: tables, staging and the recorded launch graph now, not on the third prove
            check(hg_warmup(ctx, pk, ptr::null_mut()), "hg_warmup");
            Self { ctx, pk, params, timings: std::mem::zeroed() }
        }
    }

    fn load(&self, path: &str, family: Family) -> *mut HgWitness {
        let c = CString::new(path).unwrap();
        let mut w = ptr::null_mut();
        unsafe {
            match family {
                Family::Goldilocks => check(hg_witness_from_json(&self.params, c.as_ptr(), &mut w), "hg_witness_from_json"),
                Family::Bn254 => check(hg_witness_from_json_bn254(&self.params, c.as_ptr(), &mut w), "hg_witness_from_json_bn254"),
            }
        }
        w
    }

    /// `bfv.prove::<Goldilocks, GoldilocksExt2, Pcs>(&args, pk)` on the witness stored at `path`
    /// [REF sk_encryption_circuit.rs:417-460]: returns `transcript.into_proof()`.
    pub fn prove_json(&mut self, path: &str) -> Vec<u8> {
        let w = self.load(path, Family::Goldilocks);
        let mut proof = vec![0u8; 1 << 22];
        let mut len = 0usize;
        unsafe {
            check(hg_prove(self.ctx, self.pk, w, proof.as_mut_ptr(), proof.len(), &mut len, &mut self.timings), "hg_prove");
            hg_witness_free(w);
        }
        proof.truncate(len);
        proof
    }

    /// `prove` for a run of stored witnesses under this key, the way a caller loops over ciphertexts [REF test.rs:31-44]: the
    /// library uploads and evaluates witness i+1 while it proves witness i (`hg_prove_stream`); proof i is what `prove_json(paths[i])`
    /// returns.
    pub fn prove_json_stream(&mut self, paths: &[&str]) -> Vec<Vec<u8>> {
        let ws: Vec<*mut HgWitness> = paths.iter().map(|p| self.load(p, Family::Goldilocks)).collect();
        let cap = 1usize << 20;
        let mut buf = vec![0u8; cap * ws.len().max(1)];
        let mut lens = vec![0usize; ws.len().max(1)];
        unsafe {
            check(
                hg_prove_stream(self.ctx, self.pk, ws.as_ptr() as *const *const HgWitness, ws.len(), buf.as_mut_ptr(), cap, lens.as_mut_ptr(), &mut self.timings),
                "hg_prove_stream",
            );
            for w in &ws {
                hg_witness_free(*w);
            }
        }
        (0..ws.len()).map(|i| buf[i * cap..i * cap + lens[i]].to_vec()).collect()
    }

    /// the same for the bn254 family: `bfv.prove::<Fr, Fr, Pcs>` [REF sk_encryption_circuit.rs:614-626]
    pub fn prove_json_bn254(&mut self, path: &str) -> Vec<u8> {
        let w = self.load(path, Family::Bn254);
        let mut proof = vec![0u8; 1 << 25];
        let mut len = 0usize;
        let mut ms = [0f64; 2];
        unsafe {
            check(hg_prove_bn254(self.ctx, self.pk, w, proof.as_mut_ptr(), proof.len(), &mut len, ms.as_mut_ptr()), "hg_prove_bn254");
            hg_witness_free(w);
        }
        self.timings.witness_ms = ms[0];
        self.timings.prove_ms = ms[1];
        proof.truncate(len);
        proof
    }

    /// `bfv.verify::<F, E, Pcs>(vk, inputs, args.ct0is, &proof)` [REF sk_encryption_circuit.rs:462-517]: panics on
    /// rejection like the reference (`assert_eq!` / `unwrap()` there).
    pub fn verify_json(&self, path: &str, proof: &[u8], family: Family) {
        let w = self.load(path, family);
        let rc = unsafe {
            match family {
                // the device-side verifier (2.8 ms at n=32768 k=16 against 79 ms on the host; same accept / reject decisions)
                Family::Goldilocks => hg_verify_device(self.ctx, self.pk, w, proof.as_ptr(), proof.len()),
                Family::Bn254 => hg_verify_bn254(self.pk, w, proof.as_ptr(), proof.len()),
            }
        };
        unsafe { hg_witness_free(w) };
        check(rc, "hg_verify");
        assert_eq!(rc, 0, "proof rejected");
    }

    /// `prove` on tables laid out as `get_inputs` returns them (s, e, k1: 2^L; ais, r1is: k 2^L; r2is: k 2^P; ct0is: k 2^L;
    /// canonical Goldilocks u64 values) [REF sk_encryption_circuit.rs:365-415]
    #[allow(clippy::too_many_arguments)]
    pub fn prove_args(&mut self, s: &[u64], e: &[u64], k1: &[u64], ais: &[u64], r1is: &[u64], r2is: &[u64], ct0is: &[u64]) -> Vec<u8> {
        let mut w = ptr::null_mut();
        let mut proof = vec![0u8; 1 << 22];
        let mut len = 0usize;
        unsafe {
            check(
                hg_witness_from_arrays(&self.params, s.as_ptr(), e.as_ptr(), k1.as_ptr(), ais.as_ptr(), r1is.as_ptr(), r2is.as_ptr(), ct0is.as_ptr(), &mut w),
                "hg_witness_from_arrays",
            );
            check(hg_prove(self.ctx, self.pk, w, proof.as_mut_ptr(), proof.len(), &mut len, &mut self.timings), "hg_prove");
            hg_witness_free(w);
        }
        proof.truncate(len);
        proof
    }
}

impl Drop for HipBfvEncrypt {
    fn drop(&mut self) {
        unsafe {
            hg_pk_free(self.pk);
            hg_destroy(self.ctx);
        }
    }
}
