//! Plug-in level: `impl gkr::circuit::node::Node<F, E>` for a Lasso node whose claim reduction runs on the GPU.
//!
//! The reference's circuit is built by `BfvEncryptBlock::configure`, which constructs `LassoNode::<F, E, C, M>::new(..)`
//! itself [REF bfv-gkr/src/sk_encryption_circuit.rs:205-209]; to use this node the maintainer changes that one
//! `circuit.insert(..)` to `circuit.insert(HipLassoNode::wrap(LassoNode::new(..), hip.clone()))`. Everything else
//! (`Circuit`, `prove_gkr`, Vanilla / FFT nodes, the verifier) stays the reference's CPU code, so this is the smallest
//! possible substitution - and the one the pin test uses to localise a transcript difference to the Lasso node.
//!
//! Transcript position. The reference transcript never absorbs prover messages (`write_felt` only appends, `common_felt`
//! is a no-op [REF bfv-gkr/src/transcript.rs:146-157, 180-196]), so the node's challenges are a run of the fixed Keccak
//! chain starting at the number of E challenges squeezed before the node is entered. `&mut dyn TranscriptWrite` does not
//! expose that position; `CountingTranscript` (below) wraps the prover's transcript and counts squeezes in a thread-local
//! the node reads. With an absorbing transcript the node needs the per-round form instead
//! (`hg_prove_mode` on the whole prover); this shim then refuses to run.
use crate::{bfv::HipBfvEncrypt, check, ffi::*};
use gkr::{
    circuit::node::{CombinedEvalClaim, EvalClaim, Node},
    ff_ext::ff::PrimeField,
    poly::{BoxMultilinearPoly, MultilinearPoly},
    transcript::{Transcript, TranscriptRead, TranscriptWrite},
    util::arithmetic::ExtensionField,
    Error,
};
use goldilocks::{Goldilocks, GoldilocksExt2};
use lasso_gkr::LassoNode;
use std::cell::Cell;
use std::fmt::Debug;
use std::sync::{Arc, Mutex};

const C: usize = 4; // [REF sk_encryption_circuit.rs:30]
const M: usize = 1 << 16; // [REF sk_encryption_circuit.rs:29-31]

thread_local! {
    /// E challenges squeezed so far on this thread's `CountingTranscript`
    static SQUEEZED: Cell<usize> = Cell::new(0);
}

/// Wraps the prover's transcript (`Keccak256Transcript<Vec<u8>>`) and counts `squeeze_challenge` calls. Everything is
/// forwarded unchanged, so the proof bytes are the reference's.
#[derive(Debug)]
pub struct CountingTranscript<T> {
    pub inner: T,
}
impl<T> CountingTranscript<T> {
    pub fn new(inner: T) -> Self {
        SQUEEZED.with(|c| c.set(0));
        Self { inner }
    }
    pub fn squeezed() -> usize {
        SQUEEZED.with(|c| c.get())
    }
}
impl<F: PrimeField, E: ExtensionField<F>, T: Transcript<F, E>> Transcript<F, E> for CountingTranscript<T> {
    fn squeeze_challenge(&mut self) -> E {
        SQUEEZED.with(|c| c.set(c.get() + 1));
        self.inner.squeeze_challenge()
    }
    fn common_felt(&mut self, felt: &F) {
        self.inner.common_felt(felt)
    }
}
impl<F: PrimeField, E: ExtensionField<F>, T: TranscriptWrite<F, E>> TranscriptWrite<F, E> for CountingTranscript<T> {
    fn write_felt(&mut self, felt: &F) -> Result<(), Error> {
        self.inner.write_felt(felt)
    }
    fn write_felt_ext(&mut self, felt: &E) -> Result<(), Error> {
        self.inner.write_felt_ext(felt)
    }
}
impl<F: PrimeField, E: ExtensionField<F>, T: TranscriptRead<F, E>> TranscriptRead<F, E> for CountingTranscript<T> {
    fn read_felt(&mut self) -> Result<F, Error> {
        self.inner.read_felt()
    }
    fn read_felt_ext(&mut self) -> Result<E, Error> {
        self.inner.read_felt_ext()
    }
}

/// canonical value of a Goldilocks element: `to_repr()` is the canonical little-endian byte string
/// [REF transcript.rs:183-189 reverses it to big-endian for the wire]
fn to_u64(f: &Goldilocks) -> u64 {
    let repr = f.to_repr();
    u64::from_le_bytes(repr.as_ref()[..8].try_into().unwrap())
}
fn from_u64(v: u64) -> Goldilocks {
    let mut repr = <Goldilocks as PrimeField>::Repr::default();
    repr.as_mut()[..8].copy_from_slice(&v.to_le_bytes());
    Goldilocks::from_repr_vartime(repr).expect("non-canonical limb from the C ABI")
}

/// `LassoNode<Goldilocks, GoldilocksExt2, 4, 65536>` whose `prove_claim_reduction` is `hg_lasso_prove_at`
/// [REF lasso/src/lasso.rs:57-114]
#[derive(Debug)]
pub struct HipLassoNode {
    inner: LassoNode<Goldilocks, GoldilocksExt2, C, M>,
    hip: Arc<Mutex<HipBfvEncrypt>>,
}
impl Debug for HipBfvEncrypt {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "HipBfvEncrypt(n = {}, k = {})", self.params.n, self.params.k)
    }
}

impl HipLassoNode {
    pub fn wrap(inner: LassoNode<Goldilocks, GoldilocksExt2, C, M>, hip: Arc<Mutex<HipBfvEncrypt>>) -> Self {
        Self { inner, hip }
    }
}

impl Node<Goldilocks, GoldilocksExt2> for HipLassoNode {
    fn is_input(&self) -> bool {
        false
    }
    fn log2_input_size(&self) -> usize {
        self.inner.log2_input_size()
    }
    fn log2_output_size(&self) -> usize {
        0
    }
    fn evaluate(&self, inputs: Vec<&BoxMultilinearPoly<Goldilocks, GoldilocksExt2>>) -> BoxMultilinearPoly<'static, Goldilocks, GoldilocksExt2> {
        self.inner.evaluate(inputs) // box_dense_poly([F::ZERO]) [REF lasso.rs:53-55]
    }

    fn prove_claim_reduction(
        &self,
        _claim: CombinedEvalClaim<GoldilocksExt2>,
        inputs: Vec<&BoxMultilinearPoly<Goldilocks, GoldilocksExt2>>,
        transcript: &mut dyn TranscriptWrite<Goldilocks, GoldilocksExt2>,
    ) -> Result<Vec<Vec<EvalClaim<GoldilocksExt2>>>, Error> {
        let hip = self.hip.lock().unwrap();
        // canonical little-endian u64 limbs cross the boundary
        let table: Vec<u64> = inputs[0].to_dense().iter().map(to_u64).collect();
        let nu = table.len().ilog2() as usize;
        let chain_skip = CountingTranscript::<()>::squeezed();
        let mut proof = vec![0u8; 1 << 22];
        let mut len = 0usize;
        let mut claim = vec![0u64; 2 * nu + 2];
        let mut n_chal = 0usize;
        unsafe {
            check(
                hg_lasso_prove_at(hip.ctx, hip.pk, table.as_ptr(), chain_skip, proof.as_mut_ptr(), proof.len(), &mut len, claim.as_mut_ptr()),
                "hg_lasso_prove_at",
            );
            check(hg_lasso_num_challenges(hip.pk, &mut n_chal), "hg_lasso_num_challenges");
        }
        // Bring the caller's transcript to the state the CPU node would leave it in: the node's challenges squeezed (only
        // their NUMBER matters: the chain does not depend on the messages) and its elements written, in stream order.
        for _ in 0..n_chal {
            let _: GoldilocksExt2 = transcript.squeeze_challenge();
        }
        for chunk in proof[..len].chunks_exact(8) {
            // wire format: canonical repr byte-reversed to big-endian [REF transcript.rs:183-189]
            transcript.write_felt(&from_u64(u64::from_be_bytes(chunk.try_into().unwrap())))?;
        }
        let e = |i: usize| GoldilocksExt2::from_bases(&[from_u64(claim[2 * i]), from_u64(claim[2 * i + 1])]);
        let point = (0..nu).map(e).collect::<Vec<_>>();
        Ok(vec![vec![EvalClaim::new(point, e(nu))]]) // (r, claimed_sum) for the single input [REF lasso.rs:97,113]
    }

    fn verify_claim_reduction(
        &self,
        claim: CombinedEvalClaim<GoldilocksExt2>,
        transcript: &mut dyn TranscriptRead<Goldilocks, GoldilocksExt2>,
    ) -> Result<Vec<Vec<EvalClaim<GoldilocksExt2>>>, Error> {
        self.inner.verify_claim_reduction(claim, transcript) // verification stays the reference's CPU code [REF lasso.rs:116-139]
    }
}
