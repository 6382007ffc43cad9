//! GPU-backed `r1cs::Prover`, `r1cs::Verifier`, `batch_verify` and `msm` for FindoraNetwork/ark-bulletproofs over
//! `libarkbp_hip.so` (include/arkbp.h) — the reference's own names and signatures, recording moved behind a C handle.
//!
//! STATUS: **NEVER COMPILED.**  The image this repository is built in has no Rust toolchain (profiles/r02_toolchain_probe.txt),
//! so this file has only been read, not type-checked; expect to fix a handful of borrow / trait-bound details on the first
//! `cargo build`.  The C ABI underneath is the tested artefact (tests/ drive every entry point through ctypes).
//!
//! WHERE IT GOES: this file is `src/r1cs/gpu.rs` of an ark-bulletproofs checkout and `ffi.rs` is `src/r1cs/gpu_ffi.rs`
//! (`shim/apply.sh <checkout>` copies them, adds `pub mod gpu; pub mod gpu_ffi;` to `src/r1cs/mod.rs` and a `build.rs` that
//! links `arkbp_hip`).  It has to live inside the `r1cs` module: `LinearCombination::terms` is `pub(super)`
//! (src/r1cs/linear_combination.rs:85-86).
//!
//! What maps to what (reference file:line -> entry point):
//!   `G::Group::msm(&bases, &scalars)` (17 call sites)            -> [`msm`] = `bp_msm`
//!   `Prover::new / commit / prove` (src/r1cs/prover.rs:291,327,444)               -> [`Prover`] = `bp_prover_new / _commit / _prove`
//!   `impl ConstraintSystem for Prover` (:96-268)                                  -> `bp_cs_multiply / _allocate / _allocate_multiplier / _constrain`
//!   `specify_randomized_constraints`, `challenge_scalar` (:201-267)               -> `bp_cs_specify_randomized_constraints`, `bp_cs_challenge_scalar`
//!   `Verifier::new / commit / verify` (src/r1cs/verifier.rs:252,279,549)          -> [`Verifier`] = `bp_verifier_new / _commit / _verify`
//!   `batch_verify` (src/r1cs/verifier.rs:604-691)                                 -> [`batch_verify`] = `bp_r1cs_batch_verify`
//!   `BulletproofGens` party 0 tables (src/generators.rs:150-304)                  -> `bp_gens_upload`, once per (device, capacity)
#![allow(non_snake_case)]

use super::gpu_ffi as ffi;
use super::{ConstraintSystem, LinearCombination, R1CSError, R1CSProof, RandomizableConstraintSystem, RandomizedConstraintSystem, Variable};
use crate::{BulletproofGens, PedersenGens};
use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::{BigInt, Fp256, MontBackend, MontConfig, PrimeField};
use ark_std::{rand::{CryptoRng, RngCore}, vec::Vec, UniformRand};
use core::borrow::BorrowMut;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;
use core::ptr::{null, null_mut};
use merlin::Transcript;
use std::collections::HashMap;
use std::sync::{Mutex, OnceLock};

// ------------------------------------------------------------------------------------------------------------------------------
// Curves the engine knows, and the memory layout both sides share: ark-ff's `Fp256<MontBackend<_, 4>>` is 4 x u64 little-endian
// limbs of x * 2^256 mod p — exactly what the ABI takes ("ark Montgomery words"); an affine point is x || y (8 words), the
// identity all-zero.
// ------------------------------------------------------------------------------------------------------------------------------
pub trait GpuCurve: AffineRepr {
    const CURVE_ID: c_int;
    fn scalar_words(s: &Self::ScalarField) -> [u64; 4];
    fn scalar_from_words(w: [u64; 4]) -> Self::ScalarField;
    fn point_words(p: &Self) -> [u64; 8];
    fn point_from_words(w: &[u64; 8]) -> Self;
}

fn fp_words<C: MontConfig<4>>(x: &Fp256<MontBackend<C, 4>>) -> [u64; 4] {
    (x.0).0 // BigInt<4>([u64; 4]): the Montgomery representation itself
}
fn fp_from_words<C: MontConfig<4>>(w: [u64; 4]) -> Fp256<MontBackend<C, 4>> {
    Fp256::new_unchecked(BigInt::new(w)) // no conversion: the words already are the Montgomery form
}

macro_rules! impl_gpu_curve {
    ($aff:ty, $fq:ty, $fr:ty, $id:expr) => {
        impl GpuCurve for $aff {
            const CURVE_ID: c_int = $id;
            fn scalar_words(s: &$fr) -> [u64; 4] { fp_words(s) }
            fn scalar_from_words(w: [u64; 4]) -> $fr { fp_from_words(w) }
            fn point_words(p: &Self) -> [u64; 8] {
                let mut out = [0u64; 8];
                if !p.infinity {
                    out[..4].copy_from_slice(&fp_words(&p.x));
                    out[4..].copy_from_slice(&fp_words(&p.y));
                }
                out
            }
            fn point_from_words(w: &[u64; 8]) -> Self {
                if w.iter().all(|&l| l == 0) {
                    return <$aff>::identity();
                }
                let x: $fq = fp_from_words([w[0], w[1], w[2], w[3]]);
                let y: $fq = fp_from_words([w[4], w[5], w[6], w[7]]);
                <$aff>::new_unchecked(x, y) // the engine only returns points it computed from valid inputs
            }
        }
    };
}
impl_gpu_curve!(ark_secq256k1::Affine, ark_secq256k1::Fq, ark_secq256k1::Fr, ffi::BP_CURVE_SECQ256K1);
impl_gpu_curve!(crate::curve::zorro::G1Affine, crate::curve::zorro::Fq, crate::curve::zorro::Fr, ffi::BP_CURVE_ZORRO);

fn last_error() -> String {
    unsafe {
        let p = ffi::bp_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
/// C status -> the reference's error (src/errors.rs:150-164); anything that has no counterpart there (HIP failure, no device,
/// bad argument) is a programming / deployment error and panics like the reference's own `unwrap()`s on `msm` do.
fn map_status(rc: c_int) -> Result<(), R1CSError> {
    match rc {
        ffi::BP_OK => Ok(()),
        ffi::BP_E_VERIFICATION => Err(R1CSError::VerificationError),
        ffi::BP_E_GENS_LENGTH => Err(R1CSError::InvalidGeneratorsLength),
        ffi::BP_E_FORMAT => Err(R1CSError::FormatError),
        ffi::BP_E_MISSING => Err(R1CSError::MissingAssignment),
        ffi::BP_E_NO_DEVICE => panic!("arkbp: no MI355X visible and the engine has no CPU fallback"),
        other => panic!("arkbp: status {} ({})", other, last_error()),
    }
}
/// status a randomized-constraints callback returns to the library when the gadget closure failed
const GADGET_FAILED: c_int = -100;

// ------------------------------------------------------------------------------------------------------------------------------
// Contexts: one `bp_ctx` per (curve, device, thread), generators uploaded once per capacity.
// ------------------------------------------------------------------------------------------------------------------------------
pub struct GpuContext {
    raw: *mut ffi::BpCtx,
    gens_capacity: usize,
}
unsafe impl Send for GpuContext {}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { ffi::bp_ctx_destroy(self.raw) }
    }
}
impl GpuContext {
    pub fn new<G: GpuCurve>(device: c_int) -> Self {
        let mut raw = null_mut();
        let rc = unsafe { ffi::bp_ctx_create(G::CURVE_ID, device, &mut raw) };
        map_status(rc).expect("bp_ctx_create");
        GpuContext { raw, gens_capacity: 0 }
    }
    pub fn raw(&self) -> *mut ffi::BpCtx { self.raw }
    /// installs party 0 of `bp_gens` (the only share this path uses: src/r1cs/prover.rs:504, verifier.rs:570) unless a table of at
    /// least that capacity is resident already
    pub fn ensure_gens<G: GpuCurve>(&mut self, bp_gens: &BulletproofGens<G>) {
        let cap = bp_gens.gens_capacity;
        if self.gens_capacity >= cap && cap > 0 {
            return;
        }
        let share = bp_gens.share(0);
        let mut g = Vec::with_capacity(cap * 8);
        let mut h = Vec::with_capacity(cap * 8);
        for p in share.G(cap) { g.extend_from_slice(&G::point_words(p)); }
        for p in share.H(cap) { h.extend_from_slice(&G::point_words(p)); }
        let rc = unsafe { ffi::bp_gens_upload(self.raw, g.as_ptr(), h.as_ptr(), cap) };
        map_status(rc).expect("bp_gens_upload");
        self.gens_capacity = cap;
    }
}

thread_local! {
    // (curve id) -> this thread's context on device 0; a service that drives several GPUs creates GpuContexts itself
    static CONTEXTS: core::cell::RefCell<HashMap<c_int, GpuContext>> = core::cell::RefCell::new(HashMap::new());
}
fn with_ctx<G: GpuCurve, R>(f: impl FnOnce(&mut GpuContext) -> R) -> R {
    CONTEXTS.with(|m| {
        let mut m = m.borrow_mut();
        let ctx = m.entry(G::CURVE_ID).or_insert_with(|| GpuContext::new::<G>(0));
        f(ctx)
    })
}

/// `<G::Group as VariableBaseMSM>::msm(bases, scalars)`: `Err(min_len)` on a length mismatch like ark-ec, the projective sum
/// otherwise.  Replaces the call at every site listed in SURVEY.md §8 a1.
pub fn msm<G: GpuCurve>(bases: &[G], scalars: &[G::ScalarField]) -> Result<G::Group, usize> {
    if bases.len() != scalars.len() {
        return Err(core::cmp::min(bases.len(), scalars.len()));
    }
    let mut b = Vec::with_capacity(bases.len() * 8);
    let mut s = Vec::with_capacity(scalars.len() * 4);
    for p in bases { b.extend_from_slice(&G::point_words(p)); }
    for x in scalars { s.extend_from_slice(&G::scalar_words(x)); }
    let mut out = [0u64; 8];
    with_ctx::<G, _>(|ctx| {
        let rc = unsafe { ffi::bp_msm(ctx.raw, b.as_ptr(), s.as_ptr(), bases.len(), 0, out.as_mut_ptr()) };
        map_status(rc).expect("bp_msm");
    });
    Ok(G::point_from_words(&out).into_group())
}

// ------------------------------------------------------------------------------------------------------------------------------
// merlin::Transcript <-> bp_transcript.  merlin 3.0 keeps its STROBE state private (`Transcript { strobe: Strobe128 }`,
// `Strobe128 { state: AlignedKeccakState([u8; 200]), pos: u8, pos_begin: u8, cur_flags: u8 }`), and the engine needs the state to
// continue the caller's transcript (the caller may have appended its own domain separators, e.g. benches/r1cs_secq256k1.rs:92-93).
// The bridge below reads / writes those 203 bytes through the struct's memory image; `transcript_bridge_self_test` proves on
// first use that the image is what this code assumes (it must reproduce merlin's published test vector through the engine) and
// panics with instructions otherwise.  The clean alternative is a two-line merlin patch exposing the state.
// ------------------------------------------------------------------------------------------------------------------------------
const STROBE_IMAGE: usize = 203;
unsafe fn merlin_state(t: &Transcript) -> [u8; STROBE_IMAGE] {
    let mut out = [0u8; STROBE_IMAGE];
    core::ptr::copy_nonoverlapping(t as *const Transcript as *const u8, out.as_mut_ptr(), STROBE_IMAGE);
    out
}
unsafe fn merlin_set_state(t: &mut Transcript, s: &[u8; STROBE_IMAGE]) {
    core::ptr::copy_nonoverlapping(s.as_ptr(), t as *mut Transcript as *mut u8, STROBE_IMAGE);
}
fn transcript_bridge_self_test() {
    static DONE: OnceLock<()> = OnceLock::new();
    DONE.get_or_init(|| {
        assert!(core::mem::size_of::<Transcript>() >= STROBE_IMAGE, "merlin::Transcript is smaller than a STROBE-128 state");
        let mut t = Transcript::new(b"test protocol");
        t.append_message(b"some label", b"some data");
        let h = unsafe { ffi::bp_transcript_new(b"x".as_ptr(), 1) };
        let st = unsafe { merlin_state(&t) };
        let rc = unsafe { ffi::bp_transcript_import_state(h, st.as_ptr()) };
        let mut via_engine = [0u8; 32];
        unsafe { ffi::bp_transcript_challenge_bytes(h, b"challenge\0".as_ptr() as *const _, via_engine.as_mut_ptr(), 32) };
        let mut via_merlin = [0u8; 32];
        t.challenge_bytes(b"challenge", &mut via_merlin);
        unsafe { ffi::bp_transcript_free(h) };
        assert!(rc == ffi::BP_OK && via_engine == via_merlin,
                "merlin::Transcript's memory image is not [state; 200] ++ pos ++ pos_begin ++ cur_flags with this compiler: \
                 patch merlin to expose its Strobe128 (two accessors) and route merlin_state / merlin_set_state through them");
    });
}
/// the library-side copy of a caller's transcript, kept in step with it
struct TranscriptBridge {
    handle: *mut c_void,
    host_is_ahead: bool, // the caller touched its merlin::Transcript (ConstraintSystem::transcript()) since the last sync
}
impl TranscriptBridge {
    fn new(t: &Transcript) -> Self {
        transcript_bridge_self_test();
        let handle = unsafe { ffi::bp_transcript_new(b"x".as_ptr(), 1) };
        let st = unsafe { merlin_state(t) };
        map_status(unsafe { ffi::bp_transcript_import_state(handle, st.as_ptr()) }).expect("bp_transcript_import_state");
        TranscriptBridge { handle, host_is_ahead: false }
    }
    fn to_engine(&mut self, t: &Transcript) {
        if self.host_is_ahead {
            let st = unsafe { merlin_state(t) };
            map_status(unsafe { ffi::bp_transcript_import_state(self.handle, st.as_ptr()) }).expect("bp_transcript_import_state");
            self.host_is_ahead = false;
        }
    }
    fn to_host(&mut self, t: &mut Transcript) {
        let mut st = [0u8; STROBE_IMAGE];
        map_status(unsafe { ffi::bp_transcript_export_state(self.handle as *const c_void, st.as_mut_ptr()) }).expect("bp_transcript_export_state");
        unsafe { merlin_set_state(t, &st) };
    }
}
impl Drop for TranscriptBridge {
    fn drop(&mut self) {
        unsafe { ffi::bp_transcript_free(self.handle) }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Variables and linear combinations across the boundary
// ------------------------------------------------------------------------------------------------------------------------------
fn var_out<F: PrimeField>(v: &Variable<F>) -> ffi::BpVar {
    match v {
        Variable::Committed(i) => ffi::BpVar { kind: ffi::BP_VAR_COMMITTED as u32, index: *i as u32 },
        Variable::MultiplierLeft(i) => ffi::BpVar { kind: ffi::BP_VAR_MULT_LEFT as u32, index: *i as u32 },
        Variable::MultiplierRight(i) => ffi::BpVar { kind: ffi::BP_VAR_MULT_RIGHT as u32, index: *i as u32 },
        Variable::MultiplierOutput(i) => ffi::BpVar { kind: ffi::BP_VAR_MULT_OUT as u32, index: *i as u32 },
        Variable::One() => ffi::BpVar { kind: ffi::BP_VAR_ONE as u32, index: 0 },
        Variable::Phantom(_) => unreachable!("Variable::Phantom carries no variable"),
    }
}
fn var_in<F: PrimeField>(v: ffi::BpVar) -> Variable<F> {
    match v.kind as c_int {
        ffi::BP_VAR_COMMITTED => Variable::Committed(v.index as usize),
        ffi::BP_VAR_MULT_LEFT => Variable::MultiplierLeft(v.index as usize),
        ffi::BP_VAR_MULT_RIGHT => Variable::MultiplierRight(v.index as usize),
        ffi::BP_VAR_MULT_OUT => Variable::MultiplierOutput(v.index as usize),
        _ => Variable::One(),
    }
}
fn split_terms<G: GpuCurve>(lc: &LinearCombination<G::ScalarField>) -> (Vec<ffi::BpVar>, Vec<u64>) {
    let mut vars = Vec::with_capacity(lc.terms.len());
    let mut coefs = Vec::with_capacity(lc.terms.len() * 4);
    for (v, c) in lc.terms.iter() {
        vars.push(var_out(v));
        coefs.extend_from_slice(&G::scalar_words(c));
    }
    (vars, coefs)
}

/// The recorder both `Prover` and `Verifier` wrap: a `bp_cs` handle plus the bridge to the caller's transcript.
struct Recorder<G: GpuCurve, T: BorrowMut<Transcript>> {
    raw: *mut ffi::BpCs,
    bridge: TranscriptBridge,
    transcript: T,
    callbacks: Vec<*mut c_void>, // boxed gadget closures handed to the library; freed on drop
    drop_callbacks: Vec<unsafe fn(*mut c_void)>,
    multipliers: usize,
    _g: PhantomData<G>,
}
impl<G: GpuCurve, T: BorrowMut<Transcript>> Drop for Recorder<G, T> {
    fn drop(&mut self) {
        unsafe { ffi::bp_cs_free(self.raw) };
        for (p, d) in self.callbacks.drain(..).zip(self.drop_callbacks.drain(..)) {
            unsafe { d(p) };
        }
    }
}
impl<G: GpuCurve, T: BorrowMut<Transcript>> Recorder<G, T> {
    fn new(proving: bool, mut transcript: T) -> Self {
        let bridge = TranscriptBridge::new(transcript.borrow_mut());
        let mut raw = null_mut();
        // Prover::new / Verifier::new append the "r1cs v1" domain separator (prover.rs:291-308, verifier.rs:252-263): the library does
        let rc = unsafe { if proving { ffi::bp_prover_new(G::CURVE_ID, bridge.handle, &mut raw) } else { ffi::bp_verifier_new(G::CURVE_ID, bridge.handle, &mut raw) } };
        map_status(rc).expect("bp_prover_new / bp_verifier_new");
        Recorder { raw, bridge, transcript, callbacks: Vec::new(), drop_callbacks: Vec::new(), multipliers: 0, _g: PhantomData }
    }
    fn sync_to_engine(&mut self) {
        let t: &Transcript = self.transcript.borrow();
        self.bridge.to_engine(t);
    }
    fn host_transcript(&mut self) -> &mut Transcript {
        // hand the caller its own transcript in the state the recording has reached; the next library call re-imports it
        self.sync_to_engine();
        let t = self.transcript.borrow_mut();
        self.bridge.to_host(t);
        self.bridge.host_is_ahead = true;
        self.transcript.borrow_mut()
    }
    fn multiply(&mut self, left: LinearCombination<G::ScalarField>, right: LinearCombination<G::ScalarField>)
        -> (Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>) {
        let (lv, lc) = split_terms::<G>(&left);
        let (rv, rc_) = split_terms::<G>(&right);
        let mut out = [ffi::BpVar::default(); 3];
        let rc = unsafe { ffi::bp_cs_multiply(self.raw, lv.as_ptr(), lc.as_ptr(), lv.len(), rv.as_ptr(), rc_.as_ptr(), rv.len(), out.as_mut_ptr()) };
        map_status(rc).expect("bp_cs_multiply");
        self.multipliers += 1;
        (var_in(out[0]), var_in(out[1]), var_in(out[2]))
    }
    fn allocate(&mut self, assignment: Option<G::ScalarField>) -> Result<Variable<G::ScalarField>, R1CSError> {
        let words = assignment.map(|a| G::scalar_words(&a));
        let mut out = ffi::BpVar::default();
        let rc = unsafe { ffi::bp_cs_allocate(self.raw, words.as_ref().map_or(null(), |w| w.as_ptr()), &mut out) };
        map_status(rc)?;
        if out.kind as c_int == ffi::BP_VAR_MULT_LEFT { self.multipliers += 1; }
        Ok(var_in(out))
    }
    fn allocate_multiplier(&mut self, input: Option<(G::ScalarField, G::ScalarField)>)
        -> Result<(Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>), R1CSError> {
        let words = input.map(|(l, r)| (G::scalar_words(&l), G::scalar_words(&r)));
        let (lp, rp) = words.as_ref().map_or((null(), null()), |(l, r)| (l.as_ptr(), r.as_ptr()));
        let mut out = [ffi::BpVar::default(); 3];
        map_status(unsafe { ffi::bp_cs_allocate_multiplier(self.raw, lp, rp, out.as_mut_ptr()) })?;
        self.multipliers += 1;
        Ok((var_in(out[0]), var_in(out[1]), var_in(out[2])))
    }
    fn constrain(&mut self, lc: LinearCombination<G::ScalarField>) {
        let (v, c) = split_terms::<G>(&lc);
        map_status(unsafe { ffi::bp_cs_constrain(self.raw, v.as_ptr(), c.as_ptr(), v.len()) }).expect("bp_cs_constrain");
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The randomized phase: the library calls back with the handle the closure must record on (src/r1cs/prover.rs:418-441,
// verifier.rs:353-376); `RandomizingCs` is the reference's RandomizingProver / RandomizingVerifier over that handle.
// ------------------------------------------------------------------------------------------------------------------------------
pub struct RandomizingCs<G: GpuCurve> {
    raw: *mut ffi::BpCs,
    scratch: Transcript, // ConstraintSystem::transcript() during the randomized phase: synced through the handle's transcript
    multipliers: usize,
    _g: PhantomData<G>,
}
impl<G: GpuCurve> RandomizingCs<G> {
    unsafe fn from_raw(raw: *mut ffi::BpCs) -> Self {
        let mut m = 0usize;
        let (mut q, mut c) = (0usize, 0usize);
        ffi::bp_cs_metrics(raw, &mut m, &mut q, &mut c);
        RandomizingCs { raw, scratch: Transcript::new(b"unused"), multipliers: m, _g: PhantomData }
    }
}
impl<G: GpuCurve> ConstraintSystem<G::ScalarField> for RandomizingCs<G> {
    fn transcript(&mut self) -> &mut Transcript {
        // a snapshot of the handle's transcript; appends made here are NOT carried back (the reference's randomized phase only draws
        // challenges, which go through challenge_scalar below) — a gadget that appends in phase 2 needs the merlin accessor patch
        let mut st = [0u8; STROBE_IMAGE];
        let h = unsafe { ffi::bp_cs_transcript(self.raw) };
        unsafe { ffi::bp_transcript_export_state(h as *const c_void, st.as_mut_ptr()) };
        unsafe { merlin_set_state(&mut self.scratch, &st) };
        &mut self.scratch
    }
    fn multiply(&mut self, left: LinearCombination<G::ScalarField>, right: LinearCombination<G::ScalarField>)
        -> (Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>) {
        let (lv, lc) = split_terms::<G>(&left);
        let (rv, rc_) = split_terms::<G>(&right);
        let mut out = [ffi::BpVar::default(); 3];
        map_status(unsafe { ffi::bp_cs_multiply(self.raw, lv.as_ptr(), lc.as_ptr(), lv.len(), rv.as_ptr(), rc_.as_ptr(), rv.len(), out.as_mut_ptr()) }).expect("bp_cs_multiply");
        self.multipliers += 1;
        (var_in(out[0]), var_in(out[1]), var_in(out[2]))
    }
    fn allocate(&mut self, assignment: Option<G::ScalarField>) -> Result<Variable<G::ScalarField>, R1CSError> {
        let words = assignment.map(|a| G::scalar_words(&a));
        let mut out = ffi::BpVar::default();
        map_status(unsafe { ffi::bp_cs_allocate(self.raw, words.as_ref().map_or(null(), |w| w.as_ptr()), &mut out) })?;
        if out.kind as c_int == ffi::BP_VAR_MULT_LEFT { self.multipliers += 1; }
        Ok(var_in(out))
    }
    fn allocate_multiplier(&mut self, input: Option<(G::ScalarField, G::ScalarField)>)
        -> Result<(Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>), R1CSError> {
        let words = input.map(|(l, r)| (G::scalar_words(&l), G::scalar_words(&r)));
        let (lp, rp) = words.as_ref().map_or((null(), null()), |(l, r)| (l.as_ptr(), r.as_ptr()));
        let mut out = [ffi::BpVar::default(); 3];
        map_status(unsafe { ffi::bp_cs_allocate_multiplier(self.raw, lp, rp, out.as_mut_ptr()) })?;
        self.multipliers += 1;
        Ok((var_in(out[0]), var_in(out[1]), var_in(out[2])))
    }
    fn multipliers_len(&self) -> usize { self.multipliers }
    fn constrain(&mut self, lc: LinearCombination<G::ScalarField>) {
        let (v, c) = split_terms::<G>(&lc);
        map_status(unsafe { ffi::bp_cs_constrain(self.raw, v.as_ptr(), c.as_ptr(), v.len()) }).expect("bp_cs_constrain");
    }
}
impl<G: GpuCurve> RandomizedConstraintSystem<G::ScalarField> for RandomizingCs<G> {
    fn challenge_scalar(&mut self, label: &'static [u8]) -> G::ScalarField {
        let mut l = label.to_vec();
        l.push(0);
        let mut out = [0u64; 4];
        map_status(unsafe { ffi::bp_cs_challenge_scalar(self.raw, l.as_ptr() as *const _, out.as_mut_ptr()) }).expect("bp_cs_challenge_scalar");
        G::scalar_from_words(out)
    }
}
unsafe extern "C" fn randomize_trampoline<G: GpuCurve, FF>(user: *mut c_void, cs: *mut ffi::BpCs) -> c_int
where FF: 'static + Fn(&mut RandomizingCs<G>) -> Result<(), R1CSError> {
    let cb = &*(user as *const FF);
    let mut view = RandomizingCs::<G>::from_raw(cs);
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| cb(&mut view))) {
        Ok(Ok(())) => ffi::BP_OK,
        Ok(Err(R1CSError::MissingAssignment)) => ffi::BP_E_MISSING,
        _ => GADGET_FAILED, // never unwind through C
    }
}
unsafe fn drop_boxed<FF>(p: *mut c_void) { drop(Box::from_raw(p as *mut FF)); }
fn specify<G: GpuCurve, T: BorrowMut<Transcript>, FF>(rec: &mut Recorder<G, T>, callback: FF) -> Result<(), R1CSError>
where FF: 'static + Fn(&mut RandomizingCs<G>) -> Result<(), R1CSError> {
    let boxed = Box::into_raw(Box::new(callback)) as *mut c_void;
    rec.callbacks.push(boxed);
    rec.drop_callbacks.push(drop_boxed::<FF>);
    map_status(unsafe { ffi::bp_cs_specify_randomized_constraints(rec.raw, Some(randomize_trampoline::<G, FF>), boxed) })
}

// ------------------------------------------------------------------------------------------------------------------------------
/// The engine works with `PedersenGens::default()` (it derives B and B_blinding itself, `bp_pedersen_gens`): a caller with custom
/// Pedersen generators must not be checked against the default ones silently.  Prover::new, Verifier::verify and batch_verify all
/// assert it (the reference would use `pc_gens.B` / `pc_gens.B_blinding`, verifier.rs:574-576).
fn assert_default_pc_gens<G: GpuCurve>(pc_gens: &PedersenGens<G>) {
    let (mut b, mut bb) = ([0u64; 8], [0u64; 8]);
    unsafe { ffi::bp_pedersen_gens(G::CURVE_ID, b.as_mut_ptr(), bb.as_mut_ptr()) };
    assert!(G::point_words(&pc_gens.B) == b && G::point_words(&pc_gens.B_blinding) == bb, "the engine works with PedersenGens::default()");
}

// r1cs::Prover (src/r1cs/prover.rs)
// ------------------------------------------------------------------------------------------------------------------------------
pub struct Prover<'g, G: GpuCurve, T: BorrowMut<Transcript>> {
    rec: Recorder<G, T>,
    _pc_gens: &'g PedersenGens<G>, // the engine derives PedersenGens::default() itself; kept for the signature (and checked below)
}
impl<'g, G: GpuCurve, T: BorrowMut<Transcript>> Prover<'g, G, T> {
    /// `Prover::new(pc_gens, transcript)` (prover.rs:291-308)
    pub fn new(pc_gens: &'g PedersenGens<G>, transcript: T) -> Self {
        assert_default_pc_gens::<G>(pc_gens);
        Prover { rec: Recorder::new(true, transcript), _pc_gens: pc_gens }
    }
    /// `commit(v, v_blinding) -> (V, Variable)` (prover.rs:327-341)
    pub fn commit(&mut self, v: G::ScalarField, v_blinding: G::ScalarField) -> (G, Variable<G::ScalarField>) {
        self.rec.sync_to_engine();
        let (vw, bw) = (G::scalar_words(&v), G::scalar_words(&v_blinding));
        let (mut xy, mut var) = ([0u64; 8], ffi::BpVar::default());
        map_status(unsafe { ffi::bp_prover_commit(self.rec.raw, null_mut(), vw.as_ptr(), bw.as_ptr(), 1, xy.as_mut_ptr(), &mut var) }).expect("bp_prover_commit");
        (G::point_from_words(&xy), var_in(var))
    }
    /// `prove(self, prng, bp_gens)` (prover.rs:444-451): the external rng enters as the 32 bytes `TranscriptRngBuilder::finalize`
    /// draws from it (:493) — everything else prove() samples comes out of the transcript rng inside the library
    pub fn prove<R: CryptoRng + RngCore>(mut self, prng: &mut R, bp_gens: &BulletproofGens<G>) -> Result<R1CSProof<G>, R1CSError> {
        self.rec.sync_to_engine();
        let mut rng32 = [0u8; 32];
        prng.fill_bytes(&mut rng32);
        let mut buf = vec![0u8; 1 << 16];
        let mut len = buf.len();
        let rc = with_ctx::<G, _>(|ctx| {
            ctx.ensure_gens(bp_gens);
            unsafe { ffi::bp_prover_prove(ctx.raw, self.rec.raw, rng32.as_ptr(), buf.as_mut_ptr(), &mut len, null_mut()) }
        });
        if rc == GADGET_FAILED {
            return Err(R1CSError::GadgetError { description: "a randomized-constraints closure failed".into() });
        }
        map_status(rc)?;
        let t = self.rec.transcript.borrow_mut();
        self.rec.bridge.to_host(t); // prove_and_return_transcript: the caller's transcript has absorbed the proof
        R1CSProof::from_bytes(&buf[..len])
    }
}
impl<'g, G: GpuCurve, T: BorrowMut<Transcript>> ConstraintSystem<G::ScalarField> for Prover<'g, G, T> {
    fn transcript(&mut self) -> &mut Transcript { self.rec.host_transcript() }
    fn multiply(&mut self, left: LinearCombination<G::ScalarField>, right: LinearCombination<G::ScalarField>)
        -> (Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>) { self.rec.multiply(left, right) }
    fn allocate(&mut self, assignment: Option<G::ScalarField>) -> Result<Variable<G::ScalarField>, R1CSError> { self.rec.allocate(assignment) }
    fn allocate_multiplier(&mut self, input_assignments: Option<(G::ScalarField, G::ScalarField)>)
        -> Result<(Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>), R1CSError> { self.rec.allocate_multiplier(input_assignments) }
    fn multipliers_len(&self) -> usize { self.rec.multipliers }
    fn constrain(&mut self, lc: LinearCombination<G::ScalarField>) { self.rec.constrain(lc) }
}
impl<'g, G: GpuCurve, T: BorrowMut<Transcript>> RandomizableConstraintSystem<G::ScalarField> for Prover<'g, G, T> {
    type RandomizedCS = RandomizingCs<G>;
    fn specify_randomized_constraints<FF>(&mut self, callback: FF) -> Result<(), R1CSError>
    where FF: 'static + Fn(&mut Self::RandomizedCS) -> Result<(), R1CSError> { specify(&mut self.rec, callback) }
}

// ------------------------------------------------------------------------------------------------------------------------------
// r1cs::Verifier and batch_verify (src/r1cs/verifier.rs)
// ------------------------------------------------------------------------------------------------------------------------------
pub struct Verifier<G: GpuCurve, T: BorrowMut<Transcript>> {
    rec: Recorder<G, T>,
}
impl<G: GpuCurve, T: BorrowMut<Transcript>> Verifier<G, T> {
    /// `Verifier::new(transcript)` (verifier.rs:252-263)
    pub fn new(transcript: T) -> Self { Verifier { rec: Recorder::new(false, transcript) } }
    /// `commit(V) -> Variable` (verifier.rs:279-287)
    pub fn commit(&mut self, commitment: G) -> Variable<G::ScalarField> {
        self.rec.sync_to_engine();
        let xy = G::point_words(&commitment);
        let mut var = ffi::BpVar::default();
        map_status(unsafe { ffi::bp_verifier_commit(self.rec.raw, xy.as_ptr(), 1, &mut var) }).expect("bp_verifier_commit");
        var_in(var)
    }
    /// `verify(self, proof, pc_gens, bp_gens)` (verifier.rs:549-557)
    pub fn verify(mut self, proof: &R1CSProof<G>, pc_gens: &PedersenGens<G>, bp_gens: &BulletproofGens<G>) -> Result<(), R1CSError> {
        assert_default_pc_gens::<G>(pc_gens);
        self.rec.sync_to_engine();
        let bytes = proof.to_bytes().map_err(|_| R1CSError::FormatError)?;
        let rc = with_ctx::<G, _>(|ctx| {
            ctx.ensure_gens(bp_gens);
            unsafe { ffi::bp_verifier_verify(ctx.raw, self.rec.raw, bytes.as_ptr(), bytes.len()) }
        });
        if rc == GADGET_FAILED {
            return Err(R1CSError::GadgetError { description: "a randomized-constraints closure failed".into() });
        }
        map_status(rc)
    }
}
impl<G: GpuCurve, T: BorrowMut<Transcript>> ConstraintSystem<G::ScalarField> for Verifier<G, T> {
    fn transcript(&mut self) -> &mut Transcript { self.rec.host_transcript() }
    fn multiply(&mut self, left: LinearCombination<G::ScalarField>, right: LinearCombination<G::ScalarField>)
        -> (Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>) { self.rec.multiply(left, right) }
    fn allocate(&mut self, _assignment: Option<G::ScalarField>) -> Result<Variable<G::ScalarField>, R1CSError> { self.rec.allocate(None) }
    fn allocate_multiplier(&mut self, _input_assignments: Option<(G::ScalarField, G::ScalarField)>)
        -> Result<(Variable<G::ScalarField>, Variable<G::ScalarField>, Variable<G::ScalarField>), R1CSError> { self.rec.allocate_multiplier(None) }
    fn multipliers_len(&self) -> usize { self.rec.multipliers }
    fn constrain(&mut self, lc: LinearCombination<G::ScalarField>) { self.rec.constrain(lc) }
}
impl<G: GpuCurve, T: BorrowMut<Transcript>> RandomizableConstraintSystem<G::ScalarField> for Verifier<G, T> {
    type RandomizedCS = RandomizingCs<G>;
    fn specify_randomized_constraints<FF>(&mut self, callback: FF) -> Result<(), R1CSError>
    where FF: 'static + Fn(&mut Self::RandomizedCS) -> Result<(), R1CSError> { specify(&mut self.rec, callback) }
}

/// `batch_verify(prng, instances, pc_gens, bp_gens)` (verifier.rs:604-691): one weight per instance drawn from `prng` in instance
/// order (:649), all instances in ONE library call (block pipeline + one mega-check MSM).
/// Difference from the reference in what `prng` has consumed AFTER AN ERROR: the reference draws an instance's weight only after
/// that instance's verification_scalars succeeded (:617-649) and returns on the first failure, so after an error fewer draws were
/// made; here every weight is drawn before the one library call.  On success the draws are identical.
pub fn batch_verify<'a, G, I, R, T>(prng: &mut R, instances: I, pc_gens: &PedersenGens<G>, bp_gens: &BulletproofGens<G>) -> Result<(), R1CSError>
where
    G: GpuCurve,
    R: CryptoRng + RngCore,
    T: BorrowMut<Transcript>,
    I: IntoIterator<Item = (Verifier<G, T>, &'a R1CSProof<G>)>,
{
    assert_default_pc_gens::<G>(pc_gens);
    let mut verifiers: Vec<Verifier<G, T>> = Vec::new();
    let mut bytes: Vec<u8> = Vec::new();
    let mut lens: Vec<usize> = Vec::new();
    for (mut v, proof) in instances {
        v.rec.sync_to_engine();
        let b = proof.to_bytes().map_err(|_| R1CSError::FormatError)?;
        lens.push(b.len());
        bytes.extend_from_slice(&b);
        verifiers.push(v);
    }
    if verifiers.is_empty() {
        return Ok(());
    }
    let mut alphas: Vec<u64> = Vec::with_capacity(verifiers.len() * 4);
    for _ in 0..verifiers.len() {
        alphas.extend_from_slice(&G::scalar_words(&G::ScalarField::rand(prng)));
    }
    let raws: Vec<*mut ffi::BpCs> = verifiers.iter().map(|v| v.rec.raw).collect();
    let rc = with_ctx::<G, _>(|ctx| {
        ctx.ensure_gens(bp_gens);
        unsafe { ffi::bp_r1cs_batch_verify(ctx.raw, raws.len(), raws.as_ptr(), bytes.as_ptr(), lens.as_ptr(), alphas.as_ptr(), null_mut(), null_mut()) }
    });
    if rc == GADGET_FAILED {
        return Err(R1CSError::GadgetError { description: "a randomized-constraints closure failed".into() });
    }
    map_status(rc)
}

/// keeps the registry type alive for hosts that want one context per GPU of a node instead of the per-thread default
pub struct ContextRegistry(Mutex<HashMap<(c_int, c_int), GpuContext>>);
impl ContextRegistry {
    pub fn new() -> Self { ContextRegistry(Mutex::new(HashMap::new())) }
    pub fn with<G: GpuCurve, R>(&self, device: c_int, f: impl FnOnce(&mut GpuContext) -> R) -> R {
        let mut m = self.0.lock().unwrap();
        f(m.entry((G::CURVE_ID, device)).or_insert_with(|| GpuContext::new::<G>(device)))
    }
}
