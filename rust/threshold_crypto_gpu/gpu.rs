//! `src/gpu.rs` of threshold_crypto 0.4.0 behind the optional feature "mi355x": batch forms of the crate's hot-path
//! methods on an AMD MI355X through libtc_amd.so (crate `tc_amd_sys`, generated from include/tc_amd.h).
//!
//! Drop-in: the single-item methods of the crate stay as they are (the `pairing` crate on the CPU); every method here is
//! the same computation for B independent jobs and returns exactly what B calls of the single-item method return -- the
//! library's results are bit-identical encodings of the same group elements.  The module lives INSIDE the crate because
//! the tuple fields of `PublicKey(G1)`, `Signature(G2)`, `SecretKey(Fr)`, ... are private (src/lib.rs:79,202,302) and
//! `Commitment::coeff` / `Poly::coeff` are `pub(super)` (src/poly.rs:43,432).
//!
//! NO PANICS (SURVEY 8b "Errors"; the reference's hot path returns `Result` / `Option` / `bool`, src/error.rs:7-17, src/lib.rs:608-626):
//! every method here returns `GpuResult<_>`.  A call-level failure of the library (TC_ERR_HIP from a failed hipMalloc, a lost device,
//! TC_ERR_INVALID_ARG) comes back as `Err(GpuError)`, a malformed batch (jobs of different sizes) as `Err(GpuError)` with
//! TC_ERR_INVALID_ARG, a per-job failure as that job's `Err(JobError)`.  There is no `panic!`, `assert!`, `unwrap` or `expect` in this
//! file (`tests/test_abi.py::test_rust_shim_has_no_panics` greps for them): a consensus node must not abort because a line buffer did
//! not fit in HBM.
//!
//! SOURCE ONLY: the build image of the MI355X engine has no Rust toolchain, so this file has not been compiled there;
//! `tests/test_abi.py::test_rust_bindings_match_header` keeps the FFI declarations it relies on in step with the header,
//! and `test_rust_shim_calls_existing_entry_points_with_matching_arity` checks every call below against them.
//!
//! Rows of SURVEY.md section 8(a) and where they are:
//!   A2  hash_g2                         hash_g2_batch
//!   A4  SecretKey(Share)::sign / sign_g2  SecretKey::sign_batch, sign_g2_batch, SecretKeySet::sign_shares_batch
//!   A5  SecretKeyShare::decrypt_share   SecretKeyShare::decrypt_share_no_verify_batch, decrypt_share_batch; SecretKey::decrypt_batch
//!   A6/A7 PublicKeySet::combine_signatures  PublicKeySet::combine_signatures_batch
//!   A8  PublicKeySet::decrypt           PublicKeySet::decrypt_batch
//!   A9  PublicKey::verify / verify_g2   PublicKey::verify_batch, verify_g2_batch, verify_rlc_batch (opt-in)
//!   A10 Ciphertext::verify              Ciphertext::verify_batch
//!   A11 PublicKeyShare::verify_decryption_share  PublicKeyShare::verify_decryption_share_batch
//!   A12 PublicKeySet::public_key_share  PublicKeySet::public_key_shares
//!   A13 to_bytes / from_bytes           g1_to_bytes_batch, g2_to_bytes_batch, g1_from_bytes_batch, g2_from_bytes_batch
//!   f2  share validation                PublicKeySet::verify_signature_shares_rlc (opt-in)
//!   f3  PublicKey::encrypt              PublicKey::encrypt_batch
//!   f4  Poly::commitment, BivarCommitment::row, Poly::interpolate   commitment_batch, bivar_commitment_rows, interpolate_batch
//!   (e) several GPUs of one node        GpuGroup
use crate::error::{Error, FromBytesError};
use crate::poly::{BivarCommitment, Commitment, Poly};
use crate::{
    Ciphertext, DecryptionShare, Fr, G2Affine, IntoFr, PublicKey, PublicKeySet, PublicKeyShare, SecretKey, SecretKeySet,
    SecretKeyShare, Signature, SignatureShare, G1, G2, PK_SIZE, SIG_SIZE,
};
use ff::{Field, PrimeField};
use group::{CurveAffine, CurveProjective, EncodedPoint};
use pairing::bls12_381::{FrRepr, G1Uncompressed, G2Uncompressed};
use std::os::raw::c_int;
use tc_amd_sys::*;

pub const G1_BYTES: usize = 96;
pub const G2_BYTES: usize = 192;
pub const FR_BYTES: usize = 32;

/// One context of libtc_amd.so bound to one MI355X.  There is no CPU fallback: `new` fails without a gfx950 device.
pub struct Gpu(*mut TcCtx);
unsafe impl Send for Gpu {}

/// A call-level failure: the library's return code (TC_ERR_INVALID_ARG / TC_ERR_HIP / TC_ERR_NO_DEVICE / TC_ERR_HOST, or GPU_ERR_BAD_ANSWER for
/// bytes the shim could not turn back into a group element) and its message (`tc_last_error`).  The context stays usable.
#[derive(Debug, Clone, PartialEq)]
pub struct GpuError(pub c_int, pub String);
pub type GpuResult<T> = std::result::Result<T, GpuError>;
/// the library answered with bytes that are no canonical encoding, or a status no typed operand can cause
pub const GPU_ERR_BAD_ANSWER: c_int = -1000;
impl std::fmt::Display for GpuError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "libtc_amd call failed ({}): {}", self.0, self.1)
    }
}
impl std::error::Error for GpuError {}
fn shape_error(what: &str) -> GpuError {
    GpuError(TC_ERR_INVALID_ARG, what.into())
}
fn bad_answer(what: &str) -> GpuError {
    GpuError(GPU_ERR_BAD_ANSWER, what.into())
}
/// every job of a call whose operands are typed values of this crate must come back TC_JOB_OK
fn all_ok(st: &[u8]) -> GpuResult<()> {
    match st.iter().position(|&s| s != TC_JOB_OK) {
        None => Ok(()),
        Some(j) => Err(bad_answer(&format!("job {} returned status {} for typed operands", j, st[j]))),
    }
}

impl Gpu {
    pub fn new(device: i32) -> std::result::Result<Self, GpuError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { tc_ctx_create(&mut p, device as c_int) };
        if rc == TC_OK {
            Ok(Gpu(p))
        } else {
            Err(GpuError(rc, "tc_ctx_create failed (no gfx950 HIP device?)".into()))
        }
    }
    /// Operands that are typed values of this crate are group members by construction (`from_bytes` checked them), so the
    /// per-operand membership tests a context runs by default can be switched off for them.
    pub fn trusted_operands(&self, trusted: bool) {
        unsafe { tc_ctx_set_input_checks(self.0, if trusted { 0 } else { 1 }) };
    }
    /// call-level return code -> `Err(GpuError)` with the library's message; never a panic
    fn check(&self, rc: c_int) -> GpuResult<()> {
        if rc == TC_OK {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(tc_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(GpuError(rc, msg))
    }
}
impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { tc_ctx_destroy(self.0) }
    }
}

// ---- encodings (the reference's own canonical forms) ------------------------------------------------------
fn g1_bytes(p: &G1) -> [u8; G1_BYTES] {
    let mut b = [0u8; G1_BYTES];
    b.copy_from_slice(p.into_affine().into_uncompressed().as_ref());
    b
}
fn g2_bytes(p: &G2) -> [u8; G2_BYTES] {
    let mut b = [0u8; G2_BYTES];
    b.copy_from_slice(p.into_affine().into_uncompressed().as_ref());
    b
}
fn g1_from(b: &[u8]) -> GpuResult<G1> {
    let mut u = G1Uncompressed::empty();
    if b.len() != u.as_ref().len() {
        return Err(bad_answer("G1 answer of the wrong length"));
    }
    u.as_mut().copy_from_slice(b);
    u.into_affine_unchecked().map(|a| a.into_projective()).map_err(|_| bad_answer("G1 answer is no canonical encoding"))
}
fn g2_from(b: &[u8]) -> GpuResult<G2> {
    let mut u = G2Uncompressed::empty();
    if b.len() != u.as_ref().len() {
        return Err(bad_answer("G2 answer of the wrong length"));
    }
    u.as_mut().copy_from_slice(b);
    u.into_affine_unchecked().map(|a| a.into_projective()).map_err(|_| bad_answer("G2 answer is no canonical encoding"))
}
fn g1_vec(out: &[u8]) -> GpuResult<Vec<G1>> {
    out.chunks(G1_BYTES).map(g1_from).collect()
}
fn g2_vec(out: &[u8]) -> GpuResult<Vec<G2>> {
    out.chunks(G2_BYTES).map(g2_from).collect()
}
/// Fr as 4 little-endian u64 limbs of `into_repr()` (src/serde_impl.rs:296)
fn fr_bytes(f: &Fr) -> [u8; FR_BYTES] {
    let mut b = [0u8; FR_BYTES];
    for (i, limb) in f.into_repr().as_ref().iter().enumerate() {
        b[8 * i..8 * i + 8].copy_from_slice(&limb.to_le_bytes());
    }
    b
}
fn fr_from(b: &[u8]) -> GpuResult<Fr> {
    if b.len() != FR_BYTES {
        return Err(bad_answer("Fr answer of the wrong length"));
    }
    let mut r = FrRepr::default();
    for (i, limb) in r.as_mut().iter_mut().enumerate() {
        let mut w = [0u8; 8];
        w.copy_from_slice(&b[8 * i..8 * i + 8]);
        *limb = u64::from_le_bytes(w);
    }
    Fr::from_repr(r).map_err(|_| bad_answer("Fr answer is not canonical"))
}
fn pack_messages<M: AsRef<[u8]>>(msgs: &[M]) -> (Vec<u8>, Vec<u64>) {
    let mut flat = Vec::new();
    let mut off = Vec::with_capacity(msgs.len() + 1);
    off.push(0u64);
    for m in msgs {
        flat.extend_from_slice(m.as_ref());
        off.push(flat.len() as u64);
    }
    if flat.is_empty() {
        flat.push(0);
    }
    (flat, off)
}
/// Per-job status -> the per-job result of every batch method.  The reference's hot path has no panic (only the unreachable
/// `IntoFr` expect, src/into_fr.rs:18), so none here either: TC_JOB_INVALID_ENCODING -- an operand the library could not decode or
/// that is no group member; typed values of this crate cannot cause it, a context with its membership tests on reports foreign
/// operands this way -- comes back as `JobError::Invalid(FromBytesError::Invalid)`, never as `DuplicateEntry` and never as an abort.
/// A status byte this shim does not know (a newer library, a corrupted answer) is `JobError::Internal(byte)`: distinguishable
/// from a bad operand.
fn status_to_result<T>(st: u8, v: GpuResult<T>) -> GpuResult<JobResult<T>> {
    match st {
        TC_JOB_OK => v.map(Ok),      // the answer of a job that succeeded must decode: a call-level error if it does not
        _ => Ok(wire_status_to_result(st, ()).and_then(|_| Err(WireError::Internal(st)))),
    }
}
/// Errors of the wire-level entry points: the reference's `Error` for the threshold logic, `FromBytesError::Invalid` for a
/// share that does not decode or is no group member (src/error.rs:37-41).
#[derive(Debug, PartialEq)]
pub enum WireError {
    Threshold(Error),
    Invalid(FromBytesError),
    /// a per-job status byte outside TC_JOB_*: not an operand's fault
    Internal(u8),
}
/// The per-job error of the batch methods (the typed ones too: r05).
pub type JobError = WireError;
pub type JobResult<T> = std::result::Result<T, JobError>;
fn wire_status_to_result<T>(st: u8, v: T) -> std::result::Result<T, WireError> {
    match st {
        TC_JOB_OK => Ok(v),
        TC_JOB_NOT_ENOUGH_SHARES => Err(WireError::Threshold(Error::NotEnoughShares)),
        TC_JOB_DUPLICATE_ENTRY => Err(WireError::Threshold(Error::DuplicateEntry)),
        TC_JOB_INVALID_ENCODING => Err(WireError::Invalid(FromBytesError::Invalid)),
        // (an unknown status byte: reported as such, not a panic and not blamed on an operand)
        other => Err(WireError::Internal(other)),
    }
}
/// `T: IntoFr` abscissae that all fit 64 bits travel as u64 (tc_combine_g2_batch / tc_decrypt_batch: no narrowing kernel, no
/// scratch array, no host wait inside the library -- ADVICE r04); `None` when one of them does not.
fn abscissae_as_u64(frs: &[Fr]) -> Option<Vec<u64>> {
    frs.iter()
        .map(|f| {
            let r = f.into_repr();
            let l = r.as_ref();
            if l[1..].iter().all(|&w| w == 0) { Some(l[0]) } else { None }
        })
        .collect()
}

// ---- A2: hash_g2 (src/lib.rs:691-694) ---------------------------------------------------------------------------------
pub fn hash_g2_batch<M: AsRef<[u8]>>(gpu: &Gpu, msgs: &[M]) -> GpuResult<Vec<G2>> {
    let (flat, off) = pack_messages(msgs);
    let mut out = vec![0u8; msgs.len() * G2_BYTES];
    gpu.check(unsafe { tc_hash_g2_batch(gpu.0, flat.as_ptr(), off.as_ptr(), msgs.len(), out.as_mut_ptr()) })?;
    g2_vec(&out)
}

// ---- A4: signing (src/lib.rs:372-381, 442-449) -----------------------------------------------------------------------
impl SecretKey {
    /// `sign` for B messages: hash_g2 and the multiplication both on the device (tc_sign_batch, S = 1).
    pub fn sign_batch<M: AsRef<[u8]>>(&self, gpu: &Gpu, msgs: &[M]) -> GpuResult<Vec<Signature>> {
        let (flat, off) = pack_messages(msgs);
        let mut fr = fr_bytes(&self.0);
        let (mut out, mut st) = (vec![0u8; msgs.len() * G2_BYTES], vec![0u8; msgs.len()]);
        let rc = unsafe { tc_sign_batch(gpu.0, fr.as_ptr(), flat.as_ptr(), off.as_ptr(), 1, msgs.len(), out.as_mut_ptr(), st.as_mut_ptr()) };
        fr.iter_mut().for_each(|b| *b = 0); // the reference zeroizes secrets (src/lib.rs:304-314)
        gpu.check(rc)?;
        all_ok(&st)?;
        Ok(g2_vec(&out)?.into_iter().map(Signature).collect())
    }
    /// `sign_g2` for B pre-hashed messages (tc_g2_mul_batch, S = 1).
    pub fn sign_g2_batch(&self, gpu: &Gpu, hashes: &[G2Affine]) -> GpuResult<Vec<Signature>> {
        let mut fr = fr_bytes(&self.0);
        let mut pts = Vec::with_capacity(hashes.len() * G2_BYTES);
        for h in hashes {
            pts.extend_from_slice(h.into_uncompressed().as_ref());
        }
        let (mut out, mut st) = (vec![0u8; hashes.len() * G2_BYTES], vec![0u8; hashes.len()]);
        let rc = unsafe { tc_g2_mul_batch(gpu.0, fr.as_ptr(), pts.as_ptr(), 1, hashes.len(), out.as_mut_ptr(), st.as_mut_ptr()) };
        fr.iter_mut().for_each(|b| *b = 0);
        gpu.check(rc)?;
        all_ok(&st)?;
        Ok(g2_vec(&out)?.into_iter().map(Signature).collect())
    }
}
impl SecretKeyShare {
    pub fn sign_batch<M: AsRef<[u8]>>(&self, gpu: &Gpu, msgs: &[M]) -> GpuResult<Vec<SignatureShare>> {
        Ok(self.0.sign_batch(gpu, msgs)?.into_iter().map(SignatureShare).collect())
    }
    pub fn sign_g2_batch(&self, gpu: &Gpu, hashes: &[G2Affine]) -> GpuResult<Vec<SignatureShare>> {
        Ok(self.0.sign_g2_batch(gpu, hashes)?.into_iter().map(SignatureShare).collect())
    }
}
impl SecretKeySet {
    /// The shares of B messages by per-message signer subsets, generated on the device from the key set
    /// (tc_sign_shares_g2_batch): `signers[j]` lists the n share indices of message j; out[j][k] = share(signers[j][k]).sign_g2(hashes[j]).
    pub fn sign_shares_batch(&self, gpu: &Gpu, n_nodes: usize, signers: &[Vec<u64>], hashes: &[G2Affine]) -> GpuResult<Vec<Vec<SignatureShare>>> {
        let b = hashes.len();
        let n = signers.first().map(|s| s.len()).unwrap_or(0);
        if signers.len() != b || signers.iter().any(|s| s.len() != n) {
            return Err(shape_error("one signer list of the same length per message"));
        }
        if b == 0 || n == 0 {
            return Ok(vec![Vec::new(); b]);
        }
        let mut table = Vec::with_capacity(n_nodes * FR_BYTES);
        for i in 0..n_nodes {
            table.extend_from_slice(&fr_bytes(&(self.secret_key_share(i).0).0));
        }
        let idx: Vec<u64> = signers.iter().flatten().cloned().collect();
        let mut pts = Vec::with_capacity(b * G2_BYTES);
        for h in hashes {
            pts.extend_from_slice(h.into_uncompressed().as_ref());
        }
        let (mut out, mut st) = (vec![0u8; b * n * G2_BYTES], vec![0u8; b * n]);
        let rc = unsafe { tc_sign_shares_g2_batch(gpu.0, table.as_ptr(), n_nodes, idx.as_ptr(), pts.as_ptr(), n, b, out.as_mut_ptr(), st.as_mut_ptr()) };
        table.iter_mut().for_each(|x| *x = 0);
        gpu.check(rc)?;
        all_ok(&st)?;
        out.chunks(n * G2_BYTES)
            .map(|job| -> GpuResult<Vec<SignatureShare>> { Ok(g2_vec(job)?.into_iter().map(|p| SignatureShare(Signature(p))).collect()) })
            .collect()
    }
}

// ---- A5: decryption shares (src/lib.rs:452-462) -----------------------------------------------------------------------
impl SecretKeyShare {
    /// `decrypt_share_no_verify` for B ciphertexts (tc_g1_mul_batch, S = 1).
    pub fn decrypt_share_no_verify_batch(&self, gpu: &Gpu, cts: &[Ciphertext]) -> GpuResult<Vec<DecryptionShare>> {
        let mut fr = fr_bytes(&(self.0).0);
        let mut pts = Vec::with_capacity(cts.len() * G1_BYTES);
        for ct in cts {
            pts.extend_from_slice(&g1_bytes(&ct.0));
        }
        let (mut out, mut st) = (vec![0u8; cts.len() * G1_BYTES], vec![0u8; cts.len()]);
        let rc = unsafe { tc_g1_mul_batch(gpu.0, fr.as_ptr(), pts.as_ptr(), 1, cts.len(), out.as_mut_ptr(), st.as_mut_ptr()) };
        fr.iter_mut().for_each(|b| *b = 0);
        gpu.check(rc)?;
        all_ok(&st)?;
        Ok(g1_vec(&out)?.into_iter().map(DecryptionShare).collect())
    }
    /// `decrypt_share`: `None` where the ciphertext does not verify (src/lib.rs:452-457).  ONE call: Ciphertext::verify and the
    /// multiplication both on the device (tc_decrypt_share_batch); a ciphertext that fails the check never yields [sk] u.
    pub fn decrypt_share_batch(&self, gpu: &Gpu, cts: &[Ciphertext]) -> GpuResult<Vec<Option<DecryptionShare>>> {
        let mut fr = fr_bytes(&(self.0).0);
        let (u, flat, off, w) = ciphertext_columns(cts);
        let (mut out, mut ok) = (vec![0u8; cts.len() * G1_BYTES], vec![0u8; cts.len()]);
        let rc = unsafe {
            tc_decrypt_share_batch(gpu.0, fr.as_ptr(), u.as_ptr(), flat.as_ptr(), off.as_ptr(), w.as_ptr(), cts.len(), out.as_mut_ptr(), ok.as_mut_ptr())
        };
        fr.iter_mut().for_each(|b| *b = 0);
        gpu.check(rc)?;
        out.chunks(G1_BYTES).zip(ok).map(|(c, good)| if good == 1 { g1_from(c).map(|p| Some(DecryptionShare(p))) } else { Ok(None) }).collect()
    }
}
impl SecretKey {
    /// `decrypt` for B ciphertexts (src/lib.rs:384-391): `None` where the ciphertext does not verify; verify, [sk] u and
    /// xor_with_hash in one call (tc_secret_key_decrypt_batch).
    pub fn decrypt_batch(&self, gpu: &Gpu, cts: &[Ciphertext]) -> GpuResult<Vec<Option<Vec<u8>>>> {
        let mut fr = fr_bytes(&self.0);
        let (u, flat, off, w) = ciphertext_columns(cts);
        let (mut out, mut ok) = (vec![0u8; flat.len()], vec![0u8; cts.len()]);
        let rc = unsafe {
            tc_secret_key_decrypt_batch(gpu.0, fr.as_ptr(), u.as_ptr(), flat.as_ptr(), off.as_ptr(), w.as_ptr(), cts.len(), out.as_mut_ptr(), ok.as_mut_ptr())
        };
        fr.iter_mut().for_each(|b| *b = 0);
        gpu.check(rc)?;
        Ok((0..cts.len()).map(|j| if ok[j] == 1 { Some(out[off[j] as usize..off[j + 1] as usize].to_vec()) } else { None }).collect())
    }
}
/// the columns of a ciphertext batch as the C ABI takes them: u (96 B each), the v bytes + offsets, w (192 B each)
fn ciphertext_columns(cts: &[Ciphertext]) -> (Vec<u8>, Vec<u8>, Vec<u64>, Vec<u8>) {
    let (mut u, mut w) = (Vec::with_capacity(cts.len() * G1_BYTES), Vec::with_capacity(cts.len() * G2_BYTES));
    for ct in cts {
        u.extend_from_slice(&g1_bytes(&ct.0));
        w.extend_from_slice(&g2_bytes(&ct.2));
    }
    let vs: Vec<&[u8]> = cts.iter().map(|c| c.1.as_slice()).collect();
    let (flat, off) = pack_messages(&vs);
    (u, flat, off, w)
}

// ---- A6 / A7 / A8 / A12: the key set (src/lib.rs:565-626, 719-773) -----------------------------------------------------
impl PublicKeySet {
    fn commit_bytes(&self) -> Vec<u8> {
        self.commit.coeff.iter().flat_map(|c| g1_bytes(c).to_vec()).collect()
    }
    /// Batch form of `combine_signatures` (src/lib.rs:608-615), generic over the index type like the original (`T: IntoFr`:
    /// u64 / usize, Fr, negative i32 / i64 -- src/into_fr.rs).  `jobs[j]` iterates `(index, share)` in the order the
    /// single-item method would see it (BTreeMap order); every job holds the same number of shares.  The abscissae travel as
    /// Fr values (tc_combine_g2_fr_batch); a batch whose indices all fit 64 bits runs the u64 kernels inside the library.
    pub fn combine_signatures_batch<'a, T, I>(&self, gpu: &Gpu, jobs: &[I]) -> GpuResult<Vec<JobResult<Signature>>>
    where
        T: IntoFr,
        I: Clone + IntoIterator<Item = (T, &'a SignatureShare)>,
    {
        let t = self.threshold();
        let n = jobs.first().map(|j| j.clone().into_iter().count()).unwrap_or(0);
        let (mut frs, mut shares) = (Vec::with_capacity(jobs.len() * n), Vec::with_capacity(jobs.len() * n * G2_BYTES));
        for job in jobs {
            for (i, s) in job.clone() {
                frs.push(i.into_fr());
                shares.extend_from_slice(&g2_bytes(&(s.0).0));
            }
        }
        if frs.len() != jobs.len() * n {
            return Err(shape_error("every job must hold the same number of shares"));
        }
        let (mut out, mut st) = (vec![0u8; jobs.len() * G2_BYTES], vec![0u8; jobs.len()]);
        if let Some(idx) = abscissae_as_u64(&frs) {
            gpu.check(unsafe { tc_combine_g2_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        } else {
            let idx: Vec<u8> = frs.iter().flat_map(|f| fr_bytes(f).to_vec()).collect();
            gpu.check(unsafe { tc_combine_g2_fr_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        }
        st.iter().enumerate().map(|(j, &s)| status_to_result(s, g2_from(&out[j * G2_BYTES..(j + 1) * G2_BYTES]).map(Signature))).collect()
    }
    /// The same for plain u64 indices without the detour through Fr (tc_combine_g2_batch).
    pub fn combine_signatures_batch_u64<'a, I>(&self, gpu: &Gpu, jobs: &[I]) -> GpuResult<Vec<JobResult<Signature>>>
    where
        I: Clone + IntoIterator<Item = (u64, &'a SignatureShare)>,
    {
        let t = self.threshold();
        let n = jobs.first().map(|j| j.clone().into_iter().count()).unwrap_or(0);
        let (mut idx, mut shares) = (Vec::with_capacity(jobs.len() * n), Vec::with_capacity(jobs.len() * n * G2_BYTES));
        for job in jobs {
            for (i, s) in job.clone() {
                idx.push(i);
                shares.extend_from_slice(&g2_bytes(&(s.0).0));
            }
        }
        if idx.len() != jobs.len() * n {
            return Err(shape_error("every job must hold the same number of shares"));
        }
        let (mut out, mut st) = (vec![0u8; jobs.len() * G2_BYTES], vec![0u8; jobs.len()]);
        gpu.check(unsafe { tc_combine_g2_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        st.iter().enumerate().map(|(j, &s)| status_to_result(s, g2_from(&out[j * G2_BYTES..(j + 1) * G2_BYTES]).map(Signature))).collect()
    }
    /// Wire-level `combine_signatures`: the shares as they arrive (`SignatureShare::to_bytes`, 96 bytes each; checked decode of
    /// `from_bytes`, src/lib.rs:246-252, ON the device), the combined signature as `Signature::to_bytes` (src/lib.rs:255-259).
    pub fn combine_signatures_wire_batch(&self, gpu: &Gpu, jobs: &[Vec<(u64, [u8; SIG_SIZE])>]) -> GpuResult<Vec<std::result::Result<[u8; SIG_SIZE], WireError>>> {
        let t = self.threshold();
        let n = jobs.first().map(|j| j.len()).unwrap_or(0);
        if jobs.iter().any(|j| j.len() != n) {
            return Err(shape_error("every job must hold the same number of shares"));
        }
        let (mut idx, mut shares) = (Vec::with_capacity(jobs.len() * n), Vec::with_capacity(jobs.len() * n * SIG_SIZE));
        for job in jobs {
            for (i, s) in job {
                idx.push(*i);
                shares.extend_from_slice(s);
            }
        }
        let (mut out, mut st) = (vec![0u8; jobs.len() * SIG_SIZE], vec![0u8; jobs.len()]);
        gpu.check(unsafe { tc_combine_signatures_wire_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok(st
            .iter()
            .enumerate()
            .map(|(j, &s)| {
                let mut sig = [0u8; SIG_SIZE];
                sig.copy_from_slice(&out[j * SIG_SIZE..(j + 1) * SIG_SIZE]);
                wire_status_to_result(s, sig)
            })
            .collect())
    }
    /// Batch form of `decrypt` (src/lib.rs:618-626), generic over `T: IntoFr`: per job the shares of one ciphertext; returns the
    /// plaintexts.
    pub fn decrypt_batch<'a, T, I>(&self, gpu: &Gpu, jobs: &[I], cts: &[Ciphertext]) -> GpuResult<Vec<JobResult<Vec<u8>>>>
    where
        T: IntoFr,
        I: Clone + IntoIterator<Item = (T, &'a DecryptionShare)>,
    {
        if jobs.len() != cts.len() {
            return Err(shape_error("one ciphertext per share set"));
        }
        let t = self.threshold();
        let n = jobs.first().map(|j| j.clone().into_iter().count()).unwrap_or(0);
        let (mut frs, mut shares) = (Vec::with_capacity(jobs.len() * n), Vec::with_capacity(jobs.len() * n * G1_BYTES));
        for job in jobs {
            for (i, s) in job.clone() {
                frs.push(i.into_fr());
                shares.extend_from_slice(&g1_bytes(&s.0));
            }
        }
        if frs.len() != jobs.len() * n {
            return Err(shape_error("every job must hold the same number of shares"));
        }
        let vs: Vec<&[u8]> = cts.iter().map(|c| c.1.as_slice()).collect();
        let (flat, off) = pack_messages(&vs);
        let (mut out, mut st) = (vec![0u8; flat.len()], vec![0u8; jobs.len()]);
        if let Some(idx) = abscissae_as_u64(&frs) {
            gpu.check(unsafe {
                tc_decrypt_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), flat.as_ptr(), off.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr())
            })?;
        } else {
            let idx: Vec<u8> = frs.iter().flat_map(|f| fr_bytes(f).to_vec()).collect();
            gpu.check(unsafe {
                tc_decrypt_fr_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), flat.as_ptr(), off.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr())
            })?;
        }
        Ok(st.iter().enumerate().map(|(j, &s)| wire_status_to_result(s, out[off[j] as usize..off[j + 1] as usize].to_vec())).collect())
    }
    /// Wire-level `decrypt`: the decryption shares in their 48-byte compressed form (checked decode on the device).
    pub fn decrypt_wire_batch(&self, gpu: &Gpu, jobs: &[Vec<(u64, [u8; PK_SIZE])>], cts: &[Ciphertext]) -> GpuResult<Vec<std::result::Result<Vec<u8>, WireError>>> {
        if jobs.len() != cts.len() {
            return Err(shape_error("one ciphertext per share set"));
        }
        let t = self.threshold();
        let n = jobs.first().map(|j| j.len()).unwrap_or(0);
        if jobs.iter().any(|j| j.len() != n) {
            return Err(shape_error("every job must hold the same number of shares"));
        }
        let (mut idx, mut shares) = (Vec::with_capacity(jobs.len() * n), Vec::with_capacity(jobs.len() * n * PK_SIZE));
        for job in jobs {
            for (i, s) in job {
                idx.push(*i);
                shares.extend_from_slice(s);
            }
        }
        let vs: Vec<&[u8]> = cts.iter().map(|c| c.1.as_slice()).collect();
        let (flat, off) = pack_messages(&vs);
        let (mut out, mut st) = (vec![0u8; flat.len()], vec![0u8; jobs.len()]);
        gpu.check(unsafe {
            tc_decrypt_wire_batch(gpu.0, t, n, idx.as_ptr(), shares.as_ptr(), flat.as_ptr(), off.as_ptr(), jobs.len(), out.as_mut_ptr(), st.as_mut_ptr())
        })?;
        Ok(st.iter().enumerate().map(|(j, &s)| wire_status_to_result(s, out[off[j] as usize..off[j + 1] as usize].to_vec())).collect())
    }
    /// `public_key_share(i)` for many indices (src/lib.rs:570-573 -> Commitment::evaluate, src/poly.rs:497-508).
    pub fn public_key_shares(&self, gpu: &Gpu, indices: &[u64]) -> GpuResult<Vec<PublicKeyShare>> {
        let commit = self.commit_bytes();
        let (mut out, mut st) = (vec![0u8; indices.len() * G1_BYTES], vec![0u8; indices.len()]);
        gpu.check(unsafe { tc_public_key_share_batch(gpu.0, commit.as_ptr(), self.threshold(), indices.as_ptr(), indices.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        all_ok(&st)?;
        Ok(g1_vec(&out)?.into_iter().map(|p| PublicKeyShare(PublicKey(p))).collect())
    }
    /// The share-validation loop of examples/threshold_sig.rs:115-131 for B messages x N nodes by ONE random linear
    /// combination per message (opt-in): ok[j][i] = pk_share(i).verify(&shares[j][i], msgs[j]).
    pub fn verify_signature_shares_rlc<M: AsRef<[u8]>>(&self, gpu: &Gpu, n_nodes: usize, shares: &[Vec<SignatureShare>], msgs: &[M], seed: &[u8; 32]) -> GpuResult<Vec<Vec<bool>>> {
        if shares.len() != msgs.len() || shares.iter().any(|job| job.len() != n_nodes) {
            return Err(shape_error("one share of every node per message"));
        }
        if n_nodes == 0 {
            return Ok(vec![Vec::new(); msgs.len()]);
        }
        let pks: Vec<u8> = self.public_key_shares(gpu, &(0..n_nodes as u64).collect::<Vec<_>>())?.iter().flat_map(|p| g1_bytes(&(p.0).0).to_vec()).collect();
        let mut sig = Vec::with_capacity(msgs.len() * n_nodes * G2_BYTES);
        for job in shares {
            for s in job {
                sig.extend_from_slice(&g2_bytes(&(s.0).0));
            }
        }
        let (flat, off) = pack_messages(msgs);
        let mut ok = vec![0u8; msgs.len() * n_nodes];
        let mut fallback = 0u64;
        gpu.check(unsafe {
            tc_verify_shares_rlc_batch(gpu.0, pks.as_ptr(), n_nodes, sig.as_ptr(), flat.as_ptr(), off.as_ptr(), msgs.len(), seed.as_ptr(), ok.as_mut_ptr(), &mut fallback)
        })?;
        Ok(ok.chunks(n_nodes).map(|r| r.iter().map(|&b| b == 1).collect()).collect())
    }
}

// ---- A9: verification (src/lib.rs:108-117) ----------------------------------------------------------------------------
impl PublicKey {
    /// Batch form of `verify`: one key, B (signature, message) pairs; hashing on the device.
    pub fn verify_batch<M: AsRef<[u8]>>(&self, gpu: &Gpu, sigs: &[Signature], msgs: &[M]) -> GpuResult<Vec<bool>> {
        if sigs.len() != msgs.len() {
            return Err(shape_error("one message per signature"));
        }
        let pk = g1_bytes(&self.0);
        let (flat, off) = pack_messages(msgs);
        let mut s = Vec::with_capacity(sigs.len() * G2_BYTES);
        for sig in sigs {
            s.extend_from_slice(&g2_bytes(&sig.0));
        }
        let mut ok = vec![0u8; sigs.len()];
        gpu.check(unsafe { tc_verify_sig_batch(gpu.0, pk.as_ptr(), 0 /* broadcast the key */, s.as_ptr(), flat.as_ptr(), off.as_ptr(), sigs.len(), ok.as_mut_ptr()) })?;
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
    /// Batch form of `verify_g2` (src/lib.rs:108-110): pre-hashed messages.
    pub fn verify_g2_batch(&self, gpu: &Gpu, sigs: &[Signature], hashes: &[G2Affine]) -> GpuResult<Vec<bool>> {
        if sigs.len() != hashes.len() {
            return Err(shape_error("one hash point per signature"));
        }
        let pk = g1_bytes(&self.0);
        let (mut s, mut h) = (Vec::with_capacity(sigs.len() * G2_BYTES), Vec::with_capacity(sigs.len() * G2_BYTES));
        for (sig, hash) in sigs.iter().zip(hashes) {
            s.extend_from_slice(&g2_bytes(&sig.0));
            h.extend_from_slice(hash.into_uncompressed().as_ref());
        }
        let mut ok = vec![0u8; sigs.len()];
        gpu.check(unsafe { tc_verify_g2_batch(gpu.0, pk.as_ptr(), 0, s.as_ptr(), h.as_ptr(), sigs.len(), ok.as_mut_ptr()) })?;
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
    /// The same booleans as `verify_batch` through one random linear combination per group of 64 jobs (opt-in;
    /// groups that fail are re-checked job by job).  `seed`: 32 secret random bytes drawn after the signatures arrived.
    pub fn verify_rlc_batch<M: AsRef<[u8]>>(&self, gpu: &Gpu, sigs: &[Signature], msgs: &[M], seed: &[u8; 32]) -> GpuResult<Vec<bool>> {
        if sigs.len() != msgs.len() {
            return Err(shape_error("one message per signature"));
        }
        let pk = g1_bytes(&self.0);
        let (flat, off) = pack_messages(msgs);
        let mut s = Vec::with_capacity(sigs.len() * G2_BYTES);
        for sig in sigs {
            s.extend_from_slice(&g2_bytes(&sig.0));
        }
        let mut ok = vec![0u8; sigs.len()];
        let mut fallback = 0u64;
        gpu.check(unsafe { tc_verify_sig_rlc_batch(gpu.0, pk.as_ptr(), s.as_ptr(), flat.as_ptr(), off.as_ptr(), sigs.len(), 0, seed.as_ptr(), ok.as_mut_ptr(), &mut fallback) })?;
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
    /// `encrypt_with_rng` for B messages (src/lib.rs:128-137); the Fr draws come from the caller's RNG, in order
    /// (`Fr::random(rng)` per message, as at src/lib.rs:129).
    pub fn encrypt_batch<R: rand::RngCore, M: AsRef<[u8]>>(&self, gpu: &Gpu, rng: &mut R, msgs: &[M]) -> GpuResult<Vec<Ciphertext>> {
        let pk = g1_bytes(&self.0);
        let mut r = Vec::with_capacity(msgs.len() * FR_BYTES);
        for _ in msgs {
            let f: Fr = Fr::random(rng);
            r.extend_from_slice(&fr_bytes(&f));
        }
        let (flat, off) = pack_messages(msgs);
        let (mut u, mut v, mut w, mut st) = (vec![0u8; msgs.len() * G1_BYTES], vec![0u8; flat.len()], vec![0u8; msgs.len() * G2_BYTES], vec![0u8; msgs.len()]);
        let rc = unsafe {
            tc_encrypt_batch(gpu.0, pk.as_ptr(), 0, r.as_ptr(), flat.as_ptr(), off.as_ptr(), msgs.len(), u.as_mut_ptr(), v.as_mut_ptr(), w.as_mut_ptr(), st.as_mut_ptr())
        };
        r.iter_mut().for_each(|b| *b = 0);
        gpu.check(rc)?;
        all_ok(&st)?;
        (0..msgs.len())
            .map(|j| -> GpuResult<Ciphertext> {
                Ok(Ciphertext(g1_from(&u[j * G1_BYTES..(j + 1) * G1_BYTES])?, v[off[j] as usize..off[j + 1] as usize].to_vec(), g2_from(&w[j * G2_BYTES..(j + 1) * G2_BYTES])?))
            })
            .collect()
    }
}

// ---- A10 / A11: ciphertexts (src/lib.rs:182-186, 508-512) -------------------------------------------------------------
impl Ciphertext {
    /// Batch form of `verify` (src/lib.rs:508-512).
    pub fn verify_batch(gpu: &Gpu, cts: &[Ciphertext]) -> GpuResult<Vec<bool>> {
        let (mut u, mut w) = (Vec::with_capacity(cts.len() * G1_BYTES), Vec::with_capacity(cts.len() * G2_BYTES));
        for ct in cts {
            u.extend_from_slice(&g1_bytes(&ct.0));
            w.extend_from_slice(&g2_bytes(&ct.2));
        }
        let vs: Vec<&[u8]> = cts.iter().map(|c| c.1.as_slice()).collect();
        let (flat, off) = pack_messages(&vs);
        let mut ok = vec![0u8; cts.len()];
        gpu.check(unsafe { tc_ciphertext_verify_batch(gpu.0, u.as_ptr(), flat.as_ptr(), off.as_ptr(), w.as_ptr(), cts.len(), ok.as_mut_ptr()) })?;
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
}
impl PublicKeyShare {
    /// Batch form of `verify_decryption_share` (src/lib.rs:182-186): one key share, B (share, ciphertext) pairs.
    pub fn verify_decryption_share_batch(&self, gpu: &Gpu, shares: &[DecryptionShare], cts: &[Ciphertext]) -> GpuResult<Vec<bool>> {
        if shares.len() != cts.len() {
            return Err(shape_error("one ciphertext per decryption share"));
        }
        let pk = g1_bytes(&(self.0).0);
        let (mut s, mut u, mut w) = (Vec::new(), Vec::new(), Vec::new());
        for (sh, ct) in shares.iter().zip(cts) {
            s.extend_from_slice(&g1_bytes(&sh.0));
            u.extend_from_slice(&g1_bytes(&ct.0));
            w.extend_from_slice(&g2_bytes(&ct.2));
        }
        let vs: Vec<&[u8]> = cts.iter().map(|c| c.1.as_slice()).collect();
        let (flat, off) = pack_messages(&vs);
        let mut ok = vec![0u8; shares.len()];
        gpu.check(unsafe {
            tc_verify_decryption_share_batch(gpu.0, pk.as_ptr(), 0, s.as_ptr(), u.as_ptr(), flat.as_ptr(), off.as_ptr(), w.as_ptr(), shares.len(), ok.as_mut_ptr())
        })?;
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
}

// ---- A13: wire formats (src/lib.rs:140-153, 246-259) ------------------------------------------------------------------
pub fn g1_to_bytes_batch(gpu: &Gpu, pts: &[G1]) -> GpuResult<Vec<[u8; 48]>> {
    let inp: Vec<u8> = pts.iter().flat_map(|p| g1_bytes(p).to_vec()).collect();
    let (mut out, mut st) = (vec![0u8; pts.len() * 48], vec![0u8; pts.len()]);
    gpu.check(unsafe { tc_g1_compress_batch(gpu.0, inp.as_ptr(), pts.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
    all_ok(&st)?;
    Ok(out.chunks(48).map(|c| { let mut a = [0u8; 48]; a.copy_from_slice(c); a }).collect())
}
pub fn g2_to_bytes_batch(gpu: &Gpu, pts: &[G2]) -> GpuResult<Vec<[u8; 96]>> {
    let inp: Vec<u8> = pts.iter().flat_map(|p| g2_bytes(p).to_vec()).collect();
    let (mut out, mut st) = (vec![0u8; pts.len() * 96], vec![0u8; pts.len()]);
    gpu.check(unsafe { tc_g2_compress_batch(gpu.0, inp.as_ptr(), pts.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
    all_ok(&st)?;
    Ok(out.chunks(96).map(|c| { let mut a = [0u8; 96]; a.copy_from_slice(c); a }).collect())
}
/// `PublicKey::from_bytes` for B encodings: the CHECKED decode (on the curve and in the order-r subgroup).
pub fn g1_from_bytes_batch(gpu: &Gpu, enc: &[[u8; 48]]) -> GpuResult<Vec<std::result::Result<G1, FromBytesError>>> {
    let inp: Vec<u8> = enc.iter().flat_map(|e| e.to_vec()).collect();
    let (mut out, mut st) = (vec![0u8; enc.len() * G1_BYTES], vec![0u8; enc.len()]);
    gpu.check(unsafe { tc_g1_decompress_batch(gpu.0, inp.as_ptr(), enc.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
    st.iter().enumerate().map(|(j, &s)| if s == TC_JOB_OK { g1_from(&out[j * G1_BYTES..(j + 1) * G1_BYTES]).map(Ok) } else { Ok(Err(FromBytesError::Invalid)) }).collect()
}
pub fn g2_from_bytes_batch(gpu: &Gpu, enc: &[[u8; 96]]) -> GpuResult<Vec<std::result::Result<G2, FromBytesError>>> {
    let inp: Vec<u8> = enc.iter().flat_map(|e| e.to_vec()).collect();
    let (mut out, mut st) = (vec![0u8; enc.len() * G2_BYTES], vec![0u8; enc.len()]);
    gpu.check(unsafe { tc_g2_decompress_batch(gpu.0, inp.as_ptr(), enc.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
    st.iter().enumerate().map(|(j, &s)| if s == TC_JOB_OK { g2_from(&out[j * G2_BYTES..(j + 1) * G2_BYTES]).map(Ok) } else { Ok(Err(FromBytesError::Invalid)) }).collect()
}

// ---- f4: DKG algebra (src/poly.rs:372-377, 388-417, 694-727) ------------------------------------------------------------
/// `Poly::commitment` for several polynomials at once: one fixed-base multiplication per coefficient.
pub fn commitment_batch(gpu: &Gpu, polys: &[&Poly]) -> GpuResult<Vec<Commitment>> {
    let mut fr = Vec::new();
    for p in polys {
        for c in &p.coeff {
            fr.extend_from_slice(&fr_bytes(c));
        }
    }
    let m = fr.len() / FR_BYTES;
    let (mut out, mut st) = (vec![0u8; m * G1_BYTES], vec![0u8; m]);
    let rc = unsafe { tc_g1_commitment_batch(gpu.0, fr.as_ptr(), m, out.as_mut_ptr(), st.as_mut_ptr()) };
    fr.iter_mut().for_each(|b| *b = 0);
    gpu.check(rc)?;
    all_ok(&st)?;
    let pts = g1_vec(&out)?;
    let mut at = 0;
    Ok(polys
        .iter()
        .map(|p| {
            let coeff = pts[at..at + p.coeff.len()].to_vec();   // (m = the sum of the lengths: the ranges tile pts exactly)
            at += p.coeff.len();
            Commitment { coeff }
        })
        .collect())
}
/// `BivarCommitment::row(x)` for several x.
pub fn bivar_commitment_rows(gpu: &Gpu, c: &BivarCommitment, xs: &[u64]) -> GpuResult<Vec<Commitment>> {
    let inp: Vec<u8> = c.coeff.iter().flat_map(|p| g1_bytes(p).to_vec()).collect();
    let d = c.degree();
    let (mut out, mut st) = (vec![0u8; xs.len() * (d + 1) * G1_BYTES], vec![0u8; xs.len() * (d + 1)]);
    gpu.check(unsafe { tc_bivar_commitment_row_batch(gpu.0, inp.as_ptr(), d, xs.as_ptr(), xs.len(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
    all_ok(&st)?;
    out.chunks((d + 1) * G1_BYTES).map(|row| -> GpuResult<Commitment> { Ok(Commitment { coeff: g1_vec(row)? }) }).collect()
}
/// `Poly::interpolate` for B sample sets of n points each.
pub fn interpolate_batch(gpu: &Gpu, samples: &[Vec<(Fr, Fr)>]) -> GpuResult<Vec<JobResult<Poly>>> {
    let n = samples.first().map(|s| s.len()).unwrap_or(0);
    if samples.iter().any(|job| job.len() != n) {
        return Err(shape_error("every sample set must hold the same number of points"));
    }
    if n == 0 {
        return Ok(samples.iter().map(|_| Ok(Poly::from(Vec::new()))).collect());
    }
    let (mut xs, mut ys) = (Vec::new(), Vec::new());
    for job in samples {
        for (x, y) in job {
            xs.extend_from_slice(&fr_bytes(x));
            ys.extend_from_slice(&fr_bytes(y));
        }
    }
    let (mut out, mut st) = (vec![0u8; samples.len() * n * FR_BYTES], vec![0u8; samples.len()]);
    let rc = unsafe { tc_fr_interpolate_batch(gpu.0, n, xs.as_ptr(), ys.as_ptr(), samples.len(), out.as_mut_ptr(), st.as_mut_ptr()) };
    ys.iter_mut().for_each(|b| *b = 0);
    gpu.check(rc)?;
    st.iter()
        .enumerate()
        .map(|(j, &s)| status_to_result(s, out[j * n * FR_BYTES..(j + 1) * n * FR_BYTES].chunks(FR_BYTES).map(fr_from).collect::<GpuResult<Vec<Fr>>>().map(Poly::from)))
        .collect()
}

// ---- (e): several GPUs of one node from one process ---------------------------------------------------------------------
/// One worker thread + one context per GPU inside the library; batches are sharded contiguously, the key set travels
/// by ONE RCCL broadcast over xGMI, valid counts by one all-reduce.
pub struct GpuGroup(*mut TcGroup);
unsafe impl Send for GpuGroup {}
impl GpuGroup {
    pub fn new(devices: &[i32]) -> std::result::Result<Self, GpuError> {
        let devs: Vec<c_int> = devices.iter().map(|&d| d as c_int).collect();
        let mut g = std::ptr::null_mut();
        let rc = unsafe { tc_group_create(&mut g, devs.as_ptr(), devs.len() as c_int) };
        if rc == TC_OK {
            Ok(GpuGroup(g))
        } else {
            Err(GpuError(rc, "tc_group_create failed".into()))
        }
    }
    fn check(&self, rc: c_int) -> GpuResult<()> {
        if rc == TC_OK {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(tc_group_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(GpuError(rc, msg))
    }
    pub fn size(&self) -> usize {
        unsafe { tc_group_size(self.0) as usize }
    }
    pub fn set_key_set(&self, pks: &PublicKeySet) -> GpuResult<()> {
        let commit = pks.commit_bytes();
        self.check(unsafe { tc_group_set_keyset(self.0, pks.threshold(), commit.as_ptr()) })
    }
    /// `combine_signatures` (src/lib.rs:608-615) for B jobs of n shares, sharded over the GPUs of the group.
    pub fn combine_signatures(&self, n: usize, idx: &[u64], shares: &[u8]) -> GpuResult<Vec<JobResult<Signature>>> {
        if n == 0 || idx.len() % n != 0 || shares.len() != idx.len() * G2_BYTES {
            return Err(shape_error("idx holds B x n indices, shares B x n x 192 bytes"));
        }
        let b = idx.len() / n;
        let (mut out, mut st) = (vec![0u8; b * G2_BYTES], vec![0u8; b]);
        self.check(unsafe { tc_group_combine_signatures(self.0, n, idx.as_ptr(), shares.as_ptr(), b, out.as_mut_ptr(), st.as_mut_ptr()) })?;
        st.iter().enumerate().map(|(j, &s)| status_to_result(s, g2_from(&out[j * G2_BYTES..(j + 1) * G2_BYTES]).map(Signature))).collect()
    }
    /// `public_key().verify_g2` for B jobs; returns the booleans and the all-reduced number of valid signatures.
    pub fn verify_g2(&self, sigs: &[u8], hashes: &[u8]) -> GpuResult<(Vec<bool>, u64)> {
        if sigs.len() % G2_BYTES != 0 || hashes.len() != sigs.len() {
            return Err(shape_error("sigs and hashes hold B x 192 bytes each"));
        }
        let b = sigs.len() / G2_BYTES;
        let mut ok = vec![0u8; b];
        let mut n_valid = 0u64;
        self.check(unsafe { tc_group_verify_g2(self.0, sigs.as_ptr(), hashes.as_ptr(), b, ok.as_mut_ptr(), &mut n_valid) })?;
        Ok((ok.into_iter().map(|x| x == 1).collect(), n_valid))
    }
    /// BASELINE config 5 in one call: hash, sign the selected shares on the device, combine, verify.
    pub fn sign_combine_verify<M: AsRef<[u8]>>(&self, sk_table: &[u8], n_nodes: usize, idx: &[u64], n: usize, msgs: &[M]) -> GpuResult<(Vec<Signature>, Vec<bool>, u64)> {
        let (flat, off) = pack_messages(msgs);
        let b = msgs.len();
        if sk_table.len() != n_nodes * FR_BYTES || idx.len() != b * n {
            return Err(shape_error("sk_table holds N x 32 bytes, idx B x n indices"));
        }
        let (mut sig, mut ok) = (vec![0u8; b * G2_BYTES], vec![0u8; b]);
        let mut n_valid = 0u64;
        self.check(unsafe {
            tc_group_sign_combine_verify(self.0, sk_table.as_ptr(), n_nodes, idx.as_ptr(), n, flat.as_ptr(), off.as_ptr(), b, sig.as_mut_ptr(), ok.as_mut_ptr(), &mut n_valid)
        })?;
        Ok((g2_vec(&sig)?.into_iter().map(Signature).collect(), ok.into_iter().map(|x| x == 1).collect(), n_valid))
    }
    /// bytes that crossed PCIe since the group was created: (host-to-device, device-to-host)
    pub fn transfer_bytes(&self) -> GpuResult<(u64, u64)> {
        let (mut up, mut down) = (0u64, 0u64);
        self.check(unsafe { tc_group_transfer_bytes(self.0, &mut up, &mut down) })?;
        Ok((up, down))
    }
}
impl Drop for GpuGroup {
    fn drop(&mut self) {
        unsafe { tc_group_destroy(self.0) }
    }
}
