/* tc_amd.h -- C ABI of libtc_amd.so: batched BLS12-381 threshold-crypto hot path on MI355X.
 *
 * Drop-in boundary for the data-parallel path of poanetwork/threshold_crypto 0.4.0.  Each
 * entry point is the BATCH form of one reference method (cited below, paths relative to
 * the reference repository); the reference's single-item Rust method maps to a batch of 1.
 * Plain pointers and sizes only; no C++/torch types.  Caller owns every buffer; the library
 * retains no pointer after a call returns (all calls are synchronous unless noted).
 *
 * Encodings (identical to the reference's, so results compare byte-for-byte):
 *   G1 point  : 96 B  uncompressed Zcash form  x || y           (big-endian, flag bits in byte 0)
 *   G2 point  : 192 B uncompressed Zcash form  x.c1||x.c0||y.c1||y.c0
 *               = `into_affine().into_uncompressed()`            (src/lib.rs:89,163,224,238,276)
 *   compressed: 48 B / 96 B = PublicKey::to_bytes / Signature::to_bytes (src/lib.rs:149-153,255-259)
 *   Fr scalar : 32 B little-endian canonical (4 x u64 LE limbs)  (src/serde_impl.rs:296)
 *   share idx : u64, the `T: IntoFr` index of combine_signatures/decrypt (src/into_fr.rs:16-26)
 * Decoding an uncompressed point checks range, flags and the curve equation, and BY DEFAULT every point operand
 * of every entry is also tested for order-r subgroup membership on the device before it is used (what the
 * reference does once per value in from_bytes, src/lib.rs:140-146, 246-252): a job that owns an invalid point
 * fails exactly like one with an undecodable encoding.  The scalar-multiplication and pairing kernels RELY on
 * membership (GLV/GLS ladders through phi = [-x^2] and psi = [x], the psi forms of the share combiner's
 * division, the folded cofactor constant, the 2^-63 bound of the random-linear-combination checks): an on-curve
 * point outside the subgroup would give a result the reference's plain double-and-add does not.  A caller whose
 * operands are KNOWN members -- outputs of this library's own entries, values that entered through
 * tc_g{1,2}_decompress_batch (the reference's checked decode) or were tested with
 * tc_g{1,2}_subgroup_check_batch; typically device-resident data -- opts out with
 * tc_ctx_set_input_checks(ctx, 0) and saves one 64-bit ladder per G2 operand and two per G1 operand.
 *
 * Return value: 0 = TC_OK, < 0 = call-level failure (bad argument / HIP error / no device / host failure);
 * never aborts, never throws across the boundary (every entry is a function-try-block: an exception becomes TC_ERR_HOST).
 * One abort is the ROCm runtime's own -- it kills the process when it cannot allocate a dispatch's private segments -- so a
 * call first compares the free device memory with what its size could ask for (at most ~6.5 GB for a large batch) and
 * returns TC_ERR_HIP instead of launching; TC_PRIVATE_RESERVE=<bytes> in the environment of tc_ctx_create overrides, 0 disables.  Per-job results go to `status[]`
 * (mirrors threshold_crypto::error::Error / FromBytesError, src/error.rs:7-17,37-41) and
 * `ok[]` (the `bool` of the verify methods).
 *
 * I/O residency: by default every data pointer is HOST memory and the call stages through
 * device buffers owned by the context.  After tc_ctx_set_device_io(ctx, 1) every data
 * pointer (inputs, outputs, offsets, status, ok) must be DEVICE memory on the context's GPU,
 * 8-byte aligned, and nothing crosses PCIe.  The kernels read and write them on the CONTEXT's stream and a call returns
 * once they are queued: whatever produced the operands must have finished (or run on the same stream: tc_ctx_set_stream),
 * and results are read after tc_sync or in stream order.  Output buffers should not overlap input buffers: results are written while
 * other jobs' operands are still being read.  The one supported exception is an output that overlaps a POINT operand of the
 * same call job for job (an in-place tc_g{1,2}_mul_batch with S = 1, ok[] over the head of a buffer the call reads): the
 * membership tests of that operand, which otherwise run beside the call's main kernels on the context's second stream, then
 * run in front of them on the main stream.
 *
 * Threading: a context is bound to one GPU and one HIP stream and is thread-compatible (one
 * thread at a time); distinct contexts may be used concurrently (one per GPU, or several on one GPU
 * from several host threads).  Besides its staging buffers a context holds 256 MB of device memory
 * for the per-job ladder tables of the G2 kernels, allocated by the first call that needs them.
 */
#ifndef TC_AMD_H
#define TC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_OK 0
#define TC_ERR_INVALID_ARG (-1)
#define TC_ERR_HIP (-2)
#define TC_ERR_NO_DEVICE (-3)
#define TC_ERR_HOST (-4) /* a host-side failure (std::bad_alloc, any C++ exception) stopped at the boundary */

/* per-job status codes */
#define TC_JOB_OK 0
#define TC_JOB_NOT_ENOUGH_SHARES 1 /* Error::NotEnoughShares   src/error.rs:9  */
#define TC_JOB_DUPLICATE_ENTRY 2   /* Error::DuplicateEntry    src/error.rs:12 */
#define TC_JOB_INVALID_ENCODING 3  /* FromBytesError::Invalid  src/error.rs:39 */

#define TC_G1_BYTES 96
#define TC_G2_BYTES 192
#define TC_G1_COMPRESSED_BYTES 48 /* PK_SIZE  src/lib.rs:71 */
#define TC_G2_COMPRESSED_BYTES 96 /* SIG_SIZE src/lib.rs:75 */
#define TC_FR_BYTES 32

typedef struct tc_ctx tc_ctx;

/* ---- context ---------------------------------------------------------------------------- */
/* Creates a context on HIP device `device`.  Fails with TC_ERR_NO_DEVICE when no gfx950
 * device is visible: there is no CPU fallback. */
int tc_ctx_create(tc_ctx** out, int device);
void tc_ctx_destroy(tc_ctx* ctx);
int tc_ctx_set_device_io(tc_ctx* ctx, int enabled);
int tc_ctx_get_device_io(const tc_ctx* ctx);
/* Use an externally owned HIP stream (hipStream_t passed as void*), e.g. torch's current
 * stream; NULL restores the context's own stream. */
int tc_ctx_set_stream(tc_ctx* ctx, void* hip_stream);
int tc_sync(tc_ctx* ctx);
const char* tc_last_error(const tc_ctx* ctx);
/* Milliseconds spent in the kernels of the most recent call, measured with HIP events on
 * the context's stream (0 if timing is disabled).  tc_ctx_set_timing(ctx, 1) enables it. */
int tc_ctx_set_timing(tc_ctx* ctx, int enabled);
double tc_last_kernel_ms(const tc_ctx* ctx);
/* Checked-input mode (default ON; see "Decoding" above): every uncompressed point operand of every entry is
 * tested for order-r subgroup membership on the device before it is used -- what the reference does once per
 * value in from_bytes (src/lib.rs:140-146, 246-252).  A job that owns an invalid point reports
 * TC_JOB_INVALID_ENCODING / ok = 0, as for an undecodable encoding.  Costs one 64-bit ladder per G2 point
 * (psi(P) = [x]P) and two per G1 point (phi(P) = [-x^2]P).  enabled = 0 is the explicit opt-out for operands the
 * caller knows to be members (this library's own outputs, checked decodes); results on non-members are then
 * unspecified (never memory-unsafe). */
int tc_ctx_set_input_checks(tc_ctx* ctx, int enabled);
int tc_ctx_get_input_checks(const tc_ctx* ctx);
/* A context keeps its staging and table buffers between calls (grow-only; a t = 67 combination of 131 072 jobs holds
 * ~18 GB of per-share tables, sized to a third of the HBM that was free at the call, at most 24 GiB).  tc_ctx_trim
 * waits for the context's stream and gives them back; they are allocated again on demand. */
int tc_ctx_trim(tc_ctx* ctx);
/* Bytes this context's own staging copies have moved over PCIe since it was created (host-I/O mode: every operand up,
 * every result down; device-I/O mode: the 8-byte read-back of off[B] per message-taking call and nothing else). */
int tc_ctx_transfer_bytes(const tc_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes);
/* The form choices this context makes (no counterpart in the reference: a measurement reports which kernels ran).
 * out8[0] / out8[1] = jobs from which a checked G2 decode / a hash takes two jobs per lane pair, out8[2] = the pairing form
 * (0 = by batch size: four lanes per check up to 16 384 checks, the prepared three-kernel form above; 1 quad, 2 lines,
 * 3 pair, 4 fused), out8[3] = bytes the prepared form's line buffer may take (0 = a third of the free HBM), out8[4] = 1 when
 * the membership tests of checked-input mode run on the context's second stream beside the main kernels of a call (default)
 * and 0 when they run before them; out8[5] = bytes the per-share tables of the two-stage kernels may take per call (0 = a third of the
 * free HBM, 1-24 GiB), out8[6] = free device memory a call insists on before its first launch (all ones = by the size of the call,
 * 0 = no check: see "Return value" above); out8[7] = 0 (reserved).  The defaults are the measured choices (csrc/tc_launch.h); the
 * environment variables TC_DUO_MIN, TC_PAIRING_FORM, TC_PAIRING_BUDGET, TC_CHECKS_BESIDE, TC_MSM_BUDGET, TC_PRIVATE_RESERVE (tests,
 * experiments) are read ONCE, by tc_ctx_create -- never on the launch path. */
int tc_ctx_get_tuning(const tc_ctx* ctx, uint64_t* out8);
const char* tc_version(void);

/* ---- hashing onto G2 -------------------------------------------------------------------- */
/* out[j] = hash_g2(msgs[off[j]..off[j+1]])                 pub fn hash_g2, src/lib.rs:691-694 */
int tc_hash_g2_batch(tc_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t B, uint8_t* out_g2);
/* out[j] = hash_g1_g2(g1[j], msgs[j])                      fn hash_g1_g2, src/lib.rs:697-707 */
int tc_hash_g1_g2_batch(tc_ctx* ctx, const uint8_t* g1, const uint8_t* msgs, const uint64_t* off, size_t B,
                        uint8_t* out_g2, uint8_t* status);

/* ---- scalar multiplication (share signing / decryption shares) --------------------------- */
/* out[j*S + s] = fr[s] * pts[j]   S signers x B hash points; status[j*S + s].
 * SecretKey::sign_g2 src/lib.rs:372-374, SecretKeyShare::sign_g2 :442-444 */
int tc_g2_mul_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                    uint8_t* status);
/* out[j*n + k] = sk_table[idx[j*n + k]] * hashes[j]: the signature shares of message j by the n signers of its
 * subset, out of a table of N secret key shares (SecretKeyShare::sign_g2 src/lib.rs:442-444 for each selected
 * signer) -- the on-device share generation of the (t, N, batch) workloads, so that nothing larger than the key
 * set crosses PCIe.  An index >= N fails its own output (TC_JOB_INVALID_ENCODING). */
int tc_sign_shares_g2_batch(tc_ctx* ctx, const uint8_t* sk_table, size_t N, const uint64_t* idx, const uint8_t* hashes, size_t n,
                            size_t B, uint8_t* out, uint8_t* status);
/* SecretKeyShare::decrypt_share_no_verify src/lib.rs:460-462, SecretKey::public_key :367-369 */
int tc_g1_mul_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                    uint8_t* status);
/* out[j*S + s] = fr[s] * hash_g2(msg[j])   SecretKey::sign :379-381, SecretKeyShare::sign :447-449 */
int tc_sign_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* msgs, const uint64_t* off, size_t S, size_t B,
                  uint8_t* out_g2, uint8_t* status);

/* ---- Lagrange combination ------------------------------------------------------------------ */
/* Job j holds n_per_job samples (idx[j*n + k], shares[(j*n + k)*192]) in iteration order (the
 * reference iterates a BTreeMap: ascending index).  As `interpolate` (src/lib.rs:719-767): only
 * the FIRST t+1 samples are used; n_per_job <= t gives status NOT_ENOUGH_SHARES for every job;
 * t == 0 returns the first sample; equal indices are filtered by value from the denominator.
 * PublicKeySet::combine_signatures src/lib.rs:608-615 (t = commit.degree(), :614). */
int tc_combine_g2_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                        size_t B, uint8_t* out, uint8_t* status);
/* Same in G1 (96 B points): the interpolate call of PublicKeySet::decrypt, src/lib.rs:618-625 */
int tc_combine_g1_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                        size_t B, uint8_t* out, uint8_t* status);
/* out[j] = sum_{k < n} scalars[j*n + k] * points[j*n + k]   (scalars 32 B LE, canonical).
 * Commitment::evaluate src/poly.rs:497-508 (= PublicKeySet::public_key_share src/lib.rs:570-573) is
 * this linear combination with scalars (i+1)^k over the commitment coefficients. */
int tc_g1_lincomb_batch(tc_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                        uint8_t* status);
int tc_g2_lincomb_batch(tc_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                        uint8_t* status);
/* PublicKeySet::decrypt src/lib.rs:618-626: out bytes[off[j]..off[j+1]] = xor_with_hash(combine_g1, v_j) */
int tc_decrypt_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares_g1,
                     const uint8_t* v, const uint64_t* off, size_t B, uint8_t* out, uint8_t* status);
/* `T: IntoFr` beyond u64.  combine_signatures / decrypt are generic over the index type (src/lib.rs:608-622): besides
 * u64 / usize it may be Fr itself or a negative i32 / i64 (src/into_fr.rs:10-14, 28-56: -(|x|) mod r).  idx_fr: B x n_per_job
 * abscissae as 32 B LE canonical Fr values (the IntoFr image; the interpolation point is idx_fr + 1, src/lib.rs:769-773), in
 * the iteration order of the reference's map.  A batch whose first t+1 abscissae per job all fit 64 bits runs the u64
 * kernels (small-index fast path included); otherwise every job takes the general path with coefficients from Fr
 * arithmetic.  A non-canonical abscissa (>= r) fails its job with TC_JOB_INVALID_ENCODING. */
int tc_combine_g2_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares, size_t B, uint8_t* out,
                           uint8_t* status);
int tc_combine_g1_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares, size_t B, uint8_t* out,
                           uint8_t* status);
int tc_decrypt_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares_g1, const uint8_t* v,
                        const uint64_t* off, size_t B, uint8_t* out, uint8_t* status);
/* Wire-level forms of the two combiners.  Shares arrive as they travel -- SignatureShare / Signature::to_bytes, 96 B
 * compressed G2 (src/lib.rs:255-259), DecryptionShare's 48 B compressed G1 (src/serde_impl.rs:174-218) -- and pass through the
 * CHECKED decode of from_bytes (src/lib.rs:140-146, 246-252: flags, range, on the curve, order-r subgroup) on the device:
 * only the first t+1 samples of a job, exactly the ones interpolate() uses; a job that owns an undecodable or non-member
 * share gets TC_JOB_INVALID_ENCODING and the identity.  tc_combine_signatures_wire_batch returns Signature::to_bytes of
 * the combined signature (96 B per job); tc_decrypt_wire_batch the plaintext bytes.  One call = decode + membership +
 * combine + encode on the device; the context's input-check switch does not apply (the decode IS the check). */
int tc_combine_signatures_wire_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares96, size_t B,
                                     uint8_t* out96, uint8_t* status);
int tc_decrypt_wire_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares48, const uint8_t* v,
                          const uint64_t* off, size_t B, uint8_t* out, uint8_t* status);
/* out[j] = data[j] ^ keystream(g1[j])                       fn xor_with_hash, src/lib.rs:710-715 */
int tc_xor_with_hash_batch(tc_ctx* ctx, const uint8_t* g1, const uint8_t* data, const uint64_t* off, size_t B,
                           uint8_t* out, uint8_t* status);

/* ---- pairing checks ------------------------------------------------------------------------ */
/* ok[j] = ( e(a[j], b[j]) == e(c[j], d[j]) ).  A stride of 0 broadcasts one operand to every job;
 * otherwise strides are in bytes (96 / 192 for packed arrays).  An undecodable operand gives 0.
 * PEngine::pairing(..) == PEngine::pairing(..): src/lib.rs:109, :185, :511 */
int tc_pairing_check_batch(tc_ctx* ctx, const uint8_t* a_g1, size_t a_stride, const uint8_t* b_g2, size_t b_stride,
                           const uint8_t* c_g1, size_t c_stride, const uint8_t* d_g2, size_t d_stride, size_t B,
                           uint8_t* ok);
/* ok[j] = pk.verify_g2(sig[j], hash[j]) = e(pk, hash[j]) == e(g1, sig[j])     src/lib.rs:108-110,170-172 */
int tc_verify_g2_batch(tc_ctx* ctx, const uint8_t* pk_g1, size_t pk_stride, const uint8_t* sig_g2,
                       const uint8_t* hash_g2, size_t B, uint8_t* ok);
/* ok[j] = pk.verify(sig[j], msg[j])                                              src/lib.rs:115-117,177-179 */
int tc_verify_sig_batch(tc_ctx* ctx, const uint8_t* pk_g1, size_t pk_stride, const uint8_t* sig_g2,
                        const uint8_t* msgs, const uint64_t* off, size_t B, uint8_t* ok);
/* OPT-IN fast path for validating the N signature shares of each of B messages against the N public key
 * shares (the loop of examples/threshold_sig.rs:115-131 over PublicKeyShare::verify, src/lib.rs:177-179): per
 * message ONE check  e(sum_i r_i pk_i, H(m)) == e(g1, sum_i r_i sig_i)  with 63-bit r_i (four 16-bit base-|x| digits) drawn from a ChaCha20
 * stream keyed by seed32 (HOST memory in both I/O modes: 32 bytes of fresh secret randomness per call), then
 * per-share checks only for the messages whose combined check failed.  ok[j*N + i] equals
 * PublicKeyShare::verify(pk_i, sig_{j,i}, m_j) except that a message holding an invalid share is accepted as a
 * whole with probability <= 2^-63.  sig_shares: B x N x 192, pk_shares: N x 96; n_fallback (optional, host):
 * number of messages that needed the per-share pass. */
int tc_verify_shares_rlc_batch(tc_ctx* ctx, const uint8_t* pk_shares, size_t N, const uint8_t* sig_shares, const uint8_t* msgs,
                               const uint64_t* off, size_t B, const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback);
/* Signature batches under ONE public key by random linear combination (opt-in; BASELINE config 3's shape).  The batch
 * is cut into groups of `group` jobs (0 = 64; at most 1024); a group passes with ONE check
 *     e(pk, sum_j r_j H_j) == e(g1, sum_j r_j sig_j)        instead of `group` checks of src/lib.rs:108-110,
 * r_j = 2^63 secret values from ChaCha20(seed32, j); groups that fail -- or hold an undecodable or non-member operand -- are
 * re-checked job by job, so ok[] equals tc_verify_g2_batch's (pk_stride 0) up to the 2^-63 of a wrongly passing group.
 * *n_fallback (optional) = jobs that went through the per-job checks.  seed32: 32 secret random bytes in HOST memory,
 * drawn after the signatures were received.  tc_verify_sig_rlc_batch hashes the messages on the device first
 * (PublicKey::verify, src/lib.rs:114-117). */
int tc_verify_g2_rlc_batch(tc_ctx* ctx, const uint8_t* pk, const uint8_t* sig, const uint8_t* hash, size_t B, size_t group,
                           const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback);
int tc_verify_sig_rlc_batch(tc_ctx* ctx, const uint8_t* pk, const uint8_t* sig, const uint8_t* msgs, const uint64_t* off, size_t B,
                            size_t group, const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback);
/* Decryption-share validation by one random linear combination per ciphertext (opt-in): the loop of
 * examples/threshold_enc.rs over PublicKeyShare::verify_decryption_share (src/lib.rs:182-186) for B ciphertexts x N nodes.
 * pk_shares: N x 96 B (public_key_share(i), the same for every ciphertext); shares: B x N x 96 B; u / v / off / w: the
 * ciphertexts.  ok[j * N + i] = pk_share(i).verify_decryption_share(&shares[j][i], &ct[j]) -- one combined check
 *     e(sum_i r_i share_i, hash_g1_g2(u, v)) == e(sum_i r_i pk_i, w)
 * per ciphertext (r_i: 2^63 secret values from ChaCha20(seed32, .)), share-by-share checks for the ciphertexts whose
 * combined check fails; *n_fallback (optional) = how many did.  seed32: 32 secret random bytes in HOST memory. */
int tc_verify_decryption_shares_rlc_batch(tc_ctx* ctx, const uint8_t* pk_shares, size_t N, const uint8_t* shares, const uint8_t* u,
                                          const uint8_t* v, const uint64_t* off, const uint8_t* w, size_t B, const uint8_t* seed32,
                                          uint8_t* ok, uint64_t* n_fallback);
/* ok[j] = Ciphertext(u[j], v[j], w[j]).verify() = e(g1, w) == e(u, hash_g1_g2(u, v))   src/lib.rs:508-512 */
int tc_ciphertext_verify_batch(tc_ctx* ctx, const uint8_t* u_g1, const uint8_t* v, const uint64_t* off,
                               const uint8_t* w_g2, size_t B, uint8_t* ok);
/* SecretKeyShare::decrypt_share src/lib.rs:452-457 for ONE key share and B ciphertexts: Ciphertext::verify, then [sk] u.
 * ok[j] = 1 and out_g1[j] = the DecryptionShare when ciphertext j is valid; ok[j] = 0 and the identity's encoding when it is
 * not (the reference returns None: an invalid ciphertext never yields a share).  sk_fr: 32 B LE, wiped from the staging
 * buffers after the call.  One call = hash_g1_g2 + pairing check + G1 multiplication, nothing crosses PCIe in between.
 * u and w are tested for membership in G1 / G2 ALWAYS, whatever tc_ctx_set_input_checks says: the pairing check cannot see a
 * small-order component of u, and the secret key must never multiply a point outside the order-r subgroup (the reference's
 * Ciphertext only exists after the checked decode, src/lib.rs:140-146).  A non-canonical sk_fr (>= r) gives ok[j] = 0. */
int tc_decrypt_share_batch(tc_ctx* ctx, const uint8_t* sk_fr, const uint8_t* u_g1, const uint8_t* v, const uint64_t* off,
                           const uint8_t* w_g2, size_t B, uint8_t* out_g1, uint8_t* ok);
/* SecretKey::decrypt src/lib.rs:384-391: Ciphertext::verify, g = [sk] u, xor_with_hash(g, v).  out bytes[off[j]..off[j+1]] =
 * the plaintext when ok[j] = 1, zeros when the ciphertext is invalid (None). */
int tc_secret_key_decrypt_batch(tc_ctx* ctx, const uint8_t* sk_fr, const uint8_t* u_g1, const uint8_t* v, const uint64_t* off,
                                const uint8_t* w_g2, size_t B, uint8_t* out, uint8_t* ok);
/* ok[j] = pk_share[j].verify_decryption_share(share[j], ct[j])
 *       = e(share, hash_g1_g2(u, v)) == e(pk_share, w)                                  src/lib.rs:182-186 */
int tc_verify_decryption_share_batch(tc_ctx* ctx, const uint8_t* pk_share_g1, size_t pk_stride,
                                     const uint8_t* share_g1, const uint8_t* u_g1, const uint8_t* v,
                                     const uint64_t* off, const uint8_t* w_g2, size_t B, uint8_t* ok);

/* ---- the steps either side of the path ------------------------------------------------------- */
/* (u[j], v[j], w[j]) = pk.encrypt_with_rng(rng, msg[j]) with the rng's `Fr::random` draw r[j] supplied
 * by the caller (32 B LE): u = r g1, v = msg ^ keystream(r pk), w = r hash_g1_g2(u, v).
 * PublicKey::encrypt_with_rng src/lib.rs:128-137.  v has the layout of msgs (same offsets). */
int tc_encrypt_batch(tc_ctx* ctx, const uint8_t* pk_g1, size_t pk_stride, const uint8_t* r, const uint8_t* msgs,
                     const uint64_t* off, size_t B, uint8_t* out_u, uint8_t* out_v, uint8_t* out_w, uint8_t* status);
/* out[j] = pk_set.public_key_share(idx[j]) = Commitment::evaluate(idx[j] + 1): Horner in G1 over the
 * (t+1) x 96 B commitment.  PublicKeySet::public_key_share src/lib.rs:570-573, src/poly.rs:497-508.
 * With tc_verify_sig_batch (pk_stride = 96) this is the share-validation loop of
 * examples/threshold_sig.rs:115-131 in two launches. */
int tc_public_key_share_batch(tc_ctx* ctx, const uint8_t* commit, size_t t, const uint64_t* idx, size_t M, uint8_t* out,
                              uint8_t* status);

/* ---- DKG algebra (src/poly.rs) ----------------------------------------------------------------- */
/* out[i] = coeff_fr[i] * g1: Poly::commitment src/poly.rs:372-377 and BivarPoly::commitment :625-632 (every
 * coefficient times the G1 generator).  Fixed base: a signed 4-bit window table of g1, built once per context
 * and staged in LDS by every workgroup -- 64 mixed additions and no doubling per coefficient. */
int tc_g1_commitment_batch(tc_ctx* ctx, const uint8_t* coeff_fr, size_t M, uint8_t* out, uint8_t* status);
/* BivarCommitment::row src/poly.rs:713-727 for M abscissae: out[m*(degree+1) + i] = sum_j commit[pos(i,j)] * xs[m]^j
 * with commit the (degree+1)(degree+2)/2 coefficients of a symmetric bivariate commitment in coeff_pos order
 * (src/poly.rs:746-750) and xs[m] the u64 `IntoFr` value itself (rows are requested as row(m), m = 0..N).
 * BivarCommitment::evaluate(x, y) :694-710 is row(x) followed by tc_public_key_share_batch-style Horner in y;
 * status[m*(degree+1) + i]. */
int tc_bivar_commitment_row_batch(tc_ctx* ctx, const uint8_t* commit, size_t degree, const uint64_t* xs, size_t M, uint8_t* out,
                                  uint8_t* status);
/* Poly::interpolate src/poly.rs:341-350, 388-417 in Fr: for each of B jobs the n coefficients (low degree first,
 * 32 B LE each; the reference's Poly drops trailing zeros) of the unique polynomial through the n samples
 * (xs[j][k], ys[j][k]) (32 B LE canonical each, abscissae taken as given).  A repeated abscissa -- where the
 * reference panics "sample points must be distinct" -- gives TC_JOB_DUPLICATE_ENTRY and zero coefficients. */
int tc_fr_interpolate_batch(tc_ctx* ctx, size_t n, const uint8_t* xs, const uint8_t* ys, size_t B, uint8_t* out_coeff,
                            uint8_t* status);

/* ---- membership tests ------------------------------------------------------------------------ */
/* ok[j] = 1 iff pts[j] decodes (range, flags, curve equation) and lies in the order-r subgroup: the
 * CHECKED half of `into_affine` (from_bytes, src/lib.rs:140-146, 246-252) for values that arrive
 * uncompressed.  The identity is a member. */
int tc_g1_subgroup_check_batch(tc_ctx* ctx, const uint8_t* pts96, size_t B, uint8_t* ok);
int tc_g2_subgroup_check_batch(tc_ctx* ctx, const uint8_t* pts192, size_t B, uint8_t* ok);

/* ---- wire formats --------------------------------------------------------------------------- */
/* uncompressed -> compressed: PublicKey::to_bytes src/lib.rs:149-153, Signature::to_bytes :255-259 */
int tc_g1_compress_batch(tc_ctx* ctx, const uint8_t* in96, size_t B, uint8_t* out48, uint8_t* status);
int tc_g2_compress_batch(tc_ctx* ctx, const uint8_t* in192, size_t B, uint8_t* out96, uint8_t* status);

/* compressed -> uncompressed with the CHECKED decode of PublicKey::from_bytes src/lib.rs:140-146 and
 * Signature::from_bytes :246-252 (serde: src/serde_impl.rs:187-218): flags, range, on-curve and
 * order-r subgroup membership; status = TC_JOB_INVALID_ENCODING (FromBytesError::Invalid) otherwise. */
int tc_g1_decompress_batch(tc_ctx* ctx, const uint8_t* in48, size_t B, uint8_t* out96, uint8_t* status);
int tc_g2_decompress_batch(tc_ctx* ctx, const uint8_t* in96, size_t B, uint8_t* out192, uint8_t* status);

/* ---- several GPUs of one node from ONE host process ------------------------------ */
/* A group owns one tc_ctx and one worker thread per GPU.  Jobs are independent, so a batch in host memory is
 * split into contiguous ranges, one per GPU, and there is no data-path collective; the two exchange steps of the
 * path run on RCCL (bound with dlopen at tc_group_create): the BROADCAST of the key-set parameters from rank 0 to
 * every GPU's HBM over xGMI (tc_group_set_keyset) and the ALL-REDUCE of the per-GPU valid counts
 * (tc_group_verify_g2).  Processes that prefer one process per GPU use a plain tc_ctx each and do the broadcast
 * themselves (torch.distributed / RCCL: threshold_crypto_amd/parallel.py, bench.py --gpus N).
 * devices: ndev HIP device ids; duplicate ids put several ranks on one GPU (no RCCL then: a test configuration). */
typedef struct tc_group tc_group;
int tc_group_create(tc_group** out, const int* devices, int ndev);
void tc_group_destroy(tc_group* g);
int tc_group_size(const tc_group* g);
int tc_group_uses_rccl(const tc_group* g);
tc_ctx* tc_group_ctx(tc_group* g, int rank);             /* the rank's context, for direct tc_*_batch calls */
const char* tc_group_last_error(const tc_group* g);
/* contiguous range [start, start + count) of a B-job batch that rank `rank` works on */
int tc_group_shard(const tc_group* g, size_t B, int rank, size_t* start, size_t* count);
/* bytes that crossed PCIe since the group was created: the group's own copies plus its contexts' staging copies */
int tc_group_transfer_bytes(const tc_group* g, uint64_t* h2d_bytes, uint64_t* d2h_bytes);
/* PublicKeySet { commit } src/lib.rs:539-543: (t+1) x 96 B from host memory to rank 0, then RCCL broadcast */
int tc_group_set_keyset(tc_group* g, size_t t, const uint8_t* commit);
int tc_group_get_keyset(tc_group* g, int rank, uint8_t* out_commit);
/* PublicKeySet::combine_signatures src/lib.rs:608-615 for B jobs (host buffers), threshold = the key set's */
int tc_group_combine_signatures(tc_group* g, size_t n_per_job, const uint64_t* idx, const uint8_t* shares, size_t B, uint8_t* out,
                                uint8_t* status);
/* PublicKeySet::public_key().verify_g2 src/lib.rs:565-567, 108-110 for B jobs; n_valid (optional): all-reduced count */
int tc_group_verify_g2(tc_group* g, const uint8_t* sig, const uint8_t* hash, size_t B, uint8_t* ok, uint64_t* n_valid);
/* BASELINE config 5 in one call: hash each message, sign the n shares of its signer subset ON the device from the
 * N x 32 B table of secret key shares, combine, verify under the master key.  idx: B x n ascending signer indices.
 * Per rank only its slice of msgs / off / idx and the share table go up and its slice of sig / ok comes back; the hash
 * points and the B x n x 192 B of share signatures are made and consumed in the rank's HBM (persistent per-rank device
 * buffers, one persistent worker thread per GPU).  off must start at 0 and be non-decreasing (checked before sharding). */
int tc_group_sign_combine_verify(tc_group* g, const uint8_t* sk_table, size_t N, const uint64_t* idx, size_t n, const uint8_t* msgs,
                                 const uint64_t* off, size_t B, uint8_t* sig, uint8_t* ok, uint64_t* n_valid);

#ifdef __cplusplus
}
#endif
#endif /* TC_AMD_H */
