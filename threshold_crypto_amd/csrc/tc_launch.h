// Host-side launch prototypes of the gfx950 kernels (one translation unit per kernel family
// so hipcc can build them in parallel).  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "tc_arena.h"

namespace tc {

// waves per SIMD the register allocator must leave room for (__launch_bounds__ 2nd argument).
// Kernels on Fq2 values run two lanes per job (tc_common.h): batch 65 536 gives them two waves
// per SIMD, so they are compiled for 256 registers; the G1 kernels keep the full 512.
#ifndef TC_WAVES_G2
#define TC_WAVES_G2 2
#endif
#ifndef TC_WAVES_G1
#define TC_WAVES_G1 1
#endif
// the G1 kernels of the paths beside the headline (Commitment::evaluate, BivarCommitment::row, the general-path combination and
// small linear combinations): built for 256 registers.  One lane per job gives a 65 536-job batch one wave per SIMD whatever the
// build, and from 131 072 jobs on the second wave is worth 4-15 % although 100-250 registers spill (r04: profiles/r04_g1_aux_probe.txt)
#ifndef TC_WAVES_G1_AUX
#define TC_WAVES_G1_AUX 2
#endif
constexpr int kBlock = 64;  // one wavefront per workgroup: the jobs are register/scratch heavy

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// Two jobs per lane pair (tc_duo.h: checked G2 decodes, hash_g2, hash_g1_g2) halves the lanes of a launch and lengthens each: it
// pays once the launch is large enough that the halved form still fills the SIMDs.  Measured (profiles/r05_duo_sweep.txt, one
// MI355X, ms one job per pair / two):
//   decode   32 768: 2.03 / 2.47    65 536: 3.29 / 2.68   131 072: 6.72 / 4.57   262 144: 12.90 / 9.43
//   hash_g2  32 768: 5.13 / 8.58    65 536: 8.27 / 9.02   131 072: 15.56 / 14.42 262 144: 30.43 / 27.47
// A third of a decode is Fq-only work (two exponentiations), so it wins as soon as the one-job form needs a second wave on any
// SIMD (more than 32 768 points); a hash has 24 % of it and a search loop that runs until the slowest of 64 instead of 32
// messages has its candidate, so it only wins with two waves per SIMD in the halved form (131 072 messages).
constexpr size_t kDuoMinDecode = 32768 + 1;
constexpr size_t kDuoMinHash = 131072;
inline bool duo_form(size_t jobs, size_t min_jobs) { return jobs >= min_jobs; }

// The form choices of a context.  The defaults are the measured thresholds above / in k_pairing.hip; the environment
// (TC_DUO_MIN=<jobs>, TC_PAIRING_FORM=quad|lines|pair|fused, TC_PAIRING_BUDGET=<bytes>, TC_CHECKS_BESIDE=0|1: tests and experiments) is read ONCE,
// when the context is created (tc_api.hip tuning_from_env) -- no getenv on the launch path, nothing a concurrent setenv can race
// with -- and tc_ctx_get_tuning reports what a context uses.
constexpr size_t kMaxPrivateBytesPerLane = 12288;  // >= the largest private segment of any kernel (11 136 B: k_msm_tables_g1; tests/test_abi.py)
struct Tuning {
  size_t duo_min_decode = kDuoMinDecode, duo_min_hash = kDuoMinHash;
  int pairing_form = 0;       // 0 = by batch size; 1 quad, 2 lines (prepared), 3 pair (one loop), 4 fused
  size_t pairing_budget = 0;  // bytes the prepared form's line buffer may take; 0 = a third of the HBM that is free
  // free HBM a call insists on before it launches (after its own allocations): the ROCm runtime allocates the kernels' private
  // segments (scratch) at dispatch and ABORTS THE PROCESS when it cannot (amd::roc::callbackQueue <- AqlQueue::DynamicQueueEventsHandler,
  // seen with ~0.4 GB free).  (size_t)-1 = by the size of the call (Call::guard_private); TC_PRIVATE_RESERVE=<bytes> fixes it, 0 disables.
  size_t private_reserve = (size_t)-1;
  size_t msm_budget = 0;      // bytes the per-share tables of the two-stage kernels may take per call; 0 = a third of the free HBM, 1-24 GiB (TC_MSM_BUDGET: tests)
  int checks_beside = 1;  // checked-input mode: the membership tests on a second stream beside the call's main kernels (TC_CHECKS_BESIDE=0: before them, one stream)
};

// The G1 ladder kernels keep their per-lane table in the HBM arena and fit 256 registers (two waves per SIMD) at
// EVERY batch size since the end of r04: the register-table builds they replaced (377 registers + 121 AGPRs, 7 KB of scratch per
// lane) were no faster below 65 536 jobs (profiles/r04_g1_arena_all_sizes.txt: single multiplications 7-16 % SLOWER, the
// combination 1 % faster) and are only compiled with -DTC_G1_ARENA_MIN=<jobs> (experiments).
#ifndef TC_G1_ARENA_MIN
#define TC_G1_ARENA_MIN 0
#endif
constexpr size_t kG1ArenaMinJobs = TC_G1_ARENA_MIN;
void launch_g1_mul(hipStream_t st, TableArena ta, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                   uint8_t* status);
void launch_g2_mul(hipStream_t st, TableArena ta, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                   uint8_t* status);
// out[j*n + k] = sk[idx[j*n + k]] * pts[j]: the shares of message j by its n selected signers out of N
void launch_g2_mul_gather(hipStream_t st, TableArena ta, const uint8_t* sk, size_t N, const uint64_t* idx, const uint8_t* pts, size_t n, size_t B,
                          uint8_t* out, uint8_t* status);
// many signers per message (n >= kCombMinSigners): a per-message comb in HBM (tbl: comb_table_bytes(B), ok: B bytes), then
// doubling-free multiplications (k_comb.hip)
constexpr size_t kCombMinSigners = 24;
// ... and enough messages: a comb is built by ONE lane pair per message in 64 dependent column steps (~15 ms whatever the
// batch), which only pays from a few thousand messages on; below, the per-chunk ladders of k_g2_mul_gather
constexpr size_t kCombMinBatch = 8192;
// signers a lane pair of the share-generation kernels takes: `most` (one table / one inversion shared by eight) when the
// batch fills the GPU that way; for a smaller batch the smallest number that keeps the launch within ONE round of the
// 2048 wave slots (65 536 lane pairs) -- more, shorter lane pairs, no second round with a handful of waves
inline size_t signers_per_lane_pair(size_t n, size_t B, size_t most) {
  for (size_t share = 1; share < most; share++)
    if (((n + share - 1) / share) * B <= 65536) return share;
  return most;
}
size_t comb_table_bytes(size_t B);
void launch_comb_sign(hipStream_t st, TableArena ta, const uint8_t* sk, size_t N, const uint64_t* idx, const uint8_t* pts, size_t n, size_t B,
                      int32_t* tbl, uint8_t* ok, uint8_t* out, uint8_t* status);
void launch_g1_compress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status);
void launch_g2_compress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status);
void launch_g1_decompress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status);
void launch_g2_decompress(const Tuning& tn, hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status);
// wire ingest of the combiners: the first `take` of n_per_job compressed samples per job, checked decode, compact output
// (B x take points) + one validity byte per sample; and the matching prefix of the index array
void launch_decompress_take(const Tuning& tn, hipStream_t st, bool g2, const uint8_t* in, size_t n_per_job, size_t take, size_t B, uint8_t* out, uint8_t* valid);
void launch_take_u64(hipStream_t st, const uint64_t* in, size_t n_per_job, size_t take, size_t B, uint64_t* out);

// need_general: one zeroed device word; the Lagrange stage counts the jobs the small-index fast path
// does not take, the general combine kernel leaves at once when it stays zero (k_combine.hip)
void launch_lagrange(hipStream_t st, const uint64_t* idx, size_t n_per_job, size_t t, size_t B, uint32_t* lam,
                     uint8_t* status, uint32_t* need_general);
// every coefficient of a job from ONE lane with one inversion (large thresholds); ws: lagrange_all_ws_words
size_t lagrange_all_ws_words(size_t t, size_t B);
void launch_lagrange_all(hipStream_t st, const uint64_t* idx, size_t n_per_job, size_t t, size_t B, uint32_t* lam, uint32_t* ws,
                         uint8_t* status);
// ta: the table arena (the fast path's per-lane ladder tables); may be empty when idx == nullptr (no fast path)
// cls / counters / perm (all three or none): scratch of the caller for grouping the fast path's jobs by denominator class, as in G2
void launch_combine_g1(hipStream_t st, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status, const uint32_t* need_general,
                       TableArena ta = TableArena{nullptr, nullptr}, uint8_t* cls = nullptr, uint32_t* counters = nullptr,
                       uint32_t* perm = nullptr, hipEvent_t before_main = nullptr);
// From this many jobs on the G1 fast path groups its jobs by denominator class (a wave of D = 1 jobs -- a fifth of the 4-of-10
// subsets -- skips the [1 / D] ladder).  It only pays once a SIMD sees several rounds of waves: measured 3.5 % slower at 131 072
// jobs (every wave resident at once: the launch lasts as long as two generic waves on one SIMD, grouped or not), equal at
// 262 144, 8 % faster at 524 288 (profiles/r04_g1_group_ab.txt).
constexpr size_t kG1GroupMinJobs = 524288;
// `T: IntoFr` abscissae beyond u64 (tc_combine_g{1,2}_fr_batch): idx_fr = B x n_per_job x 8 canonical LE words.
//   launch_fr_idx_narrow   idx64[i] = the value when it fits 64 bits; *wide counts the ones that do not; valid[j] (preset
//                          to 1) is cleared for a job that owns a non-canonical encoding (>= r) among its first `take`
//   launch_lagrange_fr     lambda_i of every job from the Fr abscissae x = idx_fr + 1 (one lane per coefficient)
void launch_fr_idx_narrow(hipStream_t st, const uint32_t* idx_fr, size_t n_per_job, size_t take, size_t B, uint64_t* idx64, uint32_t* wide,
                          uint8_t* valid);
void launch_lagrange_fr(hipStream_t st, const uint32_t* idx_fr, size_t n_per_job, size_t t, size_t B, uint32_t* lam, uint8_t* status);
size_t combine_group_slots(size_t B);
void launch_combine_g2(hipStream_t st, TableArena ta, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status, uint8_t* cls, uint32_t* counters,
                       uint32_t* perm, const uint32_t* need_general, hipEvent_t before_main = nullptr);
// shared_points: every job combines the SAME n points (points holds n of them) with its own n scalars
void launch_lincomb_g1(hipStream_t st, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                       uint8_t* status, bool shared_points = false);
// random-linear-combination share validation (k_check.hip): 64-bit scalars from a ChaCha20 stream keyed
// by the caller's seed; row gather / byte scatter for the per-share fallback of failed messages
void launch_rlc_scalars(hipStream_t st, const uint8_t* seed32, size_t n, uint8_t* out_fr);
void launch_gather_rows(hipStream_t st, const uint8_t* src, size_t row_bytes, const uint32_t* map, size_t rows, uint8_t* dst);
void launch_scatter_bytes(hipStream_t st, const uint8_t* src, const uint32_t* map, size_t rows, uint8_t* dst);

// large G2 linear combinations in two stages (k_msm.hip): per-share affine psi tables + digit codes in HBM
// (tbl: msm_table_bytes, codes: msm_code_bytes), then one 64-step ladder per job.  status: B bytes, jobs with
// status != TC_JOB_OK are skipped (identity out); undecodable operands set TC_JOB_INVALID_ENCODING.
constexpr size_t kMsmMinPoints = 8;  // from here on the Lagrange coefficients come from the one-inversion kernels
// jobs of a share combination that belong to the two-stage path: all of them (null members), or the ones the
// small-index fast path leaves alone -- need: the count k_lagrange made of them (zero: the kernels leave at once)
struct MsmFilter {
  const uint32_t* need = nullptr;
  const uint64_t* idx = nullptr;
  size_t n_per_job = 0, t = 0;
};
size_t msm_table_bytes(size_t n, size_t B);
size_t msm_code_bytes(size_t n, size_t B);
void launch_msm_g2(hipStream_t st, size_t n, size_t pts_stride, const uint8_t* points, const uint32_t* scalars, size_t B,
                   int32_t* tbl, uint8_t* codes, uint8_t* out, uint8_t* status, int nbits = 64, MsmFilter f = MsmFilter());

// the same two stages in G1 (threshold decryption and G1 linear combinations from kMsmMinPoints points on): codes as for G2
// (msm_code_bytes), tables of msm_table_bytes_g1
size_t msm_table_bytes_g1(size_t n, size_t B);
size_t msm_table_jobs_g1(size_t pts_stride, int nbits, size_t B);  // 1 when every job shares ONE point set and the scalars are short
void launch_msm_g1(hipStream_t st, size_t n, size_t pts_stride, const uint8_t* points, const uint32_t* scalars, size_t B, int32_t* tbl,
                   uint8_t* codes, uint8_t* out, uint8_t* status, int nbits = 128);
// random scalars for the batch validation of G1 values: a + b x^2 with a (odd), b from 32-bit draws of ChaCha20(seed, i)
void launch_rlc_scalars_g1(hipStream_t st, const uint8_t* seed32, size_t n, uint8_t* out_fr);

// opt-in operand validation (k_check.hip): valid[i] for point i = (record i / take, sample i % take) of
// records of n_per_job points `stride` bytes apart; launch_invalidate_jobs fails the jobs that own an
// invalid point (status INVALID_ENCODING + identity output, or ok = 0); job j owns record j / group
void launch_subgroup_check_g1(hipStream_t st, const uint8_t* pts, size_t stride, size_t n_per_job, size_t take, size_t n,
                              uint8_t* valid);
void launch_subgroup_check_g2(hipStream_t st, const uint8_t* pts, size_t stride, size_t n_per_job, size_t take, size_t n,
                              uint8_t* valid);
void launch_ok_and_status(hipStream_t st, const uint8_t* status, size_t B, uint8_t* ok);  // ok[j] &= status[j] == OK
void launch_invalidate_jobs(hipStream_t st, const uint8_t* valid, size_t per_job, size_t group, size_t B, uint8_t* status,
                            uint8_t* out, size_t out_bytes, uint8_t* ok);

// ws: the Miller values between the kernels of a check and, for the prepared form, the line buffer of ONE tile of checks
// (k_pairing.hip).  `tile` = checks per pass of the prepared form, sized by the caller from the memory that is free
// (pairing_tile): 61.8 KB of line products per check, 4.05 GB for the full 65 536-check tile.
struct PairingWs {
  int32_t* p;
  size_t tile;
  int form;  // pairing_form's choice for this batch (a PairingForm of k_pairing.hip)
};
int pairing_form(size_t B, const Tuning& tn);          // which kernels run a batch of B checks
bool pairing_form_needs_lines(int form);               // the prepared form: the only one with a line buffer
size_t pairing_tile(size_t B, size_t budget_bytes);    // checks per pass of the prepared form; 0 = the line buffer does not fit
size_t pairing_ws_words(size_t B, size_t tile);
void launch_pairing_check(hipStream_t st, const uint8_t* a, size_t sa, const uint8_t* b, size_t sb, const uint8_t* c,
                          size_t sc, const uint8_t* d, size_t sd, size_t B, uint8_t* ok, PairingWs ws);

// fix = false: the hash point WITHOUT its last constant multiplication (tc_gls.h g2_clear_cofactor); the
// caller folds the constant into a scalar (launch_fr_scale_cofactor_fix) or a G1 operand
// (launch_g1_scale_cofactor_fix)
void launch_hash_g2(const Tuning& tn, hipStream_t st, const uint8_t* msgs, const uint64_t* off, size_t B, uint8_t* out, bool fix = true);
void launch_hash_g1_g2(const Tuning& tn, hipStream_t st, const uint8_t* g1, const uint8_t* msgs, const uint64_t* off, size_t B,
                       uint8_t* out, uint8_t* status, bool fix = true);
void launch_fr_scale_cofactor_fix(hipStream_t st, const uint8_t* fr, size_t S, uint8_t* out);
void launch_g1_scale_cofactor_fix(hipStream_t st, const uint8_t* in, size_t stride, size_t n, uint8_t* out);
void launch_xor_with_hash(hipStream_t st, const uint8_t* g1, const uint8_t* data, const uint64_t* off, size_t B,
                          uint8_t* out, uint8_t* status);
void launch_encrypt(hipStream_t st, TableArena ta, const uint8_t* pk, size_t pk_stride, const uint8_t* r, const uint8_t* msgs,
                    const uint64_t* off, size_t B, uint8_t* out_u, uint8_t* out_v, uint8_t* out_w, uint8_t* status);
void launch_commitment_evaluate(hipStream_t st, const uint8_t* commit, size_t t, const uint64_t* idx, size_t M, uint8_t* out,
                                uint8_t* status);
// DKG algebra (k_dkg.hip): the fixed-base window table of the G1 generator is built once per context
size_t fixed_base_table_bytes();
void launch_fixed_base_table(hipStream_t st, int32_t* tbl);
void launch_g1_fixed_base(hipStream_t st, const int32_t* tbl, const uint8_t* fr, size_t M, uint8_t* out, uint8_t* status, int cus);
void launch_bivar_commitment_row(hipStream_t st, const uint8_t* commit, size_t degree, const uint64_t* xs, size_t M, uint8_t* out,
                                 uint8_t* status);
void launch_fr_interpolate(hipStream_t st, size_t n, const uint32_t* xs, const uint32_t* ys, size_t B, uint32_t* out, uint32_t* ws,
                           uint8_t* status);
void launch_fill_g1_generator(hipStream_t st, uint8_t* out96, uint8_t* out96_unfix);

}  // namespace tc
