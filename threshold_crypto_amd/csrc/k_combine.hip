// gfx950 kernels: Lagrange coefficients and the share combiner (interpolate).
#include "tc_jobs.h"
#include "tc_launch.h"
#include "tc_stage.h"

namespace tc {

// one lane per (job, sample position): lambda_i of job j.  Jobs the small-index fast path owns are
// skipped; the others are counted in *need_general (one atomic per wave) so that k_combine_general can
// leave at once when there is nothing for it to do.
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_lagrange(const uint64_t* __restrict__ idx, size_t n_per_job, size_t t,
                                                     size_t B, uint32_t* __restrict__ lam,
                                                     uint8_t* __restrict__ status, uint32_t* __restrict__ need_general) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t k = t + 1;
  const bool live = tid < B * k;
  const size_t j = live ? tid / k : 0, i = live ? tid % k : 0;
  const bool general = live && !combine_small_applies(idx + j * n_per_job, (int)t);
  const uint64_t m = __builtin_amdgcn_ballot_w64(general && i == 0);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(need_general, (uint32_t)__builtin_popcountll(m));
  if (!general) return;  // the fast path owns the job (or the lane is past the end)
  uint8_t st = job_lagrange(idx + j * n_per_job, (int)t, (int)i, lam + tid * 8);
  if (st && status) status[j] = st;
}

// `T: IntoFr` abscissae as 32-byte Fr values (tc_combine_g{1,2}_fr_batch).  k_fr_idx_narrow: the value as a u64 where it fits
// (so that a batch of ordinary indices that merely ARRIVED as Fr takes the u64 kernels, fast path included), a count of
// the ones that do not, and valid[j] = 0 for a job that owns a non-canonical one (>= r) among the first t+1.
__global__ void k_fr_idx_narrow(const uint32_t* __restrict__ idx_fr, size_t n_per_job, size_t take, size_t B, uint64_t* __restrict__ idx64,
                                uint32_t* __restrict__ wide, uint8_t* __restrict__ valid) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < B * n_per_job;
  bool is_wide = false;
  if (live) {
    const uint32_t* w = idx_fr + i * 8;
    const bool used = (i % n_per_job) < take;                      // only the first t+1 samples matter (src/lib.rs:727-730)
    is_wide = used && (w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) != 0;
    idx64[i] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    if (used && !limbs_lt_p<FrParams>(w)) valid[i / n_per_job] = 0;
  }
  const uint64_t m = __builtin_amdgcn_ballot_w64(is_wide);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(wide, (uint32_t)__builtin_popcountll(m));
}
// one lane per (job, sample position), every job (no fast path: the abscissae are arbitrary field elements)
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_lagrange_fr(const uint32_t* __restrict__ idx_fr, size_t n_per_job, size_t t, size_t B,
                                                                  uint32_t* __restrict__ lam, uint8_t* __restrict__ status) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t k = t + 1;
  if (tid >= B * k) return;
  const size_t j = tid / k, i = tid % k;
  Fr l;
  if (status[j] != TC_JOB_OK || !lagrange_coeff_at_zero_fr(idx_fr + j * n_per_job * 8, (int)t, (int)i, l)) {
    for (int w = 0; w < 8; w++) lam[tid * 8 + w] = 0;
    if (status[j] == TC_JOB_OK) status[j] = TC_JOB_DUPLICATE_ENTRY;
    return;
  }
  l.to_canonical(lam + tid * 8);
}

// Large thresholds (no job takes the small-index fast path): all t+1 coefficients of a job with ONE inversion
// (tc_threshold.h), in two kernels.
//   k_lagrange_den     one lane per (job, i): the O(t^2) part.  A 256-lane workgroup takes floor(256 / (t+1)) jobs; the
//                      denominators (integer chunk products, tc_threshold.h) and the abscissae x_i = idx_i + 1 in
//                      Montgomery form go to HBM for the second kernel.
//   k_lagrange_finish  one lane per job: prefix products, the inversion, lambda_i.
constexpr int kLagBlock = 256;
constexpr int kLagMaxN = 256;  // t + 1 above this falls back to one lane per job (k_lagrange_all)
__global__ __launch_bounds__(kLagBlock) void k_lagrange_den(const uint64_t* __restrict__ idx, size_t n_per_job, size_t t, size_t B,
                                                          uint32_t* __restrict__ xm, uint32_t* __restrict__ den) {
  const int n = (int)t + 1;
  const int jobs_per_block = kLagBlock / n;
  const int jl = (int)threadIdx.x / n, i = (int)threadIdx.x % n;
  const size_t j = (size_t)blockIdx.x * jobs_per_block + jl;
  if (jl >= jobs_per_block || j >= B) return;
  const Fr x = fr_from_u64(idx[j * n_per_job + i]) + Fr::one();
  const Fr d = lagrange_denominator(idx + j * n_per_job, n, i);
  TC_UNROLL for (int w = 0; w < 8; w++) {
    xm[(j * n + i) * 8 + w] = x.v.l[w];
    den[(j * n + i) * 8 + w] = d.v.l[w];
  }
}
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_lagrange_finish(size_t t, size_t B, const uint32_t* __restrict__ xm,
                                                                       const uint32_t* __restrict__ den, uint32_t* __restrict__ pre,
                                                                       uint32_t* __restrict__ lam, uint8_t* __restrict__ status) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= B) return;
  const size_t n = t + 1;
  const uint8_t st = lagrange_finish((int)n, xm + j * n * 8, den + j * n * 8, pre + j * n * 8, lam + j * n * 8);
  if (st && status) status[j] = st;
}
// one lane per JOB for everything (t + 1 > 256)
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_lagrange_all(const uint64_t* __restrict__ idx, size_t n_per_job, size_t t, size_t B,
                                                                    uint32_t* __restrict__ lam, uint32_t* __restrict__ ws,
                                                                    uint8_t* __restrict__ status) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= B) return;
  const uint8_t st = lagrange_all_at_zero(idx + j * n_per_job, (int)t, lam + j * (t + 1) * 8, ws + j * 4 * (t + 1) * 8);
  if (st && status) status[j] = st;
}

// ---- grouping jobs by denominator class (tc_jobs.h combine_divide) -----------------------------------
// Two small kernels turn the batch into a permutation in which every class occupies a run of whole
// waves (runs padded to kCombinePad jobs with the marker 0xffffffff), so that the lanes of a wave take
// the same branch of combine_divide.  Correctness does not depend on it: a mixed wave runs every
// branch its lanes need.
//   counters[c]      jobs of class c                               (k_combine_classify)
//   counters[4 + c]  slots of class c handed out so far            (k_combine_scatter)
// 256-lane workgroups: the four waves add their ballot counts up in LDS and the workgroup issues ONE atomic per
// class (768 same-address atomics for 65 536 jobs instead of 3072: they serialise).
constexpr uint32_t kCombinePad = 64;
constexpr int kGroupBlock = 256;
__global__ __launch_bounds__(kGroupBlock) void k_combine_classify(const uint64_t* __restrict__ idx, size_t n_per_job, size_t t, size_t B,
                                                              uint8_t* __restrict__ cls, uint32_t* __restrict__ counters) {
  __shared__ uint32_t wg_count[kCombineClasses];
  if (threadIdx.x < kCombineClasses) wg_count[threadIdx.x] = 0;
  __syncthreads();
  const size_t j = (size_t)blockIdx.x * kGroupBlock + threadIdx.x;
  const int c = (j < B) ? combine_job_class(idx + j * n_per_job, (int)t) : -1;
  if (j < B) cls[j] = (uint8_t)c;
  for (int k = 0; k < kCombineClasses; k++) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(c == k);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&wg_count[k], (uint32_t)__builtin_popcountll(m));
  }
  __syncthreads();
  if (threadIdx.x < kCombineClasses && wg_count[threadIdx.x]) atomicAdd(&counters[threadIdx.x], wg_count[threadIdx.x]);
}
__global__ __launch_bounds__(kGroupBlock) void k_combine_scatter(const uint8_t* __restrict__ cls, size_t B, uint32_t* __restrict__ counters,
                                                             uint32_t* __restrict__ perm) {
  __shared__ uint32_t wave_count[kGroupBlock / 64][kCombineClasses];
  __shared__ uint32_t wg_base[kCombineClasses];
  const size_t j = (size_t)blockIdx.x * kGroupBlock + threadIdx.x;
  const int c = (j < B) ? (int)cls[j] : -1;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t mine = 0;
  for (int k = 0; k < kCombineClasses; k++) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(c == k);
    if (lane == 0) wave_count[wave][k] = (uint32_t)__builtin_popcountll(m);
    if (c == k) mine = m;
  }
  __syncthreads();
  if (threadIdx.x < kCombineClasses) {
    // where the class's run starts: the padded sizes of the classes before it (all counts are final: the
    // classification kernel has completed), then this workgroup's share of the run
    const int k = (int)threadIdx.x;
    uint32_t start = 0, total = 0;
    for (int cc = 0; cc < k; cc++) start += (counters[cc] + kCombinePad - 1) / kCombinePad * kCombinePad;
    for (int w = 0; w < kGroupBlock / 64; w++) total += wave_count[w][k];
    wg_base[k] = total ? start + atomicAdd(&counters[4 + k], total) : 0;
  }
  __syncthreads();
  if (c >= 0) {
    uint32_t pos = wg_base[c];
    for (unsigned w = 0; w < wave; w++) pos += wave_count[w][c];
    perm[pos + (uint32_t)__builtin_popcountll(mine & ((1ull << lane) - 1ull))] = (uint32_t)j;
  }
}

template <class F, class IO, bool ARENA = false>
TC_D bool combine_fast(size_t t, const uint64_t* idx, bool live, IO& io, uint8_t* st) {
  if (t == 1) return job_combine_small_io<F, 2, IO, ARENA>(idx, live, io, st);
  if (t == 2) return job_combine_small_io<F, 3, IO, ARENA>(idx, live, io, st);
  return job_combine_small_io<F, 4, IO, ARENA>(idx, live, io, st);
}

// The combination runs as TWO kernels so that each carries only its own private segment (the general
// path's 4-share psi tables are 15 KB per lane, the fast path needs a fraction of that):
//   k_combine_fast     t in {1, 2, 3}, small distinct indices: short joint ladder + ONE division by the
//                      common denominator (tc_threshold.h lagrange_small_coeffs); leaves other jobs alone.
//                      Shares and results move through LDS a wave at a time (tc_stage.h): coalesced HBM access
//                      instead of one strided byte gather per lane.
//   k_combine_general  G1: everything else (t = 0, t >= 4, large or repeated indices): Lagrange coefficients from
//                      k_lagrange + Straus chunks; leaves at once when *need_general == 0.  G2: t = 0 only -- every other
//                      general job goes through the two-stage kernels of k_msm.hip, whose private segment is 10 KB
//                      instead of the 15.6 KB of the chunked psi ladders (concurrent contexts: INTEGRATION.md)
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_combine_fast(size_t t, size_t n_per_job, const uint64_t* __restrict__ idx,
                                                    const uint8_t* __restrict__ shares, size_t B,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                    const uint32_t* __restrict__ perm, size_t slots, TableArena ta) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  using IO = WaveRowIO<PB, L>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO::BYTES];
  uint32_t tslot = 0;
  if (L > 1) tslot = table_slot_acquire(ta);  // the G2 ladders keep their tables in the arena (tc_table.h)
  const size_t slot = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  size_t j = B;                                            // B = no job on this lane (past the end, or class padding)
  if (slot < slots) j = perm ? (size_t)perm[slot] : slot;  // grouped by denominator class, or the identity
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  IO io{lds, live ? shares + jj * n_per_job * PB : nullptr, (size_t)PB, live ? out + jj * PB : nullptr};
  uint8_t st = TC_JOB_OK;
  const bool done = combine_fast<F>(t, idx + jj * n_per_job, live, io, &st);
  if (done && status && (L == 1 || pair_leader())) status[j] = st;
  if (L > 1) table_slot_release(ta, tslot);
}

// G1 above one wave per SIMD (more than 65 536 jobs): 256 registers so that two waves share a SIMD, the [1 / D] ladder's table in
// the lane's arena entries (k_mul.hip k_g1_mul_arena explains)
// perm / slots: the jobs grouped by the class of their denominator (k_combine_classify / k_combine_scatter, as in G2) so that a
// wave of D = 1 jobs skips the [1 / D] ladder altogether (tc_jobs.h combine_divide_arena); perm = nullptr: the jobs' own order
__global__ __launch_bounds__(kBlock, 2) void k_combine_fast_g1_arena(size_t t, size_t n_per_job, const uint64_t* __restrict__ idx,
                                                                  const uint8_t* __restrict__ shares, size_t B, uint8_t* __restrict__ out,
                                                                  uint8_t* __restrict__ status, const uint32_t* __restrict__ perm, size_t slots,
                                                                  TableArena ta) {
  using IO = WaveRowIO<96, 1>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO::BYTES];
  const uint32_t tslot = table_slot_acquire(ta);
  const size_t slot = (size_t)blockIdx.x * kBlock + threadIdx.x;
  size_t j = B;                                            // B = no job on this lane (past the end, or class padding)
  if (slot < slots) j = perm ? (size_t)perm[slot] : slot;
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  IO io{lds, live ? shares + jj * n_per_job * 96 : nullptr, (size_t)96, live ? out + jj * 96 : nullptr};
  uint8_t st = TC_JOB_OK;
  const bool done = combine_fast<Fq, IO, true>(t, idx + jj * n_per_job, live, io, &st);
  if (done && status) status[j] = st;
  table_slot_release(ta, tslot);
}

template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1_AUX)) void k_combine_general(size_t t, size_t n_per_job, const uint64_t* __restrict__ idx,
                                                    const uint8_t* __restrict__ shares,
                                                    const uint32_t* __restrict__ lam, size_t B,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                    const uint32_t* __restrict__ need_general, TableArena ta) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  if (need_general && *need_general == 0) return;  // every job went through the fast path
  uint32_t tslot = 0;
  if (L > 1) tslot = table_slot_acquire(ta);
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  // idx == nullptr: every job is this kernel's (coefficients from Fr abscissae: no fast path ran)
  const bool mine = j < B && !(idx && t >= 1 && t <= 3 && combine_small_applies(idx + j * n_per_job, (int)t));  // else: k_combine_fast's
  if (mine) {
    if (status && status[j] != TC_JOB_OK) {  // lagrange stage flagged the job
      PointIO<F>::encode(Affine<F>::infinity(), out + j * PB);
    } else {
      // G2: only t = 0 reaches this kernel (launch_combine_g2)
      const uint8_t st = (L > 1) ? job_first_sample<F>(shares + j * n_per_job * PB, out + j * PB)
                                 : job_combine<Fq>((int)t, shares + j * n_per_job * PB, lam + j * (t + 1) * 8, out + j * PB);
      if (status && (L == 1 || pair_leader())) status[j] = st;
    }
  }
  if (L > 1) table_slot_release(ta, tslot);
}

// one lane per job: sum_i scalar_i * point_i with caller-supplied scalars (32 B LE each); the points of job
// j start at points + j * pts_stride (n * PB for per-job points, 0 for ONE point set shared by every job)
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1_AUX)) void k_lincomb(size_t n, const uint8_t* __restrict__ scalars,
                                                    const uint8_t* __restrict__ points, size_t pts_stride, size_t B,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status, TableArena ta) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  uint32_t tslot = 0;
  if (L > 1) tslot = table_slot_acquire(ta);
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (j < B) {
    uint8_t st = job_lincomb<F>((int)n, points + j * pts_stride, reinterpret_cast<const uint32_t*>(scalars + j * n * 32),
                                out + j * PB);
    if (status && (L == 1 || pair_leader())) status[j] = st;
  }
  if (L > 1) table_slot_release(ta, tslot);
}

void launch_lincomb_g1(hipStream_t st, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                       uint8_t* status, bool shared_points) {
  if (B) hipLaunchKernelGGL(k_lincomb<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, n, scalars, points, shared_points ? (size_t)0 : n * 96, B, out, status, TableArena{nullptr, nullptr});
}

void launch_lagrange(hipStream_t st, const uint64_t* idx, size_t n_per_job, size_t t, size_t B, uint32_t* lam,
                     uint8_t* status, uint32_t* need_general) {
  const size_t n = B * (t + 1);
  if (n) hipLaunchKernelGGL(k_lagrange, dim3(grid_for(n)), dim3(kBlock), 0, st, idx, n_per_job, t, B, lam, status, need_general);
}
void launch_fr_idx_narrow(hipStream_t st, const uint32_t* idx_fr, size_t n_per_job, size_t take, size_t B, uint64_t* idx64, uint32_t* wide,
                          uint8_t* valid) {
  if (B * n_per_job) hipLaunchKernelGGL(k_fr_idx_narrow, dim3(grid_for(B * n_per_job)), dim3(kBlock), 0, st, idx_fr, n_per_job, take, B, idx64, wide, valid);
}
void launch_lagrange_fr(hipStream_t st, const uint32_t* idx_fr, size_t n_per_job, size_t t, size_t B, uint32_t* lam, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_lagrange_fr, dim3(grid_for(B * (t + 1))), dim3(kBlock), 0, st, idx_fr, n_per_job, t, B, lam, status);
}
size_t lagrange_all_ws_words(size_t t, size_t B) { return B * 4 * (t + 1) * 8; }
void launch_lagrange_all(hipStream_t st, const uint64_t* idx, size_t n_per_job, size_t t, size_t B, uint32_t* lam, uint32_t* ws,
                         uint8_t* status) {
  if (!B) return;
  const size_t n = t + 1;
  if (n > (size_t)kLagMaxN) {
    hipLaunchKernelGGL(k_lagrange_all, dim3(grid_for(B)), dim3(kBlock), 0, st, idx, n_per_job, t, B, lam, ws, status);
    return;
  }
  uint32_t* xm = ws;                 // ws: 4 (t+1) x 8 words per job; the first three quarters are used here
  uint32_t* den = ws + B * n * 8;
  uint32_t* pre = ws + 2 * B * n * 8;
  const size_t jobs_per_block = (size_t)kLagBlock / n;
  hipLaunchKernelGGL(k_lagrange_den, dim3((unsigned)((B + jobs_per_block - 1) / jobs_per_block)), dim3(kLagBlock), 0, st, idx, n_per_job, t, B, xm,
                     den);
  hipLaunchKernelGGL(k_lagrange_finish, dim3(grid_for(B)), dim3(kBlock), 0, st, t, B, (const uint32_t*)xm, (const uint32_t*)den, pre, lam, status);
}
// need_general: one zeroed word the Lagrange stage counts non-fast jobs in (nullptr when t == 0: no
// Lagrange stage, the general kernel takes every job)
void launch_combine_g1(hipStream_t st, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status, const uint32_t* need_general, TableArena ta,
                       uint8_t* cls, uint32_t* counters, uint32_t* perm, hipEvent_t before_main) {
  if (!B) return;
  if (idx && t >= 1 && t <= 3) {
    if (!ta.mem || !ta.flags) return;   // (the arena could not be allocated: the call has failed already)
    size_t slots = B;
    if (perm && cls && counters) {
      slots = combine_group_slots(B);
      (void)hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), st);
      (void)hipMemsetAsync(perm, 0xff, slots * sizeof(uint32_t), st);
      const unsigned gb = (unsigned)((B + kGroupBlock - 1) / kGroupBlock);
      hipLaunchKernelGGL(k_combine_classify, dim3(gb), dim3(kGroupBlock), 0, st, idx, n_per_job, t, B, cls, counters);
      hipLaunchKernelGGL(k_combine_scatter, dim3(gb), dim3(kGroupBlock), 0, st, cls, B, counters, perm);
    } else {
      perm = nullptr;
    }
    if (before_main) (void)hipEventRecord(before_main, st);  // (as in launch_combine_g2: work beside this call starts behind its long kernel)
#if TC_G1_ARENA_MIN > 0
    if (B <= kG1ArenaMinJobs)
      hipLaunchKernelGGL(k_combine_fast<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, B, out, status, (const uint32_t*)nullptr, B, TableArena{nullptr, nullptr});
    else
#endif
      hipLaunchKernelGGL(k_combine_fast_g1_arena, dim3(grid_for(slots)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, B, out, status, (const uint32_t*)perm, slots, ta);
  }
  hipLaunchKernelGGL(k_combine_general<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, lam, B, out, status, need_general, TableArena{nullptr, nullptr});
}
size_t combine_group_slots(size_t B) { return B + (size_t)kCombineClasses * kCombinePad; }
// cls: B bytes, counters: 8 words, perm: combine_group_slots(B) words (scratch of the caller); pass
// perm = nullptr to run the jobs in their own order
void launch_combine_g2(hipStream_t st, TableArena ta, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status, uint8_t* cls, uint32_t* counters,
                       uint32_t* perm, const uint32_t* need_general, hipEvent_t before_main) {
  if (!B || !ta.mem || !ta.flags) return;
  if (t >= 1 && t <= 3) {
    size_t slots = B;
    if (perm) {
      slots = combine_group_slots(B);
      (void)hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), st);
      (void)hipMemsetAsync(perm, 0xff, slots * sizeof(uint32_t), st);
      const unsigned gb = (unsigned)((B + kGroupBlock - 1) / kGroupBlock);
      hipLaunchKernelGGL(k_combine_classify, dim3(gb), dim3(kGroupBlock), 0, st, idx, n_per_job, t, B, cls, counters);
      hipLaunchKernelGGL(k_combine_scatter, dim3(gb), dim3(kGroupBlock), 0, st, cls, B, counters, perm);
    }
    // before_main: recorded where the stream reaches the ONE long kernel of the call -- work the caller runs beside this call on
    // another stream (the membership tests of checked-input mode, tc_api.hip Call::run_checks) waits for it, so that it starts
    // BEHIND this launch and fills the slots its cheap waves free instead of delaying them
    if (before_main) (void)hipEventRecord(before_main, st);
    hipLaunchKernelGGL(k_combine_fast<Fq2>, dim3(grid_for(slots * kG2Lanes)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, B, out, status, (const uint32_t*)perm, slots, ta);
  }
  // t >= 1: the jobs the fast path leaves go through the two-stage kernels of k_msm.hip (the caller launches them);
  // t = 0 (the first sample is the result) is all that is left for the general kernel in G2
  if (t == 0)
    hipLaunchKernelGGL(k_combine_general<Fq2>, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, lam, B, out, status, need_general, ta);
}

}  // namespace tc

#if defined(TC_PHASE_TIMING)
// experiment builds only: the phase stamps of the last launch (tools/phase_marks_probe.py)
extern "C" int tc_debug_phase_marks(unsigned long long* out, size_t words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tc_phase_marks), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
