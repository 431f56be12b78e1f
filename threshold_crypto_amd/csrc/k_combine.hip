// gfx950 kernels: Lagrange coefficients and the share combiner (interpolate).
#include "tc_jobs.h"
#include "tc_launch.h"

namespace tc {

// one lane per (job, sample position): lambda_i of job j
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_lagrange(const uint64_t* __restrict__ idx, size_t n_per_job, size_t t,
                                                     size_t B, uint32_t* __restrict__ lam,
                                                     uint8_t* __restrict__ status, int g2) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t k = t + 1;
  if (tid >= B * k) return;
  const size_t j = tid / k, i = tid % k;
  (void)g2;
  if (combine_small_applies(idx + j * n_per_job, (int)t)) return;  // the fast path owns the job
  uint8_t st = job_lagrange(idx + j * n_per_job, (int)t, (int)i, lam + tid * 8);
  if (st && status) status[j] = st;
}

template <class F>
TC_D bool combine_fast(size_t t, const uint64_t* idx, const uint8_t* shares, uint8_t* out, uint8_t* st) {
  if (t == 1) return job_combine_small<F, 2>(idx, shares, out, st);
  if (t == 2) return job_combine_small<F, 3>(idx, shares, out, st);
  if (t == 3) return job_combine_small<F, 4>(idx, shares, out, st);
  return false;
}

// one lane per job: sum_i lambda_i * share_i
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_combine(size_t t, size_t n_per_job, const uint64_t* __restrict__ idx,
                                                    const uint8_t* __restrict__ shares,
                                                    const uint32_t* __restrict__ lam, size_t B,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (j >= B) return;
  if (status && status[j] != TC_JOB_OK) {  // lagrange stage flagged the job
    PointIO<F>::encode(Affine<F>::infinity(), out + j * PB);
    return;
  }
  uint8_t st = TC_JOB_OK;
  if (!combine_fast<F>(t, idx + j * n_per_job, shares + j * n_per_job * PB, out + j * PB, &st))
    st = job_combine<F>((int)t, shares + j * n_per_job * PB, lam + j * (t + 1) * 8, out + j * PB);
  if (status && (L == 1 || pair_leader())) status[j] = st;
}

// one lane per job: sum_i scalar_i * point_i with caller-supplied scalars (32 B LE each)
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_lincomb(size_t n, const uint8_t* __restrict__ scalars,
                                                    const uint8_t* __restrict__ points, size_t B,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (j >= B) return;
  uint8_t st = job_lincomb<F>((int)n, points + j * n * PB, reinterpret_cast<const uint32_t*>(scalars + j * n * 32),
                              out + j * PB);
  if (status && (L == 1 || pair_leader())) status[j] = st;
}

void launch_lincomb_g1(hipStream_t st, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                       uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_lincomb<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, n, scalars, points, B, out, status);
}
void launch_lincomb_g2(hipStream_t st, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                       uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_lincomb<Fq2>, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, n, scalars, points, B, out, status);
}

void launch_lagrange(hipStream_t st, const uint64_t* idx, size_t n_per_job, size_t t, size_t B, uint32_t* lam,
                     uint8_t* status, bool g2) {
  const size_t n = B * (t + 1);
  if (n) hipLaunchKernelGGL(k_lagrange, dim3(grid_for(n)), dim3(kBlock), 0, st, idx, n_per_job, t, B, lam, status, g2 ? 1 : 0);
}
void launch_combine_g1(hipStream_t st, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_combine<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, lam, B, out, status);
}
void launch_combine_g2(hipStream_t st, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                       const uint32_t* lam, size_t B, uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_combine<Fq2>, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, t, n_per_job, idx, shares, lam, B, out, status);
}

}  // namespace tc
