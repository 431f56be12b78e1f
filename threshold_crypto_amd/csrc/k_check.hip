// gfx950 kernels: opt-in validation of uncompressed operands (tc_ctx_set_input_checks) and the public
// membership tests.  The reference only ever holds G1/G2 values that passed the CHECKED decode of
// from_bytes (/root/reference/src/lib.rs:140-146, 246-252: on the curve AND in the order-r subgroup);
// the GLV/GLS ladders, the psi shortcuts of the share combiner and the folded cofactor constants rely on
// that.  A caller that feeds bytes from an untrusted source straight into the C ABI turns these on.
#include "tc_jobs.h"
#include "tc_launch.h"

namespace tc {

// valid[i] = 1 iff point i decodes (range, flags, curve equation) and lies in the order-r subgroup.
// Points are addressed as job-major records: point i = (job i / take, sample i % take) of records that
// hold n_per_job points each, so that a combine call validates exactly the first t+1 samples it uses.
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_subgroup_check(
    const uint8_t* __restrict__ pts, size_t stride, size_t n_per_job, size_t take, size_t n, uint8_t* __restrict__ valid) {
  constexpr int L = JobLanes<F>::N;
  const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (i >= n) return;
  const size_t rec = i / take, k = i % take;
  Affine<F> p;
  bool ok = PointIO<F>::decode(pts + (rec * n_per_job + k) * stride, p);
  if (!ok) p = Affine<F>::infinity();
  ok = ok && point_in_subgroup(p);
  if (L == 1 || pair_leader()) valid[i] = ok ? 1 : 0;
}

// a job with an invalid operand fails like an undecodable one: status INVALID_ENCODING and the identity
// as its output (point entries), or ok = 0 (boolean entries)
// job j owns the per_job points of record j / group (group = 1: its own record; group = S: S outputs per
// point; group >= B: one operand broadcast to every job)
__global__ void k_invalidate_jobs(const uint8_t* __restrict__ valid, size_t per_job, size_t group, size_t B,
                                  uint8_t* __restrict__ status, uint8_t* __restrict__ out, size_t out_bytes,
                                  uint8_t* __restrict__ ok) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= B) return;
  bool good = true;
  for (size_t k = 0; k < per_job; k++) good = good && valid[(j / group) * per_job + k] != 0;
  if (good) return;
  if (status && status[j] == TC_JOB_OK) status[j] = TC_JOB_INVALID_ENCODING;
  if (ok) ok[j] = 0;
  if (out) {
    for (size_t b = 0; b < out_bytes; b++) out[j * out_bytes + b] = 0;
    if (out_bytes == 96 || out_bytes == 192) out[j * out_bytes] = 0x40;  // the identity's encoding
  }
}

void launch_subgroup_check_g1(hipStream_t st, const uint8_t* pts, size_t stride, size_t n_per_job, size_t take, size_t n,
                              uint8_t* valid) {
  if (n) hipLaunchKernelGGL(k_subgroup_check<Fq>, dim3(grid_for(n)), dim3(kBlock), 0, st, pts, stride, n_per_job, take, n, valid);
}
void launch_subgroup_check_g2(hipStream_t st, const uint8_t* pts, size_t stride, size_t n_per_job, size_t take, size_t n,
                              uint8_t* valid) {
  if (n) hipLaunchKernelGGL(k_subgroup_check<Fq2>, dim3(grid_for(n * kG2Lanes)), dim3(kBlock), 0, st, pts, stride, n_per_job, take, n, valid);
}
void launch_invalidate_jobs(hipStream_t st, const uint8_t* valid, size_t per_job, size_t group, size_t B, uint8_t* status,
                            uint8_t* out, size_t out_bytes, uint8_t* ok) {
  if (B) hipLaunchKernelGGL(k_invalidate_jobs, dim3(grid_for(B)), dim3(kBlock), 0, st, valid, per_job, group ? group : 1, B, status, out, out_bytes, ok);
}

}  // namespace tc
