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

// ok[j] = ok[j] && status[j] == OK
__global__ void k_ok_and_status(const uint8_t* __restrict__ status, size_t B, uint8_t* __restrict__ ok) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < B && status[j] != TC_JOB_OK) ok[j] = 0;
}

// r[i] = d0 + d1 |x| + d2 |x|^2 + d3 |x|^3 with four 16-bit digits from ChaCha20(key = seed, block counter = i), d0
// odd: 2^63 equally likely scalars, pairwise distinct mod r (base-|x| digits are unique), so a bad share survives
// the combined check with probability <= 2^-63 -- and on G2, where psi = [x], the multiplication is a 16-column
// ladder over the psi-images (tc_msm.h short-scalar mode).  Unpredictable to whoever produced the shares as long as
// the seed is drawn after they were received.  Output: 32 B little-endian canonical scalars (r[i] < 2^208).
__global__ void k_rlc_scalars(const uint8_t* __restrict__ seed32, size_t n, uint8_t* __restrict__ out_fr) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t key[8];
  for (int w = 0; w < 8; w++)
    key[w] = (uint32_t)seed32[4 * w] | ((uint32_t)seed32[4 * w + 1] << 8) | ((uint32_t)seed32[4 * w + 2] << 16) | ((uint32_t)seed32[4 * w + 3] << 24);
  ChaChaRng rng;
  rng.init(key);
  rng.counter = i;
  const uint32_t w0 = rng.next_u32(), w1 = rng.next_u32();
  const uint64_t d[4] = {(uint64_t)(w0 & 0xffffu) | 1ull, (uint64_t)(w0 >> 16), (uint64_t)(w1 & 0xffffu), (uint64_t)(w1 >> 16)};
  // Horner in base |x| on 4 x 64-bit words: acc = ((d3 X + d2) X + d1) X + d0 < 2^208
  uint64_t acc[4] = {d[3], 0, 0, 0};
  for (int j = 2; j >= 0; j--) {
    unsigned __int128 carry = d[j];
    for (int w = 0; w < 4; w++) {
      carry += (unsigned __int128)acc[w] * BLS_X_ABS;
      acc[w] = (uint64_t)carry;
      carry >>= 64;
    }
  }
  for (int w = 0; w < 4; w++)
    for (int b = 0; b < 8; b++) out_fr[i * 32 + 8 * w + b] = (uint8_t)(acc[w] >> (8 * b));
}
// the same for values of G1, whose endomorphism is phi = [-x^2]: r[i] = a + b x^2 with a (odd) and b from two 32-bit draws
// of ChaCha20(key = seed, block counter = i) -- 2^63 equally likely scalars, pairwise distinct mod r (a + b x^2 < r), and
// a 16-step ladder in the base-4 sign-aligned form of tc_msm.h (short-scalar mode, nbits = 32).
__global__ void k_rlc_scalars_g1(const uint8_t* __restrict__ seed32, size_t n, uint8_t* __restrict__ out_fr) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t key[8];
  for (int w = 0; w < 8; w++)
    key[w] = (uint32_t)seed32[4 * w] | ((uint32_t)seed32[4 * w + 1] << 8) | ((uint32_t)seed32[4 * w + 2] << 16) | ((uint32_t)seed32[4 * w + 3] << 24);
  ChaChaRng rng;
  rng.init(key);
  rng.counter = i;
  const uint64_t a = (uint64_t)(rng.next_u32() | 1u), b = (uint64_t)rng.next_u32();
  // a + b x^2, x^2 = |x|^2 (128 bits): three words
  const unsigned __int128 x2 = (unsigned __int128)BLS_X_ABS * BLS_X_ABS;
  const uint64_t x2lo = (uint64_t)x2, x2hi = (uint64_t)(x2 >> 64);
  uint64_t acc[4];
  unsigned __int128 c = (unsigned __int128)x2lo * b + a;
  acc[0] = (uint64_t)c;
  c = (c >> 64) + (unsigned __int128)x2hi * b;
  acc[1] = (uint64_t)c;
  acc[2] = (uint64_t)(c >> 64);
  acc[3] = 0;
  for (int w = 0; w < 4; w++)
    for (int bb = 0; bb < 8; bb++) out_fr[i * 32 + 8 * w + bb] = (uint8_t)(acc[w] >> (8 * bb));
}
__global__ void k_gather_rows(const uint8_t* __restrict__ src, size_t row_words, const uint32_t* __restrict__ map, size_t rows,
                              uint8_t* __restrict__ dst) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;  // one 8-byte word per lane: coalesced within a row
  if (t >= rows * row_words) return;
  const size_t r = t / row_words, w = t % row_words;
  reinterpret_cast<uint64_t*>(dst)[t] = reinterpret_cast<const uint64_t*>(src)[(size_t)map[r] * row_words + w];
}
__global__ void k_scatter_bytes(const uint8_t* __restrict__ src, const uint32_t* __restrict__ map, size_t rows, uint8_t* __restrict__ dst) {
  const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < rows) dst[map[r]] = src[r];
}
void launch_ok_and_status(hipStream_t st, const uint8_t* status, size_t B, uint8_t* ok) {
  if (B) hipLaunchKernelGGL(k_ok_and_status, dim3(grid_for(B)), dim3(kBlock), 0, st, status, B, ok);
}
void launch_rlc_scalars(hipStream_t st, const uint8_t* seed32, size_t n, uint8_t* out_fr) {
  if (n) hipLaunchKernelGGL(k_rlc_scalars, dim3(grid_for(n)), dim3(kBlock), 0, st, seed32, n, out_fr);
}
void launch_rlc_scalars_g1(hipStream_t st, const uint8_t* seed32, size_t n, uint8_t* out_fr) {
  if (n) hipLaunchKernelGGL(k_rlc_scalars_g1, dim3(grid_for(n)), dim3(kBlock), 0, st, seed32, n, out_fr);
}
void launch_gather_rows(hipStream_t st, const uint8_t* src, size_t row_bytes, const uint32_t* map, size_t rows, uint8_t* dst) {
  const size_t words = rows * (row_bytes / 8);
  if (words) hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(words)), dim3(kBlock), 0, st, src, row_bytes / 8, map, rows, dst);
}
void launch_scatter_bytes(hipStream_t st, const uint8_t* src, const uint32_t* map, size_t rows, uint8_t* dst) {
  if (rows) hipLaunchKernelGGL(k_scatter_bytes, dim3(grid_for(rows)), dim3(kBlock), 0, st, src, map, rows, dst);
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
