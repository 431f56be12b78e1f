// gfx950 kernels: batched G1/G2 scalar multiplication and point compression.
#include "tc_jobs.h"
#include "tc_launch.h"

#include <stdlib.h>

namespace tc {

// One curve op per lane.  Lane order is signer-major (tid = s*B + j) so that the 64 lanes of
// a wave share the signer's scalar: the bit-serial double-and-add stays wave-uniform.
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_point_mul(const uint8_t* __restrict__ fr, const uint8_t* __restrict__ pts,
                                                      size_t S, size_t B, uint8_t* __restrict__ out,
                                                      uint8_t* __restrict__ status, TableArena ta) {
  constexpr int PB = PointIO<F>::BYTES;
  constexpr int L = JobLanes<F>::N;
  uint32_t tslot = 0;
  if (L > 1) tslot = table_slot_acquire(ta);  // the GLS ladder's table lives in the arena (tc_table.h)
  const size_t tid = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (tid < S * B) {
    const size_t s = tid / B, j = tid % B;
    const size_t o = j * S + s;
    uint8_t st = job_point_mul<F>(fr + s * 32, pts + j * PB, out + o * PB);
    if (status && (L == 1 || pair_leader())) status[o] = st;
  }
  if (L > 1) table_slot_release(ta, tslot);
}

// G1, batches above one wave per SIMD (more than 65 536 multiplications): built for 256 registers so that TWO waves share a SIMD
// -- a lone wave issues a v_mad_i64_i32 only every 8.75 cycles, two saturate the pipe (tools/ubench_issue) --, the 8-entry table
// of the base-4 GLV ladder in the wave's arena slot (1 KB per lane, full-line entries) instead of 224 registers.
__global__ __launch_bounds__(kBlock, 2) void k_g1_mul_arena(const uint8_t* __restrict__ fr, const uint8_t* __restrict__ pts, size_t S, size_t B,
                                                          uint8_t* __restrict__ out, uint8_t* __restrict__ status, TableArena ta) {
  const uint32_t tslot = table_slot_acquire(ta);
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = tid < S * B;
  const size_t s = live ? tid / B : 0, j = live ? tid % B : 0;
  const size_t o = j * S + s;
  // (a lane past the end multiplies job 0 again and drops the result: the ladder's loads and stores are the whole wave's)
  uint8_t sink[96];
  const uint8_t st = job_g1_mul_arena(fr + s * 32, pts + j * 96, live ? out + o * 96 : sink);
  if (live && status) status[o] = st;
  table_slot_release(ta, tslot);
}

// S > 1 scalars over the same G2 points: a lane pair takes one point and a chunk of up to
// kMulShare scalars (chunk-major lane order, so a wave shares its scalars).
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_g2_mul_shared(const uint8_t* __restrict__ fr, const uint8_t* __restrict__ pts, size_t S,
                                                                     size_t B, uint8_t* __restrict__ out, uint8_t* __restrict__ status, TableArena ta) {
  const uint32_t tslot = table_slot_acquire(ta);
  const size_t tid = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t chunks = (S + kMulShare - 1) / kMulShare;
  if (tid < chunks * B) {
    const size_t c = tid / B, j = tid % B;
    const size_t s0 = c * kMulShare;
    const int n = (int)((S - s0 < (size_t)kMulShare) ? S - s0 : (size_t)kMulShare);
    const size_t o = j * S + s0;
    job_g2_mul_shared(fr + s0 * 32, n, pts + j * 192, out + o * 192, status ? status + o : nullptr, pair_leader());
  }
  table_slot_release(ta, tslot);
}

// shares of B messages by per-message signer subsets: a lane pair takes one hash point and up to kGatherShare of the
// n selected signers (chunk-major lane order as above); out[(j * n + k)] = sk[idx[j * n + k]] * pts[j]
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_g2_mul_gather(const uint8_t* __restrict__ sk, size_t N, const uint64_t* __restrict__ idx,
                                                                     const uint8_t* __restrict__ pts, size_t n, size_t B,
                                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ status, TableArena ta,
                                                                     size_t share) {
  const uint32_t tslot = table_slot_acquire(ta);
  const size_t tid = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t chunks = (n + share - 1) / share;  // share <= kGatherShare signers per lane pair (launch_g2_mul_gather)
  if (tid < chunks * B) {
    const size_t c = tid / B, j = tid % B;
    const size_t s0 = c * share;
    const int cnt = (int)((n - s0 < share) ? n - s0 : share);
    const size_t o = j * n + s0;
    job_g2_mul_gather(sk, N, idx + o, cnt, pts + j * 192, out + o * 192, status ? status + o : nullptr, pair_leader());
  }
  table_slot_release(ta, tslot);
}

template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_compress(const uint8_t* __restrict__ in, size_t B,
                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  constexpr int L = JobLanes<F>::N;
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (j >= B) return;
  uint8_t st = job_compress<F>(in + j * PointIO<F>::BYTES, out + j * PointIO<F>::CBYTES);
  if (status && (L == 1 || pair_leader())) status[j] = st;
}

template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_decompress(const uint8_t* __restrict__ in, size_t B,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  constexpr int L = JobLanes<F>::N;
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (j >= B) return;
  uint8_t st = job_decompress<F>(in + j * PointIO<F>::CBYTES, out + j * PointIO<F>::BYTES);
  if (status && (L == 1 || pair_leader())) status[j] = st;
}

// G2: a lane pair takes TWO points (tc_duo.h) -- pair p decodes points 2p and 2p + 1
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_decompress_g2_x2(const uint8_t* __restrict__ in, size_t B, uint8_t* __restrict__ out,
                                                                       uint8_t* __restrict__ status) {
  const size_t p = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t ja = 2 * p, jb = 2 * p + 1;
  if (ja >= B) return;
  const bool has_b = jb < B;
  uint8_t sa, sb;
  job_decompress_g2_x2(in + ja * 96, in + (has_b ? jb : ja) * 96, out + ja * 192, has_b ? out + jb * 192 : nullptr, sa, sb);
  if (status && pair_leader()) {
    status[ja] = sa;
    if (has_b) status[jb] = sb;
  }
}

// Wire ingest for the combiners (tc_combine_signatures_wire_batch / tc_decrypt_wire_batch): the FIRST `take` of the n_per_job
// compressed samples of every job -- exactly the samples interpolate() uses (/root/reference/src/lib.rs:727-730) -- through the
// CHECKED decode of from_bytes (:140-146, 246-252), written compactly (job-major, take per job); valid[i] = 1 iff sample i
// decoded and is a member.  check = false leaves the membership test to the caller's batched test (k_check.hip).
template <class F>
__global__ __launch_bounds__(kBlock, (JobLanes<F>::N > 1 ? TC_WAVES_G2 : TC_WAVES_G1)) void k_decompress_take(
    const uint8_t* __restrict__ in, size_t n_per_job, size_t take, size_t n, uint8_t* __restrict__ out, uint8_t* __restrict__ valid) {
  constexpr int L = JobLanes<F>::N;
  const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (i >= n) return;
  const size_t rec = i / take, k = i % take;
  const uint8_t st = job_decompress<F>(in + (rec * n_per_job + k) * PointIO<F>::CBYTES, out + i * PointIO<F>::BYTES);
  if (L == 1 || pair_leader()) valid[i] = st == TC_JOB_OK ? 1 : 0;
}
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_decompress_take_g2_x2(const uint8_t* __restrict__ in, size_t n_per_job, size_t take,
                                                                            size_t n, uint8_t* __restrict__ out, uint8_t* __restrict__ valid) {
  const size_t p = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t ia = 2 * p;
  if (ia >= n) return;
  const bool has_b = ia + 1 < n;
  const size_t ib = has_b ? ia + 1 : ia;
  uint8_t sa, sb;
  job_decompress_g2_x2(in + ((ia / take) * n_per_job + ia % take) * 96, in + ((ib / take) * n_per_job + ib % take) * 96, out + ia * 192,
                       has_b ? out + ib * 192 : nullptr, sa, sb);
  if (pair_leader()) {
    valid[ia] = sa == TC_JOB_OK ? 1 : 0;
    if (has_b) valid[ib] = sb == TC_JOB_OK ? 1 : 0;
  }
}
// the matching prefix of the index array: out[j * take + k] = idx[j * n_per_job + k]
__global__ void k_take_u64(const uint64_t* __restrict__ in, size_t n_per_job, size_t take, size_t n, uint64_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[(i / take) * n_per_job + i % take];
}

// scalars / G1 points folded with the hash's constant FR_COFACTOR_FIX (tc_jobs.h)
__global__ void k_fr_scale_cofactor_fix(const uint8_t* __restrict__ fr, size_t S, uint8_t* __restrict__ out) {
  const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (s < S) job_fr_scale_cofactor_fix(fr + s * 32, out + s * 32);
}
__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_g1_scale_cofactor_fix(const uint8_t* __restrict__ in, size_t stride, size_t n,
                                                                            uint8_t* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < n) job_g1_scale_cofactor_fix(in + j * stride, out + j * 96);
}

__global__ void k_fill_g1_generator(uint8_t* out96, uint8_t* out96_unfix) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    g1_encode_uncompressed(g1_generator(), out96);
    g1_encode_uncompressed(jac_to_affine(g1_mul_glv(g1_generator(), FR_COFACTOR_UNFIX)), out96_unfix);
  }
}

// ta: the context's table arena (the per-lane ladder tables live there)
void launch_g1_mul(hipStream_t st, TableArena ta, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                   uint8_t* status) {
  if (!(S * B) || !ta.mem || !ta.flags) return;
#if TC_G1_ARENA_MIN > 0
  static const char* force = getenv("TC_G1_MUL_FORM");  // experiments: "arena" / "regs"
  if (force ? force[0] != 'a' : S * B <= kG1ArenaMinJobs) {
    hipLaunchKernelGGL(k_point_mul<Fq>, dim3(grid_for(S * B)), dim3(kBlock), 0, st, fr, pts, S, B, out, status, TableArena{nullptr, nullptr});
    return;
  }
#endif
  hipLaunchKernelGGL(k_g1_mul_arena, dim3(grid_for(S * B)), dim3(kBlock), 0, st, fr, pts, S, B, out, status, ta);
}
void launch_g2_mul(hipStream_t st, TableArena ta, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                   uint8_t* status) {
  if (!(S * B) || !ta.mem || !ta.flags) return;
  if (S > 1) {
    const size_t chunks = (S + kMulShare - 1) / kMulShare;
    hipLaunchKernelGGL(k_g2_mul_shared, dim3(grid_for(chunks * B * kG2Lanes)), dim3(kBlock), 0, st, fr, pts, S, B, out, status, ta);
  } else {
    hipLaunchKernelGGL(k_point_mul<Fq2>, dim3(grid_for(S * B * kG2Lanes)), dim3(kBlock), 0, st, fr, pts, S, B, out, status, ta);
  }
}
void launch_g2_mul_gather(hipStream_t st, TableArena ta, const uint8_t* sk, size_t N, const uint64_t* idx, const uint8_t* pts, size_t n, size_t B,
                          uint8_t* out, uint8_t* status) {
  if (!(n * B) || !ta.mem || !ta.flags) return;
  const size_t share = signers_per_lane_pair(n, B, kGatherShare);
  const size_t chunks = (n + share - 1) / share;
  hipLaunchKernelGGL(k_g2_mul_gather, dim3(grid_for(chunks * B * kG2Lanes)), dim3(kBlock), 0, st, sk, N, idx, pts, n, B, out, status, ta, share);
}
void launch_g1_compress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_compress<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, in, B, out, status);
}
void launch_g2_compress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_compress<Fq2>, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, in, B, out, status);
}
void launch_g1_decompress(hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_decompress<Fq>, dim3(grid_for(B)), dim3(kBlock), 0, st, in, B, out, status);
}
void launch_g2_decompress(const Tuning& tn, hipStream_t st, const uint8_t* in, size_t B, uint8_t* out, uint8_t* status) {
  if (!B) return;
  if (duo_form(B, tn.duo_min_decode)) hipLaunchKernelGGL(k_decompress_g2_x2, dim3(grid_for((B + 1) / 2 * kG2Lanes)), dim3(kBlock), 0, st, in, B, out, status);
  else hipLaunchKernelGGL(k_decompress<Fq2>, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, in, B, out, status);
}
void launch_decompress_take(const Tuning& tn, hipStream_t st, bool g2, const uint8_t* in, size_t n_per_job, size_t take, size_t B, uint8_t* out, uint8_t* valid) {
  const size_t n = B * take;
  if (!n) return;
  if (g2 && duo_form(n, tn.duo_min_decode)) hipLaunchKernelGGL(k_decompress_take_g2_x2, dim3(grid_for((n + 1) / 2 * kG2Lanes)), dim3(kBlock), 0, st, in, n_per_job, take, n, out, valid);
  else if (g2) hipLaunchKernelGGL(k_decompress_take<Fq2>, dim3(grid_for(n * kG2Lanes)), dim3(kBlock), 0, st, in, n_per_job, take, n, out, valid);
  else hipLaunchKernelGGL(k_decompress_take<Fq>, dim3(grid_for(n)), dim3(kBlock), 0, st, in, n_per_job, take, n, out, valid);
}
void launch_take_u64(hipStream_t st, const uint64_t* in, size_t n_per_job, size_t take, size_t B, uint64_t* out) {
  if (B * take) hipLaunchKernelGGL(k_take_u64, dim3(grid_for(B * take)), dim3(kBlock), 0, st, in, n_per_job, take, B * take, out);
}
void launch_fr_scale_cofactor_fix(hipStream_t st, const uint8_t* fr, size_t S, uint8_t* out) {
  if (S) hipLaunchKernelGGL(k_fr_scale_cofactor_fix, dim3(grid_for(S)), dim3(kBlock), 0, st, fr, S, out);
}
void launch_g1_scale_cofactor_fix(hipStream_t st, const uint8_t* in, size_t stride, size_t n, uint8_t* out) {
  if (n) hipLaunchKernelGGL(k_g1_scale_cofactor_fix, dim3(grid_for(n)), dim3(kBlock), 0, st, in, stride, n, out);
}
void launch_fill_g1_generator(hipStream_t st, uint8_t* out96, uint8_t* out96_unfix) {
  hipLaunchKernelGGL(k_fill_g1_generator, dim3(1), dim3(64), 0, st, out96, out96_unfix);
}

}  // namespace tc
