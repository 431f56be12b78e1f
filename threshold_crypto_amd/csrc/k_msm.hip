// gfx950 kernels: G2 linear combinations in two stages (tc_msm.h) -- the share combiner of the general path (every
// job the small-index fast path does not take; BASELINE config "t=67, N=200") and tc_g2_lincomb_batch.
#include "tc_msm.h"
#include "tc_launch.h"

namespace tc {

// which jobs of the batch are this path's: all of them, or (share combination with t <= 3) those the small-index fast
// path of k_combine_fast leaves alone; *need counts them (k_lagrange) and zero ends the kernel at once
__device__ __forceinline__ bool msm_job_taken(const MsmFilter& f, size_t j) {
  return !(f.idx && f.t >= 1 && f.t <= 3 && combine_small_applies(f.idx + j * f.n_per_job, (int)f.t));
}

// stage T: one lane pair per (job, chunk of 4 shares)
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_msm_tables(size_t n, size_t pts_stride, const uint8_t* __restrict__ points,
                                                                  const uint32_t* __restrict__ scalars, size_t B,
                                                                  int32_t* __restrict__ tbl, uint8_t* __restrict__ codes,
                                                                  uint8_t* __restrict__ status, int nbits, MsmFilter f) {
  if (f.need && *f.need == 0) return;
  const size_t chunks = msm_chunks(n);
  const size_t tid = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (tid >= B * chunks) return;
  const size_t j = tid / chunks, c = tid % chunks;
  if (!msm_job_taken(f, j)) return;
  const size_t shares4 = chunks * kMsmChunk;
  const bool ok = job_msm_tables(n, c, points + j * pts_stride, scalars + j * n * 8, tbl + j * shares4 * 8 * kMsmEntryWords,
                                 codes + j * kMsmColumns * shares4, pair_leader(), nbits);
  if (!ok && pair_leader() && status[j] == TC_JOB_OK) status[j] = TC_JOB_INVALID_ENCODING;
}

// stage L: one lane pair per job
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_msm_ladder(size_t n, size_t B, const int32_t* __restrict__ tbl,
                                                                  const uint8_t* __restrict__ codes, uint8_t* __restrict__ out,
                                                                  const uint8_t* __restrict__ status, int nbits, MsmFilter f) {
  if (f.need && *f.need == 0) return;
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j >= B) return;
  if (!msm_job_taken(f, j)) return;
  if (status[j] != TC_JOB_OK) {
    g2_encode_uncompressed(G2Affine::infinity(), out + j * 192);
    return;
  }
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  const G2Jac r = job_msm_ladder(n, tbl + j * shares4 * 8 * kMsmEntryWords, codes + j * kMsmColumns * shares4, nbits);
  g2_encode_uncompressed(jac_to_affine(r), out + j * 192);
}
// stage L for batches that would not fill the GPU with one lane pair per job: `parts` (a power of two, 2 .. 32) lane
// pairs per job, each over a range of the job's shares (tc_msm.h msm_part); the partial sums are added across the lane
// pairs of the job (they sit in one wave) in log2(parts) rounds, the first one writes the result.
__device__ __forceinline__ Fq2 msm_from_partner(const Fq2& v, int lanes) {
  Fq2 r = v;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = __shfl_xor(v.m.l[i], lanes, 64);
  return r;
}
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_msm_ladder_split(size_t n, size_t B, const int32_t* __restrict__ tbl,
                                                                        const uint8_t* __restrict__ codes, uint8_t* __restrict__ out,
                                                                        const uint8_t* __restrict__ status, int nbits, MsmFilter f,
                                                                        size_t parts) {
  if (f.need && *f.need == 0) return;
  const size_t lp = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t j = lp / parts, g = lp % parts;
  if (j >= B) return;
  if (!msm_job_taken(f, j)) return;
  if (status[j] != TC_JOB_OK) {
    if (g == 0) g2_encode_uncompressed(G2Affine::infinity(), out + j * 192);
    return;
  }
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G2Jac r = job_msm_ladder_part<true>(n, tbl + j * shares4 * 8 * kMsmEntryWords, codes + j * kMsmColumns * shares4, nbits, msm_part(n, g, parts));
  TC_NOUNROLL for (size_t d = 1; d < parts; d <<= 1) {
    const int lanes = (int)(d * kG2Lanes);
    const G2Jac o{msm_from_partner(r.x, lanes), msm_from_partner(r.y, lanes), msm_from_partner(r.z, lanes)};
    r = jac_add(r, o);
  }
  if (g == 0) g2_encode_uncompressed(jac_to_affine(r), out + j * 192);
}

// ONE point set shared by every job with short scalars (the recoding never negates a base then): one table set for the launch
__host__ __device__ inline bool msm_g1_shared_tables(size_t pts_stride, int nbits) { return pts_stride == 0 && nbits < 128; }
// ---- G1 (tc_msm.h): stage T one lane per (job, chunk of 4 shares), stage L one lane per job -- or per PART of a job
// when the batch alone does not fill the GPU (a job's `parts` lanes are adjacent; their partial sums meet in
// log2(parts) rounds of one __shfl_xor exchange + one addition).  Two waves per SIMD (256 registers).
__global__ __launch_bounds__(kBlock, 2) void k_msm_tables_g1(size_t n, size_t pts_stride, const uint8_t* __restrict__ points,
                                                           const uint32_t* __restrict__ scalars, size_t B, int32_t* __restrict__ tbl,
                                                           uint8_t* __restrict__ codes, uint8_t* __restrict__ status, int nbits) {
  const size_t chunks = msm_chunks(n);
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= B * chunks) return;
  const size_t j = tid / chunks, c = tid % chunks;
  const size_t shares4 = chunks * kMsmChunk;
  // ONE point set shared by every job, short scalars (the random linear combinations of the RLC checks): one table set, built
  // by job 0; the other jobs only recode their scalars (B x N x 1 KiB of identical tables before)
  const bool shared = msm_g1_shared_tables(pts_stride, nbits);
  const bool ok = job_msm_tables_g1(n, c, points + j * pts_stride, scalars + j * n * 8, tbl + (shared ? 0 : j * shares4 * 8 * kMsmEntryWordsG1),
                                    codes + j * kMsmColumns * shares4, nbits, !shared || j == 0);
  if (!ok && status[j] == TC_JOB_OK) status[j] = TC_JOB_INVALID_ENCODING;
}
__device__ __forceinline__ Fq msm_from_lane(const Fq& v, int lanes) {
  Fq r = v;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = __shfl_xor(v.l[i], lanes, 64);
  return r;
}
template <bool SPLIT>
__global__ __launch_bounds__(kBlock, 2) void k_msm_ladder_g1(size_t n, size_t B, const int32_t* __restrict__ tbl, const uint8_t* __restrict__ codes,
                                                           uint8_t* __restrict__ out, const uint8_t* __restrict__ status, size_t parts, int top,
                                                           int shared) {
  const size_t lp = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t j = lp / parts, g = lp % parts;
  if (j >= B) return;  // (parts divides 64: a job's lanes leave together)
  if (status[j] != TC_JOB_OK) {
    if (g == 0) g1_encode_uncompressed(G1Affine::infinity(), out + j * 96);
    return;
  }
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G1Jac r = job_msm_ladder_g1_part<SPLIT>(n, tbl + (shared ? 0 : j * shares4 * 8 * kMsmEntryWordsG1), codes + j * kMsmColumns * shares4,
                                          msm_part(n, g, parts), top);
  if (SPLIT) {
    TC_NOUNROLL for (size_t d = 1; d < parts; d <<= 1) {
      const G1Jac o{msm_from_lane(r.x, (int)d), msm_from_lane(r.y, (int)d), msm_from_lane(r.z, (int)d)};
      r = jac_add(r, o);
    }
  }
  if (g == 0) g1_encode_uncompressed(jac_to_affine(r), out + j * 96);
}

size_t msm_table_bytes_g1(size_t n, size_t B) { return B * msm_chunks(n) * kMsmChunk * 8 * kMsmEntryWordsG1 * sizeof(int32_t); }
// jobs that own a table set in a launch_msm_g1 call: all of them, or one for a shared point set in short-scalar mode
size_t msm_table_jobs_g1(size_t pts_stride, int nbits, size_t B) { return msm_g1_shared_tables(pts_stride, nbits) ? 1 : B; }
// status: B bytes, TC_JOB_OK for the jobs to run (others get the identity and keep their status)
// nbits = 128: any scalars below r.  nbits < 128 (even): odd scalars k1 + k2 x^2 with k1, k2 < 2^nbits (a job with another
// scalar fails): nbits doublings instead of 128.
void launch_msm_g1(hipStream_t st, size_t n, size_t pts_stride, const uint8_t* points, const uint32_t* scalars, size_t B, int32_t* tbl,
                   uint8_t* codes, uint8_t* out, uint8_t* status, int nbits) {
  if (!B || !n) return;
  const int top = nbits / 2;
  const int shared = msm_g1_shared_tables(pts_stride, nbits) ? 1 : 0;
  hipLaunchKernelGGL(k_msm_tables_g1, dim3(grid_for(B * msm_chunks(n))), dim3(kBlock), 0, st, n, pts_stride, points, scalars, B, tbl, codes, status, nbits);
  // lanes per job in stage L: 1 when the batch fills the 2048 wave slots by itself (131 072 lanes), else the power of two
  // that does, with at least four shares per part
  size_t parts = 1;
  while (parts < 64 && B * parts * 2 <= 131072 && parts * 2 * 4 <= n) parts *= 2;
  if (parts == 1)
    hipLaunchKernelGGL(k_msm_ladder_g1<false>, dim3(grid_for(B)), dim3(kBlock), 0, st, n, B, (const int32_t*)tbl, (const uint8_t*)codes, out,
                       (const uint8_t*)status, parts, top, shared);
  else
    hipLaunchKernelGGL(k_msm_ladder_g1<true>, dim3(grid_for(B * parts)), dim3(kBlock), 0, st, n, B, (const int32_t*)tbl, (const uint8_t*)codes, out,
                       (const uint8_t*)status, parts, top, shared);
}

size_t msm_table_bytes(size_t n, size_t B) { return B * msm_chunks(n) * kMsmChunk * 8 * kMsmEntryWords * sizeof(int32_t); }
size_t msm_code_bytes(size_t n, size_t B) { return B * kMsmColumns * msm_chunks(n) * kMsmChunk; }
// status: B bytes, TC_JOB_OK for the jobs to run (others get the identity and keep their status)
// nbits = 64: any scalars below r.  nbits < 64: odd scalars whose four base-|x| digits are below 2^nbits (a job with
// another scalar fails): nbits doublings instead of 64.
void launch_msm_g2(hipStream_t st, size_t n, size_t pts_stride, const uint8_t* points, const uint32_t* scalars, size_t B,
                   int32_t* tbl, uint8_t* codes, uint8_t* out, uint8_t* status, int nbits, MsmFilter f) {
  if (!B || !n) return;
  const size_t lanes = B * msm_chunks(n) * kG2Lanes;
  hipLaunchKernelGGL(k_msm_tables, dim3(grid_for(lanes)), dim3(kBlock), 0, st, n, pts_stride, points, scalars, B, tbl, codes, status, nbits, f);
  // lane pairs per job in stage L: 1 when the batch fills the GPU's 2048 wave slots by itself (65 536 lane pairs), else
  // the power of two that does, with at least four shares per part
  size_t parts = 1;
  while (parts < 32 && B * parts * 2 <= 65536 && parts * 2 * 4 <= n) parts *= 2;
  if (parts == 1)
    hipLaunchKernelGGL(k_msm_ladder, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, n, B, (const int32_t*)tbl, (const uint8_t*)codes, out,
                       (const uint8_t*)status, nbits, f);
  else
    hipLaunchKernelGGL(k_msm_ladder_split, dim3(grid_for(B * parts * kG2Lanes)), dim3(kBlock), 0, st, n, B, (const int32_t*)tbl,
                       (const uint8_t*)codes, out, (const uint8_t*)status, nbits, f, parts);
}

}  // namespace tc
