// gfx950 kernels: DKG algebra (tc_dkg.h) -- fixed-base commitments from an LDS-resident window table of the
// G1 generator, rows of bivariate commitments, Fr interpolation.
#include "tc_dkg.h"
#include "tc_launch.h"

namespace tc {

constexpr int kFbBlock = 256;  // four waves share one LDS copy of the table; two such workgroups per CU

__global__ void k_fixed_base_table(int32_t* __restrict__ tbl) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e < kFbWindows * kFbEntries) fixed_base_table_entry(e, tbl + (size_t)e * kFbPointWords);
}

// out[j] = fr[j] * g1.  Persistent workgroups: each stages the 56 KB table in LDS once (coalesced 16-byte
// loads) and then walks the batch with a grid stride; every addition reads its table entry with ds_read.
__global__ __launch_bounds__(kFbBlock, 2) void k_g1_fixed_base(const int32_t* __restrict__ tbl_g, const uint8_t* __restrict__ fr,
                                                              size_t M, uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  __shared__ int32_t tbl[kFbTableWords];
  {
    const int4* src = reinterpret_cast<const int4*>(tbl_g);
    int4* dst = reinterpret_cast<int4*>(tbl);
    for (int i = threadIdx.x; i < kFbTableWords / 4; i += kFbBlock) dst[i] = src[i];
  }
  __syncthreads();
  for (size_t base = (size_t)blockIdx.x * kFbBlock; base < M; base += (size_t)gridDim.x * kFbBlock) {
    const size_t j = base + threadIdx.x;
    if (j < M) {
      const uint8_t st = job_g1_fixed_base_mul((const int32_t*)tbl, fr + j * 32, out + j * 96);
      if (status) status[j] = st;
    }
  }
}

// out[(m * (degree + 1) + i)] = BivarCommitment::row(xs[m])[i]
__global__ __launch_bounds__(kBlock, TC_WAVES_G1_AUX) void k_bivar_commitment_row(const uint8_t* __restrict__ commit, size_t degree,
                                                                            const uint64_t* __restrict__ xs, size_t M,
                                                                            uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t n = degree + 1;
  if (tid >= M * n) return;
  const size_t m = tid / n, i = tid % n;
  const uint8_t st = job_bivar_commitment_row(commit, degree, i, xs[m], out + tid * 96);
  if (status) status[tid] = st;
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_fr_interpolate(size_t n, const uint32_t* __restrict__ xs, const uint32_t* __restrict__ ys,
                                                                      size_t B, uint32_t* __restrict__ out, uint32_t* __restrict__ ws,
                                                                      uint8_t* __restrict__ status) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= B) return;
  const uint8_t st = job_fr_interpolate(n, xs + j * n * 8, ys + j * n * 8, out + j * n * 8, ws + j * 2 * (n + 1) * 8);
  if (status) status[j] = st;
}

size_t fixed_base_table_bytes() { return (size_t)kFbTableWords * sizeof(int32_t); }
void launch_fixed_base_table(hipStream_t st, int32_t* tbl) {
  hipLaunchKernelGGL(k_fixed_base_table, dim3(grid_for(kFbWindows * kFbEntries)), dim3(kBlock), 0, st, tbl);
}
void launch_g1_fixed_base(hipStream_t st, const int32_t* tbl, const uint8_t* fr, size_t M, uint8_t* out, uint8_t* status, int cus) {
  if (!M) return;
  size_t blocks = (M + kFbBlock - 1) / kFbBlock;
  const size_t resident = (size_t)(cus > 0 ? cus : 256) * 2;  // two workgroups per CU hold their LDS tables at once
  if (blocks > resident) blocks = resident;
  hipLaunchKernelGGL(k_g1_fixed_base, dim3((unsigned)blocks), dim3(kFbBlock), 0, st, tbl, fr, M, out, status);
}
void launch_bivar_commitment_row(hipStream_t st, const uint8_t* commit, size_t degree, const uint64_t* xs, size_t M, uint8_t* out,
                                 uint8_t* status) {
  const size_t n = M * (degree + 1);
  if (n) hipLaunchKernelGGL(k_bivar_commitment_row, dim3(grid_for(n)), dim3(kBlock), 0, st, commit, degree, xs, M, out, status);
}
void launch_fr_interpolate(hipStream_t st, size_t n, const uint32_t* xs, const uint32_t* ys, size_t B, uint32_t* out, uint32_t* ws,
                           uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_fr_interpolate, dim3(grid_for(B)), dim3(kBlock), 0, st, n, xs, ys, B, out, ws, status);
}

}  // namespace tc
