// gfx950 kernels: the shares of a message by many signers through a per-message comb (tc_comb.h).
#include "tc_comb.h"
#include "tc_launch.h"

namespace tc {

// stage T: one lane pair per message; ok[j] = the hash point decoded
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_comb_tables(const uint8_t* __restrict__ pts, size_t B, int32_t* __restrict__ tbl,
                                                                   uint8_t* __restrict__ ok) {
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j >= B) return;
  const bool good = job_comb_tables(pts + j * 192, (tbl_word*)(tbl + j * (size_t)kCombTableWords));
  if (pair_leader()) ok[j] = good ? 1 : 0;
}

// stage S: a lane pair takes one message and up to kCombShare of its n signers (chunk-major lane order)
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_comb_sign(const uint8_t* __restrict__ sk, size_t N, const uint64_t* __restrict__ idx,
                                                                 const int32_t* __restrict__ tbl, const uint8_t* __restrict__ ok, size_t n,
                                                                 size_t B, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                                 TableArena ta, size_t share) {
  const uint32_t tslot = table_slot_acquire(ta);  // (the guarded fallback ladder reads column 0 in place: no table is built here)
  const size_t tid = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t chunks = (n + share - 1) / share;  // share <= kCombShare signers per lane pair (launch_comb_sign)
  if (tid < chunks * B) {
    const size_t c = tid / B, j = tid % B;
    const size_t s0 = c * share;
    const int cnt = (int)((n - s0 < share) ? n - s0 : share);
    const size_t o = j * n + s0;
    job_comb_sign(sk, N, idx + o, cnt, (const tbl_word*)(tbl + j * (size_t)kCombTableWords), ok[j] != 0, out + o * 192,
                  status ? status + o : nullptr, pair_leader());
  }
  table_slot_release(ta, tslot);
}

size_t comb_table_bytes(size_t B) { return B * (size_t)kCombTableWords * sizeof(int32_t); }
void launch_comb_sign(hipStream_t st, TableArena ta, const uint8_t* sk, size_t N, const uint64_t* idx, const uint8_t* pts, size_t n, size_t B,
                      int32_t* tbl, uint8_t* ok, uint8_t* out, uint8_t* status) {
  if (!(n * B) || !ta.mem || !ta.flags) return;
  hipLaunchKernelGGL(k_comb_tables, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, pts, B, tbl, ok);
  const size_t share = signers_per_lane_pair(n, B, kCombShare);
  const size_t chunks = (n + share - 1) / share;
  hipLaunchKernelGGL(k_comb_sign, dim3(grid_for(chunks * B * kG2Lanes)), dim3(kBlock), 0, st, sk, N, idx, (const int32_t*)tbl, (const uint8_t*)ok, n, B,
                     out, status, ta, share);
}

}  // namespace tc
