// gfx950 kernel: batched pairing-product check  e(a,b) == e(c,d).
#include "tc_jobs.h"
#include "tc_launch.h"

namespace tc {

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_pairing_check(const uint8_t* __restrict__ a, size_t sa,
                                                          const uint8_t* __restrict__ b, size_t sb,
                                                          const uint8_t* __restrict__ c, size_t sc,
                                                          const uint8_t* __restrict__ d, size_t sd, size_t B,
                                                          uint8_t* __restrict__ ok) {
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j >= B) return;
  const uint8_t r = job_pairing_check(a + j * sa, b + j * sb, c + j * sc, d + j * sd);
  if (pair_leader()) ok[j] = r;
}

void launch_pairing_check(hipStream_t st, const uint8_t* a, size_t sa, const uint8_t* b, size_t sb, const uint8_t* c,
                          size_t sc, const uint8_t* d, size_t sd, size_t B, uint8_t* ok) {
  if (B) hipLaunchKernelGGL(k_pairing_check, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, a, sa, b, sb, c, sc, d, sd, B, ok);
}

}  // namespace tc
