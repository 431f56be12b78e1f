// gfx950 kernel: batched pairing-product check  e(a,b) == e(c,d).
#include "tc_jobs.h"
#include "tc_launch.h"
#include "tc_stage.h"

namespace tc {

// One lane pair per check.  The four operands of the wave's 32 checks are moved through ONE LDS row buffer with
// coalesced 8-byte loads (tc_stage.h), one operand after the other; a stride of 0 broadcasts one record.
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_pairing_check(const uint8_t* __restrict__ a, size_t sa,
                                                          const uint8_t* __restrict__ b, size_t sb,
                                                          const uint8_t* __restrict__ c, size_t sc,
                                                          const uint8_t* __restrict__ d, size_t sd, size_t B,
                                                          uint8_t* __restrict__ ok) {
  using IO1 = WaveRowIO<96, kG2Lanes>;
  using IO2 = WaveRowIO<192, kG2Lanes>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO2::BYTES];
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  IO1 ia{lds, live ? a + jj * sa : nullptr, 0, nullptr}, ic{lds, live ? c + jj * sc : nullptr, 0, nullptr};
  IO2 ib{lds, live ? b + jj * sb : nullptr, 0, nullptr}, id{lds, live ? d + jj * sd : nullptr, 0, nullptr};
  const uint8_t r = job_pairing_check_io(live, ia, ib, ic, id);
  if (live && pair_leader()) ok[j] = r;
}

void launch_pairing_check(hipStream_t st, const uint8_t* a, size_t sa, const uint8_t* b, size_t sb, const uint8_t* c,
                          size_t sc, const uint8_t* d, size_t sd, size_t B, uint8_t* ok) {
  if (B) hipLaunchKernelGGL(k_pairing_check, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, a, sa, b, sb, c, sc, d, sd, B, ok);
}

}  // namespace tc
