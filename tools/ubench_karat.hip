// Experiment: one-level Karatsuba inside the lane-pair Fq2 product (tools/karat_exp.h) against the
// shipped schoolbook fq2p_mul_call.  Only products, 2 waves per SIMD as the real kernels run.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_karat.hip -o tools/ubench_karat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../threshold_crypto_amd/csrc/tc_tower.h"
#include "karat_exp.h"

using namespace tc;
#ifndef ITERS
#define ITERS 4000
#endif

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) FqRaw fq2p_mulk_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13, int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t b4, int32_t b5, int32_t b6, int32_t b7, int32_t b8, int32_t b9, int32_t b10, int32_t b11, int32_t b12, int32_t b13, int32_t odd) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  const int32_t b[FQ_LIMBS] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13};
  int32_t y[FQ_LIMBS], z[FQ_LIMBS], w[FQ_LIMBS];
  const bool o = odd != 0;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t ao = pair_swap(a[i]);
    const int32_t bo = pair_swap(b[i]);
    y[i] = o ? bo : b[i];
    w[i] = o ? b[i] : bo;
    z[i] = o ? ao : -ao;
  }
  FqRaw r;
  fq_mul2k_body(a, y, z, w, r.l);
  return r;
}
#endif

template <bool K>
__global__ __launch_bounds__(64, 2) void k_mul(const int32_t* in, int32_t* out, int jobs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t >> 1, odd = t & 1;
  if (j >= jobs) return;
  Fq x, y;
  for (int i = 0; i < FQ_LIMBS; i++) {
    x.l[i] = in[(j * 4 + 0 + odd) * FQ_LIMBS + i];
    y.l[i] = in[(j * 4 + 2 + odd) * FQ_LIMBS + i];
  }
#if defined(__HIP_DEVICE_COMPILE__)
  TC_NOUNROLL for (int it = 0; it < ITERS; it++) {
    FqRaw r = K ? fq2p_mulk_call(x.l[0], x.l[1], x.l[2], x.l[3], x.l[4], x.l[5], x.l[6], x.l[7], x.l[8], x.l[9], x.l[10], x.l[11], x.l[12], x.l[13], y.l[0], y.l[1], y.l[2], y.l[3], y.l[4], y.l[5], y.l[6], y.l[7], y.l[8], y.l[9], y.l[10], y.l[11], y.l[12], y.l[13], odd)
                  : fq2p_mul_call(x.l[0], x.l[1], x.l[2], x.l[3], x.l[4], x.l[5], x.l[6], x.l[7], x.l[8], x.l[9], x.l[10], x.l[11], x.l[12], x.l[13], y.l[0], y.l[1], y.l[2], y.l[3], y.l[4], y.l[5], y.l[6], y.l[7], y.l[8], y.l[9], y.l[10], y.l[11], y.l[12], y.l[13]);
    for (int i = 0; i < FQ_LIMBS; i++) {
      y.l[i] = x.l[i];
      x.l[i] = r.l[i];
    }
  }
#endif
  for (int i = 0; i < FQ_LIMBS; i++) out[(j * 2 + odd) * FQ_LIMBS + i] = x.l[i];
}

int main(int argc, char** argv) {
  const int jobs = argc > 1 ? atoi(argv[1]) : 65536;
  const size_t nin = (size_t)jobs * 4 * FQ_LIMBS, nout = (size_t)jobs * 2 * FQ_LIMBS;
  int32_t* h = (int32_t*)malloc(nin * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < nin; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (int32_t)(s & ((1 << FQ_RADIX) - 1));
    if (i % FQ_LIMBS == FQ_LIMBS - 1) h[i] &= 0xffff;
  }
  int32_t *d_in, *d_a, *d_b;
  hipMalloc(&d_in, nin * 4); hipMalloc(&d_a, nout * 4); hipMalloc(&d_b, nout * 4);
  hipMemcpy(d_in, h, nin * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_a = 1e30f, ms_b = 1e30f;
  for (int r = 0; r < 4; r++) {
    float ms;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mul<false>, dim3((2 * jobs + 63) / 64), dim3(64), 0, 0, d_in, d_a, jobs);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < ms_a) ms_a = ms;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mul<true>, dim3((2 * jobs + 63) / 64), dim3(64), 0, 0, d_in, d_b, jobs);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < ms_b) ms_b = ms;
  }
  int32_t* ha = (int32_t*)malloc(nout * 4); int32_t* hb = (int32_t*)malloc(nout * 4);
  hipMemcpy(ha, d_a, nout * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, d_b, nout * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < nout; i++) diff += (ha[i] != hb[i]);
  printf("{\"jobs\": %d, \"iters\": %d, \"schoolbook_ms\": %.3f, \"karatsuba_ms\": %.3f, \"ratio\": %.3f, \"limb_mismatch\": %zu}\n", jobs, ITERS, ms_a, ms_b, ms_b / ms_a, diff);
  return 0;
}
