// Does instruction-level parallelism inside ONE wave raise the v_mad_i64_i32 rate?  The field multiplier's columns
// compile to a single dependent accumulator chain (LLVM re-associates the source's "two chains"); here C chains
// (C = 1, 2, 3, 4) are kept apart with empty asm barriers and issued round-robin, at 1 / 2 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// one v_mad_i64_i32 the optimiser cannot hoist, merge or reorder (the carry-out SGPR pair is a throw-away)
#define MAD(acc, x, y)                                                                                  \
  do {                                                                                                  \
    uint64_t cc_;                                                                                       \
    asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cc_) : "v"(x), "v"(y));         \
  } while (0)

template <int C>
__global__ __launch_bounds__(64) void k(int64_t* out, const int32_t* in, int iters) {
  const int tid = blockIdx.x * 64 + threadIdx.x;
  int32_t a[14], b[14];
  for (int i = 0; i < 14; i++) {
    a[i] = in[(tid + i) & 1023];
    b[i] = in[(tid * 3 + i) & 1023];
  }
  int64_t s[4] = {tid, 1, 2, 3};
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
#pragma unroll
      for (int j = 0; j < 12 / C; j++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          MAD(s[c], a[i], b[(i + j * C + c) % 14]);
        }
      }
    }
  }
  int64_t r = 0;
  for (int c = 0; c < 4; c++) r += s[c];
  out[tid] = r;
}

template <int C>
void run(int64_t* d_out, const int32_t* d_in, int waves_per_simd, int cus, int iters = 4000) {
  const int blocks = cus * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mads = (double)blocks * 64 * iters * 14 * (12 / C) * C;
  printf("{\"op\": \"v_mad_i64_i32_chained\", \"chains\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"ms\": %.3f, \"T_mad_per_s\": %.2f}\n", C, waves_per_simd, iters,
         best, mads / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const bool peak_only = argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'p';  // --peak
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int64_t* d_out;
  int32_t* d_in;
  hipMalloc(&d_out, (size_t)p.multiProcessorCount * 4 * 8 * 64 * 8);
  hipMalloc(&d_in, 4096);
  hipMemset(d_in, 0x11, 4096);
  // the roofline denominator of bench.py: 2 chains, 4 waves per SIMD, ~40 ms per launch (sustained clock)
  run<2>(d_out, d_in, 4, p.multiProcessorCount, 36000);
  if (peak_only) return 0;
  run<2>(d_out, d_in, 2, p.multiProcessorCount, 36000);
  for (int w : {1, 2, 4}) {
    run<1>(d_out, d_in, w, p.multiProcessorCount);
    run<2>(d_out, d_in, w, p.multiProcessorCount);
    run<3>(d_out, d_in, w, p.multiProcessorCount);
    run<4>(d_out, d_in, w, p.multiProcessorCount);
  }
  return 0;
}
