// Register-only VALU throughput microbenchmark for gfx950: measures the issue rate of the
// integer multiply-add forms the Montgomery kernels can be built from, so that the roofline
// denominator (P_int) in bench.py / DESIGN.md is a MEASURED number.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096
#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[NACC];
  double dac[NACC];
  uint32_t x[NACC];
  for (int i = 0; i < NACC; i++) {
    acc[i] = (uint64_t)tid * 0x9e3779b97f4a7c15ull + i + seed;
    dac[i] = 1.0 + 1e-9 * (double)(tid + i);
    x[i] = tid * 2654435761u + i + seed;
  }
  const uint32_t b = seed | 1u;
  const double y = 1.0000001, z = 1e-12;
#pragma unroll 1
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int kx = 0; kx < NACC; kx++) {
      if (OP == 0) acc[kx] = (uint64_t)(uint32_t)acc[(kx + 1) & (NACC - 1)] * b + acc[kx];  // v_mad_u64_u32
      if (OP == 1) x[kx] = x[kx] * x[(kx + 1) & (NACC - 1)];                                 // v_mul_lo_u32
      if (OP == 2) x[kx] = __umulhi(x[kx], x[(kx + 1) & (NACC - 1)]) + 1u;                    // v_mul_hi_u32 (+add)
      if (OP == 3) dac[kx] = __builtin_fma(dac[kx], y, z);                                    // v_fma_f64
      if (OP == 4) x[kx] = __umul24(x[kx], x[(kx + 1) & (NACC - 1)]) + x[kx];                 // v_mad_u32_u24
      if (OP == 5) acc[kx] = acc[kx] + acc[(kx + 1) & (NACC - 1)];                            // 64-bit add
      if (OP == 6) x[kx] = x[kx] + x[(kx + 1) & (NACC - 1)];                                  // v_add_u32
      if (OP == 7) x[kx] = (uint32_t)__builtin_fmaf((float)0 + __uint_as_float(x[kx]), 1.0001f, 0.5f);  // placeholder f32 fma
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i] + (uint64_t)dac[i] + x[i];
  out[tid] = s;
}

template <int OP>
double run(const char* name, uint64_t* d_out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 777u + r);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ITERS * NACC;
  double rate = ops / (best * 1e-3);
  printf("{\"op\": \"%s\", \"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f}\n", name,
         best, rate, 256.0 * 4 * 64 * 2.4e9 / rate);
  return rate;
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  const int blocks = p.multiProcessorCount * 8;
  uint64_t* d_out;
  hipMalloc(&d_out, (size_t)blocks * 256 * 8);
  run<0>("v_mad_u64_u32", d_out, blocks);
  run<1>("v_mul_lo_u32", d_out, blocks);
  run<2>("v_mul_hi_u32+add", d_out, blocks);
  run<3>("v_fma_f64", d_out, blocks);
  run<4>("v_mad_u32_u24", d_out, blocks);
  run<5>("add_u64", d_out, blocks);
  run<6>("v_add_u32", d_out, blocks);
  hipFree(d_out);
  return 0;
}
