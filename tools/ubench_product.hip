// What the lane-pair product of the SHIPPED multiplier (tc_field.h fq2p_mul_call / fq2p_sqr_call, out of line, the very code the
// kernels call) reaches when a SIMD runs nothing else: w = 1, 2, 3, 4 one-wave workgroups per SIMD, register-only, >= 20 ms per
// launch.  This is the ceiling of every Fq2 kernel at its occupancy (the kernels run at two waves per SIMD):
//   rate    executed v_mad_i64_i32 lane-operations / kernel time (588 per lane and product, 392 per square),
//   frac    rate / the saturated v_mad_i64_i32 rate measured in the same process (the loop of tools/ubench_issue --peak).
// Shapes: "mul" = a chain of dependent products x = x * y; "mix" = x = x * y, y = y^2 + x (the r01 microbenchmark's loop);
// "two" = two INDEPENDENT product chains per wave (does instruction-level parallelism inside a wave help at all?).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_product.hip -o tools/ubench_product      run: tools/ubench_product
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "../threshold_crypto_amd/csrc/tc_tower.h"

using namespace tc;

struct P {
  Fq m;
};
__device__ __forceinline__ P pmul(const P& a, const P& b, int odd) {
  P r;
#if defined(__HIP_DEVICE_COMPILE__)
  FqRaw t = fq2p_mul_call(a.m.l[0], a.m.l[1], a.m.l[2], a.m.l[3], a.m.l[4], a.m.l[5], a.m.l[6], a.m.l[7], a.m.l[8], a.m.l[9], a.m.l[10], a.m.l[11], a.m.l[12], a.m.l[13], b.m.l[0], b.m.l[1], b.m.l[2], b.m.l[3], b.m.l[4], b.m.l[5], b.m.l[6], b.m.l[7], b.m.l[8], b.m.l[9], b.m.l[10], b.m.l[11], b.m.l[12], b.m.l[13]);
  for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = t.l[i];
#endif
  return r;
}
__device__ __forceinline__ P psqr(const P& a, int odd) {
  P r;
#if defined(__HIP_DEVICE_COMPILE__)
  FqRaw t = fq2p_sqr_call(a.m.l[0], a.m.l[1], a.m.l[2], a.m.l[3], a.m.l[4], a.m.l[5], a.m.l[6], a.m.l[7], a.m.l[8], a.m.l[9], a.m.l[10], a.m.l[11], a.m.l[12], a.m.l[13]);
  for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = t.l[i];
#endif
  return r;
}

template <int SHAPE>
__global__ __launch_bounds__(64) void k_product(const int32_t* in, int32_t* out, int iters) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int odd = t & 1;
  P x, y, u, v;
  volatile int32_t spill[FQ_LIMBS];
  for (int i = 0; i < FQ_LIMBS; i++) {
    x.m.l[i] = in[((t & 1023) * 4 + 0) * FQ_LIMBS + i];
    y.m.l[i] = in[((t & 1023) * 4 + 1) * FQ_LIMBS + i];
    u.m.l[i] = in[((t & 1023) * 4 + 2) * FQ_LIMBS + i];
    v.m.l[i] = in[((t & 1023) * 4 + 3) * FQ_LIMBS + i];
  }
  TC_NOUNROLL for (int it = 0; it < iters; it++) {
    if (SHAPE == 0) {
      x = pmul(x, y, odd);
    } else if (SHAPE == 1) {
      x = pmul(x, y, odd);
      P s = psqr(y, odd);
      for (int i = 0; i < FQ_LIMBS; i++) y.m.l[i] = s.m.l[i] + x.m.l[i];
    } else if (SHAPE == 2) {
      x = pmul(x, y, odd);
      u = pmul(u, v, odd);
    } else if (SHAPE == 3) {
      // what the towers put between two products: a sum, a difference and a carry normalisation of the operands (no memory)
      x = pmul(x, y, odd);
      for (int i = 0; i < FQ_LIMBS; i++) y.m.l[i] = x.m.l[i] + u.m.l[i] - v.m.l[i];
      y.m = y.m.norm();
    } else if (SHAPE == 4) {
      // a spill: 14 registers stored to scratch after a product, reloaded straight into the next product's operands
      x = pmul(x, y, odd);
      for (int i = 0; i < FQ_LIMBS; i++) spill[i] = x.m.l[i];
      asm volatile("" ::: "memory");
      for (int i = 0; i < FQ_LIMBS; i++) y.m.l[i] = spill[i] + (it & 1);
    } else {
      // a table look-up: 14 words per lane from a 64 KB per-wave table in global memory, issued before a product, used after it
      const int32_t* row = in + ((((unsigned)x.m.l[0] >> 3) & 7) * 64 + threadIdx.x) * FQ_LIMBS % (1024 * 4 * FQ_LIMBS - 64);
      int32_t tbl[FQ_LIMBS];
      for (int i = 0; i < FQ_LIMBS; i++) tbl[i] = __builtin_nontemporal_load(row + i);
      x = pmul(x, y, odd);
      for (int i = 0; i < FQ_LIMBS; i++) y.m.l[i] = (x.m.l[i] + tbl[i]) & FQ_MASK;
    }
  }
  for (int i = 0; i < FQ_LIMBS; i++) out[(size_t)t * FQ_LIMBS + i] = x.m.norm().l[i] + y.m.norm().l[i] + u.m.norm().l[i];
}

__global__ __launch_bounds__(64) void k_peak(uint64_t* out, uint32_t seed, int iters) {
  const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
  int64_t a0 = (int64_t)tid * 0x9e3779b97f4a7c15ll + seed, a1 = a0 ^ 0x5555;
  const int32_t m0 = (int32_t)(seed | 1u), m1 = (int32_t)(seed * 3u | 1u);
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 32; u++) {
      uint64_t cc;
      asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(a0), "=s"(cc) : "v"(m0), "v"(m1));
      asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(a1), "=s"(cc) : "v"(m0), "v"(m1));
    }
  }
  out[tid] = (uint64_t)(a0 + a1);
}

static float time_launch(void (*launch)(int, int, void*, void*), int blocks, int iters, void* a, void* b) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    float ms;
    hipEventRecord(e0, 0);
    launch(blocks, iters, a, b);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
template <int SHAPE>
static void launch_product(int blocks, int iters, void* in, void* out) {
  hipLaunchKernelGGL(k_product<SHAPE>, dim3(blocks), dim3(64), 0, 0, (const int32_t*)in, (int32_t*)out, iters);
}
static void launch_peak(int blocks, int iters, void* out, void*) { hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(64), 0, 0, (uint64_t*)out, 777u, iters); }

int main(int argc, char** argv) {
  const bool ceiling_only = argc > 1 && std::string(argv[1]) == "--ceiling";   // one line for bench.py: peak + the product at 1 / 2 waves
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int simds = p.multiProcessorCount * 4;
  const size_t in_words = (size_t)1024 * 4 * FQ_LIMBS;
  int32_t* h = (int32_t*)malloc(in_words * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < in_words; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (int32_t)(s & ((1 << FQ_RADIX) - 1));
    if (i % FQ_LIMBS == FQ_LIMBS - 1) h[i] &= 0xffff;
  }
  int32_t *d_in, *d_out;
  hipMalloc(&d_in, in_words * 4);
  hipMalloc(&d_out, (size_t)simds * 8 * 64 * FQ_LIMBS * 4);
  hipMemcpy(d_in, h, in_words * 4, hipMemcpyHostToDevice);
  const float peak_ms = time_launch(launch_peak, simds * 4, 40000, d_out, nullptr);
  const double peak = (double)simds * 4 * 40000 * 64 * 64 / (peak_ms * 1e-3) / 1e12;
  printf("{\"device\": \"%s\", \"simds\": %d, \"peak_T_lane_mac_per_s\": %.2f, \"peak_is\": \"v_mad_i64_i32, two chained accumulators, 4 waves per SIMD, %.1f ms\"}\n",
         p.gcnArchName, simds, peak, peak_ms);
  if (ceiling_only) {
    double rate[2];
    for (int w = 1; w <= 2; w++) {
      const float ms = time_launch(launch_product<0>, simds * w, 8000, d_in, d_out);
      rate[w - 1] = (double)simds * w * 64 * 8000 * 588.0 / (ms * 1e-3) / 1e12;
    }
    printf("{\"same_process_peak_T\": %.2f, \"product_T_1_wave\": %.2f, \"product_T_2_waves\": %.2f, \"product_frac_1_wave\": %.3f, \"product_frac_2_waves\": %.3f, "
           "\"is\": \"fq2p_mul_call alone (tools/ubench_product): what the shipped lane-pair product reaches with one / two waves per SIMD, as a fraction of the "
           "v_mad_i64_i32 rate measured in the same process\"}\n",
           peak, rate[0], rate[1], rate[0] / peak, rate[1] / peak);
    return 0;
  }
  const char* names[6] = {"mul (x = x*y)", "mix (x = x*y; y = y^2 + x)", "two independent chains (x = x*y; u = u*v)",
                          "mul + sum, difference and carry normalisation of an operand (no memory)",
                          "mul + 14 registers through scratch (store, reload into the next operands)",
                          "mul + a 14-word table row from global memory (issued before the product, used after)"};
  const double macs_per_iter[6] = {588.0, 588.0 + 392.0, 2 * 588.0, 588.0, 588.0, 588.0};
  for (int shape = 0; shape < 6; shape++)
    for (int w : {1, 2, 3, 4}) {  // 104-131 registers: four resident waves at most (three for the two-chain shape)
      const int iters = (shape == 0 ? 24000 : 14000) / (w > 2 ? (w + 1) / 2 : 1);
      float ms = 0;
      if (shape == 0) ms = time_launch(launch_product<0>, simds * w, iters, d_in, d_out);
      if (shape == 1) ms = time_launch(launch_product<1>, simds * w, iters, d_in, d_out);
      if (shape == 2) ms = time_launch(launch_product<2>, simds * w, iters, d_in, d_out);
      if (shape == 3) ms = time_launch(launch_product<3>, simds * w, iters, d_in, d_out);
      if (shape == 4) ms = time_launch(launch_product<4>, simds * w, iters, d_in, d_out);
      if (shape == 5) ms = time_launch(launch_product<5>, simds * w, iters, d_in, d_out);
      const double rate = (double)simds * w * 64 * iters * macs_per_iter[shape] / (ms * 1e-3) / 1e12;
      printf("{\"shape\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"T_lane_mac_per_s\": %.2f, \"frac_of_peak\": %.3f}\n", names[shape], w, ms, rate, rate / peak);
      fflush(stdout);
    }
  return 0;
}
