// Effective shader clock and per-instruction issue cost of the multiply-add forms on gfx950, so the
// roofline peak in bench.py / DESIGN.md carries its clock (VERDICT r01 "peak re-measured with the clock
// recorded").  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio over a
// busy kernel is the clock the chip actually sustains under that instruction mix.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <algorithm>

#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t* ticks, uint32_t seed, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[NACC];
  int64_t sacc[NACC];
  for (int i = 0; i < NACC; i++) {
    acc[i] = (uint64_t)tid * 0x9e3779b97f4a7c15ull + i + seed;
    sacc[i] = (int64_t)acc[i];
  }
  const uint32_t b = seed | 1u;
  const int32_t sb = (int32_t)(seed | 1u);
  const uint64_t t0 = __builtin_readcyclecounter();   // s_memtime
  const uint64_t r0 = wall_clock64();                 // s_memrealtime (100 MHz)
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kx = 0; kx < NACC; kx++) {
      if (OP == 4) {  // v_mad_i64_i32 as the field multiplier issues it: STABLE multiplicands, chained 64-bit accumulator
        uint64_t cc_;
        asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(sacc[kx & 1]), "=s"(cc_) : "v"(sb + kx), "v"(sb - kx));
      }
      if (OP == 0) acc[kx] = (uint64_t)(uint32_t)acc[(kx + 1) & (NACC - 1)] * b + acc[kx];        // v_mad_u64_u32
      if (OP == 1) sacc[kx] = (int64_t)(int32_t)sacc[(kx + 1) & (NACC - 1)] * sb + sacc[kx];        // v_mad_i64_i32
      if (OP == 2) acc[kx] = acc[kx] + (acc[(kx + 1) & (NACC - 1)] << 3);                          // v_lshl_add_u64
      if (OP == 3) acc[kx] = (uint64_t)((int64_t)acc[kx] >> 28) + kx;                              // v_ashrrev_i64 (+add)
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  uint64_t s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i] + (uint64_t)sacc[i];
  out[tid] = s;
  if ((threadIdx.x & 63) == 0) {
    const uint32_t w = tid >> 6;
    ticks[2 * w] = t1 - t0;
    ticks[2 * w + 1] = r1 - r0;
  }
}

template <int OP>
void run(const char* name, uint64_t* d_out, uint64_t* d_ticks, int blocks, int waves_per_simd, int iters = 8192) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nw = blocks * 4;
  std::vector<uint64_t> h(2 * nw);
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 777u + r, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int w = 0; w < nw; w++) {
    const double cycles = (double)h[2 * w], real = (double)h[2 * w + 1];
    if (real > 0) clk.push_back(cycles / (real / 100e6) / 1e9);
    cyc.push_back(cycles / ((double)iters * NACC));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double ops = (double)blocks * 256 * iters * NACC;
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"iters\": %d, \"kernel_ms\": %.4f, \"lane_ops_per_s\": %.4e, \"effective_clock_GHz_median\": %.3f, "
         "\"shader_cycles_per_instr_per_wave_median\": %.3f, \"cycles_per_instr_per_simd\": %.3f}\n",
         name, waves_per_simd, iters, best, ops / (best * 1e-3), clk[clk.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / waves_per_simd);
}

int main(int argc, char** argv) {
  const bool peak_only = argc > 1 && std::string(argv[1]) == "--peak";  // bench.py: the sustained roofline peak only
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { fprintf(stderr, "no HIP device\n"); return 1; }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("{\"device\": \"%s\", \"cus\": %d, \"max_clock_mhz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  const int cus = p.multiProcessorCount;
  uint64_t *d_out, *d_ticks;
  hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8);
  hipMalloc(&d_ticks, (size_t)cus * 8 * 4 * 16);
  // the roofline denominator: the multiplier's own instruction in the multiplier's own pattern (stable
  // multiplicands, two chained accumulators), every SIMD full (8 waves), ~20 ms per launch so that the clock is
  // the one the chip SUSTAINS under this load.  (A multiplicand that is itself a fresh result -- the pattern of the
  // "v_mad_i64_i32" / "v_mad_u64_u32" rows below and of round 1's peak -- issues ~25 % slower.)
  run<4>("v_mad_i64_i32_chained", d_out, d_ticks, cus * 8, 8, 8192 * 20);
  if (peak_only) return 0;
  run<4>("v_mad_i64_i32_chained", d_out, d_ticks, cus * 2, 2, 8192 * 4);
  run<4>("v_mad_i64_i32_chained", d_out, d_ticks, cus * 1, 1, 8192 * 4);
  run<1>("v_mad_i64_i32", d_out, d_ticks, cus * 8, 8, 8192 * 20);
  run<0>("v_mad_u64_u32", d_out, d_ticks, cus * 8, 8, 8192 * 20);
  for (int wps : {1, 2, 8}) {
    run<0>("v_mad_u64_u32", d_out, d_ticks, cus * wps, wps);
    run<1>("v_mad_i64_i32", d_out, d_ticks, cus * wps, wps);
  }
  run<2>("v_lshl_add_u64", d_out, d_ticks, cus * 2, 2);
  run<3>("v_ashrrev_i64+add", d_out, d_ticks, cus * 2, 2);
  return 0;
}
