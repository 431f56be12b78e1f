// What one VALU instruction of the field arithmetic costs a gfx950 SIMD, and how full the pipe is at 1 / 2 / 4 / 8 waves per SIMD.
//
// The roofline denominator of bench.py is the issue rate of v_mad_i64_i32 -- the multiplier's own instruction -- with every
// SIMD saturated.  Round 3 printed three incompatible figures for it: 37.5 T lane-ops/s (tools/ubench_chain, wall),
// 31.4 T (tools/ubench_clock, wall) and "3.087 cycles per instruction per SIMD" (ubench_clock, from s_memtime) = 48 T.  This
// benchmark measures BOTH views in one launch and says which is which:
//   * wall rate      lane-operations / kernel time (HIP events), the figure a kernel's achieved rate can be divided by;
//   * issue interval per wave: shader cycles (s_memtime) between the first and the last instruction of a wave / instructions,
//                    median AND maximum over all waves of the launch;
//   * clock          s_memtime / s_memrealtime (100 MHz) over the same interval: what the chip sustains under this mix.
// cycles per wave-instruction per SIMD (wall) = 1024 SIMDs x clock / (wave-instructions per second).  The per-wave MEDIAN
// interval divided by the waves per SIMD underestimates it whenever the waves of a launch do not all run for the whole launch
// (ubench_clock's 256-lane workgroups: the dispatcher does not spread 2048 of them evenly, the launch lasts as long as the
// fullest CU): that, and nothing about the pipe, is the "3.087".  One-wave workgroups, >= 20 ms per launch, loop overhead < 2 %.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue        run: tools/ubench_issue [--peak]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <string>
#include <vector>

enum { OP_MAD = 0, OP_ADD, OP_AND, OP_DPP, OP_ASHR64, OP_MULLO, OP_LSHLADD64, OP_CNDMASK, OP_COUNT };
static const char* kNames[OP_COUNT] = {"v_mad_i64_i32 (chained accumulator, stable multiplicands)", "v_add_u32", "v_and_b32",
                                       "v_mov_b32_dpp quad_perm", "v_ashrrev_i64", "v_mul_lo_u32", "v_lshl_add_u64", "v_cndmask_b32"};
constexpr int kPerIter = 64;  // instructions per loop iteration (4 chains x 16)

template <int OP>
__global__ __launch_bounds__(64) void k(uint64_t* out, uint64_t* ticks, uint32_t seed, int iters) {
  const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
  int64_t a64[4];
  uint32_t x[4], y[4];
  for (int i = 0; i < 4; i++) {
    a64[i] = (int64_t)tid * 0x9e3779b97f4a7c15ll + i + seed;
    x[i] = tid * 2654435761u + i + seed;
    y[i] = (seed | 1u) + 77u * i;
  }
  const int32_t m0 = (int32_t)(seed | 1u), m1 = (int32_t)(seed * 3u | 1u);
  const uint64_t sel = 0x5555aaaa3333ccccull ^ seed;  // a wave-uniform lane mask for the select
  const uint64_t t0 = __builtin_readcyclecounter();  // s_memtime
  const uint64_t r0 = wall_clock64();                // s_memrealtime, 100 MHz
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < kPerIter / 4; u++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        uint64_t cc;
        if (OP == OP_MAD) asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(a64[c & 1]), "=s"(cc) : "v"(m0), "v"(m1));
        if (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y[c]));
        if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y[c]));
        if (OP == OP_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[c]) : "v"(y[c]));
        if (OP == OP_ASHR64) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(a64[c]));
        if (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y[c]));
        if (OP == OP_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a64[c]) : "v"(a64[(c + 1) & 3]));
        if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y[c]), "s"(sel));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  uint64_t s = 0;
  for (int i = 0; i < 4; i++) s += (uint64_t)a64[i] + x[i];
  out[tid] = s;
  if (threadIdx.x == 0) {
    ticks[2 * blockIdx.x] = t1 - t0;
    ticks[2 * blockIdx.x + 1] = r1 - r0;
  }
}

struct Row {
  double ms, lane_ops_per_s, clock_ghz, cyc_wall, cyc_wave_median, cyc_wave_max;
};

template <int OP>
Row run(uint64_t* d_out, uint64_t* d_ticks, int cus, int waves_per_simd, int iters) {
  const int blocks = cus * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d_out, d_ticks, 777u + r, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<uint64_t> h(2 * (size_t)blocks);
  hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int w = 0; w < blocks; w++) {
    const double cycles = (double)h[2 * w], real = (double)h[2 * w + 1];
    if (real > 0) clk.push_back(cycles / (real / 100e6) / 1e9);
    cyc.push_back(cycles / ((double)iters * kPerIter));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  Row r;
  r.ms = best;
  const double wave_instr = (double)blocks * iters * kPerIter;
  r.lane_ops_per_s = wave_instr * 64 / (best * 1e-3);
  r.clock_ghz = clk[clk.size() / 2];
  r.cyc_wall = (double)cus * 4 * r.clock_ghz * 1e9 / (wave_instr / (best * 1e-3));
  r.cyc_wave_median = cyc[cyc.size() / 2];
  r.cyc_wave_max = cyc.back();
  return r;
}

template <int OP>
void report(uint64_t* d_out, uint64_t* d_ticks, int cus, int waves, int iters) {
  const Row r = run<OP>(d_out, d_ticks, cus, waves, iters);
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"T_lane_ops_per_s\": %.2f, \"clock_GHz\": %.3f, "
         "\"cycles_per_wave_instr_per_simd_wall\": %.3f, \"issue_interval_per_wave_cycles_median\": %.2f, \"issue_interval_per_wave_cycles_max\": %.2f, "
         "\"median_interval_div_waves\": %.3f}\n",
         kNames[OP], waves, r.ms, r.lane_ops_per_s / 1e12, r.clock_ghz, r.cyc_wall, r.cyc_wave_median, r.cyc_wave_max, r.cyc_wave_median / waves);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const bool peak_only = argc > 1 && std::string(argv[1]) == "--peak";
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  uint64_t *d_out, *d_ticks;
  hipMalloc(&d_out, (size_t)cus * 4 * 8 * 64 * 8);
  hipMalloc(&d_ticks, (size_t)cus * 4 * 8 * 16);
  // the roofline denominator: v_mad_i64_i32, 4 waves per SIMD, ~40 ms (sustained clock)
  report<OP_MAD>(d_out, d_ticks, cus, 4, 80000);
  if (peak_only) return 0;
  printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"max_clock_mhz\": %d}\n", p.gcnArchName, cus, cus * 4, p.clockRate / 1000);
  for (int w : {1, 2, 8}) report<OP_MAD>(d_out, d_ticks, cus, w, w == 8 ? 40000 : (w == 2 ? 80000 : 40000));
  for (int w : {1, 2, 4, 8}) {
    const int it = 40000 / w;
    report<OP_ADD>(d_out, d_ticks, cus, w, it);
    report<OP_AND>(d_out, d_ticks, cus, w, it);
    report<OP_DPP>(d_out, d_ticks, cus, w, it);
    report<OP_CNDMASK>(d_out, d_ticks, cus, w, it);
    report<OP_ASHR64>(d_out, d_ticks, cus, w, it);
    report<OP_MULLO>(d_out, d_ticks, cus, w, it);
    report<OP_LSHLADD64>(d_out, d_ticks, cus, w, it);
  }
  return 0;
}
