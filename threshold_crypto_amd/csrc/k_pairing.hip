// gfx950 kernels: batched pairing-product check  e(a,b) == e(c,d).
#include "tc_jobs.h"
#include "tc_launch.h"
#include "tc_stage.h"
#include "tc_quad.h"

#include <stdlib.h>

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

// ---- the same check as two kernels --------------------------------------------------------------------------
// The Miller loop and the final exponentiation have disjoint live sets (point arithmetic + sparse products against a
// running Fq12 / cyclotomic chains on three or four Fq12 values): compiled as ONE kernel the register allocator
// serves the union (1 994 spilled registers, 9.6 KB of scratch per lane).  As two kernels each half gets its own
// allocation; the Miller value travels through HBM in the lane-pair row layout: word w of lane l of wave v at
// fbuf[(v * kFq12Words + w) * 64 + l] -- every store / load instruction moves one full 256-byte row.
constexpr int kFq12Words = 6 * FQ_LIMBS;  // per lane: one coefficient of each of the six Fq2
#if TC_PAIR
__device__ __forceinline__ void fq12_store_rows(int32_t* __restrict__ rows, const Fq12& f) {
  const Fq2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  TC_UNROLL for (int k = 0; k < 6; k++) {
    const Fq v = c[k]->m.reduce_value();
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) rows[(k * FQ_LIMBS + i) * 64] = v.l[i];
  }
}
__device__ __forceinline__ Fq12 fq12_load_rows(const int32_t* __restrict__ rows) {
  Fq12 f;
  Fq2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  TC_UNROLL for (int k = 0; k < 6; k++) {
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) c[k]->m.l[i] = rows[(k * FQ_LIMBS + i) * 64];
  }
  return f;
}
#else
__device__ __forceinline__ void fq12_store_rows(int32_t*, const Fq12&) {}
__device__ __forceinline__ Fq12 fq12_load_rows(const int32_t*) { return Fq12::one(); }
#endif

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_miller_loop(const uint8_t* __restrict__ a, size_t sa,
                                                        const uint8_t* __restrict__ b, size_t sb,
                                                        const uint8_t* __restrict__ c, size_t sc,
                                                        const uint8_t* __restrict__ d, size_t sd, size_t B,
                                                        int32_t* __restrict__ fbuf, uint8_t* __restrict__ ok) {
  using IO1 = WaveRowIO<96, kG2Lanes>;
  using IO2 = WaveRowIO<192, kG2Lanes>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO2::BYTES];
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  IO1 ia{lds, live ? a + jj * sa : nullptr, 0, nullptr}, ic{lds, live ? c + jj * sc : nullptr, 0, nullptr};
  IO2 ib{lds, live ? b + jj * sb : nullptr, 0, nullptr}, id{lds, live ? d + jj * sd : nullptr, 0, nullptr};
  Fq12 f = Fq12::one();
  const bool good = job_miller_io(live, ia, ib, ic, id, f);
  fq12_store_rows(fbuf + (size_t)blockIdx.x * kFq12Words * 64 + threadIdx.x, f);
  if (live && pair_leader()) ok[j] = good ? 1 : 0;  // 0: an operand did not decode; the second kernel keeps it
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_final_exp(const int32_t* __restrict__ fbuf, size_t B, uint8_t* __restrict__ ok) {
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const Fq12 f = fq12_load_rows(fbuf + (size_t)blockIdx.x * kFq12Words * 64 + threadIdx.x);
  const uint8_t r = job_final_exp_is_one(f);
  if (j < B && pair_leader() && ok[j]) ok[j] = r;
}

// ---- the Miller loop as two kernels: prepared line products in HBM (tc_pairing.h) ---------------------------------------
// k_miller_lines (stage P: the two Miller points, their lines, the product of the two lines of a step) writes 68 x 5 Fq2
// coefficients per check into the wave's row block -- word w of lane l of wave v at lines[(v * kLineWords + w) * 64 + l],
// 256-byte rows, 39 KB per check --, k_miller_accumulate (stage M: the Fq12 accumulator) streams them back and leaves the
// Miller value for k_final_exp.
constexpr int kLineWords = kMillerRowSlots * FQ_LIMBS;  // per lane: operands, one parked line, 68 x 5 product coefficients
#ifndef TC_WAVES_MILLER_P
#define TC_WAVES_MILLER_P TC_WAVES_G2
#endif
#ifndef TC_WAVES_MILLER_M
#define TC_WAVES_MILLER_M TC_WAVES_G2
#endif
#if TC_PAIR
__global__ __launch_bounds__(kBlock, TC_WAVES_MILLER_P) void k_miller_lines(const uint8_t* __restrict__ a, size_t sa,
                                                         const uint8_t* __restrict__ b, size_t sb,
                                                         const uint8_t* __restrict__ c, size_t sc,
                                                         const uint8_t* __restrict__ d, size_t sd, size_t B,
                                                         int32_t* lines, uint8_t* __restrict__ ok) {  // (lines: read back through laundered pointers, tc_pairing.h rows_after: no restrict)
  using IO1 = WaveRowIO<96, kG2Lanes>;
  using IO2 = WaveRowIO<192, kG2Lanes>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO2::BYTES];
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  IO1 ia{lds, live ? a + jj * sa : nullptr, 0, nullptr}, ic{lds, live ? c + jj * sc : nullptr, 0, nullptr};
  IO2 ib{lds, live ? b + jj * sb : nullptr, 0, nullptr}, id{lds, live ? d + jj * sd : nullptr, 0, nullptr};
  const bool good = job_miller_lines_io(live, ia, ib, ic, id, Fq2Rows::at(lines + (size_t)blockIdx.x * kLineWords * 64 + threadIdx.x));
  if (live && pair_leader()) ok[j] = good ? 1 : 0;  // 0: an operand did not decode; the later kernels keep it
}

__global__ __launch_bounds__(kBlock, TC_WAVES_MILLER_M) void k_miller_accumulate(const int32_t* lines, int32_t* __restrict__ fbuf) {
  const Fq12 f = miller_accumulate(Fq2Rows::at(const_cast<int32_t*>(lines) + (size_t)blockIdx.x * kLineWords * 64 + threadIdx.x));
  fq12_store_rows(fbuf + (size_t)blockIdx.x * kFq12Words * 64 + threadIdx.x, f);
}
#else
__global__ void k_miller_lines(const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, size_t, size_t, int32_t*, uint8_t*) {}
__global__ void k_miller_accumulate(const int32_t*, int32_t*) {}
#endif

// ---- four lanes per check (tc_quad.h) -----------------------------------------------------------------------------------
// Pair A of a quad stages and decodes (a, b), pair B (c, d): two passes through the LDS row buffer with a PAIR as the
// staging unit (32 rows per wave), each pair parses one G1 and one G2 record.
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_pairing_quad(const uint8_t* __restrict__ a, size_t sa,
                                                         const uint8_t* __restrict__ b, size_t sb,
                                                         const uint8_t* __restrict__ c, size_t sc,
                                                         const uint8_t* __restrict__ d, size_t sd, size_t B,
                                                         uint8_t* __restrict__ ok) {
  using IO1 = WaveRowIO<96, kG2Lanes>;
  using IO2 = WaveRowIO<192, kG2Lanes>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[IO2::BYTES];
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kQuadLanes;
  const bool live = j < B;
  const size_t jj = live ? j : 0;
  const bool hi = quad_hi();
  IO1 i1{lds, live ? (hi ? c + jj * sc : a + jj * sa) : nullptr, 0, nullptr};
  IO2 i2{lds, live ? (hi ? d + jj * sd : b + jj * sb) : nullptr, 0, nullptr};
  const uint8_t r = job_pairing_check_quad_io(live, i1, i2);
  if (live && (threadIdx.x & (kQuadLanes - 1)) == 0) ok[j] = r;
}

// Which form runs (Tuning::pairing_form, from TC_PAIRING_FORM = quad | lines | pair | fused when the context was created, overrides
// the choice for experiments):
//   four lanes per check (k_pairing_quad) up to kQuadMaxBatch checks -- the batch alone cannot give every SIMD two waves of
//     the lane-pair kernels, and a check finishes in about 0.6 of the time (6.9 instead of 10.9 ms at 4 096 checks, 8.5
//     instead of 11.4 ms at 16 384: profiles/r03_pairing_forms.txt);
//   above it the PREPARED form: k_miller_lines -> k_miller_accumulate -> k_final_exp (r04; 206 / 142 / 4 spilled registers,
//     12.99 ms for the two Miller kernels against 13.1-13.2 ms for the one loop with its 2 238: profiles/r04_pairing_*);
//   `pair` = r03's k_miller_loop + k_final_exp, `fused` = r02's single kernel.
constexpr size_t kQuadMaxBatch = 16384;
enum PairingForm { kFormQuad, kFormLines, kFormPair, kFormFused };
int pairing_form(size_t B, const Tuning& tn) {
  if (tn.pairing_form >= 1 && tn.pairing_form <= 4) return tn.pairing_form - 1;
  return B <= kQuadMaxBatch ? kFormQuad : kFormLines;
}
bool pairing_form_needs_lines(int form) { return form == kFormLines; }
// checks per pass of the prepared form: its line buffer (kLineWords = 552 x 14 words per lane: 61.8 KB per check, 4.05 GB for
// 65 536 checks) is sized for one tile, larger batches run tile by tile.  The tile shrinks to what the caller's budget holds
// (a third of the free HBM: several contexts on one GPU, a smaller card), in whole rounds of 2 048 waves down to 16 384
// checks; below that the one-loop form (k_miller_loop + k_final_exp, no line buffer) runs instead of failing an allocation.
constexpr size_t kPreparedTile = 65536;
constexpr size_t kPreparedMinTile = 16384;
size_t pairing_tile(size_t B, size_t budget_bytes) {
  const size_t per_check = (size_t)kG2Lanes * kLineWords * sizeof(int32_t);
  size_t tile = kPreparedTile;
  while (tile > kPreparedMinTile && tile * per_check > budget_bytes) tile /= 2;
  if (tile * per_check > budget_bytes) return 0;
  return B < tile ? B : tile;
}
size_t pairing_ws_words(size_t B, size_t tile) {
  return (size_t)grid_for(B * kG2Lanes) * kFq12Words * 64 + (tile ? (size_t)grid_for(tile * kG2Lanes) * kLineWords * 64 : 0);
}

void launch_pairing_check(hipStream_t st, const uint8_t* a, size_t sa, const uint8_t* b, size_t sb, const uint8_t* c,
                          size_t sc, const uint8_t* d, size_t sd, size_t B, uint8_t* ok, PairingWs pws) {
  if (!B) return;
  int32_t* ws = pws.p;
  int form = pws.form;
  if (form == kFormLines && !pws.tile) form = kFormPair;  // no room for the line buffer
  if (form == kFormQuad) {
    hipLaunchKernelGGL(k_pairing_quad, dim3(grid_for(B * kQuadLanes)), dim3(kBlock), 0, st, a, sa, b, sb, c, sc, d, sd, B, ok);
    return;
  }
  if (form == kFormFused || !ws) {
    hipLaunchKernelGGL(k_pairing_check, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, a, sa, b, sb, c, sc, d, sd, B, ok);
    return;
  }
  if (form == kFormPair) {
    hipLaunchKernelGGL(k_miller_loop, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, a, sa, b, sb, c, sc, d, sd, B, ws, ok);
    hipLaunchKernelGGL(k_final_exp, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, (const int32_t*)ws, B, ok);
    return;
  }
  // prepared form: lines (stage P) -> accumulator (stage M) tile by tile through one line buffer, then ONE final exponentiation launch
  int32_t* lines = ws + (size_t)grid_for(B * kG2Lanes) * kFq12Words * 64;
  for (size_t lo = 0; lo < B; lo += pws.tile) {
    const size_t cnt = (B - lo < pws.tile) ? B - lo : pws.tile;
    const unsigned grid = grid_for(cnt * kG2Lanes);
    int32_t* fb = ws + (size_t)grid_for(lo * kG2Lanes) * kFq12Words * 64;
    hipLaunchKernelGGL(k_miller_lines, dim3(grid), dim3(kBlock), 0, st, a + lo * sa, sa, b + lo * sb, sb, c + lo * sc, sc, d + lo * sd, sd, cnt,
                       lines, ok + lo);
    hipLaunchKernelGGL(k_miller_accumulate, dim3(grid), dim3(kBlock), 0, st, (const int32_t*)lines, fb);
  }
  hipLaunchKernelGGL(k_final_exp, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, (const int32_t*)ws, B, ok);
}

}  // namespace tc
