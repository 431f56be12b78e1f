// Common macros for the gfx950 device library.
//
// Every arithmetic routine in csrc/*.h is written once, as TC_HD (host+device) code: hipcc
// compiles it for gfx950 (the product), and tests/hostsim compiles the very same headers
// with g++ so the logic can be differential-tested against the oracle in the GPU-less
// build container.  The host compilation is a TEST HARNESS only: libtc_amd.so (the product)
// contains no CPU compute path and fails loudly without a HIP device.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TC_HD __host__ __device__ __forceinline__
#define TC_HD_NOINLINE __host__ __device__ __attribute__((noinline)) inline
#define TC_D __device__ __forceinline__
#define TC_CONST static constexpr
#else
#define TC_HD inline
#define TC_HD_NOINLINE __attribute__((noinline)) inline
#define TC_D inline
#define TC_CONST static constexpr
#endif

// Jacobian point operations are inlined into the scalar-multiplication loops so the accumulator
// stays in registers across iterations (measured on MI355X: -11 % kernel time vs. real
// functions with by-reference operands; -DTC_NOINLINE_JAC restores those for experiments).
#if defined(TC_NOINLINE_JAC)
#define TC_JAC_ATTR TC_HD_NOINLINE
#else
#define TC_JAC_ATTR TC_HD
#endif
// Fq6 routines are inlined into the Fq12 routines (one level less of by-reference operand
// traffic, no intermediate value reductions; measured on MI355X: pairing check -14 %).
// -DTC_NOINLINE_FQ6 restores real functions for experiments.
#if defined(TC_NOINLINE_FQ6)
#define TC_FQ6_ATTR TC_HD_NOINLINE
#define TC_FQ6_OUT(x) (x).reduce_value()
#else
#define TC_FQ6_ATTR TC_HD
#define TC_FQ6_OUT(x) (x).norm()
#endif
// Fq12 multiplication / sparse multiplication / squaring / cyclotomic squaring and the Miller-loop steps are inlined
// into their loops, so the 12 (per lane: 6) coefficient registers of the accumulator stay in
// VGPRs across iterations instead of travelling through scratch by reference (measured on
// MI355X, lane-pair build: pairing check 36.8 -> 29.9 ms; the one-lane-per-job build, whose
// Fq12 needs 180 registers, was faster with real functions).  -DTC_NOINLINE_FQ12 /
// -DTC_NOINLINE_MILLER / -DTC_NOINLINE_CYCLO restore real functions for experiments.
#if defined(TC_NOINLINE_FQ12)
#define TC_FQ12_ATTR TC_HD_NOINLINE
#else
#define TC_FQ12_ATTR TC_HD
#endif
#if defined(TC_NOINLINE_MILLER)
#define TC_MILLER_ATTR TC_HD_NOINLINE
#else
#define TC_MILLER_ATTR TC_HD
#endif
#if defined(TC_NOINLINE_CYCLO)
#define TC_CYCLO_ATTR TC_HD_NOINLINE
#else
#define TC_CYCLO_ATTR TC_HD
#endif
#if defined(TC_NOINLINE_FQ12MUL)
#define TC_FQ12MUL_ATTR TC_HD_NOINLINE
#else
#define TC_FQ12MUL_ATTR TC_HD
#endif
#if defined(TC_INLINE_EXPX)
#define TC_EXPX_ATTR TC_HD
#else
#define TC_EXPX_ATTR TC_HD_NOINLINE
#endif

// ---- Fq2 across a lane pair ------------------------------------------------------------------
// In the hipcc build every job that touches Fq2 (all G2 arithmetic, the pairing, hash_g2) is
// worked on by TWO adjacent lanes: the even lane holds the c0 coefficient of every Fq2 value,
// the odd lane c1 (tc_tower.h).  Linear Fq2 operations are component-wise, so each lane does
// half of them; a product costs each lane two 14x14 limb products and ONE Montgomery reduction
// (tc_field.h fq2p_mul_call) with the partner's limbs fetched over DPP.  Per lane that halves
// registers, scratch and latency; a batch of B jobs fills 2B lanes, so batch 65 536 gives the
// MI355X two waves per SIMD instead of one.  Both lanes of a pair always take the same
// branches (every predicate is reduced over the pair).  G1-only kernels keep one lane per job.
// The g++ build of the same headers (tests/hostsim, a test harness) keeps both coefficients
// in one object and runs the identical coefficient formulas one after the other.
#if defined(__HIPCC__) && !defined(TC_NO_PAIR)
#define TC_PAIR 1
#else
#define TC_PAIR 0
#endif

namespace tc {
#if TC_PAIR
constexpr int kG2Lanes = 2;  // lanes per job in kernels that work on Fq2 values
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int pair_odd() { return (int)(threadIdx.x & 1u); }
// value held by the partner lane (quad_perm [1,0,3,2]: a full-rate VALU move)
#if defined(TC_PAIR_SWIZZLE)
__device__ __forceinline__ int32_t pair_swap(int32_t v) { return __builtin_amdgcn_ds_swizzle(v, 0x80B1); }
#else
__device__ __forceinline__ int32_t pair_swap(int32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
#endif
#else  // host pass of hipcc: never executed, only has to parse
inline int pair_odd() { return 0; }
inline int32_t pair_swap(int32_t v) { return v; }
#endif
// predicates reduced over the pair (the swap executes unconditionally on both lanes)
TC_HD bool pair_all(bool c) {
  const int32_t mine = c ? 1 : 0;
  const int32_t theirs = pair_swap(mine);
  return (mine & theirs) != 0;
}
TC_HD bool pair_any(bool c) {
  const int32_t mine = c ? 1 : 0;
  const int32_t theirs = pair_swap(mine);
  return (mine | theirs) != 0;
}
#else
constexpr int kG2Lanes = 1;
TC_HD int pair_odd() { return 0; }
TC_HD bool pair_all(bool c) { return c; }
TC_HD bool pair_any(bool c) { return c; }
#endif
// the lane that writes per-job scalars (status bytes, booleans, G1 outputs) in a G2 kernel
TC_HD bool pair_leader() { return pair_odd() == 0; }

// Data-dependent loops are written wave-uniform:  while (wave_any(!done)) { if (!done) {...} }.
// A loop whose lanes leave at different iterations costs the same on SIMD hardware (the wave
// runs until its slowest lane is done), and it keeps every value that must survive the loop
// out of the hands of the register allocator while its lane is masked off: measured on
// gfx950 / ROCm 7.2, a lane-divergent retry loop inlined into the hash_g1_g2 kernel returned
// corrupted live-out values for lanes that had left early (profiles/r01_d_pair_notes.md).
TC_HD bool wave_any(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(c) != 0;
#else
  return c;
#endif
}
// A value every lane of the wave holds identically (a constant handed to an out-of-line function
// arrives in a VGPR): read it from the first lane so loops steered by it compile to scalar
// branches instead of exec-masked ones.
TC_HD uint64_t wave_uniform(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (uint64_t)lo | ((uint64_t)hi << 32);
#else
  return v;
#endif
}
// Time-sliced issue priority.  A SIMD arbitrates its resident waves oldest-first: of the two waves these
// kernels keep per SIMD the older one runs at the speed of a lone wave and the younger one gets the leftover
// issue slots (about half speed: profiles/r02_phase_timing.txt), so the older wave of every SIMD finishes early
// and the kernel's last third runs ONE wave per SIMD.  Alternating s_setprio between the two waves in slices of
// 2^17 shader cycles -- (clock slice + hardware wave slot) parity -- makes them advance together, and both
// finish at 4/3 of a lone wave's time instead of 1 and 3/2.  Called once per iteration of the long loops.
TC_HD void tc_fair() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TC_NO_FAIR)
  const uint32_t slice = (uint32_t)(__builtin_readcyclecounter() >> 17);
  const uint32_t slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));  // HW_REG_HW_ID, WAVE_ID = bits [3:0]
  if ((slice + slot) & 1u) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}
}  // namespace tc

#if defined(__HIP_DEVICE_COMPILE__)
#define TC_UNROLL _Pragma("unroll")
#define TC_NOUNROLL _Pragma("nounroll")
#else
#define TC_UNROLL
#define TC_NOUNROLL
#endif

// -DTC_PHASE_TIMING (experiment builds only, tools/phase_marks_probe.py): lane 0 of every workgroup stamps the shader
// clock at phase boundaries of the combine job into a per-translation-unit device array
#if defined(TC_PHASE_TIMING) && defined(__HIPCC__)
static __device__ unsigned long long tc_phase_marks[4096 * 16];
#if defined(__HIP_DEVICE_COMPILE__)
#define TC_MARK(k)                                                                                        \
  do {                                                                                                    \
    if (threadIdx.x == 0) tc_phase_marks[(blockIdx.x & 4095u) * 16 + (k)] = __builtin_readcyclecounter(); \
  } while (0)
// wall clock (s_memrealtime, 100 MHz, the same counter on every XCD): slots 8.. hold absolute start / end times
#define TC_MARK_WALL(k)                                                                       \
  do {                                                                                        \
    if (threadIdx.x == 0) tc_phase_marks[(blockIdx.x & 4095u) * 16 + (k)] = wall_clock64();  \
  } while (0)
#else
#define TC_MARK(k) ((void)0)
#define TC_MARK_WALL(k) ((void)0)
#endif
#else
#define TC_MARK(k) ((void)0)
#define TC_MARK_WALL(k) ((void)0)
#endif

#include "tc_constants.h"
