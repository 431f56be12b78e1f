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
// Fq12 sparse multiplication / squaring used by the Miller loop: -DTC_INLINE_FQ12 inlines them.
#if defined(TC_INLINE_FQ12)
#define TC_FQ12_ATTR TC_HD
#else
#define TC_FQ12_ATTR TC_HD_NOINLINE
#endif
// Miller-loop doubling / addition steps: same switch (-DTC_INLINE_MILLER).
#if defined(TC_INLINE_MILLER)
#define TC_MILLER_ATTR TC_HD
#else
#define TC_MILLER_ATTR TC_HD_NOINLINE
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define TC_UNROLL _Pragma("unroll")
#define TC_NOUNROLL _Pragma("nounroll")
#else
#define TC_UNROLL
#define TC_NOUNROLL
#endif

#include "tc_constants.h"
