// Tower-level lazy reduction, priced before it is built into the kernels.
//
// The shipped Fq6 product (tc_tower.h Fq6::operator*, Karatsuba over Fq2) calls the lane-pair multiplier six times: each call is two
// 14 x 14 limb products and ONE interleaved Montgomery reduction per lane (588 v_mad), 3 528 per Fq6 product.  The lazy form keeps the
// six Karatsuba partial products UNREDUCED -- a lane's coefficient of an Fq2 product as 28 carried limbs (`Wide`) -- forms the three
// output coefficients in that double width (sums, differences, the multiplication by the non-residue as one DPP exchange per limb)
// and reduces each output ONCE: 6 x 392 + 3 x (196 + 28) = 3 024 v_mad (-14 %), at the price of 28-limb linear operations, a
// separate carry pass per partial product and three more 28-register values alive.
//
// This program runs both forms as nothing else -- register-only chains x <- x * y (an Fq6 each), one-wave workgroups, 1 ... 3 waves per
// SIMD -- checks that they agree (canonical values, first iteration) and prints the time per Fq6 product of each.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_tower.hip -o tools/ubench_tower      run: tools/ubench_tower
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../threshold_crypto_amd/csrc/tc_tower.h"

using namespace tc;

constexpr int WL = 2 * FQ_LIMBS;  // 28 limbs of 28 bits: a product before its reduction
struct Wide {
  int32_t l[WL];
};

// the wide routines are INLINED by default: out of line a 28-limb result comes back through memory (sret: seven scratch_store_dwordx4
// in the callee, seven loads in the caller -- measured with -DTC_WIDE_CALLS), which the register-only comparison must not pay
#if defined(TC_WIDE_CALLS)
#define TC_WIDE_ATTR __attribute__((noinline))
#else
#define TC_WIDE_ATTR __forceinline__
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// this lane's coefficient of the Fq2 product a * b, NOT reduced: even lane a0 b0 - a1 b1, odd lane a1 b0 + a0 b1 (the operand set-up of
// fq2p_mul_call), 27 column sums carried into 28 limbs
__device__ TC_WIDE_ATTR Wide fq2p_mulw_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13, int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t b4, int32_t b5, int32_t b6, int32_t b7, int32_t b8, int32_t b9, int32_t b10, int32_t b11, int32_t b12, int32_t b13) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  const int32_t b[FQ_LIMBS] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13};
  int32_t y[FQ_LIMBS], z[FQ_LIMBS], w[FQ_LIMBS];
  const int32_t mneg = pair_even_mask();
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t ao = pair_swap(a[i]);
    y[i] = __builtin_amdgcn_mov_dpp(b[i], 0xA0, 0xF, 0xF, true);
    w[i] = __builtin_amdgcn_mov_dpp(b[i], 0xF5, 0xF, 0xF, true);
    z[i] = (ao ^ mneg) - mneg;
  }
  Wide r;
  int64_t carry = 0;
  constexpr int N = FQ_LIMBS;
  TC_UNROLL for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = (k < N) ? 0 : (k - N + 1);
    const int hi = (k < N) ? k : (N - 1);
    int64_t s = carry;
    TC_UNROLL for (int i = lo; i <= hi; i++) {
      s += (int64_t)a[i] * y[k - i]; TC_PIN(s);
      s += (int64_t)z[i] * w[k - i]; TC_PIN(s);
    }
    r.l[k] = (int32_t)((uint32_t)s & (uint32_t)FQ_MASK);
    carry = s >> FQ_RADIX;
  }
  r.l[2 * N - 1] = (int32_t)carry;
  return r;
}
// Montgomery reduction of a wide value with lazy limbs (|limb| < 2^31): T + m p = 0 mod R, result (T + m p) / R
__device__ TC_WIDE_ATTR FqRaw fq_redcw_call(const Wide& t) {
  constexpr int N = FQ_LIMBS;
  int32_t m[N];
  FqRaw out;
  int64_t carry = 0;
  TC_UNROLL for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = (k < N) ? 0 : (k - N + 1);
    const int hi = (k < N) ? k : (N - 1);
    int64_t s = carry + (int64_t)t.l[k];
    if (k < N) {
      TC_UNROLL for (int i = 0; i < k; i++) { s += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s); }
      m[k] = (int32_t)(((uint32_t)s * FQL_INV) & (uint32_t)FQ_MASK);
      s += (int64_t)m[k] * FQL_P[0];
      carry = s >> FQ_RADIX;
    } else {
      TC_UNROLL for (int i = lo; i <= hi; i++) { s += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s); }
      out.l[k - N] = (int32_t)((uint32_t)s & (uint32_t)FQ_MASK);
      carry = s >> FQ_RADIX;
    }
  }
  out.l[N - 1] = (int32_t)(carry + (int64_t)t.l[2 * N - 1]);
  return out;
}
__device__ __forceinline__ Wide mulw(const Fq2& a, const Fq2& b) {
  return fq2p_mulw_call(a.m.l[0], a.m.l[1], a.m.l[2], a.m.l[3], a.m.l[4], a.m.l[5], a.m.l[6], a.m.l[7], a.m.l[8], a.m.l[9], a.m.l[10], a.m.l[11], a.m.l[12], a.m.l[13], b.m.l[0], b.m.l[1], b.m.l[2], b.m.l[3], b.m.l[4], b.m.l[5], b.m.l[6], b.m.l[7], b.m.l[8], b.m.l[9], b.m.l[10], b.m.l[11], b.m.l[12], b.m.l[13]);
}
__device__ __forceinline__ Wide wadd(const Wide& a, const Wide& b) {
  Wide r;
  TC_UNROLL for (int i = 0; i < WL; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
__device__ __forceinline__ Wide wsub(const Wide& a, const Wide& b) {
  Wide r;
  TC_UNROLL for (int i = 0; i < WL; i++) r.l[i] = a.l[i] - b.l[i];
  return r;
}
// times the non-residue 1 + u on the pair: (c0 - c1 | c0 + c1)
__device__ __forceinline__ Wide wxi(const Wide& a) {
  Wide r;
  const int32_t mneg = pair_even_mask();  // even lane: subtract the partner, odd lane: add it
  TC_UNROLL for (int i = 0; i < WL; i++) {
    const int32_t o = pair_swap(a.l[i]);
    r.l[i] = a.l[i] + ((o ^ mneg) - mneg);
  }
  return r;
}
__device__ __forceinline__ Fq2 redcw(const Wide& t) {
  const FqRaw r = fq_redcw_call(t);
  Fq2 o;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) o.m.l[i] = r.l[i];
  return o;
}
// the lazy Fq6 product: 6 unreduced Karatsuba products, 3 reductions
__device__ __forceinline__ Fq6 fq6_mul_lazy(const Fq6& a, const Fq6& b) {
  const Wide t0 = mulw(a.c0, b.c0);
  const Wide t1 = mulw(a.c1, b.c1);
  const Wide t2 = mulw(a.c2, b.c2);
  Fq6 r;
  r.c0 = redcw(wadd(t0, wxi(wsub(wsub(mulw(a.c1 + a.c2, b.c1 + b.c2), t1), t2))));
  r.c1 = redcw(wadd(wsub(wsub(mulw(a.c0 + a.c1, b.c0 + b.c1), t0), t1), wxi(t2)));
  r.c2 = redcw(wadd(wsub(wsub(mulw(a.c0 + a.c2, b.c0 + b.c2), t0), t2), t1));
  return r;
}
#else
__device__ inline Fq6 fq6_mul_lazy(const Fq6& a, const Fq6&) { return a; }  // (host pass: only has to parse)
#endif

// canonical words of an Fq (both forms are lazy representations: compare values)
__device__ void canon(const Fq& a, uint32_t* w) { a.to_canonical(w); }

template <int LAZY>
__global__ __launch_bounds__(64, 2) void k_tower(const int32_t* in, int32_t* out, int iters, unsigned* mismatches) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  Fq6 x, y;
  Fq2* xs[3] = {&x.c0, &x.c1, &x.c2};
  Fq2* ys[3] = {&y.c0, &y.c1, &y.c2};
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < FQ_LIMBS; i++) {
      xs[c]->m.l[i] = in[((t & 1023) * 6 + c) * FQ_LIMBS + i];
      ys[c]->m.l[i] = in[((t & 1023) * 6 + 3 + c) * FQ_LIMBS + i];
    }
  if (mismatches) {  // agreement of the two forms on this lane's data
    const Fq6 p = x * y, q = fq6_mul_lazy(x, y);
    const Fq2* ps[3] = {&p.c0, &p.c1, &p.c2};
    const Fq2* qs[3] = {&q.c0, &q.c1, &q.c2};
    unsigned bad = 0;
    for (int c = 0; c < 3; c++) {
      uint32_t u[12], v[12];
      canon(ps[c]->m, u);
      canon(qs[c]->m, v);
      for (int i = 0; i < 12; i++) bad |= u[i] ^ v[i];
    }
    if (bad) atomicAdd(mismatches, 1u);
  }
  TC_NOUNROLL for (int it = 0; it < iters; it++) {
    if (LAZY) x = fq6_mul_lazy(x, y);
    else x = x * y;
  }
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < FQ_LIMBS; i++) out[((size_t)t * 3 + c) * FQ_LIMBS + i] = xs[c]->m.norm().l[i];
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int simds = p.multiProcessorCount * 4;
  const size_t in_words = (size_t)1024 * 6 * FQ_LIMBS;
  int32_t* h = (int32_t*)malloc(in_words * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < in_words; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (int32_t)(s & ((1 << FQ_RADIX) - 1));
    if (i % FQ_LIMBS == FQ_LIMBS - 1) h[i] &= 0xffff;  // value below p
  }
  int32_t *d_in, *d_out;
  unsigned* d_bad;
  hipMalloc(&d_in, in_words * 4);
  hipMalloc(&d_out, (size_t)simds * 4 * 64 * 3 * FQ_LIMBS * 4);
  hipMalloc(&d_bad, 4);
  hipMemset(d_bad, 0, 4);
  hipMemcpy(d_in, h, in_words * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tower<0>, dim3(simds), dim3(64), 0, 0, (const int32_t*)d_in, d_out, 1, d_bad);
  unsigned bad = 0;
  hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
  printf("{\"device\": \"%s\", \"simds\": %d, \"lanes_where_the_two_forms_disagree\": %u, \"of\": %d}\n", p.gcnArchName, simds, bad, simds * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 3000;
  for (int w = 1; w <= 2; w++) {
    float ms[2];
    for (int lazy = 0; lazy < 2; lazy++) {
      float best = 1e30f;
      for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        if (lazy) hipLaunchKernelGGL(k_tower<1>, dim3(simds * w), dim3(64), 0, 0, (const int32_t*)d_in, d_out, iters, (unsigned*)nullptr);
        else hipLaunchKernelGGL(k_tower<0>, dim3(simds * w), dim3(64), 0, 0, (const int32_t*)d_in, d_out, iters, (unsigned*)nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (t < best) best = t;
      }
      ms[lazy] = best;
    }
    printf("{\"waves_per_simd\": %d, \"fq6_products_per_lane_pair\": %d, \"shipped_ms\": %.3f, \"lazy_ms\": %.3f, \"lazy_over_shipped\": %.3f, "
           "\"shipped_v_mad_per_lane\": 3528, \"lazy_v_mad_per_lane\": 3024}\n", w, iters, ms[0], ms[1], ms[1] / ms[0]);
    fflush(stdout);
  }
  return 0;
}
