// Prime-field arithmetic for gfx950.
//
//   Fq (BLS12-381 base field, 381 bit) -- the hot field: REDUCED RADIX, 14 signed limbs of
//      28 bits in 32-bit VGPRs, Montgomery form with R = 2^392.  On gfx950 a VALU carry-out
//      feeding a VALU carry-in costs two wait states (the compiler pads every v_add_co ->
//      v_addc pair), so saturated 32-bit limbs make every add/sub/multiply-accumulate a serial,
//      padded carry chain.  With 28-bit limbs nothing carries:
//        * a + b, a - b, -a, 2a are 14 independent v_add/v_sub (lazy, no reduction);
//        * a product column is a chain of v_mad_i64_i32 into one 64-bit accumulator
//          (14 * 2^56 * 9.14 = 2^63), the carry into the next column is one 64-bit shift;
//        * reduction is interleaved per column (product scanning), quotient digit
//          m_k = column * (-p^-1) mod 2^28.
//      14 limbs are the fewest that hold 381 bits with any headroom: a limb product is 196
//      multiply-adds (15 x 26-bit limbs, the previous layout: 225; measured -9..11 % kernel time).
//      The price is a tight budget: lazy limbs must stay below 8 * 2^28 (int32) and a
//      multiplication needs B_a * B_b <= 8 (sum of the two products of an Fq2 coefficient <= 8),
//      so formulas apply small multiples AFTER products and pass sums through `norm()` (one
//      parallel carry pass) where needed.  Every such decision is checked, not trusted: the
//      interval analysis of tests/hostsim -DTC_BOUND_CHECK walks every job body.
//   Fr (scalar field, 255 bit) -- off the hot path (Lagrange coefficients only): plain
//      saturated 8 x u32 Montgomery (R = 2^256), compiler-scheduled.
//
// Device replacement for pairing::bls12_381::Fq and ff's derived Fr, which the reference
// reaches through the engine seam at /root/reference/src/lib.rs:60-67.  No MFMA: this is
// wide-integer modular arithmetic, not a dense contraction.
#pragma once
#include "tc_common.h"

#if defined(TC_BOUND_CHECK)
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
#endif

namespace tc {

#if defined(TC_BOUND_CHECK)
// host-only (tests/hostsim built with -DTC_BOUND_CHECK -O0): report the call chain of a
// multiplication whose operand bounds could overflow a column accumulator
inline void tc_bound_fail(float a, float b) {
  void* bt[24];
  int n = backtrace(bt, 24);
  fprintf(stderr, "TC_BOUND_CHECK: operand bounds %.2f x %.2f exceed the multiplier's limit\n", a, b);
  backtrace_symbols_fd(bt, n, 2);
  abort();
}
#endif

// =======================================================================================
// generic saturated Montgomery over N x u32 (used for Fr; FqParams kept for range checks)
// =======================================================================================
struct FqParams {
  static constexpr int N = 12;
  TC_HD static uint32_t p(int i) { return FQ_P[i]; }
};

struct FrParams {
  static constexpr int N = 8;
  TC_HD static uint32_t p(int i) { return FR_P[i]; }
  TC_HD static uint32_t one(int i) { return FR_ONE[i]; }
  TC_HD static uint32_t r2(int i) { return FR_R2[i]; }
  TC_HD static uint32_t pm2(int i) { return FR_P_MINUS_2[i]; }
  static constexpr uint32_t inv = FR_INV32;
  static constexpr int bits = 255;
};

template <class PR>
struct Mont {
  static constexpr int N = PR::N;
  uint32_t l[N];
};

// canonical limbs < p ?
template <class PR>
TC_HD bool limbs_lt_p(const uint32_t* limbs) {
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < PR::N; i++) {
    uint64_t d = (uint64_t)limbs[i] - PR::p(i) - borrow;
    borrow = (uint32_t)(d >> 63);
  }
  return borrow != 0;
}

template <class PR>
TC_HD void mont_cond_sub_p(Mont<PR>& a) {
  constexpr int N = PR::N;
  uint32_t t[N];
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - PR::p(i) - borrow;
    t[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  TC_UNROLL for (int i = 0; i < N; i++) a.l[i] = borrow ? a.l[i] : t[i];
}

template <class PR>
TC_HD Mont<PR> mont_add(const Mont<PR>& a, const Mont<PR>& b) {
  constexpr int N = PR::N;
  Mont<PR> r;
  uint32_t c = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
    r.l[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  mont_cond_sub_p(r);  // the modulus leaves the top bit clear: no carry out of limb N-1
  return r;
}

template <class PR>
TC_HD Mont<PR> mont_sub(const Mont<PR>& a, const Mont<PR>& b) {
  constexpr int N = PR::N;
  Mont<PR> r;
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
    r.l[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  uint32_t mask = 0u - borrow;
  uint32_t c = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)r.l[i] + (PR::p(i) & mask) + c;
    r.l[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  return r;
}

// CIOS Montgomery product (top limb of the modulus < 2^31: no (N+2)-th limb needed)
template <class PR>
TC_HD_NOINLINE Mont<PR> mont_mul(const Mont<PR>& a, const Mont<PR>& b) {
  constexpr int N = PR::N;
  uint32_t t[N + 1];
  TC_UNROLL for (int i = 0; i <= N; i++) t[i] = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
    TC_UNROLL for (int j = 0; j < N; j++) {
      uint64_t s = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    uint32_t tn = t[N] + (uint32_t)c;
    const uint32_t m = t[0] * PR::inv;
    uint64_t s = (uint64_t)m * PR::p(0) + t[0];
    c = s >> 32;
    TC_UNROLL for (int j = 1; j < N; j++) {
      s = (uint64_t)m * PR::p(j) + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = s >> 32;
    }
    s = (uint64_t)tn + c;
    t[N - 1] = (uint32_t)s;
    t[N] = (uint32_t)(s >> 32);
  }
  Mont<PR> r;
  TC_UNROLL for (int i = 0; i < N; i++) r.l[i] = t[i];
  mont_cond_sub_p(r);
  return r;
}

// a^e with e given by a word accessor; fixed 4-bit windows, uniform control flow
template <class F, class EXP>
TC_HD F field_pow_fixed(const F& a, EXP e, int nbits) {
  F tbl[16];
  tbl[0] = F::one();
  tbl[1] = a;
  TC_NOUNROLL for (int i = 2; i < 16; i++) tbl[i] = (i & 1) ? tbl[i - 1] * a : tbl[i >> 1].sqr();
  int top = ((nbits + 3) / 4) * 4 - 4;
  F r = tbl[(e(top >> 5) >> (top & 31)) & 15];
  TC_NOUNROLL for (int pos = top - 4; pos >= 0; pos -= 4) {
    tc_fair();
    r = r.sqr();
    r = r.sqr();
    r = r.sqr();
    r = r.sqr();
    uint32_t w = (e(pos >> 5) >> (pos & 31)) & 15;
    if (w) r = r * tbl[w];
  }
  return r;
}

struct Fr {
  Mont<FrParams> v;
  static constexpr int N = 8;
  TC_HD static Fr zero() {
    Fr r;
    TC_UNROLL for (int i = 0; i < N; i++) r.v.l[i] = 0;
    return r;
  }
  TC_HD static Fr one() {
    Fr r;
    TC_UNROLL for (int i = 0; i < N; i++) r.v.l[i] = FR_ONE[i];
    return r;
  }
  TC_HD bool is_zero() const {
    uint32_t o = 0;
    TC_UNROLL for (int i = 0; i < N; i++) o |= v.l[i];
    return o == 0;
  }
  TC_HD bool operator==(const Fr& b) const {
    uint32_t o = 0;
    TC_UNROLL for (int i = 0; i < N; i++) o |= (v.l[i] ^ b.v.l[i]);
    return o == 0;
  }
  TC_HD bool operator!=(const Fr& b) const { return !(*this == b); }
  TC_HD Fr operator+(const Fr& b) const { return Fr{mont_add(v, b.v)}; }
  TC_HD Fr operator-(const Fr& b) const { return Fr{mont_sub(v, b.v)}; }
  TC_HD Fr operator*(const Fr& b) const { return Fr{mont_mul(v, b.v)}; }
  TC_HD Fr sqr() const { return Fr{mont_mul(v, v)}; }
  TC_HD static Fr from_canonical(const uint32_t* limbs) {
    Mont<FrParams> a, r2;
    TC_UNROLL for (int i = 0; i < N; i++) {
      a.l[i] = limbs[i];
      r2.l[i] = FR_R2[i];
    }
    return Fr{mont_mul(a, r2)};
  }
  TC_HD void to_canonical(uint32_t* limbs) const {
    Mont<FrParams> o;
    TC_UNROLL for (int i = 0; i < N; i++) o.l[i] = (i == 0) ? 1u : 0u;
    Mont<FrParams> r = mont_mul(v, o);
    TC_UNROLL for (int i = 0; i < N; i++) limbs[i] = r.l[i];
  }
  // Fermat inverse a^(r-2); 0 -> 0
  TC_HD_NOINLINE Fr inv() const {
    return field_pow_fixed(*this, [](int i) { return FR_P_MINUS_2[i]; }, 255);
  }
};

// Fr from u64 (IntoFr for u64, /root/reference/src/into_fr.rs:16-20)
TC_HD Fr fr_from_u64(uint64_t x) {
  uint32_t l[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0};
  return Fr::from_canonical(l);
}

// =======================================================================================
// Fq: 14 x 28-bit signed limbs, R = 2^392
// =======================================================================================
constexpr int FQ_LIMBS = FQ_LIMBS_GEN;
constexpr int FQ_RADIX = FQ_RADIX_GEN;
constexpr int32_t FQ_MASK = (1 << FQ_RADIX) - 1;
constexpr float FQ_MAX_BOUND_PRODUCT = 8.14f;  // B_a * B_b allowed at a multiplication: 14 * 2^56 * (8 + 1) < 2^63

#if defined(TC_COUNT_OPS)
// host-only (tests/hostsim): multiplications / squarings executed, for the "ours M/unit"
// column of DESIGN.md and bench.py's executed-MAC roofline.  In the lane-pair build an Fq
// operation INSIDE an Fq2 method is one lane's half of that method (counted once: "split"); an
// Fq operation outside (inversions, square-root exponentiations, G1 work in a G2 kernel) is
// executed by both lanes ("local", counted twice for a G2 kernel by tests/count_ops.py).
inline uint64_t g_tc_mul_count = 0, g_tc_sqr_count = 0, g_tc_mul2_count = 0;
inline uint64_t g_tc_split_mul_count = 0, g_tc_split_sqr_count = 0;
inline int g_tc_split_depth = 0;
struct TcSplitScope {
  TcSplitScope() { g_tc_split_depth++; }
  ~TcSplitScope() { g_tc_split_depth--; }
};
#define TC_SPLIT_SCOPE TcSplitScope tc_split_scope_
#else
#define TC_SPLIT_SCOPE
#endif

struct Fq;
TC_HD Fq fq_mul(const Fq& a, const Fq& b);
TC_HD Fq fq_sqr(const Fq& a);

struct Fq {
  int32_t l[FQ_LIMBS];
#if defined(TC_BOUND_CHECK)
  // host-only static-analysis aid: every limb lies in [blo, bhi] * 2^26.  The interval is data
  // independent (it follows the operation sequence), so any test that walks a code path
  // proves that path never overflows a column accumulator.
  // bval: |value| <= bval * p.  Lazy sums also grow the VALUE; only a multiplication (or
  // reduce_value()) brings it back to ~p: a product lands in (-V_a V_b p / 512, p + V_a V_b p / 512).
  float blo, bhi, bval;
  TC_HD void set_range(float lo, float hi) {
    blo = lo;
    bhi = hi;
    // the limbs themselves are int32: |l_i| < 2^31 = 32 * 2^26 (keep a margin for norm()'s carry-in)
    if (lo < -7.9f || hi > 7.9f) tc_bound_fail(lo, hi);
  }
  TC_HD void set_val(float v) { bval = v; }
  TC_HD float lo() const { return blo; }
  TC_HD float hi() const { return bhi; }
  TC_HD float val() const { return bval; }
  TC_HD float bound() const { return (-blo > bhi) ? -blo : bhi; }
#else
  TC_HD void set_range(float, float) {}
  TC_HD void set_val(float) {}
  TC_HD float lo() const { return 0.f; }
  TC_HD float hi() const { return 0.f; }
  TC_HD float val() const { return 0.f; }
  TC_HD float bound() const { return 0.f; }
#endif

  TC_HD static Fq from_limbs(const int32_t* c) {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = c[i];
    r.set_range(0.f, 1.f);
    r.set_val(1.f);
    return r;
  }
  TC_HD static Fq zero() {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = 0;
    r.set_range(0.f, 0.f);
    r.set_val(0.f);
    return r;
  }
  TC_HD static Fq one() { return from_limbs(FQL_ONE); }

  // ---- lazy linear operations: no carries, no reduction --------------------------------
  TC_HD Fq operator+(const Fq& b) const {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = l[i] + b.l[i];
    r.set_range(lo() + b.lo(), hi() + b.hi());
    r.set_val(val() + b.val());
    return r;
  }
  TC_HD Fq operator-(const Fq& b) const {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = l[i] - b.l[i];
    r.set_range(lo() - b.hi(), hi() - b.lo());
    r.set_val(val() + b.val());
    return r;
  }
  TC_HD Fq operator-() const {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = -l[i];
    r.set_range(-hi(), -lo());
    r.set_val(val());
    return r;
  }
  TC_HD Fq dbl() const {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = l[i] * 2;
    r.set_range(2.f * lo(), 2.f * hi());
    r.set_val(2.f * val());
    return r;
  }
  // one parallel carry pass: limbs back to [0, 2^26) + a carry of a few units; the top limb
  // keeps the sign.  The value is unchanged.
  TC_HD Fq norm() const {
    Fq r;
    r.l[0] = l[0] & FQ_MASK;
    TC_UNROLL for (int i = 1; i < FQ_LIMBS - 1; i++) r.l[i] = (l[i] & FQ_MASK) + (l[i - 1] >> FQ_RADIX);
    r.l[FQ_LIMBS - 1] = l[FQ_LIMBS - 1] + (l[FQ_LIMBS - 2] >> FQ_RADIX);
    r.set_range(-0.001f, 1.001f);
    r.set_val(val());
#if defined(TC_BOUND_CHECK)
    if (val() > 300.f) tc_bound_fail(val(), -1.f);  // the top limb (~ V * 2^17.7) must stay below 2^26 too
#endif
    return r;
  }
  // Brings the VALUE back to about [0, 2p] by subtracting k*p, k estimated from the top limb
  // (any integer k keeps the residue), and carries the limbs.  Lazy sums grow the value as
  // well as the limbs; products shrink it again (|a b| / R), but towers of Karatsuba sums
  // (Fq2 -> Fq6 -> Fq12) and linear feedback (cyclotomic squaring) can outrun that, so the
  // big tower/point routines reduce their outputs with this instead of norm().
  // Requires |value| <= 300 p.  ~105 VALU instructions, no multiplier-sized work.
  TC_HD Fq reduce_value() const {
    Fq n = norm();
    const int32_t k = (int32_t)__builtin_floorf((float)n.l[FQ_LIMBS - 1] * (1.0f / (float)FQL_P[FQ_LIMBS - 1]));
    int64_t t[FQ_LIMBS];
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) t[i] = (int64_t)n.l[i] - (int64_t)k * FQL_P[i];
    Fq r;
    r.l[0] = (int32_t)((uint32_t)t[0] & (uint32_t)FQ_MASK);
    TC_UNROLL for (int i = 1; i < FQ_LIMBS - 1; i++)
      r.l[i] = (int32_t)((uint32_t)t[i] & (uint32_t)FQ_MASK) + (int32_t)(t[i - 1] >> FQ_RADIX);
    r.l[FQ_LIMBS - 1] = (int32_t)(t[FQ_LIMBS - 1] + (t[FQ_LIMBS - 2] >> FQ_RADIX));
    r.set_range(-0.001f, 1.001f);
    r.set_val(2.1f);
    return r;
  }
  TC_HD Fq operator*(const Fq& b) const { return fq_mul(*this, b); }
  TC_HD Fq sqr() const { return fq_sqr(*this); }
  TC_HD static Fq select(bool c, const Fq& a, const Fq& b) {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = c ? a.l[i] : b.l[i];
    r.set_range(a.lo() < b.lo() ? a.lo() : b.lo(), a.hi() > b.hi() ? a.hi() : b.hi());
    r.set_val(a.val() > b.val() ? a.val() : b.val());
    return r;
  }

  // select on a predicate that is LANE IDENTITY, not data (which pair of a quad this is: tc_quad.h): the same instruction,
  // but the interval bookkeeping of the bound-check build follows the operand this lane really takes
  TC_HD static Fq select_lane(bool c, const Fq& a, const Fq& b) {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = c ? a.l[i] : b.l[i];
    const Fq& pick = c ? a : b;
    r.set_range(pick.lo(), pick.hi());
    r.set_val(pick.val());
    return r;
  }

  // Zero test in two steps.  value = k p with |k| <= 300 (the value bound every operand
  // obeys) forces the low 26 bits of the value -- l[0] & mask, whatever the lazy upper limbs
  // hold -- to be k p mod 2^26, i.e. (l[0] * p^-1) mod 2^26 must land within 300 of zero.
  // Four VALU instructions reject all but ~2^-16 of the non-zero inputs; only survivors pay for
  // the full reduction.  (Point additions test three coordinates per call.)
  TC_HD bool maybe_zero() const {
#if defined(TC_BOUND_CHECK)
    if (val() > 300.f) tc_bound_fail(val(), -2.f);
#endif
    const uint32_t t = ((uint32_t)l[0] * FQL_INV) & (uint32_t)FQ_MASK;  // = -k mod 2^26
    return ((t + 300u) & (uint32_t)FQ_MASK) <= 600u;
  }
  // the same test on the low TWO limbs (56 bits of k p): no call needed afterwards, false positives ~2^-46
  TC_HD bool maybe_zero56() const {
    const uint64_t mask = (1ull << 56) - 1;
    const uint64_t v = (uint64_t)(int64_t)l[0] + ((uint64_t)(int64_t)l[1] << FQ_RADIX);  // (mod 2^64: no shift of a negative value)
    const uint64_t t = (v * FQ_INV56) & mask;
    return ((t + 300u) & mask) <= 600u;
  }
  TC_HD_NOINLINE bool is_zero_full() const;
  TC_HD bool is_zero() const { return maybe_zero() && is_zero_full(); }
  TC_HD bool operator==(const Fq& b) const { return (*this - b).is_zero(); }
  TC_HD bool operator!=(const Fq& b) const { return !(*this == b); }
  TC_HD_NOINLINE Fq inv() const;  // 0 -> 0
  TC_HD static Fq from_canonical(const uint32_t* words12);
  TC_HD void to_canonical(uint32_t* words12) const;
  TC_HD static Fq from_mont384(const uint32_t* words12);
};

// ---- the multiplier ----------------------------------------------------------------------
// Product-scanning Montgomery multiplication.  Column k gathers the carry from column k-1, sum a_i b_{k-i} and
// sum m_i p_{k-i} in one v_mad chain (see TC_PIN).
//   T = a*b + m*p,  T = 0 mod R,  result = T / R  in (-p/4, 5p/4)
// TC_PIN(sum) gives a partial column sum a second (ephemeral) use, so LLVM's reassociation pass treats it as a leaf
// and keeps the accumulation order written here -- carry first.  Left alone it sorts the carry (the deepest value) to
// the END of the column: the products then start from zero and the carry costs one v_lshl_add_u64 per column.
// The assumed fact is unconditionally true (squares are 0, 1 or 4 mod 8, also after wrap-around) and no code is
// generated for it.
#if defined(__HIP_DEVICE_COMPILE__)
#define TC_PIN(v) __builtin_assume((uint64_t)(v) * (uint64_t)(v) != 2u)
#else
#define TC_PIN(v)
#endif
template <bool SQUARE>
TC_HD void fq_mul_body(const int32_t* a, const int32_t* b, int32_t* out) {
  constexpr int N = FQ_LIMBS;
  int32_t m[N];
  int32_t a2[N];
  if (SQUARE) {
    TC_UNROLL for (int i = 0; i < N; i++) a2[i] = a[i] * 2;
  }
  int64_t carry = 0;
  TC_UNROLL for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = (k < N) ? 0 : (k - N + 1);
    const int hi = (k < N) ? k : (N - 1);
    int64_t s1 = carry;
    if (SQUARE) {
      // pairs i < j with i + j = k, doubled, plus the diagonal term
      TC_UNROLL for (int i = lo; i <= hi; i++) {
        const int j = k - i;
        if (i < j) { s1 += (int64_t)a[i] * a2[j]; TC_PIN(s1); }
        if (i == j) { s1 += (int64_t)a[i] * a[i]; TC_PIN(s1); }
      }
    } else {
      TC_UNROLL for (int i = lo; i <= hi; i++) { s1 += (int64_t)a[i] * b[k - i]; TC_PIN(s1); }
    }
    if (k < N) {
      TC_UNROLL for (int i = 0; i < k; i++) { s1 += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s1); }
      int64_t s = s1;
      m[k] = (int32_t)(((uint32_t)s * FQL_INV) & (uint32_t)FQ_MASK);
      s += (int64_t)m[k] * FQL_P[0];
      carry = s >> FQ_RADIX;
    } else {
      TC_UNROLL for (int i = lo; i <= hi; i++) { s1 += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s1); }
      int64_t s = s1;
      out[k - N] = (int32_t)((uint32_t)s & (uint32_t)FQ_MASK);
      carry = s >> FQ_RADIX;
    }
  }
  out[N - 1] = (int32_t)carry;
}

// Two products, one reduction:  out = (x*y + z*w + m*p) / R.  This is what one lane of an
// Fq2 lane pair computes for its own coefficient of a product (c0 = a0 b0 - a1 b1 or
// c1 = a0 b1 + a1 b0): 588 multiply-adds instead of the 784 two separate Montgomery
// multiplications would take.
// Column bound: 14 * 2^56 * (Bx By + Bz Bw + 1) < 2^63  <=>  Bx By + Bz Bw < 8.14.
constexpr float FQ_MAX_BOUND_PRODUCT2 = 8.14f;
TC_HD void fq_mul2_body(const int32_t* x, const int32_t* y, const int32_t* z, const int32_t* w, int32_t* out) {
  constexpr int N = FQ_LIMBS;
  int32_t m[N];
  int64_t carry = 0;
  TC_UNROLL for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = (k < N) ? 0 : (k - N + 1);
    const int hi = (k < N) ? k : (N - 1);
    // ONE accumulator chain per column, started from the previous column's carry: every term becomes a
    // v_mad_i64_i32 / v_mad_u64_u32 whose addend is the running sum, and no separate 64-bit add is left (TC_PIN above;
    // saves 26 v_lshl_add_u64 per multiplication).  Chained multiply-adds issue back to back
    // (tools/ubench_chain: 1 chain = 4 chains), two waves per SIMD fill the multiplier.
    int64_t s1 = carry;
    TC_UNROLL for (int i = lo; i <= hi; i++) {
      s1 += (int64_t)x[i] * y[k - i]; TC_PIN(s1);
      s1 += (int64_t)z[i] * w[k - i]; TC_PIN(s1);
    }
    if (k < N) {
      TC_UNROLL for (int i = 0; i < k; i++) { s1 += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s1); }
      int64_t s = s1;
      m[k] = (int32_t)(((uint32_t)s * FQL_INV) & (uint32_t)FQ_MASK);
      s += (int64_t)m[k] * FQL_P[0];
      carry = s >> FQ_RADIX;
    } else {
      TC_UNROLL for (int i = lo; i <= hi; i++) { s1 += (int64_t)m[i] * FQL_P[k - i]; TC_PIN(s1); }
      int64_t s = s1;
      out[k - N] = (int32_t)((uint32_t)s & (uint32_t)FQ_MASK);
      carry = s >> FQ_RADIX;
    }
  }
  out[N - 1] = (int32_t)carry;
}

#if defined(__HIP_DEVICE_COMPILE__)
// On the device the multiplier is a REAL function, not inlined into every tower/curve routine
// (that makes the pairing kernel's code object several MB and hipcc compile times unbounded).
// Operands travel as scalar i32 arguments so the AMDGPU calling convention keeps all 30 of
// them in VGPRs (aggregates beyond 16 registers would go through scratch memory); the 15-limb
// result comes back in VGPRs too.
// (-DTC_INLINE_MUL, experiments only: the lane-pair multiplier inlined into its callers -- no argument moves, operand
// broadcasts shared between products; measured on k_msm_ladder: 6 % fewer instructions, 0.7 % faster, 4x the code)
#if defined(TC_INLINE_MUL)
#define TC_MULCALL_ATTR __forceinline__
#else
#define TC_MULCALL_ATTR __attribute__((noinline)) inline
#endif
struct FqRaw {
  int32_t l[FQ_LIMBS];
};
__device__ __attribute__((noinline)) inline FqRaw fq_mul_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13, int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t b4, int32_t b5, int32_t b6, int32_t b7, int32_t b8, int32_t b9, int32_t b10, int32_t b11, int32_t b12, int32_t b13) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  const int32_t b[FQ_LIMBS] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13};
  FqRaw r;
  fq_mul_body<false>(a, b, r.l);
  return r;
}
__device__ __attribute__((noinline)) inline FqRaw fq_sqr_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  FqRaw r;
  fq_mul_body<true>(a, a, r.l);
  return r;
}

#if TC_PAIR
// ---- Fq2 spread over a lane pair (tc_common.h) ---------------------------------------------
// A product needs the partner's coefficients: they come over DPP inside the callee, so a call
// still passes only its own 14 + 14 limbs in VGPRs.
// The lane parity is NOT an argument (it was until r04): a value every product of a kernel needs is live from the first
// instruction to the last, the 256-register kernels kept it in SCRATCH, and the register allocator reloaded it straight into the
// argument register before every call -- one exposed scratch round trip per product (36 of the 46 products of a Miller step in
// k_miller_accumulate waited on exactly that load).  Four instructions in the callee instead: the lane id from two
// v_mbcnt (a workgroup is a whole number of waves, so its parity is threadIdx.x's), its low bit, minus one.
__device__ __forceinline__ int32_t pair_even_mask() {  // even lane of a pair: -1, odd lane: 0
  return (int32_t)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 1u) - 1;
}
//   even lane: c0 = a0 b0 - a1 b1 = mine*b0 + (-other)*b1
//   odd  lane: c1 = a1 b0 + a0 b1 = mine*b0 +   other *b1
// with b0 / b1 broadcast to both lanes of the pair (two DPP moves per limb) and the partner's a negated on the even
// lane by xor/subtract with a lane mask (the xor carries the DPP swap): 56 prologue instructions, no branch.
__device__ TC_MULCALL_ATTR FqRaw fq2p_mul_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13, int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t b4, int32_t b5, int32_t b6, int32_t b7, int32_t b8, int32_t b9, int32_t b10, int32_t b11, int32_t b12, int32_t b13) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  const int32_t b[FQ_LIMBS] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13};
  int32_t y[FQ_LIMBS], z[FQ_LIMBS], w[FQ_LIMBS];
  const int32_t mneg = pair_even_mask();  // even lane: -1 (negate the partner's coefficient), odd lane: 0
#if defined(TC_MUL_SWIZZLE)
  // experiment (r04): the operand exchange through the LDS crossbar (ds_swizzle, no memory access) instead of DPP moves: a
  // v_mov_b32_dpp occupies the VALU as long as a multiply-add (4.2 cycles, tools/ubench_issue), a swizzle issues on the LDS pipe
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t ao = __builtin_amdgcn_ds_swizzle(a[i], 0x80B1);
    y[i] = __builtin_amdgcn_ds_swizzle(b[i], 0x80A0);
    w[i] = __builtin_amdgcn_ds_swizzle(b[i], 0x80F5);
    z[i] = (ao ^ mneg) - mneg;
  }
#else
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t ao = pair_swap(a[i]);
    y[i] = __builtin_amdgcn_mov_dpp(b[i], 0xA0, 0xF, 0xF, true);  // b0 on both lanes
    w[i] = __builtin_amdgcn_mov_dpp(b[i], 0xF5, 0xF, 0xF, true);  // b1 on both lanes
    z[i] = (ao ^ mneg) - mneg;
  }
#endif
  FqRaw r;
  fq_mul2_body(a, y, z, w, r.l);
  return r;
}
//   even lane: c0 = (a0 + a1)(a0 - a1);   odd lane: c1 = (2 a1) a0
__device__ TC_MULCALL_ATTR FqRaw fq2p_sqr_call(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13) {
  const int32_t a[FQ_LIMBS] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13};
  int32_t x[FQ_LIMBS], y[FQ_LIMBS];
  int32_t even = pair_even_mask();  // even lane: all ones
  asm("" : "+v"(even));   // opaque: keeps the AND (one v_and_b32_dpp) from becoming a compare and a select
#if defined(TC_MUL_SWIZZLE)
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t a1 = __builtin_amdgcn_ds_swizzle(a[i], 0x80F5);
    const int32_t a0 = __builtin_amdgcn_ds_swizzle(a[i], 0x80A0);
    x[i] = a[i] + a1;
    y[i] = a0 - (a1 & even);                  // even lane: a0 - a1; odd lane: a0
  }
#else
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t a1 = __builtin_amdgcn_mov_dpp(a[i], 0xF5, 0xF, 0xF, true);  // the odd lane's coefficient, on both lanes
    const int32_t a0 = __builtin_amdgcn_mov_dpp(a[i], 0xA0, 0xF, 0xF, true);  // the even lane's
    x[i] = a[i] + a1;                         // a0 + a1 | 2 a1
    y[i] = a0 - (pair_swap(a[i]) & even);     // a0 - a1 | a0
  }
#endif
  FqRaw r;
  fq_mul_body<false>(x, y, r.l);
  return r;
}
// the squaring INLINED (r04): for the one loop that is short enough to stay in the instruction cache with six of them in its body
// -- the compressed cyclotomic squaring chain of the final exponentiation (tc_tower.h CycloCompressed::sqr_t): no argument /
// result moves, no call, and the six independent squarings of a step interleave
__device__ __forceinline__ FqRaw fq2p_sqr_inl(const int32_t* a, int32_t odd) {
  int32_t x[FQ_LIMBS], y[FQ_LIMBS];
  int32_t even = odd - 1;
  asm("" : "+v"(even));
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int32_t a1 = __builtin_amdgcn_mov_dpp(a[i], 0xF5, 0xF, 0xF, true);
    const int32_t a0 = __builtin_amdgcn_mov_dpp(a[i], 0xA0, 0xF, 0xF, true);
    x[i] = a[i] + a1;
    y[i] = a0 - (pair_swap(a[i]) & even);
  }
  FqRaw r;
  fq_mul_body<false>(x, y, r.l);
  return r;
}
#endif  // TC_PAIR
#endif

TC_HD Fq fq_mul(const Fq& a, const Fq& b) {
#if defined(TC_BOUND_CHECK)
  if (a.bound() * b.bound() > (float)FQ_MAX_BOUND_PRODUCT) tc_bound_fail(a.bound(), b.bound());
  if (a.val() > 300.f || b.val() > 300.f) tc_bound_fail(-a.val(), -b.val());  // top limb < 2^26
#endif
  Fq r;
#if defined(__HIP_DEVICE_COMPILE__)
  FqRaw t = fq_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11], b.l[12], b.l[13]);
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = t.l[i];
#else
  fq_mul_body<false>(a.l, b.l, r.l);
#if defined(TC_COUNT_OPS)
  g_tc_mul_count++;
  if (g_tc_split_depth) g_tc_split_mul_count++;
#endif
#endif
  r.set_range(0.f, 1.f);
  r.set_val(1.f + a.val() * b.val() / 2048.f);  // |a b| / R + p, p / R < 2^-9
  return r;
}

TC_HD Fq fq_sqr(const Fq& a) {
#if defined(TC_BOUND_CHECK)
  if (a.bound() * a.bound() > (float)FQ_MAX_BOUND_PRODUCT) tc_bound_fail(a.bound(), a.bound());
  if (a.val() > 300.f) tc_bound_fail(-a.val(), -a.val());
#endif
  Fq r;
#if defined(__HIP_DEVICE_COMPILE__)
  FqRaw t = fq_sqr_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13]);
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = t.l[i];
#else
  fq_mul_body<true>(a.l, a.l, r.l);
#if defined(TC_COUNT_OPS)
  g_tc_sqr_count++;
  if (g_tc_split_depth) g_tc_split_sqr_count++;
#endif
#endif
  r.set_range(0.f, 1.f);
  r.set_val(1.f + a.val() * a.val() / 2048.f);
  return r;
}

// (x*y + z*w) / R with ONE reduction: the coefficient formula of an Fq2 product.  The hipcc build
// reaches it through fq2p_mul_call (one coefficient per lane of a pair); this wrapper is the
// g++ (test-harness) form, with the static bound bookkeeping.
TC_HD Fq fq_mul2(const Fq& x, const Fq& y, const Fq& z, const Fq& w) {
#if defined(TC_BOUND_CHECK)
  if (x.bound() * y.bound() + z.bound() * w.bound() > (float)FQ_MAX_BOUND_PRODUCT2) {
    fprintf(stderr, "fq_mul2 operand limb intervals: x [%.3f, %.3f] y [%.3f, %.3f] z [%.3f, %.3f] w [%.3f, %.3f]\n", x.lo(), x.hi(),
            y.lo(), y.hi(), z.lo(), z.hi(), w.lo(), w.hi());
    tc_bound_fail(x.bound() * y.bound(), z.bound() * w.bound());
  }
  if (x.val() > 300.f || y.val() > 300.f || z.val() > 300.f || w.val() > 300.f) tc_bound_fail(-x.val(), -y.val());
#endif
  Fq r;
  fq_mul2_body(x.l, y.l, z.l, w.l, r.l);
#if defined(TC_COUNT_OPS)
  g_tc_mul2_count++;
#endif
  r.set_range(0.f, 1.f);
  r.set_val(1.f + (x.val() * y.val() + z.val() * w.val()) / 2048.f);
  return r;
}

// a / R mod p as the UNIQUE representative in [0, p]: limbs fully carried, all in [0, 2^26).
// (T = a + m p with m in [0, R): T / R > -1 and <= p for |a| < R.)
TC_HD void fq_redc_full(const Fq& a, int32_t* out) {
  int32_t one[FQ_LIMBS];
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) one[i] = (i == 0) ? 1 : 0;
  Fq an = a.norm();  // any lazily grown input is fine: bring limbs back below 2^27 first
  // the column loop masks limbs 0..13 and carries into the next column, so the output is
  // already fully carried: digits in [0, 2^26), top limb >= 0
  fq_mul_body<false>(an.l, one, out);
}

TC_HD_NOINLINE bool Fq::is_zero_full() const {
  int32_t t[FQ_LIMBS];
  fq_redc_full(*this, t);
  int32_t z = 0, e = 0;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    z |= t[i];
    e |= (t[i] ^ FQL_P[i]);
  }
  return z == 0 || e == 0;
}

// Fermat inverse a^(p-2): ~381 squarings + ~110 products.  Kept for the cross-check in
// tests/hostsim; the library inverts with Fq::inv() below.
TC_HD_NOINLINE Fq fq_inv_fermat(const Fq& a) {
  return field_pow_fixed(a.norm(), [](int i) { return FQ_P_MINUS_2[i]; }, 381);
}

TC_HD void words12_to_limbs(const uint32_t* w, int32_t* l);

// y^-1 mod p (0 -> 0): Pornin's optimised binary GCD (eprint 2020/972).  Plain binary GCD keeps
// a = y u,  b = y v  (mod p)  while halving a / replacing it by (a - b) / 2; after at most
// 2 * 381 - 1 steps a = 0, b = 1 and v = 1 / y.  Here 30 steps at a time run on 64-bit
// APPROXIMATIONS of a and b (their exact low 31 bits and the top 33 bits of the longer one) and
// only produce four small factors with |f0| + |g0|, |f1| + |g1| <= 2^30; the long numbers are then
// updated in one go,
//     (a, b) <- ((a f0 + b g0) / 2^30, (a f1 + b g1) / 2^30)       exact; a wrong comparison in
//                                                                  the approximation shows up as
//                                                                  a negative value and is undone
//     (u, v) <- the same combination, divided by 2^30 modulo p     (one Montgomery-style step)
// At most 26 rounds of 30 steps; the loop ends when every lane of the wave has reached a = 0.
// The long numbers are 13 limbs of 30 bits (low limbs in [0, 2^30), the top limb carries the sign): a limb times a
// factor is ONE v_mad_i64_i32, the division by 2^30 drops a limb instead of shifting twelve words, and u, v stay in
// (-2p, p) between rounds instead of being made canonical (the signed-limb form of libsecp256k1's modinv32, applied
// to Pornin's loop) -- 1.5x fewer instructions per round than the 12 x u32 form this replaces.
constexpr int FQ_INV_LIMBS = 13;
constexpr int32_t FQ_INV_MASK = (1 << 30) - 1;
// bits [30 i, 30 i + 30) of p; p^-1 mod 2^30
TC_HD constexpr int32_t fq_p30(int i) {
  const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
  uint64_t v = (uint64_t)FQ_P[wi] >> sh;
  if (wi + 1 < 12) v |= (uint64_t)FQ_P[wi + 1] << (32 - sh);
  return (int32_t)((uint32_t)v & (uint32_t)FQ_INV_MASK);
}
constexpr uint32_t FQ_P_INV30 = (0u - FQ_INV32) & (uint32_t)FQ_INV_MASK;  // FQ_INV32 = -p^-1 mod 2^32

// a limb as an int32 the compiler knows nothing about: known to be non-negative, its signed 32 x 32 -> 64 multiply-add
// (one v_mad_i64_i32) would be rewritten as an unsigned one plus a correction
TC_HD int32_t fq_inv_limb(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}
// (x f + y g) / 2^30 for both factor pairs, exact (the low 30 bits of the sums are zero): x, y >= 0 in, results with a
// signed top limb out
TC_HD void fq_inv_update_ab(const int32_t* x, const int32_t* y, int32_t f0, int32_t g0, int32_t f1, int32_t g1, int32_t* o0,
                            int32_t* o1) {
  int64_t c0 = 0, c1 = 0;
  TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) {
    const int32_t xi = fq_inv_limb(x[i]), yi = fq_inv_limb(y[i]);
    c0 += (int64_t)xi * f0;
    c0 += (int64_t)yi * g0;
    c1 += (int64_t)xi * f1;
    c1 += (int64_t)yi * g1;
    if (i == 0) {
      c0 >>= 30;
      c1 >>= 30;
      continue;
    }
    o0[i - 1] = (int32_t)c0 & FQ_INV_MASK;
    o1[i - 1] = (int32_t)c1 & FQ_INV_MASK;
    c0 >>= 30;
    c1 >>= 30;
  }
  o0[FQ_INV_LIMBS - 1] = (int32_t)c0;
  o1[FQ_INV_LIMBS - 1] = (int32_t)c1;
}
// x <- |x|; true if x was negative
TC_HD bool fq_inv_abs(int32_t* x) {
  const bool neg = x[FQ_INV_LIMBS - 1] < 0;
  const int32_t m = neg ? FQ_INV_MASK : 0;
  int32_t carry = neg ? 1 : 0;
  TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS - 1; i++) {
    const int32_t t = (x[i] ^ m) + carry;
    x[i] = t & FQ_INV_MASK;
    carry = (int32_t)((uint32_t)t >> 30);
  }
  x[FQ_INV_LIMBS - 1] = (x[FQ_INV_LIMBS - 1] ^ (neg ? -1 : 0)) + carry;
  return neg;
}
// (x f + y g) / 2^30 mod p for x, y in (-2p, p): a multiple of p is added so that the low 30 bits vanish and the
// result is in (-2p, p) again
TC_HD void fq_inv_update_uv(const int32_t* x, const int32_t* y, int32_t f, int32_t g, int32_t* o) {
  const int32_t sx = x[FQ_INV_LIMBS - 1] >> 31, sy = y[FQ_INV_LIMBS - 1] >> 31;
  int32_t md = (sx & f) + (sy & g);  // a negative operand counts as operand + p
  int64_t c = (int64_t)fq_inv_limb(x[0]) * f;
  c += (int64_t)fq_inv_limb(y[0]) * g;
  md -= (int32_t)((FQ_P_INV30 * (uint32_t)c + (uint32_t)md) & (uint32_t)FQ_INV_MASK);
  c = (c + (int64_t)fq_p30(0) * md) >> 30;
  TC_UNROLL for (int i = 1; i < FQ_INV_LIMBS; i++) {
    c += (int64_t)fq_inv_limb(x[i]) * f;
    c += (int64_t)fq_inv_limb(y[i]) * g;
    c += (int64_t)fq_p30(i) * md;
    o[i - 1] = (int32_t)c & FQ_INV_MASK;
    c >>= 30;
  }
  o[FQ_INV_LIMBS - 1] = (int32_t)c;
}

// y: 13 limbs of a canonical value; out: 1 / y (or 0) in (-2p, p), signed top limb
TC_HD void fq_inv_limbs30(const int32_t* y, int32_t* out) {
  int32_t a[FQ_INV_LIMBS], b[FQ_INV_LIMBS], u[FQ_INV_LIMBS], v[FQ_INV_LIMBS];
  TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) {
    a[i] = y[i];
    b[i] = fq_p30(i);
    u[i] = (i == 0) ? 1 : 0;
    v[i] = 0;
  }
  TC_NOUNROLL for (int round = 0; round < 26; round++) {
    if ((round & 7) == 0) tc_fair();
    // ---- approximations: the three limbs below the highest non-zero limb of a | b ---------------
    uint32_t hw = 0;
    int top = 0;
    TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) {
      const uint32_t w = (uint32_t)(a[i] | b[i]);
      top = w ? i : top;
      hw = w ? w : hw;
    }
    uint32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0;
    TC_UNROLL for (int i = 2; i < FQ_INV_LIMBS; i++) {
      const bool here = top == i;
      ah = here ? (uint32_t)a[i] : ah;
      am = here ? (uint32_t)a[i - 1] : am;
      al = here ? (uint32_t)a[i - 2] : al;
      bh = here ? (uint32_t)b[i] : bh;
      bm = here ? (uint32_t)b[i - 1] : bm;
      bl = here ? (uint32_t)b[i - 2] : bl;
    }
    const int len = hw ? 32 - __builtin_clz(hw) : 0;        // bits of the top limb
    const bool big = 30 * top + len > 64;                   // else a and b are exact in 64 bits
    const uint64_t alo = (uint64_t)(uint32_t)a[0] | ((uint64_t)(uint32_t)a[1] << 30) | ((uint64_t)(uint32_t)a[2] << 60);
    const uint64_t blo = (uint64_t)(uint32_t)b[0] | ((uint64_t)(uint32_t)b[1] << 30) | ((uint64_t)(uint32_t)b[2] << 60);
    // top 33 bits of the longer one's length: (three limbs >> 27) >> len
    const uint64_t ahi = (((((uint64_t)ah << 30) | am) << 3) | (al >> 27)) >> len;
    const uint64_t bhi = (((((uint64_t)bh << 30) | bm) << 3) | (bl >> 27)) >> len;
    uint64_t abar = big ? ((alo & 0x7fffffffull) | (ahi << 31)) : alo;
    uint64_t bbar = big ? ((blo & 0x7fffffffull) | (bhi << 31)) : blo;
    // ---- 30 steps on the approximations ------------------------------------------------------
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    TC_UNROLL for (int i = 0; i < 30; i++) {
      const bool odd = (abar & 1ull) != 0;
      const bool swp = odd && (abar < bbar);
      const uint64_t ta = swp ? bbar : abar, tb = swp ? abar : bbar;
      const int32_t nf0 = swp ? f1 : f0, ng0 = swp ? g1 : g0, nf1 = swp ? f0 : f1, ng1 = swp ? g0 : g1;
      abar = (ta - (odd ? tb : 0)) >> 1;
      bbar = tb;
      f0 = nf0 - (odd ? nf1 : 0);
      g0 = ng0 - (odd ? ng1 : 0);
      f1 = nf1 * 2;  // (not `<< 1`: the factors are signed)
      g1 = ng1 * 2;
    }
    // ---- (a, b) <- exact combinations / 2^30, made non-negative -------------------------------
    int32_t na[FQ_INV_LIMBS], nb[FQ_INV_LIMBS];
    fq_inv_update_ab(a, b, f0, g0, f1, g1, na, nb);
    const bool sa = fq_inv_abs(na), sb = fq_inv_abs(nb);
    f0 = sa ? -f0 : f0;
    g0 = sa ? -g0 : g0;
    f1 = sb ? -f1 : f1;
    g1 = sb ? -g1 : g1;
    // ---- (u, v) <- the same combinations / 2^30 mod p -----------------------------------------
    int32_t nu[FQ_INV_LIMBS], nv[FQ_INV_LIMBS];
    fq_inv_update_uv(u, v, f0, g0, nu);
    fq_inv_update_uv(u, v, f1, g1, nv);
    uint32_t left = 0;
    TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) {
      a[i] = na[i];
      b[i] = nb[i];
      u[i] = nu[i];
      v[i] = nv[i];
      left |= (uint32_t)na[i];
    }
    // a = 0: b = gcd = 1 and further rounds leave v's residue alone.  761 steps are the worst case; random inputs are
    // done after 17-20 rounds (18.4 on average), so a wave usually leaves here after 19 or 20 of the 26.
    if (!wave_any(left != 0)) break;
  }
  TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) out[i] = v[i];
}

TC_HD_NOINLINE Fq Fq::inv() const {
  int32_t t[FQ_LIMBS];
  fq_redc_full(*this, t);  // the canonical value in [0, p], 28-bit limbs
  int32_t e = 0;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) e |= (t[i] ^ FQL_P[i]);
  int32_t y[FQ_INV_LIMBS], r[FQ_INV_LIMBS];
  TC_UNROLL for (int i = 0; i < FQ_INV_LIMBS; i++) {  // 28-bit limbs -> 30-bit limbs (p itself counts as 0)
    const int bit = 30 * i, j = bit / FQ_RADIX, sh = bit % FQ_RADIX;
    uint32_t w = (uint32_t)t[j] >> sh;
    if (j + 1 < FQ_LIMBS) w |= (uint32_t)t[j + 1] << (FQ_RADIX - sh);
    if (j + 2 < FQ_LIMBS && 2 * FQ_RADIX - sh < 30) w |= (uint32_t)t[j + 2] << (2 * FQ_RADIX - sh);
    y[i] = e ? (int32_t)(w & (uint32_t)FQ_INV_MASK) : 0;
  }
  fq_inv_limbs30(y, r);
  Fq x;  // 30-bit limbs (value in (-2p, p), signed top limb) -> 28-bit limbs, the top one keeps the sign
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int bit = FQ_RADIX * i, j = bit / 30, sh = bit % 30;
    if (i == FQ_LIMBS - 1) {
      x.l[i] = r[j] >> sh;  // j = 12: arithmetic shift of the signed top limb
    } else {
      uint32_t w = (uint32_t)r[j] >> sh;
      if (j + 1 < FQ_INV_LIMBS) w |= (uint32_t)r[j + 1] << (30 - sh);
      x.l[i] = (int32_t)(w & (uint32_t)FQ_MASK);
    }
  }
  x.set_range(-0.001f, 1.f);
  x.set_val(2.f);
  return fq_mul(x, Fq::from_limbs(FQL_R2));
}

// 12 canonical u32 words (an integer < 2^384) -> plain limbs
TC_HD void words12_to_limbs(const uint32_t* w, int32_t* l) {
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int bit = FQ_RADIX * i;
    const int wi = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)w[wi] >> sh;
    if (wi + 1 < 12) v |= (uint64_t)w[wi + 1] << (32 - sh);
    l[i] = (int32_t)((uint32_t)v & (uint32_t)FQ_MASK);
  }
}

TC_HD void limbs_to_words12(const int32_t* l, uint32_t* w) {
  TC_UNROLL for (int i = 0; i < 12; i++) w[i] = 0;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    const int bit = FQ_RADIX * i;
    const int wi = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)(uint32_t)l[i] << sh;
    if (wi < 12) w[wi] |= (uint32_t)v;
    if (wi + 1 < 12) w[wi + 1] |= (uint32_t)(v >> 32);
  }
}

// canonical integer (12 little-endian u32 words, < p) -> Montgomery form
TC_HD Fq Fq::from_canonical(const uint32_t* words12) {
  Fq a;
  words12_to_limbs(words12, a.l);
  a.set_range(0.f, 1.f);
  a.set_val(1.f);
  return fq_mul(a, Fq::from_limbs(FQL_R2));
}

// Montgomery form -> canonical integer in [0, p)
TC_HD void Fq::to_canonical(uint32_t* words12) const {
  int32_t t[FQ_LIMBS];
  fq_redc_full(*this, t);
  int32_t e = 0;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) e |= (t[i] ^ FQL_P[i]);
  if (e == 0) {
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) t[i] = 0;  // the representative p is 0
  }
  limbs_to_words12(t, words12);
}

// A 384-bit pattern that IS a Montgomery representation w.r.t. 2^384 (what ff_derive's
// random() yields): value = pattern * 2^-384, so the R = 2^392 form is pattern * 2^8
// = montmul(pattern, 2^400 mod p).
TC_HD Fq Fq::from_mont384(const uint32_t* words12) {
  Fq a;
  words12_to_limbs(words12, a.l);
  a.set_range(0.f, 1.f);
  a.set_val(1.f);
  return fq_mul(a, Fq::from_limbs(FQL_FIX384));
}

// lexicographic "y > -y" test on the canonical value: y > (p-1)/2
TC_HD bool fq_canonical_gt_half(const uint32_t* y) {
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < 12; i++) {
    uint64_t d = (uint64_t)FQ_HALF_P_CANON[i] - y[i] - borrow;
    borrow = (uint32_t)(d >> 63);
  }
  return borrow != 0;
}

}  // namespace tc
