// Prime-field arithmetic for gfx950: Montgomery form over 32-bit limbs.
//   Fq  (BLS12-381 base field, 381 bit): 12 x u32, R = 2^384
//   Fr  (scalar field, 255 bit):          8 x u32, R = 2^256
// Replaces (on device) pairing::bls12_381::Fq and ff's derived Fr, which the reference
// reaches through the engine seam at /root/reference/src/lib.rs:60-67.
//
// One field element lives in one lane's VGPRs; the inner product step is the gfx950
// 32x32+64 -> 64 multiply-add (v_mad_u64_u32).  No MFMA: this is wide-integer modular
// arithmetic, not a dense contraction.
#pragma once
#include "tc_common.h"

namespace tc {

struct FqParams {
  static constexpr int N = 12;
  TC_HD static uint32_t p(int i) { return FQ_P[i]; }
  TC_HD static uint32_t one(int i) { return FQ_ONE[i]; }
  TC_HD static uint32_t r2(int i) { return FQ_R2[i]; }
  TC_HD static uint32_t pm2(int i) { return FQ_P_MINUS_2[i]; }
  static constexpr uint32_t inv = FQ_INV32;
  static constexpr int bits = 381;
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

  TC_HD static Mont zero() {
    Mont r;
    TC_UNROLL for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  TC_HD static Mont one() {
    Mont r;
    TC_UNROLL for (int i = 0; i < N; i++) r.l[i] = PR::one(i);
    return r;
  }
  TC_HD bool is_zero() const {
    uint32_t o = 0;
    TC_UNROLL for (int i = 0; i < N; i++) o |= l[i];
    return o == 0;
  }
  TC_HD bool operator==(const Mont& b) const {
    uint32_t o = 0;
    TC_UNROLL for (int i = 0; i < N; i++) o |= (l[i] ^ b.l[i]);
    return o == 0;
  }
  TC_HD bool operator!=(const Mont& b) const { return !(*this == b); }
};

// r = a - p if a >= p else a     (a < 2p)
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
  // both moduli leave the top bit of the top limb clear, so no carry out of limb N-1
  mont_cond_sub_p(r);
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

template <class PR>
TC_HD Mont<PR> mont_neg(const Mont<PR>& a) {
  constexpr int N = PR::N;
  Mont<PR> r;
  uint32_t borrow = 0;
  uint32_t nz = 0;
  TC_UNROLL for (int i = 0; i < N; i++) nz |= a.l[i];
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)PR::p(i) - a.l[i] - borrow;
    r.l[i] = nz ? (uint32_t)d : 0u;
    borrow = (uint32_t)(d >> 63);
  }
  return r;
}

template <class PR>
TC_HD Mont<PR> mont_dbl(const Mont<PR>& a) {
  return mont_add(a, a);
}

// Montgomery product a*b*R^-1 mod p, coarsely-integrated operand scanning (CIOS).  The top
// limb of both moduli is < 2^31, so the running value never needs an (N+2)-th limb.
template <class PR>
TC_HD void mont_mul_body(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  constexpr int N = PR::N;
  uint32_t t[N + 1];
  TC_UNROLL for (int i = 0; i <= N; i++) t[i] = 0;
  TC_UNROLL for (int i = 0; i < N; i++) {
    uint64_t c = 0;
    const uint32_t bi = b[i];
    TC_UNROLL for (int j = 0; j < N; j++) {
      uint64_t s = (uint64_t)a[j] * bi + t[j] + c;
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
  TC_UNROLL for (int i = 0; i < N; i++) out[i] = r.l[i];
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(TC_INLINE_MUL)
// On the device the product is a REAL function (not inlined into every tower/curve routine:
// that makes the pairing kernel's code object several MB and hipcc compile times unbounded).
// Operands travel as 2N scalar u32 arguments so the AMDGPU calling convention keeps them in
// VGPRs (aggregates beyond 16 registers would be passed through scratch memory).
__device__ __attribute__((noinline)) inline Mont<FqParams> fq_mul_call(
    uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
    uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3,
    uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
  const uint32_t a[12] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11};
  const uint32_t b[12] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11};
  Mont<FqParams> r;
  mont_mul_body<FqParams>(a, b, r.l);
  return r;
}
__device__ __attribute__((noinline)) inline Mont<FrParams> fr_mul_call(
    uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
    uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7) {
  const uint32_t a[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
  const uint32_t b[8] = {b0, b1, b2, b3, b4, b5, b6, b7};
  Mont<FrParams> r;
  mont_mul_body<FrParams>(a, b, r.l);
  return r;
}
TC_HD Mont<FqParams> mont_mul(const Mont<FqParams>& a, const Mont<FqParams>& b) {
  return fq_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10],
                     a.l[11], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9],
                     b.l[10], b.l[11]);
}
TC_HD Mont<FrParams> mont_mul(const Mont<FrParams>& a, const Mont<FrParams>& b) {
  return fr_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], b.l[0], b.l[1], b.l[2],
                     b.l[3], b.l[4], b.l[5], b.l[6], b.l[7]);
}
#else
template <class PR>
TC_HD Mont<PR> mont_mul(const Mont<PR>& a, const Mont<PR>& b) {
  Mont<PR> r;
  mont_mul_body<PR>(a.l, b.l, r.l);
  return r;
}
#endif

template <class PR>
TC_HD Mont<PR> mont_sqr(const Mont<PR>& a) {
  return mont_mul(a, a);
}

// canonical integer (little-endian limbs) -> Montgomery form
template <class PR>
TC_HD Mont<PR> mont_from_canonical(const uint32_t* limbs) {
  Mont<PR> a, r2;
  TC_UNROLL for (int i = 0; i < PR::N; i++) {
    a.l[i] = limbs[i];
    r2.l[i] = PR::r2(i);
  }
  return mont_mul(a, r2);
}

// Montgomery form -> canonical integer
template <class PR>
TC_HD void mont_to_canonical(const Mont<PR>& a, uint32_t* limbs) {
  Mont<PR> o;
  TC_UNROLL for (int i = 0; i < PR::N; i++) o.l[i] = (i == 0) ? 1u : 0u;
  Mont<PR> r = mont_mul(a, o);
  TC_UNROLL for (int i = 0; i < PR::N; i++) limbs[i] = r.l[i];
}

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

// a^e, e given as nlimbs little-endian u32 words via accessor E(i); fixed 4-bit windows,
// uniform control flow (all lanes run the same exponent).
template <class F, class EXP>
TC_HD F field_pow_fixed(const F& a, EXP e, int nbits) {
  F tbl[16];
  tbl[0] = F::one();
  tbl[1] = a;
  TC_NOUNROLL for (int i = 2; i < 16; i++) tbl[i] = (i & 1) ? tbl[i - 1] * a : tbl[i >> 1].sqr();
  int top = ((nbits + 3) / 4) * 4 - 4;
  F r = tbl[(e(top >> 5) >> (top & 31)) & 15];
  TC_NOUNROLL for (int pos = top - 4; pos >= 0; pos -= 4) {
    r = r.sqr();
    r = r.sqr();
    r = r.sqr();
    r = r.sqr();
    uint32_t w = (e(pos >> 5) >> (pos & 31)) & 15;
    if (w) r = r * tbl[w];
  }
  return r;
}

// ---- Fq / Fr value types with operators ---------------------------------------------
template <class PR>
struct Fe {
  Mont<PR> v;
  static constexpr int N = PR::N;
  TC_HD static Fe zero() { return Fe{Mont<PR>::zero()}; }
  TC_HD static Fe one() { return Fe{Mont<PR>::one()}; }
  TC_HD bool is_zero() const { return v.is_zero(); }
  TC_HD bool operator==(const Fe& b) const { return v == b.v; }
  TC_HD bool operator!=(const Fe& b) const { return v != b.v; }
  TC_HD Fe operator+(const Fe& b) const { return Fe{mont_add(v, b.v)}; }
  TC_HD Fe operator-(const Fe& b) const { return Fe{mont_sub(v, b.v)}; }
  TC_HD Fe operator-() const { return Fe{mont_neg(v)}; }
  TC_HD Fe operator*(const Fe& b) const { return Fe{mont_mul(v, b.v)}; }
  TC_HD Fe sqr() const { return Fe{mont_sqr(v)}; }
  TC_HD Fe dbl() const { return Fe{mont_dbl(v)}; }
  TC_HD static Fe from_canonical(const uint32_t* limbs) { return Fe{mont_from_canonical<PR>(limbs)}; }
  TC_HD void to_canonical(uint32_t* limbs) const { mont_to_canonical(v, limbs); }
  // Fermat inverse a^(p-2); 0 -> 0
  TC_HD_NOINLINE Fe inv() const {
    return field_pow_fixed(*this, [](int i) { return PR::pm2(i); }, PR::bits);
  }
  TC_HD static Fe select(bool c, const Fe& a, const Fe& b) {
    Fe r;
    TC_UNROLL for (int i = 0; i < N; i++) r.v.l[i] = c ? a.v.l[i] : b.v.l[i];
    return r;
  }
};

using Fq = Fe<FqParams>;
using Fr = Fe<FrParams>;

// Fq from u64 (IntoFr for u64, /root/reference/src/into_fr.rs:16-20)
TC_HD Fr fr_from_u64(uint64_t x) {
  uint32_t l[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0};
  return Fr::from_canonical(l);
}

// lexicographic "y > -y" test on the canonical value: y > (p-1)/2
TC_HD bool fq_canonical_gt_half(const uint32_t* y) {
  // returns y > (p-1)/2
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < 12; i++) {
    uint64_t d = (uint64_t)FQ_HALF_P_CANON[i] - y[i] - borrow;
    borrow = (uint32_t)(d >> 63);
  }
  return borrow != 0;
}

}  // namespace tc
