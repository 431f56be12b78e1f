// Extension tower Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(u+1)), Fq12 = Fq6[w]/(w^2-v)
// (device replacement for pairing::bls12_381::{Fq2,Fq6,Fq12}).
#pragma once
#include "tc_field.h"

namespace tc {

#if TC_PAIR
// hipcc build: ONE coefficient per lane (tc_common.h).  Even lane: c0 (real part), odd lane: c1.
struct Fq2 {
  Fq m;
  TC_HD static bool odd() { return pair_odd() != 0; }
  // the partner lane's coefficient
  TC_HD Fq other() const {
    Fq r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = pair_swap(m.l[i]);
    return r;
  }
  // both coefficients, on both lanes (codecs, square roots)
  TC_HD static Fq2 make(const Fq& re, const Fq& im) { return Fq2{Fq::select(odd(), im, re)}; }
  TC_HD Fq re() const { return Fq::select(odd(), other(), m); }
  TC_HD Fq im() const { return Fq::select(odd(), m, other()); }
  TC_HD static Fq2 zero() { return Fq2{Fq::zero()}; }
  TC_HD static Fq2 one() { return make(Fq::one(), Fq::zero()); }
  TC_HD bool is_zero() const { return pair_all(m.is_zero()); }
  TC_HD bool operator==(const Fq2& b) const { return (*this - b).is_zero(); }
  TC_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
  TC_HD Fq2 operator+(const Fq2& b) const { return Fq2{m + b.m}; }
  TC_HD Fq2 operator-(const Fq2& b) const { return Fq2{m - b.m}; }
  TC_HD Fq2 operator-() const { return Fq2{-m}; }
  TC_HD Fq2 dbl() const { return Fq2{m.dbl()}; }
  TC_HD Fq2 conj() const { return Fq2{Fq::select(odd(), -m, m)}; }
  TC_HD Fq2 norm() const { return Fq2{m.norm()}; }
  TC_HD Fq2 reduce_value() const { return Fq2{m.reduce_value()}; }
  // each lane: two limb products, one reduction (tc_field.h fq2p_mul_call)
  TC_HD Fq2 operator*(const Fq2& b) const {
    Fq2 r;
#if defined(__HIP_DEVICE_COMPILE__)
    const Fq& a = m;
    FqRaw t = fq2p_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13], b.m.l[0], b.m.l[1], b.m.l[2], b.m.l[3], b.m.l[4], b.m.l[5], b.m.l[6], b.m.l[7], b.m.l[8], b.m.l[9], b.m.l[10], b.m.l[11], b.m.l[12], b.m.l[13]);
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = t.l[i];
#else
    r = b;
#endif
    return r;
  }
  TC_HD Fq2 sqr() const {
    Fq2 r;
#if defined(__HIP_DEVICE_COMPILE__)
    const Fq& a = m;
    FqRaw t = fq2p_sqr_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13]);
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = t.l[i];
#else
    r = *this;
#endif
    return r;
  }
  // the squaring with its body inlined (tc_field.h fq2p_sqr_inl) -- EXPERIMENT, -DTC_INLINE_CYCLO_SQR: measured in r04 on the
  // compressed cyclotomic chain (6 squarings per step, 5 % fewer instructions, no calls): verify_g2 2 % SLOWER than with the
  // out-of-line body (profiles/r04_cyclo_inline_ab.txt), like every inlining of the multiplier before it
  TC_HD Fq2 sqr_inl() const {
    Fq2 r;
#if defined(__HIP_DEVICE_COMPILE__) && defined(TC_INLINE_CYCLO_SQR)
    FqRaw t = fq2p_sqr_inl(m.l, pair_odd());
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = t.l[i];
#else
    r = sqr();
#endif
    return r;
  }
  TC_HD Fq2 scale(const Fq& k) const { return Fq2{m * k}; }
  // times the non-residue (1 + u): (c0 - c1, c0 + c1)
  TC_HD Fq2 mul_xi() const {
    const Fq o = other();
    return Fq2{m + Fq::select(odd(), o, -o)};
  }
  // c0^2 + c1^2 (the Fq2 norm), on both lanes
  TC_HD Fq norm_fq() const {
    const Fq sq = m.sqr();
    return sq + Fq2{sq}.other();
  }
  TC_HD_NOINLINE Fq2 inv() const {
    const Fq sq = m.sqr();
    const Fq t = (sq + Fq2{sq}.other()).inv();  // 1 / (c0^2 + c1^2), computed by both lanes
    const Fq r = m * t;
    return Fq2{Fq::select(odd(), -r, r)};
  }
  TC_HD static Fq2 select(bool c, const Fq2& a, const Fq2& b) { return Fq2{Fq::select(c, a.m, b.m)}; }
  TC_HD static Fq2 select_lane(bool c, const Fq2& a, const Fq2& b) { return Fq2{Fq::select_lane(c, a.m, b.m)}; }
};
#else
struct Fq2 {
  Fq c0, c1;
  TC_HD static Fq2 make(const Fq& re, const Fq& im) { return Fq2{re, im}; }
  TC_HD Fq re() const { return c0; }
  TC_HD Fq im() const { return c1; }
  TC_HD static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  TC_HD static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  TC_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  TC_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  TC_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
  TC_HD Fq2 operator+(const Fq2& b) const { return Fq2{c0 + b.c0, c1 + b.c1}; }
  TC_HD Fq2 operator-(const Fq2& b) const { return Fq2{c0 - b.c0, c1 - b.c1}; }
  TC_HD Fq2 operator-() const { return Fq2{-c0, -c1}; }
  TC_HD Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
  TC_HD Fq2 conj() const { return Fq2{c0, -c1}; }
  TC_HD Fq2 norm() const { return Fq2{c0.norm(), c1.norm()}; }
  TC_HD Fq2 reduce_value() const { return Fq2{c0.reduce_value(), c1.reduce_value()}; }
#if defined(__HIP_DEVICE_COMPILE__)
  // one-lane-per-job device build (-DTC_NO_PAIR): Karatsuba, 3 Fq mul.  TIMING EXPERIMENTS ONLY -- its
  // outputs have wider limb intervals than the coefficient formulas the 28-bit bound analysis covers
  TC_HD Fq2 operator*(const Fq2& b) const {
    Fq aa = c0 * b.c0;
    Fq bb = c1 * b.c1;
    Fq o = (c0 + c1) * (b.c0 + b.c1);
    return Fq2{aa - bb, o - aa - bb};
  }
#else
  // the two coefficient formulas of the lane-pair build, one after the other
  TC_HD Fq2 operator*(const Fq2& b) const {
    return Fq2{fq_mul2(c0, b.c0, -c1, b.c1), fq_mul2(c1, b.c0, c0, b.c1)};
  }
#endif
  // c0 = (c0 + c1)(c0 - c1), c1 = (2 c1) c0
  TC_HD Fq2 sqr() const {
    TC_SPLIT_SCOPE;
    return Fq2{(c0 + c1) * (c0 - c1), c1.dbl() * c0};
  }
  TC_HD Fq2 sqr_inl() const { return sqr(); }
  TC_HD Fq2 scale(const Fq& k) const {
    TC_SPLIT_SCOPE;
    return Fq2{c0 * k, c1 * k};
  }
  // times the non-residue (1 + u)
  TC_HD Fq2 mul_xi() const { return Fq2{c0 - c1, c0 + c1}; }
  TC_HD Fq norm_fq() const {
    TC_SPLIT_SCOPE;
    return c0.sqr() + c1.sqr();
  }
  TC_HD_NOINLINE Fq2 inv() const {
    const Fq t = norm_fq().inv();
    return scale(t).conj();
  }
  TC_HD static Fq2 select(bool c, const Fq2& a, const Fq2& b) {
    return Fq2{Fq::select(c, a.c0, b.c0), Fq::select(c, a.c1, b.c1)};
  }
  TC_HD static Fq2 select_lane(bool c, const Fq2& a, const Fq2& b) {
    return Fq2{Fq::select_lane(c, a.c0, b.c0), Fq::select_lane(c, a.c1, b.c1)};
  }
};
#endif

struct Fq6 {
  Fq2 c0, c1, c2;
  TC_HD static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  TC_HD static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  TC_HD bool operator==(const Fq6& b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
  TC_HD Fq6 operator+(const Fq6& b) const { return Fq6{c0 + b.c0, c1 + b.c1, c2 + b.c2}; }
  TC_HD Fq6 operator-(const Fq6& b) const { return Fq6{c0 - b.c0, c1 - b.c1, c2 - b.c2}; }
  TC_HD Fq6 operator-() const { return Fq6{-c0, -c1, -c2}; }
  TC_HD Fq6 norm() const { return Fq6{c0.norm(), c1.norm(), c2.norm()}; }
  TC_HD Fq6 reduce_value() const { return Fq6{c0.reduce_value(), c1.reduce_value(), c2.reduce_value()}; }
  TC_FQ6_ATTR Fq6 operator*(const Fq6& b) const {
    Fq2 t0 = c0 * b.c0;
    Fq2 t1 = c1 * b.c1;
    Fq2 t2 = c2 * b.c2;
    Fq6 r;
    r.c0 = t0 + ((c1 + c2) * (b.c1 + b.c2) - t1 - t2).mul_xi();
    r.c1 = (c0 + c1) * (b.c0 + b.c1) - t0 - t1 + t2.mul_xi();
    r.c2 = (c0 + c2) * (b.c0 + b.c2) - t0 - t2 + t1;
    return TC_FQ6_OUT(r);
  }
  // CH-SQR2 (Chung-Hasan): 2 mul + 3 sqr in Fq2
  TC_FQ6_ATTR Fq6 sqr() const {
    Fq2 s0 = c0.sqr();
    Fq2 ab = c0 * c1;
    Fq2 s1 = ab.dbl();
    Fq2 s2 = (c0 - c1 + c2).norm().sqr();
    Fq2 bc = c1 * c2;
    Fq2 s3 = bc.dbl();
    Fq2 s4 = c2.sqr();
    Fq6 r;
    r.c0 = s0 + s3.mul_xi();
    r.c1 = s1 + s4.mul_xi();
    r.c2 = s1 + s2 + s3 - s0 - s4;
    return TC_FQ6_OUT(r);
  }
  TC_HD Fq6 mul_by_v() const { return Fq6{c2.mul_xi(), c0, c1}; }
  // sparse: times (b0 + b1 v)
  TC_FQ6_ATTR Fq6 mul_by_01(const Fq2& b0, const Fq2& b1) const {
    Fq2 aa = c0 * b0;
    Fq2 bb = c1 * b1;
    Fq6 r;
    r.c0 = ((c1 + c2) * b1 - bb).mul_xi() + aa;
    r.c1 = (c0 + c1) * (b0 + b1) - aa - bb;
    r.c2 = (c0 + c2) * b0 - aa + bb;
    return TC_FQ6_OUT(r);
  }
  // sparse: times (b1 v + b2 v^2)
  TC_FQ6_ATTR Fq6 mul_by_12(const Fq2& b1, const Fq2& b2) const {
    Fq2 bb = c1 * b1;
    Fq2 cc = c2 * b2;
    Fq6 r;
    r.c0 = ((c1 + c2) * (b1 + b2) - bb - cc).mul_xi();
    r.c1 = (c0 + c1) * b1 - bb + cc.mul_xi();
    r.c2 = (c0 + c2) * b2 - cc + bb;
    return TC_FQ6_OUT(r);
  }
  // sparse: times (b1 v)
  TC_FQ6_ATTR Fq6 mul_by_1(const Fq2& b1) const { return TC_FQ6_OUT((Fq6{(c2 * b1).mul_xi(), c0 * b1, c1 * b1})); }
  TC_HD_NOINLINE Fq6 inv() const {
    // (operands carry-normalised: the caller may hand in a difference of products)
    const Fq6 n = this->norm();
    Fq2 t0 = (n.c0.sqr() - (n.c1 * n.c2).mul_xi());
    Fq2 t1 = (n.c2.sqr().mul_xi() - n.c0 * n.c1);
    Fq2 t2 = (n.c1.sqr() - n.c0 * n.c2);
    Fq2 d = (n.c0 * t0 + (n.c2 * t1 + n.c1 * t2).mul_xi()).norm();
    Fq2 di = d.inv();
    return Fq6{t0 * di, t1 * di, t2 * di};
  }
};

TC_HD Fq2 frob_coeff(int k, int i) {  // gamma_k[i], i = 1..5
  const int32_t* a0 = (k == 1) ? FROBL_1_C0[i - 1] : (k == 2) ? FROBL_2_C0[i - 1] : FROBL_3_C0[i - 1];
  const int32_t* a1 = (k == 1) ? FROBL_1_C1[i - 1] : (k == 2) ? FROBL_2_C1[i - 1] : FROBL_3_C1[i - 1];
  return Fq2::make(Fq::from_limbs(a0), Fq::from_limbs(a1));
}

struct Fq12 {
  Fq6 c0, c1;
  TC_HD static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
  TC_HD bool operator==(const Fq12& b) const { return c0 == b.c0 && c1 == b.c1; }
  TC_FQ12MUL_ATTR Fq12 operator*(const Fq12& b) const {
    Fq6 t0 = c0 * b.c0;
    Fq6 t1 = c1 * b.c1;
    Fq12 r;
    r.c1 = (c0 + c1).norm() * (b.c0 + b.c1).norm() - t0 - t1;
    r.c0 = t0 + t1.mul_by_v();
    return r.reduce_value();
  }
  // complex squaring over Fq6: 2 Fq6 mul
  TC_FQ12_ATTR Fq12 sqr() const {
    Fq6 ab = c0 * c1;
    Fq6 t = (c0 + c1).norm() * (c0 + c1.mul_by_v()).norm() - ab - ab.mul_by_v();
    return Fq12{t, ab + ab}.norm();
  }
  TC_HD Fq12 conj() const { return Fq12{c0, (-c1).norm()}; }
  TC_HD Fq12 norm() const { return Fq12{c0.norm(), c1.norm()}; }
  TC_HD Fq12 reduce_value() const { return Fq12{c0.reduce_value(), c1.reduce_value()}; }
  TC_HD_NOINLINE Fq12 inv() const {
    Fq6 t = (c0.sqr() - c1.sqr().mul_by_v()).inv();
    return Fq12{c0 * t, -(c1 * t)};
  }
  // sparse multiplication by (d0 + d1 v) + (d4 v) w -- the Miller-loop line shape
  TC_FQ12_ATTR Fq12 mul_by_014(const Fq2& d0, const Fq2& d1, const Fq2& d4) const {
    Fq6 aa = c0.mul_by_01(d0, d1);
    Fq6 bb = c1.mul_by_1(d4);
    Fq2 o = d1 + d4;
    Fq12 r;
    r.c1 = (c1 + c0).norm().mul_by_01(d0, o.norm()) - aa - bb;
    r.c0 = bb.mul_by_v() + aa;
    return r.norm();
  }
  // (d0 + d1 v + d4 v w) * (e0 + e1 v + e4 v w): the product of two Miller-loop lines, 6 Fq2
  // products; the result has no w coefficient of degree 0 in v:  c1 = (0, *, *)
  TC_HD static Fq12 line_product(const Fq2& d0, const Fq2& d1, const Fq2& d4, const Fq2& e0, const Fq2& e1,
                                 const Fq2& e4) {
    Fq2 t0 = d0 * e0;
    Fq2 t1 = d1 * e1;
    Fq2 t3 = d4 * e4;
    Fq2 t2 = (d0 + d1) * (e0 + e1) - t0 - t1;
    Fq2 u = (d0 + d4) * (e0 + e4) - t0 - t3;
    Fq2 w = (d1 + d4) * (e1 + e4) - t1 - t3;
    Fq12 r;
    r.c0 = Fq6{(t0 + t3.mul_xi()).norm(), t2, t1};
    r.c1 = Fq6{Fq2::zero(), u, w.norm()};
    return r;
  }
  // times an element whose c1 is (0, *, *) (a line product): 6 + 5 + 6 Fq2 products
  TC_FQ12_ATTR Fq12 mul_by_line_product(const Fq12& l) const {
    Fq6 t0 = c0 * l.c0;
    Fq6 t1 = c1.mul_by_12(l.c1.c1, l.c1.c2);
    Fq12 r;
    r.c1 = (c0 + c1).norm() * (l.c0 + l.c1).norm() - t0 - t1;
    r.c0 = t0 + t1.mul_by_v();
    return r.norm();
  }
  // a^(q^k), k in {1,2,3}
  TC_HD_NOINLINE Fq12 frobenius(int k) const {
    const bool cj = (k & 1);
    Fq12 r;
    r.c0.c0 = cj ? c0.c0.conj() : c0.c0;
    r.c0.c1 = (cj ? c0.c1.conj() : c0.c1) * frob_coeff(k, 2);
    r.c0.c2 = (cj ? c0.c2.conj() : c0.c2) * frob_coeff(k, 4);
    r.c1.c0 = (cj ? c1.c0.conj() : c1.c0) * frob_coeff(k, 1);
    r.c1.c1 = (cj ? c1.c1.conj() : c1.c1) * frob_coeff(k, 3);
    r.c1.c2 = (cj ? c1.c2.conj() : c1.c2) * frob_coeff(k, 5);
    return r;
  }
  // Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part
  // of the final exponentiation): 9 Fq2 squarings' worth instead of 2 Fq6 mul.
  TC_CYCLO_ATTR Fq12 cyclotomic_sqr() const {
    // view as three Fq4 = Fq2[s]/(s^2 - xi) pairs: (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2)
    Fq2 z0 = c0.c0, z4 = c0.c1, z3 = c0.c2, z2 = c1.c0, z1 = c1.c1, z5 = c1.c2;
    Fq2 t0, t1, t2, t3;
    // fp4 square (z0, z1)
    {
      Fq2 a2 = z0.sqr(), b2 = z1.sqr();
      t0 = (b2.mul_xi() + a2).norm();
      t1 = (z0 + z1).sqr() - a2 - b2;
    }
    z0 = (t0 - z0).dbl() + t0;
    z1 = (t1 + z1).dbl() + t1;
    {
      Fq2 a2 = z2.sqr(), b2 = z3.sqr();
      t0 = (b2.mul_xi() + a2).norm();
      t1 = (z2 + z3).sqr() - a2 - b2;
    }
    {
      Fq2 a2 = z4.sqr(), b2 = z5.sqr();
      t2 = (b2.mul_xi() + a2).norm();
      t3 = (z4 + z5).sqr() - a2 - b2;
    }
    z4 = (t0 - z4).dbl() + t0;
    z5 = (t1 + z5).dbl() + t1;
    Fq2 t3x = t3.mul_xi().norm();
    z2 = (t3x + z2).dbl() + t3x;
    z3 = (t2 - z3).dbl() + t2;
    Fq12 r;
    // z' = 3t -+ 2z feeds z back linearly: squaring after squaring the VALUE would double, so
    // bring it back to ~p here (tc_field.h reduce_value)
    r.c0.c0 = z0.reduce_value(); r.c0.c1 = z4.reduce_value(); r.c0.c2 = z3.reduce_value();
    r.c1.c0 = z2.reduce_value(); r.c1.c1 = z1.reduce_value(); r.c1.c2 = z5.reduce_value();
    return r;
  }
};

// Karabina's compressed form of an element of the cyclotomic subgroup: four of its six Fq2
// coefficients, in the Granger-Scott numbering of cyclotomic_sqr() above
//     f = z0 + z2 w + z4 w^2 + z1 w^3 + z3 w^4 + z5 w^5        (w^6 = xi).
// The squaring formulas for (z2, z3, z4, z5) never read (z0, z1): 6 Fq2 squarings instead of 9.
// decompress() (tc_pairing.h) recovers z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2) and
// z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1.
struct CycloCompressed {
  Fq2 z2, z3, z4, z5;
  TC_HD static CycloCompressed from(const Fq12& f) { return CycloCompressed{f.c1.c0, f.c0.c2, f.c0.c1, f.c1.c2}; }
  // REDUCE = false leaves the outputs carry-normalised only: z' = 3 t -+ 2 z doubles the VALUE per squaring, so the
  // chain of tc_pairing.h pulls it back (reduce_value, ~105 instructions per coefficient) every kCycloReduceEvery-th
  // squaring instead of every time -- as often as the interval analysis (tests/hostsim -DTC_BOUND_CHECK) demands
  template <bool REDUCE>
  TC_CYCLO_ATTR CycloCompressed sqr_t() const {
    Fq2 t0, t1, t2, t3;
    {
      Fq2 a2 = z2.sqr_inl(), b2 = z3.sqr_inl();
      t0 = (b2.mul_xi() + a2).norm();
      t1 = (z2 + z3).sqr_inl() - a2 - b2;
    }
    {
      Fq2 a2 = z4.sqr_inl(), b2 = z5.sqr_inl();
      t2 = (b2.mul_xi() + a2).norm();
      t3 = (z4 + z5).sqr_inl() - a2 - b2;
    }
    const Fq2 t3x = t3.mul_xi().norm();
    CycloCompressed r;
    r.z4 = (t0 - z4).dbl() + t0;
    r.z5 = (t1 + z5).dbl() + t1;
    r.z2 = (t3x + z2).dbl() + t3x;
    r.z3 = (t2 - z3).dbl() + t2;
    if (REDUCE) {
      r.z4 = r.z4.reduce_value(); r.z5 = r.z5.reduce_value(); r.z2 = r.z2.reduce_value(); r.z3 = r.z3.reduce_value();
    } else {
      r.z4 = r.z4.norm(); r.z5 = r.z5.norm(); r.z2 = r.z2.norm(); r.z3 = r.z3.norm();
    }
    return r;
  }
  TC_HD CycloCompressed sqr() const { return sqr_t<true>(); }
  TC_HD CycloCompressed reduced() const {
    return CycloCompressed{z2.reduce_value(), z3.reduce_value(), z4.reduce_value(), z5.reduce_value()};
  }
};
#ifndef TC_CYCLO_REDUCE_EVERY
#define TC_CYCLO_REDUCE_EVERY 3
#endif
constexpr int kCycloReduceEvery = TC_CYCLO_REDUCE_EVERY;

}  // namespace tc
