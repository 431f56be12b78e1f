// G1 / G2 group arithmetic (Jacobian coordinates), templated over the coordinate field.
// Device replacement for pairing::bls12_381::{G1,G2,G1Affine,G2Affine} as reached from
// /root/reference/src/lib.rs:373 (sign_g2), :461 (decrypt_share), :764 (interpolate).
#pragma once
#include "tc_tower.h"

namespace tc {

// Output conditioning of point coordinates (tc_field.h): one carry pass.  (The doubling's x3 = f - 8 t is the widest
// sum of the point formulas -- value ~ 14 p -- but it only ever enters products, which pull the value back; the
// interval analysis of tests/hostsim -DTC_BOUND_CHECK accepts the carry pass for Fq and Fq2 alike.)
TC_HD Fq coord_out(const Fq& a) { return a.norm(); }
TC_HD Fq2 coord_out(const Fq2& a) { return a.norm(); }
// one carry pass only, where the bound checker (tests/hostsim -DTC_BOUND_CHECK) accepts it
template <class F>
TC_HD F coord_norm(const F& a) { return a.norm(); }

template <class F>
struct Affine {
  F x, y;
  bool inf;
  TC_HD static Affine infinity() { return Affine{F::zero(), F::one(), true}; }
};

template <class F>
struct Jac {
  F x, y, z;  // infinity <=> z == 0
  TC_HD static Jac infinity() { return Jac{F::zero(), F::one(), F::zero()}; }
  TC_HD bool is_inf() const { return z.is_zero(); }
  TC_HD static Jac from_affine(const Affine<F>& a) {
    Jac r{a.x, a.y, F::one()};
    if (a.inf) r = infinity();
    return r;
  }
  TC_HD static Jac select(bool c, const Jac& a, const Jac& b) {
    return Jac{F::select(c, a.x, b.x), F::select(c, a.y, b.y), F::select(c, a.z, b.z)};
  }
};

TC_HD bool maybe_zero56(const Fq& a) { return a.maybe_zero56(); }
TC_HD bool maybe_zero56(const Fq2& a) {
#if TC_PAIR
  return pair_all(a.m.maybe_zero56());  // zero <=> both coefficients zero
#else
  return a.c0.maybe_zero56() && a.c1.maybe_zero56();
#endif
}
// The 28-bit limbs leave three bits of headroom in an int32 (|limb| < 8 * 2^28) and a
// multiplication wants B_a * B_b <= 8 (tc_field.h), so the formulas below keep small multiples
// AFTER the products they scale and pass sums through norm() (one carry pass) where the interval
// analysis of tests/hostsim -DTC_BOUND_CHECK asks for it.

// dbl-2009-l (a = 0): 2M + 5S.  Doubling the identity (z = 0) yields z3 = 0 again.
template <class F>
TC_JAC_ATTR Jac<F> jac_dbl(const Jac<F>& p) {
  F a = p.x.sqr();
  F b = p.y.sqr();
  F c = b.sqr();
  F t = ((p.x + b).sqr() - a - c).norm();  // d = 2 t
  F e = (a.dbl() + a).norm();
  F f = e.sqr();
  Jac<F> r;
  r.z = coord_norm((p.y * p.z).dbl());
  r.x = coord_out(f - t.dbl().dbl());
  r.y = coord_norm(e * (t.dbl() - r.x) - c.dbl().dbl().norm().dbl());
  return r;
}

// madd-2007-bl with the exceptional cases handled (p = inf, q = inf, p = +-q).
//   h = u2 - x1, r = s2 - y1 (the formulas' r is 2 r), hh = h^2, j = h hh (theirs: 4 j), v = x1 hh (4 v)
//   x3 = 4 (r^2 - j - 2 v),  y3 = 2 (r (4 v - x3) - 4 y1 j),  z3 = (z1 + h)^2 - z1^2 - hh
template <class F>
TC_JAC_ATTR Jac<F> jac_add_mixed(const Jac<F>& p, const Affine<F>& q) {
  if (q.inf) return p;
  F z1z1 = p.z.sqr();
  F u2 = q.x * z1z1;
  F s2 = q.y * p.z * z1z1;
  F h = u2 - p.x;
  F rr = s2 - p.y;
  const bool p_inf = p.is_inf();
  const bool same_x = h.is_zero();
  if (!p_inf && same_x) {
    // p == q -> double; p == -q -> infinity
    if (rr.is_zero()) return jac_dbl(p);
    return Jac<F>::infinity();
  }
  F hh = h.sqr();
  F j = h * hh;
  F v = p.x * hh;
  Jac<F> r;
  r.x = coord_norm((rr.sqr() - j - v.dbl()).norm().dbl().dbl());
  r.y = coord_norm((rr * (v.dbl().dbl() - r.x) - (p.y * j).dbl().dbl()).norm().dbl());
  r.z = coord_norm((p.z + h).norm().sqr() - z1z1 - hh);
  if (p_inf) r = Jac<F>{q.x, q.y, F::one()};
  return r;
}

// The same addition for ladder loops: the generic case only, no branch.  A lane whose operands MAY hit a special case
// (p or q at infinity, p = +-q; two-limb zero tests, never missed, wrongly raised with probability ~2^-46) sets `exc`;
// its result is then meaningless and the caller redoes that ladder with jac_add_mixed.
template <class F>
TC_JAC_ATTR Jac<F> jac_add_mixed_generic(const Jac<F>& p, const Affine<F>& q, bool& exc) {
  F z1z1 = p.z.sqr();
  F u2 = q.x * z1z1;
  F s2 = q.y * p.z * z1z1;
  F h = u2 - p.x;
  F rr = s2 - p.y;
  exc = exc || q.inf || maybe_zero56(h) || maybe_zero56(p.z);
  F hh = h.sqr();
  F j = h * hh;
  F v = p.x * hh;
  Jac<F> r;
  r.x = coord_norm((rr.sqr() - j - v.dbl()).norm().dbl().dbl());
  r.y = coord_norm((rr * (v.dbl().dbl() - r.x) - (p.y * j).dbl().dbl()).norm().dbl());
  r.z = coord_norm((p.z + h).norm().sqr() - z1z1 - hh);
  return r;
}

// mmadd-2007-bl: two AFFINE points (z1 = z2 = 1), 4M + 2S; same rescaling as jac_add_mixed
// (h = x2 - x1, r = y2 - y1, z3 = 2 h).  Table builders start with these.
template <class F>
TC_JAC_ATTR Jac<F> jac_add_affine(const Affine<F>& p, const Affine<F>& q) {
  if (q.inf) return Jac<F>::from_affine(p);
  if (p.inf) return Jac<F>::from_affine(q);
  F h = q.x - p.x;
  F rr = q.y - p.y;
  if (h.is_zero()) {
    if (rr.is_zero()) return jac_dbl(Jac<F>::from_affine(p));
    return Jac<F>::infinity();
  }
  F hh = h.sqr();
  F j = h * hh;
  F v = p.x * hh;
  Jac<F> r;
  r.x = coord_norm((rr.sqr() - j - v.dbl()).norm().dbl().dbl());
  r.y = coord_norm((rr * (v.dbl().dbl() - r.x) - (p.y * j).dbl().dbl()).norm().dbl());
  r.z = coord_norm(h.dbl());
  return r;
}

// add-2007-bl with the exceptional cases handled (same rescaling as above: r is half the
// formulas' r, i = hh is a quarter of theirs).
template <class F>
TC_JAC_ATTR Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  const bool p_inf = p.is_inf();
  const bool q_inf = q.is_inf();
  F z1z1 = p.z.sqr();
  F z2z2 = q.z.sqr();
  F u1 = p.x * z2z2;
  F u2 = q.x * z1z1;
  F s1 = p.y * q.z * z2z2;
  F s2 = q.y * p.z * z1z1;
  F h = u2 - u1;
  F rr = s2 - s1;
  if (!p_inf && !q_inf && h.is_zero()) {
    if (rr.is_zero()) return jac_dbl(p);
    return Jac<F>::infinity();
  }
  F hh = h.sqr();
  F j = h * hh;
  F v = u1 * hh;
  Jac<F> r;
  r.x = coord_norm((rr.sqr() - j - v.dbl()).norm().dbl().dbl());
  r.y = coord_norm((rr * (v.dbl().dbl() - r.x) - (s1 * j).dbl().dbl()).norm().dbl());
  r.z = coord_norm(((p.z + q.z).sqr() - z1z1 - z2z2) * h);
  if (q_inf) r = p;
  if (p_inf) r = q;
  return r;
}

// add-2007-bl for ladder loops: the generic case only (see jac_add_mixed_generic); p at infinity is the CALLER's
// business (ladders that start from the identity carry a `started` flag instead of testing Z)
template <class F>
TC_JAC_ATTR Jac<F> jac_add_generic(const Jac<F>& p, const Jac<F>& q, bool& exc) {
  F z1z1 = p.z.sqr();
  F z2z2 = q.z.sqr();
  F u1 = p.x * z2z2;
  F u2 = q.x * z1z1;
  F s1 = p.y * q.z * z2z2;
  F s2 = q.y * p.z * z1z1;
  F h = u2 - u1;
  F rr = s2 - s1;
  exc = exc || maybe_zero56(h) || maybe_zero56(p.z) || maybe_zero56(q.z);
  F hh = h.sqr();
  F j = h * hh;
  F v = u1 * hh;
  Jac<F> r;
  r.x = coord_norm((rr.sqr() - j - v.dbl()).norm().dbl().dbl());
  r.y = coord_norm((rr * (v.dbl().dbl() - r.x) - (s1 * j).dbl().dbl()).norm().dbl());
  r.z = coord_norm(((p.z + q.z).sqr() - z1z1 - z2z2) * h);
  return r;
}

template <class F>
TC_HD Jac<F> jac_neg(const Jac<F>& p) {
  return Jac<F>{p.x, (-p.y), p.z};
}

template <class F>
TC_HD_NOINLINE Affine<F> jac_to_affine(const Jac<F>& p) {
  if (p.is_inf()) return Affine<F>::infinity();
  F zi = p.z.inv();
  F zi2 = zi.sqr();
  return Affine<F>{p.x * zi2, p.y * zi2 * zi, false};
}

// n <= N Jacobian points to affine with ONE inversion (Montgomery's trick); infinity stays infinity
template <class F, int N>
TC_HD void jac_batch_to_affine(const Jac<F>* in, Affine<F>* out, int n) {
  F pre[N];
  bool inf[N];
  F acc = F::one();
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    inf[i] = in[i].is_inf();
    pre[i] = acc;
    acc = acc * F::select(inf[i], F::one(), in[i].z);
  }
  F inv = acc.inv();
  TC_NOUNROLL for (int i = n - 1; i >= 0; i--) {
    const F zi = inv * pre[i];
    inv = inv * F::select(inf[i], F::one(), in[i].z);
    const F zi2 = zi.sqr();
    out[i] = Affine<F>{in[i].x * zi2, in[i].y * (zi2 * zi), inf[i]};
    if (inf[i]) out[i] = Affine<F>::infinity();
  }
}

// Brings n <= CAP Jacobian points to ONE common Z without an inversion:
//     out[i] = (X_i F_i^2, Y_i F_i^3),   F_i = prod_{k != i} Z_k,   Zc = prod_k Z_k  (returned)
// (X_i, Y_i, Z_i) ~ (X_i F_i^2, Y_i F_i^3, Zc), so every out[i] is an AFFINE point of the
// isomorphic curve y^2 = x^3 + b Zc^6.  The a = 0 group law never reads b: a ladder can run on
// those affine points with mixed additions, and its result (X, Y, Z) is (X, Y, Z Zc) on the
// original curve.  Points at infinity stay flagged and count as Z = 1.
template <class F, int CAP = 16>
TC_HD F jac_batch_to_common_z(const Jac<F>* in, Affine<F>* out, int n) {
  F pre[CAP];
  bool inf[CAP];
  F acc = F::one();
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    inf[i] = in[i].is_inf();
    pre[i] = acc;
    acc = acc * F::select(inf[i], F::one(), in[i].z);
  }
  const F zc = acc;
  F suf = F::one();
  TC_NOUNROLL for (int i = n - 1; i >= 0; i--) {
    const F f = pre[i] * suf;
    suf = suf * F::select(inf[i], F::one(), in[i].z);
    const F f2 = f.sqr();
    out[i] = Affine<F>{coord_norm(in[i].x * f2), coord_norm(in[i].y * (f2 * f)), inf[i]};
  }
  return zc;
}
// an affine point of the original curve on the curve scaled by zc (zc2 = zc^2, zc3 = zc^3)
template <class F>
TC_HD Affine<F> affine_scale_z(const Affine<F>& p, const F& zc2, const F& zc3) {
  return Affine<F>{coord_norm(p.x * zc2), coord_norm(p.y * zc3), p.inf};
}

// k * P for a scalar given as nwords little-endian u32 words (bits above nbits are zero).
// MSB-first double-and-add with mixed additions (the CurveAffine::mul shape of group 0.6).
// Control flow depends on the scalar: use when the scalar is wave-uniform (shared secret
// key share, cofactor) so that no lane diverges.
template <class F, class SC>
TC_HD Jac<F> jac_mul_affine_uniform(const Affine<F>& p, SC word, int nbits) {
  Jac<F> acc = Jac<F>::infinity();
  TC_NOUNROLL for (int i = nbits - 1; i >= 0; i--) {
    acc = jac_dbl(acc);
    if ((word(i >> 5) >> (i & 31)) & 1u) acc = jac_add_mixed(acc, p);
  }
  return acc;
}

// [k] pa for a WAVE-UNIFORM 64-bit k whose leading one is bit `top` (fixed scalars: |x|, the cofactor constants):
// top doublings, a mixed addition per further one bit.  The loop adds with the branch-free generic formula; a lane
// that may have met a special case (pa at infinity or of small order: P = +-Q can then happen) redoes its ladder with
// jac_add_mixed.  The caller multiplies Z by the base's when pa is (X, Y) of a Jacobian point.
template <class F>
TC_HD_NOINLINE Jac<F> jac_ladder_uniform_safe(const Affine<F>& pa, uint64_t k, int top) {
  Jac<F> acc = Jac<F>::from_affine(pa);
  TC_NOUNROLL for (int bit = top - 1; bit >= 0; bit--) {
    acc = jac_dbl(acc);
    if ((k >> bit) & 1ull) acc = jac_add_mixed(acc, pa);
  }
  return acc;
}
template <class F>
TC_HD Jac<F> jac_ladder_uniform(const Affine<F>& pa, uint64_t k, int top) {
  Jac<F> acc = Jac<F>::from_affine(pa);
  bool exc = pa.inf;
  TC_NOUNROLL for (int bit = top - 1; bit >= 0; bit--) {
    tc_fair();
    acc = jac_dbl(acc);
    if ((k >> bit) & 1ull) acc = jac_add_mixed_generic(acc, pa, exc);
  }
  if (wave_any(exc)) acc = Jac<F>::select(exc, jac_ladder_uniform_safe(pa, k, top), acc);
  return acc;
}

// y^2 == x^3 + b ?
template <class F>
TC_HD bool affine_on_curve(const Affine<F>& p, const F& b) {
  if (p.inf) return true;
  return p.y.sqr() == p.x.sqr() * p.x + b;
}

TC_HD Fq g1_b() { return Fq::from_limbs(FQL_B1); }
TC_HD Fq2 g2_b() { return Fq2::make(g1_b(), g1_b()); }

TC_HD Affine<Fq> g1_generator() {
  return Affine<Fq>{Fq::from_limbs(G1_GENL_X), Fq::from_limbs(G1_GENL_Y), false};
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1Jac = Jac<Fq>;
using G2Jac = Jac<Fq2>;

}  // namespace tc
