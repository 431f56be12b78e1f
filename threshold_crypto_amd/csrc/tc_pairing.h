// Optimal-ate pairing product check on BLS12-381 (device side).
// Replaces the `PEngine::pairing(a, b) == PEngine::pairing(c, d)` pattern of
// /root/reference/src/lib.rs:109 (verify_g2), :185 (verify_decryption_share), :511
// (Ciphertext::verify).  The reference computes two full pairings and compares Fq12 values;
// only the boolean is observable, so the device evaluates
//     FE( ML(a, b) * ML(-c, d) ) == 1
// with ONE shared Miller-loop accumulator (shared Fq12 squarings) and ONE final
// exponentiation using cyclotomic squarings.  Pairs with a point at infinity contribute 1,
// as pairing 0.16's miller_loop skips them.
#pragma once
#include "tc_curve.h"

namespace tc {

struct LineCoeffs {
  Fq2 c0, c1, c2;
};

// Algorithm 26 of eprint 2010/354 (Jacobian doubling + tangent line), a = 0.  Small multiples are
// applied after the products and sums are carry-normalised where the bound analysis asks (tc_curve.h).
TC_MILLER_ATTR LineCoeffs miller_doubling_step(G2Jac& r) {
  Fq2 tmp0 = r.x.sqr();
  Fq2 tmp1 = r.y.sqr();
  Fq2 tmp2 = tmp1.sqr();
  Fq2 t3h = ((tmp1 + r.x).sqr() - tmp0 - tmp2).norm();  // tmp3 = 2 t3h
  Fq2 tmp4 = (tmp0.dbl() + tmp0).norm();
  Fq2 tmp6 = r.x + tmp4;
  Fq2 tmp5 = tmp4.sqr();
  Fq2 zsq = r.z.sqr();
  Fq2 nx = (tmp5 - t3h.dbl().dbl()).reduce_value();
  Fq2 nz = ((r.z + r.y).sqr() - tmp1 - zsq).norm();
  Fq2 ny = ((t3h.dbl() - nx) * tmp4 - tmp2.dbl().dbl().norm().dbl()).norm();
  LineCoeffs l;
  l.c1 = (-((tmp4 * zsq).dbl()));
  l.c2 = (tmp6.sqr() - tmp0 - tmp5 - tmp1.dbl().dbl()).norm();
  l.c0 = (nz * zsq).dbl();
  r.x = nx;
  r.y = ny;
  r.z = nz;
  return l;
}

// Algorithm 27 of eprint 2010/354 (mixed addition + chord line).
TC_MILLER_ATTR LineCoeffs miller_addition_step(G2Jac& r, const G2Affine& q) {
  Fq2 zsq = r.z.sqr();
  Fq2 ysq = q.y.sqr();
  Fq2 t0 = zsq * q.x;
  Fq2 t1 = ((q.y + r.z).sqr() - ysq - zsq) * zsq;
  Fq2 t2 = t0 - r.x;
  Fq2 t3 = t2.sqr();
  Fq2 t5q = t3 * t2;   // t5 = 4 t5q
  Fq2 t6 = (t1 - r.y.dbl()).norm();
  Fq2 t9 = t6 * q.x;
  Fq2 t7q = t3 * r.x;  // t7 = 4 t7q
  Fq2 nx = (t6.sqr() - (t5q + t7q.dbl()).norm().dbl().dbl()).norm();
  Fq2 nz = ((r.z + t2).norm().sqr() - zsq - t3).norm();
  Fq2 t10 = q.y + nz;
  Fq2 t8 = (t7q.dbl().dbl() - nx) * t6;
  Fq2 ny = (t8 - (r.y * t5q).dbl().dbl().norm().dbl()).norm();
  t10 = t10.sqr() - ysq - nz.sqr();
  t9 = (t9.dbl() - t10).norm();
  LineCoeffs l;
  l.c0 = nz.dbl();
  l.c1 = (-t6).dbl();
  l.c2 = t9;
  r.x = nx;
  r.y = ny;
  r.z = nz;
  return l;
}

// f *= line evaluated at the G1 point p  (the `ell` of pairing 0.16)
TC_HD void miller_ell(Fq12& f, const LineCoeffs& l, const G1Affine& p) {
  f = f.mul_by_014(l.c2, l.c1.scale(p.x), l.c0.scale(p.y));
}

// f *= product of the lines of all pairs.  Two lines are multiplied with each other first (6 Fq2
// products) and f by the result (17) -- 23 instead of the 2 x 13 of two sparse multiplications.
template <int NP>
TC_HD void miller_apply_lines(Fq12& f, const LineCoeffs* l, const G1Affine* ps, const bool* skip) {
  if (NP == 2) {
    if (!skip[0] && !skip[1]) {
      const Fq12 lp = Fq12::line_product(l[0].c2, l[0].c1.scale(ps[0].x), l[0].c0.scale(ps[0].y), l[1].c2,
                                         l[1].c1.scale(ps[1].x), l[1].c0.scale(ps[1].y));
      f = f.mul_by_line_product(lp);
      return;
    }
  }
  TC_UNROLL for (int k = 0; k < NP; k++)
    if (!skip[k]) miller_ell(f, l[k], ps[k]);
}

// Product Miller loop over NP pairs (NP = 2 for every check on the path).
template <int NP>
TC_HD Fq12 miller_loop(const G1Affine* ps, const G2Affine* qs) {
  Fq12 f = Fq12::one();
  G2Jac r[NP];
  bool skip[NP];
  TC_UNROLL for (int k = 0; k < NP; k++) {
    skip[k] = ps[k].inf || qs[k].inf;
    r[k] = G2Jac{qs[k].x, qs[k].y, Fq2::one()};
  }
  // nothing to accumulate (every pair has an identity operand): the product of pairings is 1.
  // (Also keeps the bare squarings of the loop below, whose outputs are only carry-normalised,
  // always paired with a line multiplication that reduces the value.)
  bool all_skipped = true;
  TC_UNROLL for (int k = 0; k < NP; k++) all_skipped = all_skipped && skip[k];
  if (all_skipped) return f;
  const uint64_t xs = BLS_X_ABS >> 1;
  LineCoeffs l[NP];
  TC_NOUNROLL for (int i = 61; i >= 0; i--) {  // bit 62 is the leading one
    TC_UNROLL for (int k = 0; k < NP; k++)
      if (!skip[k]) l[k] = miller_doubling_step(r[k]);
    miller_apply_lines<NP>(f, l, ps, skip);
    if ((xs >> i) & 1ull) {
      TC_UNROLL for (int k = 0; k < NP; k++)
        if (!skip[k]) l[k] = miller_addition_step(r[k], qs[k]);
      miller_apply_lines<NP>(f, l, ps, skip);
    }
    f = f.sqr();
  }
  TC_UNROLL for (int k = 0; k < NP; k++)
    if (!skip[k]) l[k] = miller_doubling_step(r[k]);
  miller_apply_lines<NP>(f, l, ps, skip);
  return f.conj();  // x < 0
}

// f^|x| followed by conjugation (x < 0), for f in the cyclotomic subgroup
TC_EXPX_ATTR Fq12 cyclotomic_exp_by_x(const Fq12& f, uint64_t x) {
  Fq12 r = f;
  bool started = false;
  TC_NOUNROLL for (int i = 63; i >= 0; i--) {
    if (started) r = r.cyclotomic_sqr();
    if ((x >> i) & 1ull) {
      if (started) r = r * f;
      started = true;
    }
  }
  return r.conj();
}

// f^(3 (q^12-1)/r): easy part, then the y0..y3 hard-part chain with cyclotomic squarings.
// (The factor 3 is a property of this chain; 3 does not divide r, so "== 1" is unaffected.)
TC_HD_NOINLINE Fq12 final_exponentiation(const Fq12& f) {
  Fq12 r = f.conj() * f.inv();  // f^(q^6-1)
  r = r.frobenius(2) * r;       // ^(q^2+1): r is now in the cyclotomic subgroup
  const uint64_t x = BLS_X_ABS;
  Fq12 y0 = r.cyclotomic_sqr();
  Fq12 y1 = cyclotomic_exp_by_x(y0, x);
  Fq12 y2 = cyclotomic_exp_by_x(y1, x >> 1);
  Fq12 y3 = r.conj();
  y1 = (y1 * y3).conj() * y2;
  y2 = cyclotomic_exp_by_x(y1, x);
  y3 = cyclotomic_exp_by_x(y2, x);
  y1 = y1.conj();
  y3 = y3 * y1;
  y1 = y1.conj().frobenius(3);
  y2 = y2.frobenius(2);
  y1 = y1 * y2;
  y2 = cyclotomic_exp_by_x(y3, x) * y0 * r;
  y1 = y1 * y2;
  y2 = y3.frobenius(1);
  return y1 * y2;
}

// e(a, b) == e(c, d)
TC_HD bool pairing_check(const G1Affine& a, const G2Affine& b, const G1Affine& c, const G2Affine& d) {
  G1Affine ps[2] = {a, G1Affine{c.x, (-c.y), c.inf}};
  G2Affine qs[2] = {b, d};
  Fq12 f = miller_loop<2>(ps, qs);
  return final_exponentiation(f) == Fq12::one();
}

}  // namespace tc
