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

// The Miller-loop point T runs in HOMOGENEOUS projective coordinates (X : Y : Z), stored with the
// third coordinate doubled (zt = 2 Z) so that the twist constant b' = 4 (1 + u) turns into small
// multiples:  9 b' Z^2 = 9 xi zt^2,  27 b'^2 Z^4 = 3 (3 xi zt^2)^2.
//   doubling (3M + 6S; Costello-Lange-Naehrig / Aranha et al. 2011, a = 0):
//     X3 = 2 XY (Y^2 - 9 b' Z^2),  Y3 = (Y^2 + 9 b' Z^2)^2 - 108 b'^2 Z^4,  Z3 = 8 Y^3 Z
//     tangent at T, evaluated at P = (xP, yP), up to a factor of Fq2 (which the final
//     exponentiation removes):   2 Y Z yP w^3  -  3 X^2 xP w^2  +  (Y^2 - 3 b' Z^2)
//   mixed addition of the affine Q (11M + 2S):  theta = Y - y2 Z, lambda = X - x2 Z, ... and the
//     chord  lambda yP w^3 - theta xP w^2 + (theta x2 - lambda y2)
// (the pairing crate's Jacobian steps, Algorithms 26/27 of eprint 2010/354, cost 3M + 8S and
// 7M + 8S; only the boolean of the product check is observable, so any correct line will do.)
// Small multiples are applied after the products and sums are carry-normalised where the bound
// analysis asks (tc_curve.h).
TC_MILLER_ATTR LineCoeffs miller_doubling_step(G2Jac& r) {
  const Fq2 B = r.y.sqr();
  const Fq2 C = r.z.sqr();
  const Fq2 xc = C.mul_xi();
  const Fq2 E = (xc.dbl() + xc).norm();  // 3 b' Z^2 = 3 xi zt^2
  const Fq2 F = (E.dbl() + E).norm();    // 9 b' Z^2
  const Fq2 XY = r.x * r.y;
  const Fq2 H = (r.y + r.z).sqr() - B - C;  // 2 Y zt
  const Fq2 J = r.x.sqr();
  LineCoeffs l;
  l.c0 = H;
  l.c1 = -(J.dbl() + J).dbl();
  l.c2 = (B - E).dbl().norm();
  r.x = (XY * (B - F)).dbl().norm();
  const Fq2 e4 = E.sqr().dbl().dbl().norm();  // 4 E^2
  r.y = ((B + F).sqr() - (e4.dbl() + e4)).norm();
  r.z = (B * H).dbl().dbl().norm();
  return l;
}

TC_MILLER_ATTR LineCoeffs miller_addition_step(G2Jac& r, const G2Affine& q) {
  const Fq2 X2 = r.x.dbl(), Y2 = r.y.dbl();
  const Fq2 theta = (Y2 - q.y * r.z).norm();
  const Fq2 lambda = (X2 - q.x * r.z).norm();
  const Fq2 C = theta.sqr();
  const Fq2 D = lambda.sqr();
  const Fq2 E = lambda * D;
  const Fq2 F = r.z * C;
  const Fq2 G = X2 * D;
  const Fq2 Hh = E + F - G.dbl();
  LineCoeffs l;
  l.c0 = lambda;
  l.c1 = -theta;
  l.c2 = (theta * q.x - lambda * q.y).norm();
  r.x = (lambda * Hh).norm();
  r.y = (theta * (G - Hh) - E * Y2).norm();
  r.z = (r.z * E).dbl().norm();
  return l;
}

// f *= line evaluated at the G1 point p  (the `ell` of pairing 0.16)
TC_HD void miller_ell(Fq12& f, const LineCoeffs& l, const G1Affine& p) {
  f = f.mul_by_014(l.c2, l.c1.scale(p.x), l.c0.scale(p.y));
}

// Where the loop finds its operands: the caller's arrays (registers).  (r03 experiment, profiles/r03_pairing_forms.txt: a
// provider that re-read the G1 coordinates from LDS and the G2 points from HBM rows wherever the loop uses them -- 112 of
// the loop's 280 live registers per lane no longer live -- made the kernel 4 % SLOWER; the provider interface stays.)
struct MillerArrayOps {
  const G1Affine* ps;
  const G2Affine* qs;
  TC_HD Fq px(int k) const { return ps[k].x; }
  TC_HD Fq py(int k) const { return ps[k].y; }
  TC_HD G2Affine q(int k) const { return qs[k]; }
};

// f *= line evaluated at the G1 point of pair k  (the `ell` of pairing 0.16)
template <class OPS>
TC_HD void miller_ell_ops(Fq12& f, const LineCoeffs& l, const OPS& ops, int k) {
  f = f.mul_by_014(l.c2, l.c1.scale(ops.px(k)), l.c0.scale(ops.py(k)));
}

// f *= product of the lines of all pairs.  Two lines are multiplied with each other first (6 Fq2
// products) and f by the result (17) -- 23 instead of the 2 x 13 of two sparse multiplications.
// `fresh` (wave-uniform): f is still 1, the product of the two lines simply becomes f.
template <int NP, class OPS>
TC_HD void miller_apply_lines(Fq12& f, const LineCoeffs* l, const OPS& ops, const bool* skip, bool fresh = false) {
  if (NP == 2) {
    if (!skip[0] && !skip[1]) {
      const Fq12 lp = Fq12::line_product(l[0].c2, l[0].c1.scale(ops.px(0)), l[0].c0.scale(ops.py(0)), l[1].c2,
                                         l[1].c1.scale(ops.px(1)), l[1].c0.scale(ops.py(1)));
      if (fresh) f = lp;
      else f = f.mul_by_line_product(lp);
      return;
    }
  }
  TC_UNROLL for (int k = 0; k < NP; k++)
    if (!skip[k]) miller_ell_ops(f, l[k], ops, k);
}

// Product Miller loop over NP pairs (NP = 2 for every check on the path).  skip[k]: pair k has an operand at infinity.
template <int NP, class OPS>
TC_HD Fq12 miller_loop_ops(const OPS& ops, const bool* skip) {
  Fq12 f = Fq12::one();
  G2Jac r[NP];
  TC_UNROLL for (int k = 0; k < NP; k++) {
    const G2Affine q = ops.q(k);
    r[k] = G2Jac{q.x, q.y, Fq2::one().dbl().norm()};  // (x : y : 1), third coordinate doubled
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
    tc_fair();
    TC_UNROLL for (int k = 0; k < NP; k++)
      if (!skip[k]) l[k] = miller_doubling_step(r[k]);
    miller_apply_lines<NP>(f, l, ops, skip, i == 61);
    if ((xs >> i) & 1ull) {
      TC_UNROLL for (int k = 0; k < NP; k++)
        if (!skip[k]) l[k] = miller_addition_step(r[k], ops.q(k));
      miller_apply_lines<NP>(f, l, ops, skip);
    }
    f = f.sqr();
  }
  TC_UNROLL for (int k = 0; k < NP; k++)
    if (!skip[k]) l[k] = miller_doubling_step(r[k]);
  miller_apply_lines<NP>(f, l, ops, skip);
  return f.conj();  // x < 0
}
template <int NP>
TC_HD Fq12 miller_loop(const G1Affine* ps, const G2Affine* qs) {
  bool skip[NP];
  TC_UNROLL for (int k = 0; k < NP; k++) skip[k] = ps[k].inf || qs[k].inf;
  return miller_loop_ops<NP>(MillerArrayOps{ps, qs}, skip);
}

// =====================================================================================================================
// The same product Miller loop as TWO stages that meet in memory (k_pairing.hip: k_miller_lines, k_miller_accumulate) --
// the split pairing 0.16 makes between G2Prepared and miller_loop (what /root/reference/src/lib.rs:109, :185, :511 reach
// through PEngine::pairing), carried one step further:
//   stage P  G2 point arithmetic only: the 68 steps (63 doublings, 5 additions) of BOTH Miller points.  Per step the two
//            lines are evaluated at their G1 points and multiplied WITH EACH OTHER (6 Fq2 products); the five non-trivial
//            Fq2 coefficients of that product are what goes to memory: lines prepared for THIS check.
//   stage M  the Fq12 accumulator only:  f <- f * (line product)  per step,  f <- f^2  per loop iteration.  Neither the
//            Miller points nor any operand of the check is live there.
// As ONE loop the live set was accumulator (84 registers per lane) + two Miller points (84) + four operands (112) before
// a single temporary: 2 238 spilled registers, 25.7 GB of scratch traffic per 65 536 checks.
constexpr int kMillerSteps = 68;         // 63 doublings + 5 additions (|x| >> 1 has five set bits below its leading one)
constexpr int kLineProductCoeffs = 5;    // c0 = (a0, a1, a2), c1 = (0, b1, b2)

// Where the prepared coefficients live.  Device (lane-pair build): this lane's column of the wave's row block -- word w
// of lane l at p[w * 64 + l], so every store / load instruction moves one full 256-byte row.  Host build (test harness): a
// plain array.
#if TC_PAIR
#if defined(__HIP_DEVICE_COMPILE__)
// (explicitly GLOBAL: a pointer that went through rows_after's empty asm is no longer known to point into global memory,
// and a generic pointer makes every access a flat_load that also has to wait for the LDS counter)
typedef __attribute__((address_space(1))) int32_t* RowPtr;
#else
typedef int32_t* RowPtr;
#endif
struct Fq2Rows {
  RowPtr p;
  TC_HD static Fq2Rows at(int32_t* q) { return Fq2Rows{(RowPtr)q}; }
  TC_HD void put(int k, const Fq2& v) const {
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) p[(k * FQ_LIMBS + i) * 64] = v.m.l[i];
  }
  TC_HD Fq2 get(int k) const {
    Fq2 r;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.m.l[i] = p[(k * FQ_LIMBS + i) * 64];
    return r;
  }
  // an Fq value both lanes of the pair hold (G1 coordinates): each lane keeps its own copy in its own column
  TC_HD void put_fq(int k, const Fq& v) const { put(k, Fq2{v}); }
  TC_HD Fq get_fq(int k) const { return get(k).m; }
};
#else
struct Fq2Rows {
  Fq2* p;
  TC_HD static Fq2Rows at(Fq2* q) { return Fq2Rows{q}; }
  TC_HD void put(int k, const Fq2& v) const { p[k] = v; }
  TC_HD Fq2 get(int k) const { return p[k]; }
  TC_HD void put_fq(int k, const Fq& v) const { p[k] = Fq2{v, Fq::zero()}; }
  TC_HD Fq get_fq(int k) const { return p[k].c0; }
};
#endif

// Row slots of a lane's block: the operands of the check (G1 coordinates, the affine G2 points: read where a step uses
// them instead of being held in 112 registers for the whole loop), the 68 scaled lines of the FIRST pair (stage P runs the
// two Miller points one after the other: pass 1 leaves the first pair's lines here, pass 2 multiplies each with the second
// pair's line of the same step), then the 68 x 5 line-product coefficients stage M reads.
constexpr int kRowPx = 0, kRowPy = 2, kRowQx = 4, kRowQy = 6, kRowFirst = 8, kRowProducts = kRowFirst + 3 * 68;
constexpr int kMillerRowSlots = kRowProducts + kMillerSteps * kLineProductCoeffs;

TC_HD Fq2Rows rows_after(const Fq2Rows& rows, const Fq2& v);

// one step of one Miller point: the point advances, its line is evaluated at the pair's G1 point: (e0, e1 v, e4 v w).  A
// skipped pair (an operand at infinity) contributes the unit line; its point arithmetic runs on a harmless stand-in.
template <bool ADD>
TC_HD void miller_point_step(G2Jac& r, int k, bool skip, const Fq2Rows& rows, Fq2& e0, Fq2& e1, Fq2& e4) {
  LineCoeffs l;
  if (ADD) {
    const Fq2Rows rq = rows_after(rows, r.z);
    l = miller_addition_step(r, G2Affine{rq.get(kRowQx + k), rq.get(kRowQy + k), false});
  } else {
    l = miller_doubling_step(r);
  }
  const Fq2Rows rp = rows_after(rows, l.c2);
  e0 = Fq2::select(skip, Fq2::one(), l.c2);
  e1 = Fq2::select(skip, Fq2::zero(), l.c1.scale(rp.get_fq(kRowPx + k)));
  e4 = Fq2::select(skip, Fq2::zero(), l.c0.scale(rp.get_fq(kRowPy + k)));
}

// pass 1: the first pair's line of step s goes to its rows
template <bool ADD>
TC_HD void miller_first_step(G2Jac& r, bool skip, const Fq2Rows& rows, int s) {
  Fq2 e0, e1, e4;
  miller_point_step<ADD>(r, 0, skip, rows, e0, e1, e4);
  rows.put(kRowFirst + 3 * s + 0, e0);
  rows.put(kRowFirst + 3 * s + 1, e1);
  rows.put(kRowFirst + 3 * s + 2, e4);
}

// pass 2: the second pair's line of step s times the first pair's (read back where the product uses it):
// (d0 + d1 v + d4 v w) (e0 + e1 v + e4 v w), the six Fq2 products of Fq12::line_product
template <bool ADD>
TC_HD void miller_second_step(G2Jac& r, bool skip, const Fq2Rows& rows, int s) {
  Fq2 e0, e1, e4;
  miller_point_step<ADD>(r, 1, skip, rows, e0, e1, e4);
  const int k = kRowProducts + s * kLineProductCoeffs, d = kRowFirst + 3 * s;
  const Fq2Rows ra = rows_after(rows, e4);
  const Fq2 d0 = ra.get(d + 0), d1 = ra.get(d + 1), d4 = ra.get(d + 2);
  const Fq2 t0 = d0 * e0;
  const Fq2 t1 = d1 * e1;
  const Fq2 t3 = d4 * e4;
  rows.put(k + 0, (t0 + t3.mul_xi()).norm());
  rows.put(k + 2, t1);
  rows.put(k + 1, ((d0 + d1) * (e0 + e1) - t0 - t1).norm());
  rows.put(k + 3, ((d0 + d4) * (e0 + e4) - t0 - t3).norm());
  rows.put(k + 4, ((d1 + d4) * (e1 + e4) - t1 - t3).norm());
}

template <int PASS>
TC_HD void miller_lines_pass(const G2Affine& q, bool skip, const Fq2Rows& rows) {
  G2Jac r{q.x, q.y, Fq2::one().dbl().norm()};  // (x : y : 1), third coordinate doubled
  const uint64_t xs = BLS_X_ABS >> 1;
  int s = 0;
  TC_NOUNROLL for (int i = 61; i >= 0; i--) {  // bit 62 is the leading one
    tc_fair();
    if (PASS == 0) miller_first_step<false>(r, skip, rows, s++);
    else miller_second_step<false>(r, skip, rows, s++);
    if ((xs >> i) & 1ull) {
      if (PASS == 0) miller_first_step<true>(r, skip, rows, s++);
      else miller_second_step<true>(r, skip, rows, s++);
    }
  }
  if (PASS == 0) miller_first_step<false>(r, skip, rows, s++);
  else miller_second_step<false>(r, skip, rows, s++);
}

// qs: the two affine G2 points, ps: the two G1 points (the second already negated)
TC_HD void miller_prepare_lines(const G1Affine* ps, const G2Affine* qs, const bool* skip, const Fq2Rows& rows) {
  TC_UNROLL for (int k = 0; k < 2; k++) {
    rows.put_fq(kRowPx + k, ps[k].x);
    rows.put_fq(kRowPy + k, ps[k].y);
    rows.put(kRowQx + k, qs[k].x);
    rows.put(kRowQy + k, qs[k].y);
  }
  miller_lines_pass<0>(qs[0], skip[0], rows);
  miller_lines_pass<1>(qs[1], skip[1], rows_after(rows, Fq2::zero()));
}

TC_HD Fq12 miller_line_product_at(const Fq2Rows& rows, int s) {
  const int k = kRowProducts + s * kLineProductCoeffs;
  Fq12 l;
  l.c0 = Fq6{rows.get(k + 0), rows.get(k + 1), rows.get(k + 2)};
  l.c1 = Fq6{Fq2::zero(), rows.get(k + 3), rows.get(k + 4)};
  return l;
}

// The loads of stage M are pinned BEHIND the value they are first needed after (an empty asm that makes the row pointer
// "depend" on that value): left alone, the scheduler hoists all 70 loads of a step to the top of the iteration, finds no
// registers for them beside the accumulator, and copies them from HBM straight into scratch.
TC_HD Fq2Rows rows_after(const Fq2Rows& rows, const Fq2& v) {  // (declared above)
#if TC_PAIR && defined(__HIP_DEVICE_COMPILE__)
  RowPtr p = rows.p;
  asm volatile("" : "+v"(p) : "v"(v.m.l[0]));
  return Fq2Rows{p};
#else
  (void)v;
  return rows;
#endif
}

// f <- f * (line product of step s), the coefficients streamed from their rows where the Karatsuba product over Fq6 uses
// them: (a0, a1, a2) for t0 = f.c0 * l.c0, (b1, b2) for t1 = f.c1 * l.c1, and all five once more for the middle product
// -- 70 more words per step out of L2 instead of 70 registers held across eleven multiplications.
TC_FQ12_ATTR Fq12 miller_mul_by_line_product_rows(const Fq12& f, const Fq2Rows& rows, int s) {
  const int k = kRowProducts + s * kLineProductCoeffs;
  Fq6 t0;
  {
    const Fq6 a{rows.get(k + 0), rows.get(k + 1), rows.get(k + 2)};
    t0 = f.c0 * a;
  }
  Fq6 t1;
  {
    const Fq2Rows r1 = rows_after(rows, t0.c2);
    const Fq2 b1 = r1.get(k + 3), b2 = r1.get(k + 4);
    t1 = f.c1.mul_by_12(b1, b2);
  }
  Fq12 r;
  {
    const Fq2Rows r2 = rows_after(rows, t1.c2);
    const Fq6 m{r2.get(k + 0), (r2.get(k + 1) + r2.get(k + 3)).norm(), (r2.get(k + 2) + r2.get(k + 4)).norm()};
    r.c1 = (f.c0 + f.c1).norm() * m - t0 - t1;
  }
  r.c0 = t0 + t1.mul_by_v();
  return r.norm();
}

// stage M: the accumulator over the prepared line products (the loop of miller_loop_ops without its point arithmetic)
TC_HD Fq12 miller_accumulate(const Fq2Rows& rows) {
  const uint64_t xs = BLS_X_ABS >> 1;
  int s = 0;
  Fq12 f = miller_line_product_at(rows, s++);  // f = 1 times the first product
  TC_NOUNROLL for (int i = 61; i >= 0; i--) {
    tc_fair();
    if (i != 61) f = miller_mul_by_line_product_rows(f, rows_after(rows, f.c0.c0), s++);
    if ((xs >> i) & 1ull) f = miller_mul_by_line_product_rows(f, rows_after(rows, f.c0.c0), s++);
    f = f.sqr();
  }
  f = miller_mul_by_line_product_rows(f, rows_after(rows, f.c0.c0), s++);
  return f.conj();  // x < 0
}

// Three compressed cyclotomic elements back to Fq12 with ONE shared inversion (Montgomery's trick).
//   z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2),   or  2 z4 z5 / z3  when z2 = 0;
//   z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1.
// z2 = z3 = 0 only happens for the identity (then z4 = z5 = 0 as well): its denominator is
// replaced by 1 in the shared product and z1 = 0, z0 = 1 come out of the same formulas.
TC_HD_NOINLINE void cyclotomic_decompress3(const CycloCompressed* c, Fq12* out) {
  Fq2 num[3], den[3];
  bool dz[3];
  TC_NOUNROLL for (int i = 0; i < 3; i++) {
    const bool z2_zero = c[i].z2.is_zero();
    const Fq2 s4 = c[i].z4.sqr();
    const Fq2 n_a = (c[i].z5.sqr().mul_xi() + (s4.dbl() + s4) - c[i].z3.dbl()).norm();
    const Fq2 n_b = (c[i].z4 * c[i].z5).dbl();
    num[i] = Fq2::select(z2_zero, n_b, n_a);
    const Fq2 d = Fq2::select(z2_zero, c[i].z3, c[i].z2.dbl().dbl().norm());
    dz[i] = d.is_zero();
    den[i] = Fq2::select(dz[i], Fq2::one(), d);
  }
  const Fq2 p01 = den[0] * den[1];
  const Fq2 all = (p01 * den[2]).inv();
  const Fq2 t = all * den[2];
  Fq2 inv[3];
  inv[2] = all * p01;
  inv[1] = t * den[0];
  inv[0] = t * den[1];
  TC_NOUNROLL for (int i = 0; i < 3; i++) {
    const Fq2 z1 = Fq2::select(dz[i], Fq2::zero(), num[i] * inv[i]);
    const Fq2 m34 = c[i].z3 * c[i].z4;
    const Fq2 z0 = ((z1.sqr().dbl() + c[i].z2 * c[i].z5 - (m34.dbl() + m34)).norm().mul_xi() + Fq2::one()).norm();
    out[i].c0.c0 = z0;
    out[i].c0.c1 = c[i].z4;
    out[i].c0.c2 = c[i].z3;
    out[i].c1.c0 = c[i].z2;
    out[i].c1.c1 = z1;
    out[i].c1.c2 = c[i].z5;
  }
}

// f^|x| followed by conjugation (x < 0), for f in the cyclotomic subgroup.  |x| (and |x| >> 1) has
// six set bits p0 < ... < p5 with long gaps below p2 (16, 48, 57) and short ones above: the powers
// f^(2^p0), f^(2^p1), f^(2^p2) come from ONE chain of compressed squarings (6 Fq2 squarings each,
// Karabina) and are decompressed together; the remaining six squarings run on full elements
// (Granger-Scott, 9 each).
TC_EXPX_ATTR Fq12 cyclotomic_exp_by_x(const Fq12& f, uint64_t x_arg) {
  const uint64_t x = wave_uniform(x_arg);  // |x| or |x| >> 1, the same in every lane
#if defined(TC_PLAIN_CYCLO_EXP)  // experiment switch: 63 full squarings, no decompression
  Fq12 r0 = f;
  bool started = false;
  TC_NOUNROLL for (int i = 63; i >= 0; i--) {
    if (started) r0 = r0.cyclotomic_sqr();
    if ((x >> i) & 1ull) {
      if (started) r0 = r0 * f;
      started = true;
    }
  }
  return r0.conj();
#endif
  CycloCompressed saved[3];
  int bit = 0;
  {
    CycloCompressed c = CycloCompressed::from(f);
    int ns = 0;
    int since = 0;  // squarings since the value was last pulled back (tc_tower.h CycloCompressed::sqr_t)
    TC_NOUNROLL for (;; bit++) {
      tc_fair();
      if ((x >> bit) & 1ull) {
        saved[ns++] = since ? c.reduced() : c;
        if (ns == 3) break;
      }
      if (since + 1 == kCycloReduceEvery) {
        c = c.sqr_t<true>();
        since = 0;
      } else {
        c = c.sqr_t<false>();
        since++;
      }
    }
  }
  Fq12 pw[3];
  cyclotomic_decompress3(saved, pw);
  Fq12 r = pw[0] * pw[1] * pw[2];
  Fq12 t = pw[2];
  TC_NOUNROLL for (bit++; bit < 64; bit++) {
    tc_fair();
    t = t.cyclotomic_sqr();
    if ((x >> bit) & 1ull) r = r * t;
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
