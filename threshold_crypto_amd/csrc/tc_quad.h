// The pairing check on FOUR lanes per job (a "quad": lanes 4q .. 4q+3 of a wave).
//
// The lane-pair layout of tc_common.h gives an Fq2 value to two lanes.  A product pairing check carries two Miller
// points, two G1 operands and an Fq12 accumulator: on one lane pair that is more than 512 registers of live state at
// the peaks of the loop (profiles/r03_*: 1 931 spilled registers at 256, still 626 at 512), and the second wave of
// the SIMD -- which the multiplier needs to issue at full rate -- pays for it in scratch traffic.  Here a job owns two
// lane pairs, "pair A" (lanes 0, 1 of the quad) and "pair B" (lanes 2, 3):
//   * the two pairings of the check split naturally: pair A runs the point arithmetic of e(a, b), pair B that of
//     e(-c, d) -- the same instruction stream on different data, no exchange at all;
//   * an Fq12 value is DISTRIBUTED: pair A holds its c0 (an Fq6), pair B its c1.  Products run as two Fq6-sized halves
//     at a time (a0 b0 on A while a1 b1 runs on B; the Karatsuba middle term with three of its six Fq2 products on
//     each pair), squarings as the two Fq6 products of the complex method, one per pair;
//   * the long chains of the final exponentiation run on Karabina's compressed form with (z2, z3) on pair A and
//     (z4, z5) on pair B: three Fq2 squarings per step and pair instead of six.
// Values cross between the pairs with one DPP move per limb (quad_perm [2, 3, 0, 1]); every exchange is symmetric (both
// pairs give and take in the same instruction).  Per lane the live state halves; the lanes of a check double, so the
// batch of 65 536 checks is 4 096 waves (two rounds of two waves per SIMD).
//
// The g++ build (tests/hostsim) keeps an Fq2 whole (TC_PAIR = 0) and runs the two PAIRS of a quad as two host threads
// that meet at every exchange -- the same code, the same order of operations, with the interval analysis of
// -DTC_BOUND_CHECK carried through the exchanges.
#pragma once
#include "tc_pairing.h"

#if !defined(__HIPCC__)
#include <condition_variable>
#include <mutex>
#endif

namespace tc {

constexpr int kQuadLanes = 2 * kG2Lanes;  // lanes per check in the quad kernels (device: 4)

#if defined(__HIPCC__)
TC_HD bool quad_hi() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (threadIdx.x & 2u) != 0;
#else  // host pass of hipcc: only has to parse
  return false;
#endif
}
TC_HD int32_t quad_xchg(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
#else
  return v;
#endif
}
TC_HD Fq quad_other(const Fq& v) {
  Fq r = v;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = quad_xchg(v.l[i]);
  return r;
}
TC_HD Fq2 quad_other(const Fq2& v) { return Fq2{quad_other(v.m)}; }
TC_HD bool quad_all(bool c) {
  const int32_t mine = c ? 1 : 0;
  return (mine & quad_xchg(mine)) != 0;
}
#else
// g++ test build: the two pairs of a quad are two threads; an exchange is a rendezvous
struct QuadSim {
  std::mutex m;
  std::condition_variable cv;
  const void* slot[2] = {nullptr, nullptr};
  int count = 0;
  unsigned long gen = 0;
  void barrier(std::unique_lock<std::mutex>& lk) {
    const unsigned long g = gen;
    if (++count == 2) {
      count = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};
inline thread_local QuadSim* tl_quad_sim = nullptr;
inline thread_local int tl_quad_hi = 0;
inline bool quad_hi() { return tl_quad_hi != 0; }
template <class T>
inline T quad_exchange_host(const T& v) {
  QuadSim* s = tl_quad_sim;
  std::unique_lock<std::mutex> lk(s->m);
  s->slot[tl_quad_hi] = &v;
  s->barrier(lk);
  T r = *static_cast<const T*>(s->slot[1 - tl_quad_hi]);
  s->barrier(lk);  // nobody leaves (and drops its value) before the partner has copied it
  return r;
}
inline Fq quad_other(const Fq& v) { return quad_exchange_host(v); }
inline Fq2 quad_other(const Fq2& v) { return quad_exchange_host(v); }
inline bool quad_all(bool c) {
  const int mine = c ? 1 : 0;
  return (mine & quad_exchange_host(mine)) != 0;
}
#endif

TC_HD Fq6 quad_other(const Fq6& v) { return Fq6{quad_other(v.c0), quad_other(v.c1), quad_other(v.c2)}; }
// (the predicate is always "which pair of the quad": lane identity, tc_field.h select_lane)
TC_HD Fq6 fq6_select(bool c, const Fq6& a, const Fq6& b) {
  return Fq6{Fq2::select_lane(c, a.c0, b.c0), Fq2::select_lane(c, a.c1, b.c1), Fq2::select_lane(c, a.c2, b.c2)};
}

// An Fq12 value c0 + c1 w spread over the quad: pair A holds c0, pair B holds c1.
struct QFq12 {
  Fq6 h;
  TC_HD static QFq12 one() { return QFq12{fq6_select(quad_hi(), Fq6::zero(), Fq6::one())}; }
  TC_HD static QFq12 from(const Fq12& f) { return QFq12{fq6_select(quad_hi(), f.c1, f.c0)}; }  // both pairs hold f
  TC_HD QFq12 conj() const { return QFq12{fq6_select(quad_hi(), (-h).norm(), h)}; }
  TC_HD QFq12 reduced() const { return QFq12{h.reduce_value()}; }
  // the whole value on both pairs
  TC_HD Fq12 gather() const {
    const Fq6 o = quad_other(h);
    return Fq12{fq6_select(quad_hi(), o, h), fq6_select(quad_hi(), h, o)};
  }
  TC_HD bool is_one() const {
    const bool mine = quad_hi() ? (h == Fq6::zero()) : (h == Fq6::one());
    return quad_all(mine);
  }
};

// the Karatsuba middle term of an Fq12 product, (s0 + s1 v + s2 v^2)(m0 + m1 v + m2 v^2) with both operands known to
// both pairs: pair A computes the three diagonal products, pair B the three cross products, pair B assembles -- the
// value returned is meaningful on pair B only
TC_HD Fq6 q_middle_on_b(const Fq6& s, const Fq6& m) {
  const bool hi = quad_hi();
  const Fq2 x0 = Fq2::select_lane(hi, s.c1 + s.c2, s.c0), y0 = Fq2::select_lane(hi, m.c1 + m.c2, m.c0);
  const Fq2 x1 = Fq2::select_lane(hi, s.c0 + s.c1, s.c1), y1 = Fq2::select_lane(hi, m.c0 + m.c1, m.c1);
  const Fq2 x2 = Fq2::select_lane(hi, s.c0 + s.c2, s.c2), y2 = Fq2::select_lane(hi, m.c0 + m.c2, m.c2);
  const Fq2 p0 = x0 * y0, p1 = x1 * y1, p2 = x2 * y2;  // A: s0 m0, s1 m1, s2 m2;  B: the (1,2), (0,1), (0,2) cross products
  const Fq2 q0 = quad_other(p0), q1 = quad_other(p1), q2 = quad_other(p2);
  // on B (q = A's diagonal terms): the Fq6 product as in Fq6::operator*
  Fq6 r;
  r.c0 = q0 + (p0 - q1 - q2).mul_xi();
  r.c1 = p1 - q0 - q1 + q2.mul_xi();
  r.c2 = p2 - q0 - q2 + q1;
  return TC_FQ6_OUT(r);
}

// a * b
TC_HD QFq12 q_mul(const QFq12& a, const QFq12& b) {
  const bool hi = quad_hi();
  const Fq6 t = a.h * b.h;  // A: a0 b0;  B: a1 b1
  const Fq6 sa = (a.h + quad_other(a.h)).norm(), sb = (b.h + quad_other(b.h)).norm();
  const Fq6 m2 = q_middle_on_b(sa, sb);
  const Fq6 to = quad_other(t);
  // A: c0 = a0 b0 + v a1 b1;  B: c1 = (a0 + a1)(b0 + b1) - a0 b0 - a1 b1
  const Fq6 ra = t + to.mul_by_v();
  const Fq6 rb = m2 - to - t;
  return QFq12{fq6_select(hi, rb, ra).reduce_value()};
}

// a^2 (complex method over Fq6: the two Fq6 products on the two pairs at once)
TC_HD QFq12 q_sqr(const QFq12& a) {
  const bool hi = quad_hi();
  const Fq6 o = quad_other(a.h);
  // A: (c0 + c1)(c0 + v c1);  B: c0 c1
  const Fq6 lhs = fq6_select(hi, o, (a.h + o).norm());
  const Fq6 rhs = fq6_select(hi, a.h, (a.h + o.mul_by_v()).norm());
  const Fq6 pr = lhs * rhs;
  const Fq6 po = quad_other(pr);  // A receives c0 c1
  const Fq6 ra = pr - po - po.mul_by_v();
  const Fq6 rb = pr + pr;
  return QFq12{fq6_select(hi, rb, ra).norm()};
}

// f * l for l whose c1 is (0, *, *) -- the product of two Miller-loop lines, known to both pairs
TC_HD QFq12 q_mul_by_line_product(const QFq12& f, const Fq12& l) {
  const bool hi = quad_hi();
  const Fq6 t = f.h * fq6_select(hi, l.c1, l.c0);  // A: c0 l.c0;  B: c1 l.c1
  const Fq6 s = (f.h + quad_other(f.h)).norm(), m = (l.c0 + l.c1).norm();
  const Fq6 m2 = q_middle_on_b(s, m);
  const Fq6 to = quad_other(t);
  const Fq6 ra = t + to.mul_by_v();
  const Fq6 rb = m2 - to - t;
  return QFq12{fq6_select(hi, rb, ra).norm()};
}

// The product of the two pairs' lines (d0 + d1 v + d4 v w on each pair, already scaled by the pair's G1 point): three
// of the six Fq2 products on each pair, the sparse result on both (Fq12::line_product has the formulas).
struct QLine {
  Fq2 d0, d1, d4;
};
TC_HD Fq12 q_line_product(const QLine& mine) {
  const bool hi = quad_hi();
  const QLine e{quad_other(mine.d0), quad_other(mine.d1), quad_other(mine.d4)};
  const QLine& d = mine;  // the product is symmetric in the two lines: "d" is this pair's, "e" the other's
  // A: d0 e0, d1 e1, d4 e4;  B: (d0 + d1)(e0 + e1), (d0 + d4)(e0 + e4), (d1 + d4)(e1 + e4)
  const Fq2 p0 = Fq2::select_lane(hi, d.d0 + d.d1, d.d0) * Fq2::select_lane(hi, e.d0 + e.d1, e.d0);
  const Fq2 p1 = Fq2::select_lane(hi, d.d0 + d.d4, d.d1) * Fq2::select_lane(hi, e.d0 + e.d4, e.d1);
  const Fq2 p2 = Fq2::select_lane(hi, d.d1 + d.d4, d.d4) * Fq2::select_lane(hi, e.d1 + e.d4, e.d4);
  const Fq2 o0 = quad_other(p0), o1 = quad_other(p1), o2 = quad_other(p2);
  const Fq2 t0 = Fq2::select_lane(hi, o0, p0), t1 = Fq2::select_lane(hi, o1, p1), t3 = Fq2::select_lane(hi, o2, p2);
  const Fq2 x01 = Fq2::select_lane(hi, p0, o0), x04 = Fq2::select_lane(hi, p1, o1), x14 = Fq2::select_lane(hi, p2, o2);
  Fq12 r;
  r.c0 = Fq6{(t0 + t3.mul_xi()).norm(), x01 - t0 - t1, t1};
  r.c1 = Fq6{Fq2::zero(), x04 - t0 - t3, (x14 - t1 - t3).norm()};
  return r;
}

// a^(q^k), k in {1, 2, 3}: every coefficient times its constant (pair A's first constant is 1)
TC_HD_NOINLINE QFq12 q_frobenius(const QFq12& a, int k) {
  const bool hi = quad_hi();
  const bool cj = (k & 1);
  const Fq2 g0 = Fq2::select_lane(hi, frob_coeff(k, 1), Fq2::one());
  const Fq2 g1 = Fq2::select_lane(hi, frob_coeff(k, 3), frob_coeff(k, 2));
  const Fq2 g2 = Fq2::select_lane(hi, frob_coeff(k, 5), frob_coeff(k, 4));
  QFq12 r;
  r.h.c0 = (cj ? a.h.c0.conj() : a.h.c0) * g0;
  r.h.c1 = (cj ? a.h.c1.conj() : a.h.c1) * g1;
  r.h.c2 = (cj ? a.h.c2.conj() : a.h.c2) * g2;
  return r;
}

// 1 / a:  t = (c0^2 - v c1^2)^-1,  (c0 t, -c1 t)
TC_HD_NOINLINE QFq12 q_inv(const QFq12& a) {
  const bool hi = quad_hi();
  const Fq6 sq = a.h.sqr();  // A: c0^2;  B: c1^2
  const Fq6 so = quad_other(sq);
  const Fq6 d = fq6_select(hi, so - sq.mul_by_v(), sq - so.mul_by_v());  // the same value on both pairs
  const Fq6 t = d.inv();
  const Fq6 p = a.h * t;
  return QFq12{fq6_select(hi, (-p).norm(), p)};
}

// ---- Miller loop ---------------------------------------------------------------------------------------------------------
// Every pair brings ONE pairing of the product: its G1 point p (pair B: already negated) and its G2 point q.  A pair with
// an operand at infinity contributes the constant line 1 (pairing 0.16's miller_loop skips such pairs).
TC_HD QLine q_scaled_line(const LineCoeffs& l, const G1Affine& p, bool skip) {
  QLine r{l.c2, l.c1.scale(p.x), l.c0.scale(p.y)};
  r.d0 = Fq2::select(skip, Fq2::one(), r.d0);
  r.d1 = Fq2::select(skip, Fq2::zero(), r.d1);
  r.d4 = Fq2::select(skip, Fq2::zero(), r.d4);
  return r;
}
TC_HD QFq12 q_miller_loop(const G1Affine& p, const G2Affine& q) {
  const bool skip = p.inf || q.inf;
  G2Jac r{q.x, q.y, Fq2::one().dbl().norm()};  // (x : y : 1), third coordinate doubled (tc_pairing.h)
  QFq12 f = QFq12::one();
  const uint64_t xs = BLS_X_ABS >> 1;
  TC_NOUNROLL for (int i = 61; i >= 0; i--) {  // bit 62 is the leading one
    tc_fair();
    {
      const LineCoeffs l = miller_doubling_step(r);
      const Fq12 lp = q_line_product(q_scaled_line(l, p, skip));
      if (i == 61) f = QFq12::from(lp);
      else f = q_mul_by_line_product(f, lp);
    }
    if ((xs >> i) & 1ull) {
      const LineCoeffs l = miller_addition_step(r, q);
      const Fq12 lp = q_line_product(q_scaled_line(l, p, skip));
      f = q_mul_by_line_product(f, lp);
    }
    f = q_sqr(f);
  }
  const LineCoeffs l = miller_doubling_step(r);
  f = q_mul_by_line_product(f, q_line_product(q_scaled_line(l, p, skip)));
  return f.conj();  // x < 0
}

// ---- final exponentiation ----------------------------------------------------------------------------------------------
// Karabina's compressed form across the quad: pair A holds (z2, z3), pair B holds (z4, z5) (tc_tower.h CycloCompressed
// has the numbering); a squaring is the same three Fq2 squarings on both pairs and one exchange of the two results.
struct QCyclo {
  Fq2 a, b;  // A: z2, z3;  B: z4, z5
  // from the distributed Fq12 (A: c0 = (z0, z4, z3), B: c1 = (z2, z1, z5)): A needs z2, B needs z4 -- one exchange
  TC_HD static QCyclo from(const QFq12& f) {
    const bool hi = quad_hi();
    const Fq2 give = Fq2::select_lane(hi, f.h.c0, f.h.c1);  // A gives z4 (its c0.c1), B gives z2 (its c1.c0)
    const Fq2 got = quad_other(give);
    return QCyclo{got, f.h.c2};  // A: (z2, z3 = c0.c2);  B: (z4, z5 = c1.c2)
  }
  template <bool REDUCE>
  TC_CYCLO_ATTR QCyclo sqr_t() const {
    const bool hi = quad_hi();
    const Fq2 a2 = a.sqr(), b2 = b.sqr();
    const Fq2 t_even = (b2.mul_xi() + a2).norm();  // A: t0;  B: t2
    const Fq2 t_odd = (a + b).sqr() - a2 - b2;     // A: t1;  B: t3
    const Fq2 o_even = quad_other(t_even), o_odd = quad_other(t_odd);
    // A: z2' = 3 xi t3 + 2 z2, z3' = 3 t2 - 2 z3;   B: z4' = 3 t0 - 2 z4, z5' = 3 t1 + 2 z5
    const Fq2 u = Fq2::select_lane(hi, o_even, o_odd.mul_xi().norm());
    const Fq2 v = Fq2::select_lane(hi, o_odd, o_even);
    const Fq2 sa = Fq2::select_lane(hi, -a, a), sb = Fq2::select_lane(hi, b, -b);
    QCyclo r;
    r.a = (u + sa).dbl() + u;
    r.b = (v + sb).dbl() + v;
    if (REDUCE) {
      r.a = r.a.reduce_value();
      r.b = r.b.reduce_value();
    } else {
      r.a = r.a.norm();
      r.b = r.b.norm();
    }
    return r;
  }
  TC_HD QCyclo reduced() const { return QCyclo{a.reduce_value(), b.reduce_value()}; }
  // the four coefficients on both pairs, in tc_tower.h's form
  TC_HD CycloCompressed gather() const {
    const bool hi = quad_hi();
    const Fq2 oa = quad_other(a), ob = quad_other(b);
    return CycloCompressed{Fq2::select_lane(hi, oa, a), Fq2::select_lane(hi, ob, b), Fq2::select_lane(hi, a, oa), Fq2::select_lane(hi, b, ob)};
  }
};

// f^|x| followed by conjugation (x < 0), f in the cyclotomic subgroup; the structure of tc_pairing.h
// cyclotomic_exp_by_x: one chain of compressed squarings up to the third set bit, the three saved powers decompressed
// with one inversion (by both pairs, on gathered values), the six remaining squarings on full elements
TC_EXPX_ATTR QFq12 q_exp_by_x(const QFq12& f, uint64_t x_arg) {
  const uint64_t x = wave_uniform(x_arg);
  QCyclo saved[3];
  int bit = 0;
  {
    QCyclo c = QCyclo::from(f);
    int ns = 0;
    int since = 0;
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
  QFq12 pw[3];
  {
    CycloCompressed full[3];
    TC_NOUNROLL for (int i = 0; i < 3; i++) full[i] = saved[i].gather();
    Fq12 out[3];
    cyclotomic_decompress3(full, out);
    TC_NOUNROLL for (int i = 0; i < 3; i++) pw[i] = QFq12::from(out[i]);
  }
  QFq12 r = q_mul(q_mul(pw[0], pw[1]), pw[2]);
  QFq12 t = pw[2];
  TC_NOUNROLL for (bit++; bit < 64; bit++) {
    tc_fair();
    t = q_sqr(t).reduced();
    if ((x >> bit) & 1ull) r = q_mul(r, t);
  }
  return r.conj();
}

// f^(3 (q^12 - 1) / r): the chain of tc_pairing.h final_exponentiation on distributed values
TC_HD_NOINLINE QFq12 q_final_exponentiation(const QFq12& f) {
  QFq12 r = q_mul(f.conj(), q_inv(f));  // f^(q^6 - 1)
  r = q_mul(q_frobenius(r, 2), r);      // ^(q^2 + 1)
  const uint64_t x = BLS_X_ABS;
  QFq12 y0 = q_sqr(r).reduced();
  QFq12 y1 = q_exp_by_x(y0, x);
  QFq12 y2 = q_exp_by_x(y1, x >> 1);
  QFq12 y3 = r.conj();
  y1 = q_mul(q_mul(y1, y3).conj(), y2);
  y2 = q_exp_by_x(y1, x);
  y3 = q_exp_by_x(y2, x);
  y1 = y1.conj();
  y3 = q_mul(y3, y1);
  y1 = q_frobenius(y1.conj(), 3);
  y2 = q_frobenius(y2, 2);
  y1 = q_mul(y1, y2);
  y2 = q_mul(q_mul(q_exp_by_x(y3, x), y0), r);
  y1 = q_mul(y1, y2);
  y2 = q_frobenius(y3, 1);
  return q_mul(y1, y2);
}

// e(a, b) == e(c, d) with pair A given (a, b) and pair B given (c, d): each pair passes ITS G1 and G2 operand
TC_HD bool q_pairing_check(const G1Affine& g1, const G2Affine& g2) {
  const bool hi = quad_hi();
  const G1Affine p{g1.x, Fq::select_lane(hi, (-g1.y).norm(), g1.y), g1.inf};  // pair B brings e(-c, d)
  const QFq12 f = q_miller_loop(p, g2);
  return q_final_exponentiation(f).is_one();
}

// The job body: this PAIR's two operands come through IO objects (tc_jobs.h DirectIO, tc_stage.h WaveRowIO with a pair as
// the staging unit); every lane of the wave runs it, live or not (the exchanges are wave-wide instructions).
template <class IO1, class IO2>
TC_HD uint8_t job_pairing_check_quad_io(bool live, IO1& g1io, IO2& g2io) {
  G1Affine p = G1Affine::infinity();
  G2Affine q = G2Affine::infinity();
  bool ok = live;
  const uint8_t* e = g1io.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, p);
  e = g2io.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, q);
  ok = quad_all(ok);
  if (!ok) {  // an operand of the check did not decode: the job fails; the lanes still walk the (empty) product
    p = G1Affine::infinity();
    q = G2Affine::infinity();
  }
  const bool r = q_pairing_check(p, q);
  return (ok && r) ? 1 : 0;
}

}  // namespace tc
