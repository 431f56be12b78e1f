// Square roots in Fq and Fq2 (q = 3 mod 4), compressed-point decoding and subgroup checks.
//   * hash_g2 needs "is x^3 + b a square, and a root of it" (G2Affine::get_point_from_x of
//     pairing 0.16; /root/reference/src/lib.rs:693);
//   * from_bytes (/root/reference/src/lib.rs:140-146, 246-252) needs the checked decode of the
//     48 / 96-byte compressed forms: on-curve AND in the order-r subgroup.
// The reference's Fq2::sqrt is Algorithm 9 of eprint 2012/685 (two Fq2 exponentiations).  Which
// root comes out is not observable -- callers re-select by lexicographic order -- so the device
// takes roots through the norm instead: two Fq exponentiations, and a non-square is rejected
// without any (binary Jacobi symbol of the norm: the common case inside hash_g2's retry loop).
#pragma once
#include "tc_codec.h"
#include "tc_gls.h"
#include "tc_duo.h"

namespace tc {

// a^((q-3)/4).  For a square a != 0:  a * w is a square root of a and w = 1 / (a * w).
TC_HD_NOINLINE Fq fq_pow_qm3d4(const Fq& a) {
  return field_pow_fixed(a, [](int i) { return FQ_P_MINUS_3_DIV_4[i]; }, 379);
}

// root of a in Fq; false if a is not a square.  inv_root (optional) receives 1/root.
TC_HD bool fq_sqrt(const Fq& a, Fq& root, Fq* inv_root = nullptr) {
  Fq w = fq_pow_qm3d4(a);
  root = w * a;
  if (inv_root) *inv_root = w;
  return root.sqr() == a;
}

TC_HD Fq fq_half(const Fq& a) { return a * Fq::from_limbs(FQL_INV2); }

// a - b - borrow_in on 32-bit words; the device form is the hardware's borrow chain (v_sub_co / v_subb_co)
TC_HD uint32_t sub_borrow(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int bo;
  const uint32_t r = __builtin_subc(a, b, bin, &bo);
  bout = bo;
  return r;
#else
  const uint64_t t = (uint64_t)a - b - bin;
  bout = (uint32_t)(t >> 63);
  return (uint32_t)t;
#endif
}

// Legendre symbol (a / q) by the binary Jacobi algorithm on the canonical integer: subtractions with borrow, selects
// and a funnel shift on 12 x u32 (~100 instructions per step), ~5x cheaper than the exponentiation a^((q-1)/2).
// The Montgomery factor R = 2^392 is a square, so the symbol of the representative is the
// symbol of the value.  Wave-uniform loop (tc_common.h wave_any), branch-free body.
// Returns +1, -1, or 0 for a = 0.
TC_HD_NOINLINE int fq_legendre(const Fq& a) {
  uint32_t x[12], n[12];
  a.to_canonical(x);
  TC_UNROLL for (int i = 0; i < 12; i++) n[i] = FQ_P[i];
  uint32_t s = 0;  // sign bit of the running symbol
  uint32_t any = 0;
  TC_UNROLL for (int i = 0; i < 12; i++) any |= x[i];
  bool nz = any != 0;
  TC_NOUNROLL while (wave_any(nz)) {
    // (a lane that is done holds x = 0: nothing below changes its x, n or s)
    // x odd: (x / n) = (x - n / n), after reciprocity if x < n
    const bool odd = (x[0] & 1u) != 0;
    uint32_t d1[12], d2[12];  // x - n, n - x
    uint32_t b1 = 0, b2 = 0;
    TC_UNROLL for (int i = 0; i < 12; i++) {
      d1[i] = sub_borrow(x[i], n[i], b1, b1);
      d2[i] = sub_borrow(n[i], x[i], b2, b2);
    }
    const bool lt = b1 != 0;  // x < n
    const bool swap = odd && lt;
    if (swap && (x[0] & 3u) == 3u && (n[0] & 3u) == 3u) s ^= 1u;
    TC_UNROLL for (int i = 0; i < 12; i++) {
      const uint32_t xi = x[i];
      x[i] = odd ? (lt ? d2[i] : d1[i]) : xi;
      n[i] = swap ? xi : n[i];
    }
    // x is even now (or zero): strip up to 31 factors of two, (2 / n) = -1 iff n = 3, 5 mod 8
    const int k = x[0] ? __builtin_ctz(x[0]) : 31;
    const uint32_t n8 = n[0] & 7u;
    if (nz && (k & 1) && (n8 == 3u || n8 == 5u)) s ^= 1u;
    any = 0;
    TC_UNROLL for (int i = 0; i < 12; i++) {
      const uint32_t hi = (i + 1 < 12) ? x[i + 1] : 0u;
      x[i] = (uint32_t)((((uint64_t)hi << 32) | x[i]) >> k);  // one funnel shift per word
      any |= x[i];
    }
    nz = any != 0;
  }
  uint32_t rest = 0;
  TC_UNROLL for (int i = 1; i < 12; i++) rest |= n[i];
  const bool n_is_one = n[0] == 1 && rest == 0;
  return n_is_one ? (s ? -1 : 1) : 0;
}

// "is a square" for two Fq values at once: the lane-pair build evaluates one per lane
TC_HD void fq_is_square_2(const Fq& a, const Fq& b, bool& sa, bool& sb) {
#if TC_PAIR
  const bool odd = pair_odd() != 0;
  const int mine = fq_legendre(Fq::select(odd, b, a));
  const int theirs = pair_swap(mine);
  sa = (odd ? theirs : mine) >= 0;
  sb = (odd ? mine : theirs) >= 0;
#else
  sa = fq_legendre(a) >= 0;
  sb = fq_legendre(b) >= 0;
#endif
}

// a in Fq2 is a square  <=>  its norm a0^2 + a1^2 is a square in Fq  (a1 = 0: always, one of
// a0, -a0 is a square in Fq).  norm = a.norm_fq().
TC_HD bool fq2_is_square(const Fq2& a, const Fq& norm) {
  if (a.im().is_zero()) return true;
  return fq_legendre(norm) >= 0;
}

// Square root of a SQUARE a of Fq2 through the norm:
//   n = sqrt(a0^2 + a1^2),  x0 = sqrt((a0 + n) / 2) or sqrt((a0 - n) / 2) -- exactly one of the two
//   is a square when a1 != 0 --,  x1 = a1 / (2 x0).
// Two Fq exponentiations; in the lane-pair build the two candidates for x0 are tried by the two
// lanes at the same time.
TC_HD_NOINLINE Fq2 fq2_sqrt_of_square(const Fq2& a, const Fq& norm) {
  const Fq re = a.re(), im = a.im();
  if (im.is_zero()) {
    Fq s;
    if (fq_sqrt(re, s)) return Fq2::make(s, Fq::zero());
    return Fq2::make(Fq::zero(), s);  // s^2 = -a0, (s u)^2 = a0
  }
  Fq n;
  fq_sqrt(norm, n);
  const Fq dp = fq_half(re + n), dm = fq_half(re - n);
  Fq x0, x0inv;
#if TC_PAIR
  const bool odd = pair_odd() != 0;
  const bool mine_ok = fq_sqrt(Fq::select(odd, dm, dp), x0, &x0inv);
  x0 = Fq::select(mine_ok, x0, Fq2{x0}.other());
  x0inv = Fq::select(mine_ok, x0inv, Fq2{x0inv}.other());
  // (exactly one lane succeeds: dp * dm = -a1^2 / 4 is a non-square)
#else
  if (!fq_sqrt(dp, x0, &x0inv)) fq_sqrt(dm, x0, &x0inv);
#endif
  return Fq2::make(x0, fq_half(im * x0inv));
}

// root of a in Fq2; false if a is not a square
TC_HD bool fq2_sqrt(const Fq2& a, Fq2& out) {
  const Fq norm = a.norm_fq();
  if (!fq2_is_square(a, norm)) return false;
  out = fq2_sqrt_of_square(a, norm);
  return true;
}

// P in G1  <=>  phi(P) = [-x^2] P   (M. Scott, eprint 2021/1130, the G1 test for BLS12 curves):
// two 64-bit ladders instead of the 255-bit [r]P.  tests/ cross-check it against [r]P on points of
// every prime order dividing the G1 cofactor.
TC_HD bool g1_in_subgroup(const G1Affine& p) {
  if (p.inf) return true;
  G1Jac x2p = g1_mul_by_x_abs(g1_mul_by_x_abs(G1Jac::from_affine(p)));  // [x^2] P
  return jac_add_mixed(x2p, g1_phi(p)).is_inf();                         // phi(P) + [x^2] P == 0
}

// P in G2  <=>  psi(P) = [x] P   (M. Scott, "A note on group membership tests for G1, G2 and GT
// on BLS pairing-friendly curves", 2021): one 64-bit ladder instead of a 255-bit one.
TC_HD bool g2_in_subgroup(const G2Affine& p) {
  if (p.inf) return true;
  G2Jac xp = g2_mul_by_x_abs(G2Jac::from_affine(p));  // [|x|] P = -[x] P
  return jac_add_mixed(xp, g2_psi(p)).is_inf();        // psi(P) + [|x|] P == 0
}

// ---- checked decode of the compressed forms (EncodedPoint::into_affine) ------------------------
TC_HD bool g1_decode_compressed(const uint8_t* b, G1Affine& p) {
  const uint8_t f = b[0];
  if (!(f & 0x80)) return false;
  if (f & 0x40) {
    uint32_t o = f & 0x3f;
    for (int i = 1; i < 48; i++) o |= b[i];
    p = G1Affine::infinity();
    return o == 0;
  }
  Fq x, y;
  if (!fq_from_be48(b, true, x)) return false;
  if (!fq_sqrt(x.sqr() * x + g1_b(), y)) return false;
  const bool greatest = (f & 0x20) != 0;
  if (fq_lex_largest(y) != greatest) y = -y;
  p = G1Affine{x, y, false};
  return g1_in_subgroup(p);
}

TC_HD bool g2_decode_compressed(const uint8_t* b, G2Affine& p) {
  const uint8_t f = b[0];
  if (!(f & 0x80)) return false;
  if (f & 0x40) {
    uint32_t o = f & 0x3f;
    for (int i = 1; i < 96; i++) o |= b[i];
    p = G2Affine::infinity();
    return o == 0;
  }
  Fq2 x, y;
  if (!fq2_from_be96(b, true, x)) return false;
  if (!fq2_sqrt(x.sqr() * x + g2_b(), y)) return false;
  const bool greatest = (f & 0x20) != 0;
  if (fq2_lex_largest(y) != greatest) y = -y;
  p = G2Affine{x, y, false};
  return g2_in_subgroup(p);
}

// ---- two jobs per lane pair (tc_duo.h) ----------------------------------------------------------
// Square roots of TWO Fq2 values through their norms: per value two Fq exponentiations, each executed by ONE lane (lane 0
// for a, lane 1 for b) instead of by both.
//   n = sqrt(a0^2 + a1^2);  d = (a0 + n) / 2  (d = a0 when a1 = 0);  w = d^((q-3)/4), s = w d:
//     d a square (s^2 = d, w = 1/s):       root = (s, a1 w / 2)
//     d a non-square (s^2 = -d, w = -1/s): (a0 - n) / 2 = -a1^2 / (4 d) = (a1 / (2 s))^2, so  root = (-a1 w / 2, s)
// -- both candidates of the one-job form (fq2_sqrt_of_square) out of ONE exponentiation.  ok = the result squares to the
// input (a non-square norm leaves garbage in n and the comparison fails).  Which root comes out is not observable: every
// caller re-selects by lexicographic order.
TC_HD void fq2_sqrt_x2(const Fq2& a, const Fq2& b, Fq2& ya, Fq2& yb, bool& oka, bool& okb) {
  Fq na, nb;
  {
    const Duo<Fq> nm = duo_from(a.norm_fq(), b.norm_fq());
    Duo<Fq> n;
    duo_each([&](int s) { n.at(s) = fq_pow_qm3d4(nm.at(s)) * nm.at(s); });
    duo_to(n, na, nb);
  }
  const Fq are = a.re(), aim = a.im(), bre = b.re(), bim = b.im();
  const Fq da = Fq::select(aim.is_zero(), are, fq_half(are + na));
  const Fq db = Fq::select(bim.is_zero(), bre, fq_half(bre + nb));
  const Duo<Fq> d = duo_from(da, db);
  Duo<Fq> w, s;
  Duo<bool> sq;
  duo_each([&](int k) {
    w.at(k) = fq_pow_qm3d4(d.at(k));
    s.at(k) = w.at(k) * d.at(k);
    sq.at(k) = s.at(k).sqr() == d.at(k);
  });
  Fq wa, wb, sa, sb;
  bool sqa, sqb;
  duo_to(w, wa, wb);
  duo_to(s, sa, sb);
  duo_to(sq, sqa, sqb);
  const Fq ha = fq_half(aim * wa), hb = fq_half(bim * wb);
  ya = Fq2::make(Fq::select(sqa, sa, -ha), Fq::select(sqa, ha, sa));
  yb = Fq2::make(Fq::select(sqb, sb, -hb), Fq::select(sqb, hb, sb));
  oka = ya.sqr() == a;
  okb = yb.sqr() == b;
}

// 1 / a and 1 / b: the Fq inversions of the two norms on one lane each (0 -> 0)
TC_HD void fq2_inv_x2(const Fq2& a, const Fq2& b, Fq2& ia, Fq2& ib) {
  const Duo<Fq> nm = duo_from(a.norm_fq(), b.norm_fq());
  Duo<Fq> t;
  duo_each([&](int s) { t.at(s) = nm.at(s).inv(); });
  Fq ta, tb;
  duo_to(t, ta, tb);
  ia = a.scale(ta).conj();
  ib = b.scale(tb).conj();
}
// two Jacobian G2 points to affine, the inversions shared out as above
TC_HD void jac_to_affine_x2(const G2Jac& p, const G2Jac& q, G2Affine& pa, G2Affine& qa) {
  const bool pinf = p.is_inf(), qinf = q.is_inf();
  Fq2 zp, zq;
  fq2_inv_x2(p.z, q.z, zp, zq);
  const Fq2 zp2 = zp.sqr(), zq2 = zq.sqr();
  pa = G2Affine{p.x * zp2, p.y * zp2 * zp, false};
  qa = G2Affine{q.x * zq2, q.y * zq2 * zq, false};
  if (pinf) pa = G2Affine::infinity();
  if (qinf) qa = G2Affine::infinity();
}

// (out of line for the two-point decode: two inlined copies cost the kernel 147 spilled registers)
TC_HD_NOINLINE bool g2_in_subgroup_call(const G2Affine& p) { return g2_in_subgroup(p); }

// The checked decode of TWO compressed G2 points (EncodedPoint::into_affine of pairing 0.16 behind from_bytes,
// /root/reference/src/lib.rs:246-252): same verdicts and points as g2_decode_compressed on each, one control flow for the
// pair -- a point that fails early keeps a stand-in through the arithmetic and is replaced at the end.
//   kind: 0 = malformed, 1 = the identity, 2 = a finite point (x parsed, in range)
TC_HD int g2_parse_compressed(const uint8_t* b, Fq2& x, bool& greatest) {
  const uint8_t f = b[0];
  greatest = (f & 0x20) != 0;
  uint32_t o = f & 0x3f;
  for (int i = 1; i < 96; i++) o |= b[i];
  const bool in_range = fq2_from_be96(b, true, x);
  if (!(f & 0x80)) return 0;
  if (f & 0x40) return o == 0 ? 1 : 0;
  return in_range ? 2 : 0;
}
TC_HD void g2_decode_compressed_x2(const uint8_t* ba, const uint8_t* bb, G2Affine& pa, G2Affine& pb, bool& oka, bool& okb) {
  Fq2 xa, xb;
  bool ga, gb;
  const int ka = g2_parse_compressed(ba, xa, ga), kb = g2_parse_compressed(bb, xb, gb);
  // (a malformed or identity encoding walks on with x = 0: y^2 = b has whatever fate it has, the result is discarded)
  if (ka != 2) xa = Fq2::zero();
  if (kb != 2) xb = Fq2::zero();
  Fq2 ya, yb;
  bool sqa, sqb;
  fq2_sqrt_x2(xa.sqr() * xa + g2_b(), xb.sqr() * xb + g2_b(), ya, yb, sqa, sqb);
  if (fq2_lex_largest(ya) != ga) ya = -ya;
  if (fq2_lex_largest(yb) != gb) yb = -yb;
  pa = G2Affine{xa, ya, false};
  pb = G2Affine{xb, yb, false};
  const bool ina = g2_in_subgroup_call(pa);
  const bool inb = g2_in_subgroup_call(pb);
  oka = ka == 2 ? (sqa && ina) : ka == 1;
  okb = kb == 2 ? (sqb && inb) : kb == 1;
  if (ka != 2 || !oka) pa = G2Affine::infinity();
  if (kb != 2 || !okb) pb = G2Affine::infinity();
}

}  // namespace tc
