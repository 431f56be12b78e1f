// Square roots in Fq and Fq2 (q = 3 mod 4), compressed-point decoding and subgroup checks.
//   * hash_g2 needs "is x^3 + b a square, and a root of it" (G2Affine::get_point_from_x of
//     pairing 0.16; /root/reference/src/lib.rs:693);
//   * from_bytes (/root/reference/src/lib.rs:140-146, 246-252) needs the checked decode of the
//     48 / 96-byte compressed forms: on-curve AND in the order-r subgroup.
// The reference's Fq2::sqrt is Algorithm 9 of eprint 2012/685 (two Fq2 exponentiations).  Which
// root comes out is not observable -- callers re-select by lexicographic order -- so the device
// takes roots through the norm instead: three Fq exponentiations at most, and a non-square is
// rejected after ONE of them (that is the common case inside hash_g2's retry loop).
#pragma once
#include "tc_codec.h"
#include "tc_gls.h"

namespace tc {

// a^((q-3)/4).  For a square a != 0:  a * w is a square root of a and w = 1 / (a * w).
TC_HD_NOINLINE Fq fq_pow_qm3d4(const Fq& a) {
  return field_pow_fixed(a.norm(), [](int i) { return FQ_P_MINUS_3_DIV_4[i]; }, 379);
}

// root of a in Fq; false if a is not a square.  inv_root (optional) receives 1/root.
TC_HD bool fq_sqrt(const Fq& a, Fq& root, Fq* inv_root = nullptr) {
  Fq w = fq_pow_qm3d4(a);
  root = w * a;
  if (inv_root) *inv_root = w;
  return root.sqr() == a;
}

TC_HD Fq fq_half(const Fq& a) { return a * Fq::from_limbs(FQ26_INV2); }

// Square root in Fq2 in two steps, so that rejection sampling (hash_g2) can run only the cheap
// squareness test inside its retry loop and finish the root once, after the loop, for all lanes
// of the wave together:
//   begin : a is a square in Fq2  <=>  its norm a0^2 + a1^2 is a square in Fq; n = sqrt(norm)
//   finish: x0 = sqrt((a0 +- n) / 2), x1 = a1 / (2 x0)
TC_HD bool fq2_sqrt_begin(const Fq2& a, Fq& n) {
  if (a.im().is_zero()) {
    n = Fq::zero();
    return true;  // a = a0 in Fq: one of a0, -a0 is a square in Fq, so a is a square in Fq2
  }
  return fq_sqrt(a.re().sqr() + a.im().sqr(), n);
}

TC_HD_NOINLINE Fq2 fq2_sqrt_finish(const Fq2& a, const Fq& n) {
  if (a.im().is_zero()) {
    Fq s;
    if (fq_sqrt(a.re(), s)) return Fq2::make(s, Fq::zero());
    return Fq2::make(Fq::zero(), s);  // s^2 = -a0, (s u)^2 = a0
  }
  Fq delta = fq_half(a.re() + n);
  Fq x0, x0inv;
  if (!fq_sqrt(delta, x0, &x0inv)) {
    delta = (delta - n).norm();  // (a0 - n) / 2: exactly one of the two is a square (a1 != 0)
    fq_sqrt(delta, x0, &x0inv);
  }
  return Fq2::make(x0, fq_half(a.im() * x0inv));
}

// root of a in Fq2; false if a is not a square
TC_HD bool fq2_sqrt(const Fq2& a, Fq2& out) {
  Fq n;
  if (!fq2_sqrt_begin(a, n)) return false;
  out = fq2_sqrt_finish(a, n);
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
  p = G1Affine{x, y.norm(), false};
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
  p = G2Affine{x, y.norm(), false};
  return g2_in_subgroup(p);
}

}  // namespace tc
