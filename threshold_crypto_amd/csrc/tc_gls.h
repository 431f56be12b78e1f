// G2 scalar multiplication through the psi endomorphism (4-dimensional GLS), and fast
// cofactor clearing.  Not in the reference (pairing 0.16 multiplies bit by bit over all 255 /
// 507 bits); these compute the SAME group elements with ~4x fewer doublings, so the encodings
// that leave the C ABI are unchanged.
//
//   psi(x, y) = (conj(x) * cx, conj(y) * cy)   untwist o Frobenius o twist on E'(Fq2)
//   on G2:  psi(P) = [q mod r] P = [x] P = -[|x|] P        (x = -0xd201000000010000)
//   so for k = d0 + d1|x| + d2|x|^2 + d3|x|^3 (base-|x| digits, each < 2^64):
//       [k] P = d0 P - d1 psi(P) + d2 psi^2(P) - d3 psi^3(P)
//   and one joint double-and-add over 64 bits with an 8-entry table (sign-aligned digits,
//   sac_recode4) replaces the 255-bit ladder.
//   on all of E'(Fq2):  psi^2 - t psi + q = 0, which gives (Budroni-Pintore)
//       [x^2-x-1] P + [x-1] psi(P) + psi^2(2P) = [3(x^2-1) h2] P,
//   and because [h2]P lies in the order-r subgroup, [h2] P = [c] of that with
//   c = (3(x^2-1))^-1 mod r: the reference's 507-bit scale_by_cofactor becomes two 64-bit
//   ladders and one GLS multiplication.
#pragma once
#include "tc_table.h"

namespace tc {

TC_HD Fq2 psi_cx() { return Fq2::make(Fq::from_limbs(PSI_CX_C0), Fq::from_limbs(PSI_CX_C1)); }
TC_HD Fq2 psi_cy() { return Fq2::make(Fq::from_limbs(PSI_CY_C0), Fq::from_limbs(PSI_CY_C1)); }

TC_HD G2Affine g2_psi(const G2Affine& p) {
  if (p.inf) return p;
  return G2Affine{(p.x.conj() * psi_cx()), (p.y.conj() * psi_cy()), false};
}

TC_HD_NOINLINE G2Jac g2_psi(const G2Jac& p) {
  return G2Jac{(p.x.conj() * psi_cx()), (p.y.conj() * psi_cy()), p.z.conj().norm()};
}

// (u1 : u0) / |x| for u1 < |x|: quotient and remainder by multiplication with the precomputed reciprocal
// (Moeller-Granlund, "Improved division by invariant integers", 2-by-1 division; |x| has its top bit set)
TC_HD uint64_t div_by_x_abs(uint64_t u1, uint64_t u0, uint64_t* rem) {
  typedef unsigned __int128 u128;
  const uint64_t X = BLS_X_ABS;
  const u128 q = (u128)BLS_X_RECIP * u1 + (((u128)u1 << 64) | u0);
  uint64_t q1 = (uint64_t)(q >> 64) + 1;
  const uint64_t q0 = (uint64_t)q;
  uint64_t r = u0 - q1 * X;
  if (r > q0) {
    q1--;
    r += X;
  }
  if (r >= X) {
    q1++;
    r -= X;
  }
  *rem = r;
  return q1;
}
// k (8 little-endian u32 words, < r < |x|^4) -> four base-|x| digits: three long divisions by the 64-bit |x|, a word at
// a time (12 reciprocal divisions; the bit-serial form this replaces took ~770 shift/compare/subtract iterations,
// 2-3 % of a G2 scalar multiplication).
TC_HD void gls_decompose(const uint32_t* k, uint64_t* d) {
  uint64_t n[4] = {(uint64_t)k[0] | ((uint64_t)k[1] << 32), (uint64_t)k[2] | ((uint64_t)k[3] << 32),
                   (uint64_t)k[4] | ((uint64_t)k[5] << 32), (uint64_t)k[6] | ((uint64_t)k[7] << 32)};
  TC_UNROLL for (int digit = 0; digit < 3; digit++) {
    uint64_t rem = 0;  // n = q * |x| + rem, the quotient replaces n
    TC_UNROLL for (int i = 3; i >= 0; i--) n[i] = div_by_x_abs(rem, n[i], &rem);
    d[digit] = rem;
  }
  d[3] = n[0];  // k < r < |x|^4  =>  the last quotient fits one word
}

// Sign-aligned recoding of four 64-bit scalars (GLV-SAC, Faz-Hernandez, Longa, Sanchez 2013):
//   d0 (odd)  = sum_{i <= 64} s_i 2^i            s_64 = +1,  s_i = 2 bit_{i+1}(d0) - 1  for i < 64
//   d_j       = sum_{i <= 64} s_i u_j[i] 2^i     u_j[i] in {0, 1}      (j = 1, 2, 3)
// so that column i of the joint ladder adds  s_i (B0 + u_1[i] B1 + u_2[i] B2 + u_3[i] B3):  a table
// of 8 sums that all contain B0 (7 additions to build) instead of the 15 subset sums (11), and no
// zero columns.  An even d0 is replaced by d0 + 1; the caller subtracts B0 at the end (fix).
struct SacDigits {
  uint64_t neg;   // bit i: s_i = -1 (i < 64)
  uint64_t u[3];  // bit i of u[j-1]: u_j[i] (i < 64)
  uint32_t top;   // bit j-1: u_j[64]
  bool fix;
};
// nbits < 64: the same recoding for SHORT digits (every d_j < 2^nbits): columns 0 .. nbits, the leading +1 at
// column nbits -- a ladder of nbits doublings instead of 64 (tc_msm.h uses 16-bit digits for the random scalars of
// the batch share validation).
TC_HD SacDigits sac_recode4(const uint64_t* d, int nbits = 64) {
  SacDigits r;
  r.fix = (d[0] & 1ull) == 0;
  r.neg = ~((d[0] | 1ull) >> 1);
  r.top = 0;
  TC_NOUNROLL for (int j = 0; j < 3; j++) {
    uint64_t k = d[j + 1], u = 0;
    TC_NOUNROLL for (int i = 0; i < nbits; i++) {
      const uint64_t odd = k & 1ull;
      u |= odd << i;
      k = (k >> 1) + (odd & (r.neg >> i));  // (k - s_i) / 2
    }
    r.u[j] = u;
    r.top |= (uint32_t)k << j;  // k - 1 <= (d_j - 1) / 2^64 < 1: k is 0 or 1
  }
  return r;
}

// sum_i d_i * B_i for four AFFINE base points and 64-bit scalars: joint double-and-add over the
// 8-entry table B0 + (subset sums of B1, B2, B3) with sign-aligned digits (above).  The 7 proper
// sums are brought to one common Z (no inversion: jac_batch_to_common_z) and B0 scaled to it, so
// every addition of the 64-step ladder is a mixed one on the isomorphic curve (7M + 4S instead of
// 11M + 5S in Fq2), and a table entry is two coordinates.  The table itself lives in HBM (tc_table.h): the
// kernel must hold a table slot (table_slot_acquire) while any of this runs.
struct G2SacTable {
  tbl_word* mem;  // this lane pair's 8 entries in the table arena (tc_table.h): B0 + (subset sums of B1, B2, B3),
                  // affine on the curve scaled by zc
  Fq2 zc;
  TC_HD G2Affine entry(uint32_t m) const { return tbl_load_g2(mem + m * kTblEntryWords); }
};
TC_HD void g2_sac_table(const G2Affine* base, G2SacTable& t) {
  G2Jac sums[7];
  G2Affine sums_aff[7];
  TC_NOUNROLL for (int m = 1; m < 8; m++) {
    const int low = __builtin_ctz((unsigned)m);
    const int rest = m & (m - 1);
    sums[m - 1] = rest ? jac_add_mixed(sums[rest - 1], base[low + 1]) : jac_add_affine(base[0], base[low + 1]);
  }
  t.zc = jac_batch_to_common_z(sums, sums_aff, 7);
  const Fq2 zc2 = t.zc.sqr();
  const Fq2 zc3 = zc2 * t.zc;
  t.mem = pair_table();
  tbl_store_g2(t.mem, affine_scale_z(base[0], zc2, zc3));
  TC_NOUNROLL for (int m = 1; m < 8; m++) tbl_store_g2(t.mem + m * kTblEntryWords, sums_aff[m - 1]);
}
// the ladder with every special case of the addition handled (the slow path of g2_sac_ladder)
TC_HD_NOINLINE G2Jac g2_sac_ladder_safe(const G2SacTable& t, const SacDigits& sd) {
  G2Jac acc = G2Jac::from_affine(t.entry(sd.top));
  TC_NOUNROLL for (int bit = 63; bit >= 0; bit--) {
    acc = jac_dbl(acc);
    const uint32_t m = (uint32_t)((sd.u[0] >> bit) & 1) | ((uint32_t)((sd.u[1] >> bit) & 1) << 1) |
                       ((uint32_t)((sd.u[2] >> bit) & 1) << 2);
    G2Affine e = t.entry(m);
    e.y = Fq2::select((sd.neg >> bit) & 1, -e.y, e.y);
    acc = jac_add_mixed(acc, e);
  }
  return acc;
}
TC_HD G2Jac g2_sac_ladder(const G2SacTable& t, const uint64_t* d) {
  const SacDigits sd = sac_recode4(d);
  G2Jac acc = G2Jac::from_affine(t.entry(sd.top));
  bool exc = acc.is_inf();
  TC_NOUNROLL for (int bit = 63; bit >= 0; bit--) {
    tc_fair();
    acc = jac_dbl(acc);
    const uint32_t m = (uint32_t)((sd.u[0] >> bit) & 1) | ((uint32_t)((sd.u[1] >> bit) & 1) << 1) |
                       ((uint32_t)((sd.u[2] >> bit) & 1) << 2);
    G2Affine e = t.entry(m);
    e.y = Fq2::select((sd.neg >> bit) & 1, -e.y, e.y);
    acc = jac_add_mixed_generic(acc, e, exc);
  }
  if (wave_any(exc)) acc = G2Jac::select(exc, g2_sac_ladder_safe(t, sd), acc);
  if (wave_any(sd.fix)) {
    G2Affine e = t.entry(0);
    e.y = -e.y;
    acc = G2Jac::select(sd.fix, jac_add_mixed(acc, e), acc);
  }
  acc.z = coord_norm(acc.z * t.zc);
  return acc;
}
// out-of-line forms for callers that run several ladders over one table (tc_jobs.h job_g2_mul_shared)
TC_HD_NOINLINE void g2_sac_table_call(const G2Affine* base, G2SacTable& t) { g2_sac_table(base, t); }
TC_HD_NOINLINE G2Jac g2_sac_ladder_call(const G2SacTable& t, const uint64_t* d) { return g2_sac_ladder(t, d); }
TC_HD_NOINLINE G2Jac g2_joint_mul4(const G2Affine* base, const uint64_t* d) {
  G2SacTable t;
  g2_sac_table(base, t);
  TC_MARK(3);
  return g2_sac_ladder(t, d);
}

// the four psi-images the digits multiply:  P, -psi(P), psi^2(P), -psi^3(P)
TC_HD void g2_gls_bases(const G2Affine& p, G2Affine* base) {
  base[0] = p;
  base[1] = g2_psi(p);
  base[2] = g2_psi(base[1]);
  base[3] = g2_psi(base[2]);
  base[1].y = (-base[1].y).norm();
  base[3].y = (-base[3].y).norm();
}

// [k] P for P in G2 (the order-r subgroup), k < r given as 8 LE u32 words
// The ladder wants an odd first digit, and d0 = k mod 2 because |x| is even: an even k is
// replaced by the odd r - k and the result negated.
TC_HD bool gls_decompose_odd(const uint32_t* k, uint64_t* d) {
  const bool flip = (k[0] & 1u) == 0;
  uint32_t kk[8];
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)FR_P[i] - k[i] - borrow;
    borrow = (uint32_t)(t >> 63);
    kk[i] = flip ? (uint32_t)t : k[i];
  }
  gls_decompose(kk, d);
  return flip;
}
TC_HD G2Jac g2_mul_gls(const G2Affine& p, const uint32_t* k) {
  uint64_t d[4];
  const bool flip = gls_decompose_odd(k, d);
  G2Affine base[4];
  g2_gls_bases(p, base);
  G2Jac r = g2_joint_mul4(base, d);
  r.y = Fq2::select(flip, -r.y, r.y);
  return r;
}
// Jacobian input without an inversion: scale (X, Y, Z) by conj(Z) so that the third coordinate
// becomes the norm N = Z conj(Z), an element of Fq.  psi keeps a real Z, so all four psi-images
// share it and are AFFINE points of the isomorphic curve y^2 = x^3 + b N^6 -- the a = 0 group
// law never reads b -- and the result only needs its Z multiplied by N to come back.
TC_HD G2Jac g2_gls_digits_mul(const G2Jac& p, const uint64_t* d) {
  const Fq2 l = p.z.conj();
  const Fq2 l2 = l.sqr();
  const G2Affine q{coord_norm(p.x * l2), coord_norm(p.y * (l2 * l)), p.is_inf()};
  G2Affine base[4];
  g2_gls_bases(q, base);
  G2Jac r = g2_joint_mul4(base, d);
  r.z = coord_norm(r.z.scale(p.z.norm_fq()));
  return r;
}
TC_HD G2Jac g2_mul_gls(const G2Jac& p, const uint32_t* k) {
  uint64_t d[4];
  const bool flip = gls_decompose_odd(k, d);
  G2Jac r = g2_gls_digits_mul(p, d);
  r.y = Fq2::select(flip, -r.y, r.y);
  return r;
}

// [|x|] P by the 64-bit ladder (|x| has Hamming weight 6: 63 doublings, 5 additions)
// (the base enters as the affine point (X, Y) of the isomorphic curve y^2 = x^3 + b Z^6, so the five
// additions are mixed ones; the result's Z is multiplied by the base's)
TC_HD_NOINLINE G2Jac g2_mul_by_x_abs(const G2Jac& p) {
  const G2Affine pa{p.x, p.y, p.is_inf()};
  G2Jac acc = jac_ladder_uniform(pa, BLS_X_ABS, 63);
  acc.z = coord_norm(acc.z * p.z);
  return acc;
}

// [h2] P for ANY point of E'(Fq2): the value the reference's scale_by_cofactor returns.
// fix = false stops before the last multiplication and returns Q' = [3(x^2-1) h2] P, the point of G2
// with [h2] P = [c] Q', c = FR_COFACTOR_FIX: callers that multiply the hash point by a scalar or pair
// it with a G1 point fold c into that scalar / that point instead (tc_jobs.h).
TC_HD G2Jac g2_clear_cofactor(const G2Affine& pa, bool fix = true) {
  const G2Jac p = G2Jac::from_affine(pa);
  G2Jac t1 = jac_neg(g2_mul_by_x_abs(p));            // [x] P          (x < 0)
  G2Jac t2 = g2_psi(p);                              // psi(P)
  G2Jac t3 = g2_psi(g2_psi(jac_dbl(p)));             // psi^2(2P)
  t3 = jac_add(t3, jac_neg(t2));                     // psi^2(2P) - psi(P)
  t2 = jac_add(t1, t2);                              // [x]P + psi(P)
  t2 = jac_neg(g2_mul_by_x_abs(t2));                 // [x^2]P + [x]psi(P)
  t3 = jac_add(t3, t2);
  t3 = jac_add(t3, jac_neg(t1));
  t3 = jac_add(t3, jac_neg(p));                      // = [3(x^2-1) h2] P, in G2
  if (!fix) return t3;
  // [c] t3, c = (3 (x^2 - 1))^-1 mod r.  On G2, where psi = [x]:  1 / (x^2 - 1) = x^-4 = x^8 and
  // 1 / 3 = ((x - 1) / 3) (x - 1)^-1 = ((x - 1) / 3) (-x^3 - x^2)  (3 divides x - 1), x^6 = -1, so
  //     [c] Q = [(|x| + 1) / 3] (-psi^4 (Q + psi(Q))):
  // one 63-bit ladder on one point (62 doublings + 27 additions) instead of the 4-dimensional one
  // (64 + 64 and a table).
  G2Jac s = jac_add(t3, g2_psi(t3));
  s = g2_psi(g2_psi(g2_psi(g2_psi(s))));
  s = jac_neg(s);
  s.x = s.x.norm();
  s.y = s.y.norm();
  s.z = s.z.norm();
  // (X, Y) of s is an AFFINE point of the isomorphic curve y^2 = x^3 + b Z^6 (tc_curve.h
  // jac_batch_to_common_z): the ladder adds it with mixed additions and its result gets Z back
  const G2Affine sa{s.x, s.y, s.is_inf()};
  G2Jac acc = jac_ladder_uniform(sa, G2_COFACTOR_FIX_SHORT, 62);  // bit 62 is the leading one
  acc.z = coord_norm(acc.z * s.z);
  return acc;
}

// ---- G1: 2-dimensional GLV through phi(x, y) = (beta x, y) --------------------------------------
//   on G1:  phi(P) = [-x^2] P,  so with k = k1 + k2 x^2  (k1 = k mod x^2, k2 = k div x^2, both
//   below 2^128):   [k] P = [k1] P + [k2] (-phi(P)):  one joint 128-bit ladder over {P, -phi P, P - phi P}
//   instead of the 255-bit one.
TC_HD G1Affine g1_phi(const G1Affine& p) {
  if (p.inf) return p;
  return G1Affine{p.x * Fq::from_limbs(G1_BETA), p.y, false};
}

typedef unsigned __int128 tc_u128;

// k (8 LE u32 words, < r) -> (k1, k2) with k = k1 + k2 * x^2: from the base-|x| digits of k (gls_decompose above),
// k1 = d0 + d1 |x|, k2 = d2 + d3 |x|
TC_HD void glv_decompose(const uint32_t* k, tc_u128* k1, tc_u128* k2) {
  uint64_t d[4];
  gls_decompose(k, d);
  *k1 = (tc_u128)d[1] * BLS_X_ABS + d[0];
  *k2 = (tc_u128)d[3] * BLS_X_ABS + d[2];
}

// k (8 canonical LE words, k <= r) in the base-4 sign-aligned form of the 2-dimensional decomposition:
//     k' = k1 + k2 x^2,  k1 odd:   k1 = sum_i s_i 2^i,  k2 = sum_i s_i u_i 2^i   (i = 0 .. 128, s_i = +-1, s_128 = +1, u_i in {0, 1})
// k' = k when k is odd, r - k otherwise (r is odd, x^2 is even, so k1 = k' mod x^2 is odd): returns true in that case and the
// caller negates the base point.  *neg: bit i set <=> s_i = -1 (i < 128); *u: the u_i (i < 128); *top = u_128.
TC_HD bool glv_recode_sign_aligned(const uint32_t* k, tc_u128* neg_out, tc_u128* u_out, bool* top) {
  const bool flip = (k[0] & 1u) == 0;
  uint32_t kk[8];
  uint32_t borrow = 0;
  TC_UNROLL for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)FR_P[i] - k[i] - borrow;
    borrow = (uint32_t)(t >> 63);
    kk[i] = flip ? (uint32_t)t : k[i];
  }
  tc_u128 k1, k2;
  glv_decompose(kk, &k1, &k2);
  const tc_u128 neg = ~((k1 | 1) >> 1);
  tc_u128 u = 0;
  TC_NOUNROLL for (int i = 0; i < 128; i++) {
    const tc_u128 odd = k2 & 1;
    u |= odd << i;
    k2 = (k2 >> 1) + (odd & (neg >> i));
  }
  *neg_out = neg;
  *u_out = u;
  *top = k2 != 0;
  return flip;
}

// [k] P for P in G1, k < r: the columns of the sign-aligned form two at a time (base 4).  Columns 2c and 2c+1 contribute
//     4^c sigma (A P + B phi'),  phi' = -phi(P) = [x^2] P,  sigma = s_{2c+1},  A = 2 + e in {1, 3},  B = e u_{2c} + 2 u_{2c+1},
//     e = s_{2c} s_{2c+1}:   one of { P, P - phi', P + 2 phi', P + phi', 3P, 3P + phi', 3P + 2 phi', 3P + 3 phi' }
// (index u_{2c} + 2 u_{2c+1}, + 4 when e = +1), added or subtracted; column 128 starts the accumulator with P or P + phi'.
// 64 steps of two doublings and ONE mixed addition -- 128 doublings + 64 additions + a 7-addition table instead of the 128
// doublings + 128 additions of the binary joint ladder (a wave takes the addition of a column whenever one of its lanes
// does).  The table is brought to one common Z without an inversion (tc_curve.h jac_batch_to_common_z: its entries are
// affine points of an isomorphic curve, the result gets the common Z back).  Same recoding as the two-stage G1 kernels
// (tc_msm.h msm_g1_recode), which keep their tables in HBM.
// ARENA: the table goes to this lane's 1 KB of the wave's arena slot (tc_table.h lane_table: full-line entries) instead of staying in
// 224 registers -- the form of the kernels built for TWO waves per SIMD (256 registers), which batches above one wave per SIMD
// take; the caller's kernel must hold a slot (table_slot_acquire).
template <bool ARENA>
TC_HD G1Jac g1_mul_glv_impl(const G1Affine& p, const uint32_t* k) {
  if (!ARENA && p.inf) return G1Jac::infinity();
  tc_u128 neg, u;
  bool top;
  const bool flip = glv_recode_sign_aligned(k, &neg, &u, &top);
  G1Affine b = p;
  b.y = Fq::select(flip, -p.y, p.y).norm();
  const G1Jac p2 = jac_dbl(G1Jac::from_affine(b));
  const G1Jac p3 = jac_add_mixed(p2, b);
  const Fq beta = Fq::from_limbs(G1_BETA);
  const G1Affine m1{b.x * beta, b.y, false};                  // phi(P) = -phi'(P)
  const G1Affine f1{m1.x, (-b.y).norm(), false};              // phi'(P)
  const G1Jac f2{p2.x * beta, (-p2.y).norm(), p2.z};          // phi'(2P) = 2 phi'(P)
  const G1Jac f3{p3.x * beta, (-p3.y).norm(), p3.z};          // phi'(3P)
  G1Jac e[7];
  e[0] = jac_add_affine(b, m1);    // 1: P - phi'
  e[1] = jac_add_mixed(f2, b);     // 2: P + 2 phi'
  e[2] = jac_add_affine(b, f1);    // 3: P + phi'
  e[3] = p3;                       // 4: 3P
  e[4] = jac_add_mixed(p3, f1);    // 5: 3P + phi'
  e[5] = jac_add(p3, f2);          // 6: 3P + 2 phi'
  e[6] = jac_add(p3, f3);          // 7: 3P + 3 phi'
  G1Affine tbl[8];
  const Fq zc = jac_batch_to_common_z<Fq, 8>(e, tbl + 1, 7);
  const Fq zc2 = zc.sqr();
  tbl[0] = affine_scale_z(b, zc2, zc2 * zc);
  tbl_word* mem = nullptr;
  if (ARENA) {
    mem = lane_table();
    TC_UNROLL for (int m = 0; m < 8; m++) tbl_store_g1(mem + m * kG1EntryWords, tbl[m]);
  }
  G1Jac acc = G1Jac::from_affine(ARENA ? tbl_load_g1(mem + (top ? 3 : 0) * kG1EntryWords) : tbl[top ? 3 : 0]);
  TC_NOUNROLL for (int c = 63; c >= 0; c--) {
    tc_fair();
    acc = jac_dbl(jac_dbl(acc));
    const uint32_t n = (uint32_t)(neg >> 126) & 3u, w = (uint32_t)(u >> 126) & 3u;  // bit 1: column 2c+1, bit 0: column 2c
    neg <<= 2;
    u <<= 2;
    const bool sub = (n >> 1) != 0;
    const uint32_t m = w | ((((n >> 1) ^ n) & 1u) ? 0u : 4u);
    G1Affine t = ARENA ? tbl_load_g1(mem + m * kG1EntryWords) : tbl[m];
    t.y = Fq::select(sub, -t.y, t.y);
    acc = jac_add_mixed(acc, t);
  }
  acc.z = coord_norm(acc.z * zc);
  if (ARENA && p.inf) acc = G1Jac::infinity();  // (uniform control flow up to here: the slot's stores and loads are the wave's)
  return acc;
}
TC_HD_NOINLINE G1Jac g1_mul_glv(const G1Affine& p, const uint32_t* k) { return g1_mul_glv_impl<false>(p, k); }
TC_HD_NOINLINE G1Jac g1_mul_glv_arena(const G1Affine& p, const uint32_t* k) { return g1_mul_glv_impl<true>(p, k); }
// the same for a Jacobian P (the share combiner's final [D^-1] step)
// (no inversion: (X, Y) is an affine point of the isomorphic curve y^2 = x^3 + b Z^6, phi acts on
// it the same way, and the a = 0 group law never reads b; the result's Z is multiplied by Z)
TC_HD G1Jac g1_mul_glv(const G1Jac& p, const uint32_t* k) {
  G1Jac r = g1_mul_glv(G1Affine{p.x, p.y, p.is_inf()}, k);
  r.z = coord_norm(r.z * p.z);
  return r;
}

// [|x|] P on G1 by the 64-bit ladder
TC_HD_NOINLINE G1Jac g1_mul_by_x_abs(const G1Jac& p) {
  const G1Affine pa{p.x, p.y, p.is_inf()};
  G1Jac acc = jac_ladder_uniform(pa, BLS_X_ABS, 63);
  acc.z = coord_norm(acc.z * p.z);
  return acc;
}

}  // namespace tc
