// DKG algebra (device side): fixed-base G1 multiplication for Poly::commitment / BivarPoly::commitment
// (/root/reference/src/poly.rs:372-377, 625-632), Horner evaluation of commitment rows for
// BivarCommitment::row / evaluate (:694-727) and the Fr-only Poly::interpolate (:341-350, 388-417).
//
// Poly::commitment multiplies ONE point -- the G1 generator -- by every coefficient, so the window table is
// the same for every job of every batch: it is built once per context and staged in LDS by each workgroup
// (k_dkg.hip).  With signed 4-bit windows
//     k = sum_{w < 64} d_w 16^w,   d_w in [-8, 8],        T[w][m - 1] = [m 16^w] g1   (m = 1 .. 8)
// a multiplication is 64 mixed additions and NO doubling (the reference's CurveAffine::mul: 255 doublings +
// ~127 additions; the variable-base GLV ladder of tc_gls.h: 128 + ~96).  512 affine entries x 112 B = 56 KB
// of LDS, i.e. two 256-lane workgroups per CU.
#pragma once
#include "tc_jobs.h"

namespace tc {

constexpr int kFbWindows = 64;                                      // 4-bit windows of a 256-bit scalar
constexpr int kFbEntries = 8;                                       // |digit| = 1 .. 8
constexpr int kFbPointWords = 2 * FQ_LIMBS;                         // x, y: 28 x int32
constexpr int kFbTableWords = kFbWindows * kFbEntries * kFbPointWords;  // 14 336 words = 57 344 B

// entry e = 8 w + (m - 1) of the table: [m 16^w] g1, affine, limbs carry-normalised
TC_HD void fixed_base_table_entry(int e, int32_t* out28) {
  const int w = e / kFbEntries, m = e % kFbEntries + 1;
  uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int bit = 4 * w;
  k[bit >> 5] = (uint32_t)m << (bit & 31);  // m <= 8 and 4 | bit: never straddles a word
  const G1Affine p = jac_to_affine(g1_mul_glv(g1_generator(), k));
  const Fq x = p.x.norm(), y = p.y.norm();
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    out28[i] = x.l[i];
    out28[FQ_LIMBS + i] = y.l[i];
  }
}

// [k] g1 from the window table (tbl: kFbTableWords int32 in LDS or global memory), k < r as 8 LE words
template <class TBL>
TC_HD G1Jac g1_fixed_base_mul(TBL tbl, const uint32_t* k) {
  G1Jac acc = G1Jac::infinity();
  uint32_t carry = 0;
  TC_NOUNROLL for (int w = 0; w < kFbWindows; w++) {
    const uint32_t v = ((k[w >> 3] >> (4 * (w & 7))) & 15u) + carry;  // 0 .. 16
    const bool neg = v > 8;
    carry = neg ? 1u : 0u;
    const uint32_t mag = neg ? 16u - v : v;  // 0 .. 8
    // k < r < 2^255: the top nibble is at most 7, so the last window absorbs its carry (mag <= 8)
    G1Affine e;
    e.inf = mag == 0;
    const int base = (w * kFbEntries + (int)(mag ? mag - 1 : 0)) * kFbPointWords;
    TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
      e.x.l[i] = tbl[base + i];
      e.y.l[i] = tbl[base + FQ_LIMBS + i];
    }
    e.x.set_range(0.f, 1.001f);
    e.x.set_val(2.1f);
    e.y.set_range(0.f, 1.001f);
    e.y.set_val(2.1f);
    e.y = Fq::select(neg, (-e.y).norm(), e.y);
    acc = jac_add_mixed(acc, e);
  }
  return acc;
}

// out = fr * g1   (Poly::commitment src/poly.rs:372-377: `G1Affine::one().mul(*c)`)
template <class TBL>
TC_HD uint8_t job_g1_fixed_base_mul(TBL tbl, const uint8_t* fr_le32, uint8_t* out96) {
  uint32_t k[8];
  if (!fr_from_le32(fr_le32, k)) {
    g1_encode_uncompressed(G1Affine::infinity(), out96);
    return TC_JOB_INVALID_ENCODING;
  }
  g1_encode_uncompressed(jac_to_affine(g1_fixed_base_mul(tbl, k)), out96);
  return TC_JOB_OK;
}

// position of coefficient (i, j) of a symmetric bivariate polynomial (coeff_pos, src/poly.rs:746-750)
TC_HD size_t bivar_coeff_pos(size_t i, size_t j) {
  const size_t lo = i < j ? i : j, hi = i < j ? j : i;
  return lo + hi * (hi + 1) / 2;
}

// res <- res * x  for a 64-bit x (the Horner step of Commitment::evaluate / BivarCommitment::row)
TC_HD G1Jac g1_mul_u64(const G1Jac& p, uint64_t x) {
  G1Jac acc = G1Jac::infinity();
  bool started = false;
  TC_NOUNROLL for (int bit = 63; bit >= 0; bit--) {
    if (started) acc = jac_dbl(acc);
    if ((x >> bit) & 1ull) {
      acc = started ? jac_add(acc, p) : p;
      started = true;
    }
  }
  return acc;
}

// BivarCommitment::row(x)[i] = sum_j commit[pos(i, j)] x^j   (src/poly.rs:713-727), x = IntoFr for u64 (the
// value itself, src/into_fr.rs:16-20), evaluated by Horner from the top coefficient down.
TC_HD uint8_t job_bivar_commitment_row(const uint8_t* commit, size_t degree, size_t i, uint64_t x, uint8_t* out96) {
  G1Affine c;
  bool ok = g1_decode_uncompressed(commit + bivar_coeff_pos(i, degree) * 96, c);
  G1Jac res = G1Jac::from_affine(c);
  TC_NOUNROLL for (size_t jj = degree; jj-- > 0;) {
    G1Jac scaled = g1_mul_u64(res, x);
    ok &= g1_decode_uncompressed(commit + bivar_coeff_pos(i, jj) * 96, c);
    res = jac_add_mixed(scaled, c);
  }
  if (!ok) {
    g1_encode_uncompressed(G1Affine::infinity(), out96);
    return TC_JOB_INVALID_ENCODING;
  }
  g1_encode_uncompressed(jac_to_affine(res), out96);
  return TC_JOB_OK;
}

// ---- Poly::interpolate in Fr (src/poly.rs:388-417 compute_interpolation), one lane per polynomial ------------
// xs, ys: n x 8 canonical LE words; out: n coefficients (low degree first, canonical words; the reference's
// Poly drops trailing zeros, the caller strips them); ws: 2 (n + 1) x 8 words of scratch.  The sample-by-sample
// construction of the reference: `poly` is right on the samples seen so far, `base` vanishes on them.
TC_HD Fr fr_horner(const uint32_t* coeffs_mont, size_t len, const Fr& x) {
  Fr r = Fr::zero();
  TC_NOUNROLL for (size_t k = len; k-- > 0;) {
    Fr c;
    TC_UNROLL for (int i = 0; i < 8; i++) c.v.l[i] = coeffs_mont[k * 8 + i];
    r = r * x + c;
  }
  return r;
}
TC_HD uint8_t job_fr_interpolate(size_t n, const uint32_t* xs, const uint32_t* ys, uint32_t* out, uint32_t* ws) {
  uint32_t* poly = ws;                 // Montgomery form while we work
  uint32_t* base = ws + (n + 1) * 8;
  bool ok = true;
  TC_NOUNROLL for (size_t s = 0; s < n; s++) ok &= limbs_lt_p<FrParams>(xs + s * 8) && limbs_lt_p<FrParams>(ys + s * 8);
  if (!ok || n == 0) {
    TC_NOUNROLL for (size_t k = 0; k < n * 8; k++) out[k] = 0;
    return n == 0 ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  }
  auto put = [](uint32_t* dst, const Fr& v) { TC_UNROLL for (int i = 0; i < 8; i++) dst[i] = v.v.l[i]; };
  auto get = [](const uint32_t* src) { Fr v; TC_UNROLL for (int i = 0; i < 8; i++) v.v.l[i] = src[i]; return v; };
  const Fr x0 = Fr::from_canonical(xs);
  put(poly, Fr::from_canonical(ys));
  put(base, Fr::zero() - x0);
  put(base + 8, Fr::one());
  size_t len_poly = 1, len_base = 2;
  bool dup = false;
  TC_NOUNROLL for (size_t s = 1; s < n; s++) {
    const Fr x = Fr::from_canonical(xs + s * 8), y = Fr::from_canonical(ys + s * 8);
    const Fr bv = fr_horner(base, len_base, x);
    dup = dup || bv.is_zero();  // "sample points must be distinct" (src/poly.rs:404)
    const Fr diff = (y - fr_horner(poly, len_poly, x)) * bv.inv();
    // base *= diff; poly += base
    TC_NOUNROLL for (size_t k = 0; k < len_base; k++) {
      const Fr b = get(base + k * 8) * diff;
      put(base + k * 8, b);
      put(poly + k * 8, (k < len_poly ? get(poly + k * 8) : Fr::zero()) + b);
    }
    len_poly = len_base;
    // base *= (X - x), from the top coefficient down
    put(base + len_base * 8, get(base + (len_base - 1) * 8));
    TC_NOUNROLL for (size_t k = len_base - 1; k > 0; k--) put(base + k * 8, get(base + (k - 1) * 8) - x * get(base + k * 8));
    put(base, Fr::zero() - x * get(base));
    len_base++;
  }
  TC_NOUNROLL for (size_t k = 0; k < n; k++) {
    if (dup) {
      TC_UNROLL for (int i = 0; i < 8; i++) out[k * 8 + i] = 0;
    } else {
      get(poly + k * 8).to_canonical(out + k * 8);
    }
  }
  return dup ? TC_JOB_DUPLICATE_ENTRY : TC_JOB_OK;
}

}  // namespace tc
