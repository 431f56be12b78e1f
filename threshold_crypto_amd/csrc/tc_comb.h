// Many scalars over ONE G2 point: the shares of a message by its t + 1 selected signers (SecretKeyShare::sign_g2,
// /root/reference/src/lib.rs:442-444, once per signer; BASELINE config "t=67, N=200": 68 multiplications of every hash
// point).  A sign-aligned 4-dimensional GLS multiplication (tc_gls.h) is
//     [k] P = sum_{c <= 64} s_c T[m_c] 2^c,        T = the 8-entry table  B0 + (subsets of B1, B2, B3)  of psi-images of P,
// 64 doublings + 65 additions per scalar -- and the doublings do not depend on the scalar.  With enough scalars per point
// they are done ONCE, on the table:
//   stage T  one lane pair per message: column 0 of the comb is T itself (truly affine: one inversion), column c is twice
//            column c-1 by AFFINE doublings (lambda = 3 x^2 / 2 y), the eight inversions of a column shared (Montgomery's
//            trick).  65 columns x 8 entries x 256 B = 133 KB per message, in HBM, in the entry layout of tc_table.h.
//   stage S  one lane pair per (message, 8 signers): per signer 65 MIXED additions  acc += +-comb[c][m_c],  no doubling,
//            with the branch-free generic addition (tc_curve.h); the 8 results share one inversion.
// Per signer 65 mixed additions instead of 64 doublings + 65 additions; per message 64 x 8 affine doublings + 64
// inversions, which 68 signers repay three times over.  Same group elements, same bytes as the per-scalar ladders.
#pragma once
#include "tc_jobs.h"

namespace tc {

constexpr int kCombColumns = 65;                                      // digit columns 0 .. 64 (64: the leading +1)
constexpr int kCombTableWords = kCombColumns * 8 * kTblEntryWords;    // per message
constexpr int kCombShare = 8;                                         // signers per lane pair in stage S

// Stage T.  false: the point does not decode (every entry is then the identity and stage S fails the message's shares).
TC_HD bool job_comb_tables(const uint8_t* pt192, tbl_word* tbl) {
  G2Affine p;
  const bool ok = g2_decode_uncompressed(pt192, p);
  if (!ok) p = G2Affine::infinity();
  G2Affine e[8];
  {
    G2Affine base[4];
    g2_gls_bases(p, base);
    G2Jac sums[7];
    TC_NOUNROLL for (int m = 1; m < 8; m++) {
      const int low = __builtin_ctz((unsigned)m);
      const int rest = m & (m - 1);
      sums[m - 1] = rest ? jac_add_mixed(sums[rest - 1], base[low + 1]) : jac_add_affine(base[0], base[low + 1]);
    }
    e[0] = base[0];
    jac_batch_to_affine<Fq2, 7>(sums, e + 1, 7);
  }
  TC_NOUNROLL for (int m = 0; m < 8; m++) {
    e[m].x = e[m].x.norm();
    e[m].y = e[m].y.norm();
    tbl_store_g2(tbl + m * kTblEntryWords, e[m]);
  }
  TC_NOUNROLL for (int c = 1; c < kCombColumns; c++) {
    tc_fair();
    // 1 / (2 y) of the eight entries with one inversion (an entry at infinity counts as 1 and stays at infinity; 2 y = 0
    // needs a point of order two, which E'(Fq2) does not have: its cofactor is odd)
    Fq2 pre[8];
    Fq2 acc = Fq2::one();
    TC_NOUNROLL for (int m = 0; m < 8; m++) {
      pre[m] = acc;
      acc = acc * Fq2::select(e[m].inf, Fq2::one(), e[m].y.dbl());
    }
    Fq2 inv = acc.inv();
    TC_NOUNROLL for (int m = 7; m >= 0; m--) {
      const Fq2 d = Fq2::select(e[m].inf, Fq2::one(), e[m].y.dbl());
      const Fq2 di = inv * pre[m];
      inv = inv * d;
      const Fq2 xx = e[m].x.sqr();
      const Fq2 lam = ((xx.dbl() + xx).norm()) * di;            // 3 x^2 / (2 y)
      // (x3 = lambda^2 - 2 x feeds its own value back doubled: pull it to ~p every column, tc_field.h reduce_value)
      const Fq2 x3 = (lam.sqr() - e[m].x.dbl()).reduce_value();
      const Fq2 y3 = (lam * (e[m].x - x3) - e[m].y).reduce_value();
      e[m].x = x3;
      e[m].y = y3;
      tbl_store_g2(tbl + (size_t)(c * 8 + m) * kTblEntryWords, e[m]);
    }
  }
  return ok;
}

// Stage S: out[s] = sk[idx[s]] * P for n <= kCombShare signer indices into a table of N secret key shares, from the
// message's comb.  An index >= N or a scalar >= r fails its own output, an undecodable point (table_ok = false) all n.
TC_HD void job_comb_sign(const uint8_t* sk_table, size_t N, const uint64_t* idx, int n, const tbl_word* tbl, bool table_ok, uint8_t* out,
                         uint8_t* status, bool leader) {
  G2Jac res[kCombShare];
  bool ok[kCombShare];
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t i = idx[s];
    ok[s] = table_ok && i < N;
    if (i < N) ok[s] = fr_from_le32(sk_table + 32 * i, k) && ok[s];
    uint64_t d[4];
    const bool flip = gls_decompose_odd(k, d);
    const SacDigits sd = sac_recode4(d);
    const G2Affine top = tbl_load_g2(tbl + (size_t)(64 * 8 + sd.top) * kTblEntryWords);
    G2Jac acc = G2Jac::from_affine(top);
    bool exc = top.inf;
    TC_NOUNROLL for (int bit = 63; bit >= 0; bit--) {
      if ((bit & 15) == 15) tc_fair();
      const uint32_t m = (uint32_t)((sd.u[0] >> bit) & 1) | ((uint32_t)((sd.u[1] >> bit) & 1) << 1) |
                         ((uint32_t)((sd.u[2] >> bit) & 1) << 2);
      G2Affine e = tbl_load_g2(tbl + (size_t)(bit * 8 + m) * kTblEntryWords);
      e.y = Fq2::select((sd.neg >> bit) & 1, -e.y, e.y);
      acc = jac_add_mixed_generic(acc, e, exc);
    }
    if (wave_any(exc)) {
      // a lane that may have met a special case: the guarded 64-step ladder over column 0 (= the plain table)
      G2SacTable t;
      t.mem = const_cast<tbl_word*>(tbl);
      t.zc = Fq2::one();
      acc = G2Jac::select(exc, g2_sac_ladder_safe(t, sd), acc);
    }
    if (wave_any(sd.fix)) {
      G2Affine e0 = tbl_load_g2(tbl);
      e0.y = -e0.y;
      acc = G2Jac::select(sd.fix, jac_add_mixed(acc, e0), acc);
    }
    acc.y = Fq2::select(flip, -acc.y, acc.y);
    res[s] = G2Jac::select(ok[s], acc, G2Jac::infinity());
  }
  G2Affine aff[kCombShare];
  jac_batch_to_affine<Fq2, kCombShare>(res, aff, n);
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    g2_encode_uncompressed(aff[s], out + (size_t)s * 192);
    if (status && leader) status[s] = ok[s] ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  }
}

}  // namespace tc
