// Large linear combinations in G2 (t+1 >= 8 shares per job: BASELINE config "t=67, N=200"), two stages:
//
//   stage T  one lane pair per (job, chunk of 4 shares): every scalar is split into its four base-|x| digits
//            (tc_gls.h), each share gets its 8-entry sign-aligned table  B0 + {subset sums of B1, B2, B3}  of
//            psi-images, and the 4 x 7 proper sums of the chunk are brought to AFFINE form with one shared
//            inversion.  Tables (256 B per entry) and per-column digit codes go to HBM.
//   stage L  one lane pair per job: ONE chain of 64 doublings for all t+1 shares; per column and share one
//            table look-up (a single coalesced 64-byte row per coordinate and lane) and one MIXED addition.
//
// Per share that is 64 mixed additions + 7 table additions + 1/4 inversion (a chunked ladder with its tables in per-lane
// scratch -- the form this replaced for every G2 combination -- needs 16 more doublings per share and 15.6 KB of private
// segment per lane).  Same group element, same bytes as interpolate's sum_i lambda_i S_i (/root/reference/src/lib.rs:764).
#pragma once
#include "tc_jobs.h"

namespace tc {

constexpr int kMsmChunk = 4;          // shares per stage-T lane pair
constexpr int kMsmEntryWords = kTblEntryWords;  // one table entry (tc_table.h): 256 B
constexpr int kMsmColumns = 65;       // digit columns 0 .. 64 (column 64: the leading +1 of the sign-aligned form);
                                      // short scalars (all four base-|x| digits below 2^nbits) use columns 0 .. nbits

TC_HD size_t msm_chunks(size_t n) { return (n + kMsmChunk - 1) / kMsmChunk; }

// table entries in the layout of tc_table.h (the single-point ladders of tc_gls.h keep theirs the same way)
TC_HD void msm_store_entry(int32_t* e, const G2Affine& p) { tbl_store_g2((tbl_word*)e, p); }
TC_HD G2Affine msm_load_entry(const int32_t* e) { return tbl_load_g2((const tbl_word*)e); }

// Stage T for the chunk `c` of job data: points (n x 192 B), scalars (n x 8 canonical words).
//   tbl    this job's tables: (4 * chunks) shares x 8 entries x 64 words
//   codes  this job's digit codes: 65 columns x (4 * chunks) shares, one byte each: bits 0..2 table index,
//          bit 3 = subtract
// Returns false when a point or scalar of the chunk does not decode (the job then fails as a whole).
TC_HD bool job_msm_tables(size_t n, size_t c, const uint8_t* points, const uint32_t* scalars, int32_t* tbl, uint8_t* codes,
                          bool leader, int nbits = 64) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G2Jac sums[7 * kMsmChunk];
  G2Affine b0[kMsmChunk];
  SacDigits sd[kMsmChunk];
  bool ok = true;
  TC_NOUNROLL for (int k = 0; k < kMsmChunk; k++) {
    const size_t s = c * kMsmChunk + k;
    G2Affine p = G2Affine::infinity();
    uint32_t sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s < n) {
      ok &= g2_decode_uncompressed(points + s * 192, p);
      for (int w = 0; w < 8; w++) sc[w] = scalars[s * 8 + w];
      ok &= limbs_lt_p<FrParams>(sc);
    }
    if (!ok) p = G2Affine::infinity();
    uint64_t d[4];
    bool flip = gls_decompose_odd(sc, d);
    if (nbits < 64) {
      // short-scalar mode: the caller promises odd scalars d0 + d1 |x| + d2 |x|^2 + d3 |x|^3 with digits below
      // 2^nbits (never flipped: r - k would not be short); anything else fails the job
      uint64_t dd[4];
      gls_decompose(sc, dd);
      bool fits = (sc[0] & 1u) != 0 || s >= n;
      TC_UNROLL for (int j = 0; j < 4; j++) fits = fits && (dd[j] >> nbits) == 0;
      if (s >= n) dd[0] = 1, dd[1] = dd[2] = dd[3] = 0;
      ok &= fits;
      TC_UNROLL for (int j = 0; j < 4; j++) d[j] = fits ? dd[j] : (uint64_t)(j == 0);
      flip = false;
    }
    sd[k] = sac_recode4(d, nbits);
    G2Affine base[4];
    g2_gls_bases(p, base);
    TC_NOUNROLL for (int j = 0; j < 4; j++) base[j].y = Fq2::select(flip, -base[j].y, base[j].y).norm();
    b0[k] = base[0];
    TC_NOUNROLL for (int m = 1; m < 8; m++) {
      const int low = __builtin_ctz((unsigned)m);
      const int rest = m & (m - 1);
      sums[7 * k + m - 1] = rest ? jac_add_mixed(sums[7 * k + rest - 1], base[low + 1]) : jac_add_affine(base[0], base[low + 1]);
    }
  }
  G2Affine aff[7 * kMsmChunk];
  jac_batch_to_affine<Fq2, 7 * kMsmChunk>(sums, aff, 7 * kMsmChunk);
  TC_NOUNROLL for (int k = 0; k < kMsmChunk; k++) {
    const size_t s = c * kMsmChunk + k;
    int32_t* t = tbl + s * 8 * kMsmEntryWords;
    msm_store_entry(t, b0[k]);
    TC_NOUNROLL for (int m = 1; m < 8; m++) msm_store_entry(t + m * kMsmEntryWords, aff[7 * k + m - 1]);
    if (leader) {
      TC_NOUNROLL for (int col = 0; col < nbits; col++) {
        const uint32_t m = (uint32_t)((sd[k].u[0] >> col) & 1) | ((uint32_t)((sd[k].u[1] >> col) & 1) << 1) |
                           ((uint32_t)((sd[k].u[2] >> col) & 1) << 2);
        codes[(size_t)col * shares4 + s] = (uint8_t)(m | (((sd[k].neg >> col) & 1) << 3));
      }
      codes[(size_t)nbits * shares4 + s] = (uint8_t)sd[k].top;  // top column: every share adds +tbl[top]
    }
  }
  return ok;
}

// Stage L can split a job over `parts` lane pairs (small batches: k_msm.hip): part g sums the shares
// [g n / parts, (g + 1) n / parts) -- never empty for parts <= n --, every part runs ceil(n / parts) look-ups per column
// (the last one masked in the shorter parts, so that the lanes of a wave keep one trip count), and the caller adds the
// partial sums.  parts = 1: the whole job.
struct MsmPart {
  size_t s0, s1, trips;
};
TC_HD MsmPart msm_part(size_t n, size_t g = 0, size_t parts = 1) { return MsmPart{g * n / parts, (g + 1) * n / parts, (n + parts - 1) / parts}; }

// Stage L: sum over the shares of one part from the job's tables and digit codes -- every special case of the
// addition handled (the slow path of job_msm_ladder below, and the g++ reference of the fast one)
TC_HD_NOINLINE G2Jac job_msm_ladder_safe(size_t n, const int32_t* tbl, const uint8_t* codes, int nbits, MsmPart part) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G2Jac acc = G2Jac::infinity();
  TC_NOUNROLL for (int col = nbits; col >= 0; col--) {
    if (col != nbits) acc = jac_dbl(acc);
    const uint8_t* cc = codes + (size_t)col * shares4;
    TC_NOUNROLL for (size_t t = 0; t < part.trips; t++) {
      const bool take = part.s0 + t < part.s1;
      const size_t s = take ? part.s0 + t : part.s0;
      const uint32_t code = cc[s];
      G2Affine e = msm_load_entry(tbl + (s * 8 + (code & 7)) * kMsmEntryWords);
      // (the top column adds every entry as it is: no sign select there, so that the accumulator starts from a
      // carry-normalised y -- the lazy-limb budget of the first real addition depends on it)
      if (col != nbits) e.y = Fq2::select((code >> 3) & 1, -e.y, e.y);
      acc = G2Jac::select(take, jac_add_mixed(acc, e), acc);
    }
  }
  return acc;
}
// The fast form: the accumulator starts from the part's first share's top entry, the padding shares (entries at
// infinity, s >= n) are not visited, and the additions are the branch-free generic ones (tc_curve.h
// jac_add_mixed_generic); a part that may have met a special case (an operand at infinity, P = +-Q) is redone by
// job_msm_ladder_safe.  SPLIT = false: the part is the whole job (no masked look-ups).
template <bool SPLIT>
TC_HD G2Jac job_msm_ladder_part(size_t n, const int32_t* tbl, const uint8_t* codes, int nbits, MsmPart part) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  bool exc = false;
  G2Jac acc;
  TC_NOUNROLL for (int col = nbits; col >= 0; col--) {
    tc_fair();
    const uint8_t* cc = codes + (size_t)col * shares4;
    size_t t = 0;
    if (col != nbits) {
      acc = jac_dbl(acc);
    } else {
      const G2Affine e0 = msm_load_entry(tbl + (part.s0 * 8 + (size_t)(cc[part.s0] & 7)) * kMsmEntryWords);
      acc = G2Jac::from_affine(e0);
      exc = e0.inf;
      t = 1;
    }
    TC_NOUNROLL for (; t < part.trips; t++) {
      const bool take = !SPLIT || part.s0 + t < part.s1;
      const size_t s = take ? part.s0 + t : part.s0;
      const uint32_t code = cc[s];
      G2Affine e = msm_load_entry(tbl + (s * 8 + (code & 7)) * kMsmEntryWords);
      if (col != nbits) e.y = Fq2::select((code >> 3) & 1, -e.y, e.y);
      if (SPLIT) {
        bool hit = false;
        const G2Jac sum = jac_add_mixed_generic(acc, e, hit);
        exc = exc || (take && hit);
        acc = G2Jac::select(take, sum, acc);
      } else {
        acc = jac_add_mixed_generic(acc, e, exc);
      }
    }
  }
  if (wave_any(exc)) acc = G2Jac::select(exc, job_msm_ladder_safe(n, tbl, codes, nbits, part), acc);
  return acc;
}
TC_HD G2Jac job_msm_ladder(size_t n, const int32_t* tbl, const uint8_t* codes, int nbits = 64) {
  return job_msm_ladder_part<false>(n, tbl, codes, nbits, msm_part(n));
}

// ---- the same two stages in G1 (threshold decryption at large thresholds: PublicKeySet::decrypt, ---------------------
// /root/reference/src/lib.rs:618-626, at t = 67; tc_g1_lincomb_batch from 8 points on) ---------------------------------
// G1 has the 2-dimensional GLV decomposition k = k1 + k2 x^2 with 128-bit halves (tc_gls.h), sign-aligned over 129
// columns:  k1 = sum_i s_i 2^i,  k2 = sum_i s_i u_i 2^i  (s_i = +-1, s_128 = +1, u_i in {0, 1}).  To run the SAME ladder as
// G2 -- 64 steps, one table look-up and one mixed addition per share and step, an 8-entry table per share -- the columns
// are taken two at a time (base 4): columns 2c and 2c+1 contribute
//     4^c sigma (A P + B phi'),   phi' = -phi(P) = [x^2] P,  sigma = s_{2c+1},
//     A = 2 + e in {1, 3},  B = e u_{2c} + 2 u_{2c+1},  e = s_{2c} s_{2c+1}
// i.e. one of  { P, P - phi', P + 2 phi', P + phi',  3P, 3P + phi', 3P + 2 phi', 3P + 3 phi' }  (index u_{2c} + 2 u_{2c+1},
// + 4 when e = +1), added or subtracted; the top column (i = 128) starts the accumulator with P or P + phi'.  A step of
// the ladder is TWO doublings + the additions: per share 64 mixed additions + a table of six affine additions + 1/2
// inversion, per job 128 doublings -- instead of 255 doublings + ~240 additions per chunk of four shares in the per-lane
// Straus ladders (tc_threshold.h straus_chunk).  One lane per job (G1), tables of 128 B per entry in HBM.
constexpr int kMsmEntryWordsG1 = 32;  // x: 14 limbs + 2 words (word 15: infinity flag), y: 14 limbs + 2 words

TC_HD void msm_store_entry_g1(int32_t* e, const G1Affine& p) {
  const Fq x = p.x.norm(), y = p.y.norm();
#if defined(TC_BOUND_CHECK)
  if (x.val() > 2.1f || y.val() > 2.1f) tc_bound_fail(x.val(), y.val());
#endif
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    e[i] = x.l[i];
    e[16 + i] = y.l[i];
  }
  e[14] = 0;
  e[15] = p.inf ? 1 : 0;
  e[30] = e[31] = 0;
}
TC_HD G1Affine msm_load_entry_g1(const int32_t* e) {
  G1Affine p;
  p.x = tbl_load_fq((const tbl_word*)e);
  p.y = tbl_load_fq((const tbl_word*)e + 16);
  p.inf = e[15] != 0;
  return p;
}
// k (8 canonical words, < r) -> the 65 column codes of the base-4 sign-aligned form above, written `stride` bytes apart:
// bits 0..2 table index, bit 3 = subtract (columns 0 .. 63); column 64: the index of the starting entry (0 or 3).
// Returns true when r - k was recoded instead of k (k even; k1 = k mod x^2 must be odd): the caller negates the bases.
// nbits < 128 (even): SHORT scalars -- the caller promises k = k1 + k2 x^2 with k1 odd and k1, k2 < 2^nbits (the random
// scalars of the batch validation of decryption shares); columns 0 .. nbits / 2 - 1 and the starting entry at column
// nbits / 2, never flipped; *fits = false when the promise does not hold (the job fails).
TC_HD bool msm_g1_recode(const uint32_t* k, uint8_t* codes, size_t stride, int nbits = 128, bool* fits = nullptr) {
  if (nbits < 128) {
    tc_u128 k1, k2;
    glv_decompose(k, &k1, &k2);
    const bool good = (k1 & 1) != 0 && (k1 >> nbits) == 0 && (k2 >> nbits) == 0;
    if (fits) *fits = good;
    if (!good) {
      k1 = 1;
      k2 = 0;
    }
    const tc_u128 neg = ~((k1 | 1) >> 1);
    tc_u128 u = 0;
    TC_NOUNROLL for (int i = 0; i < nbits; i++) {
      const tc_u128 odd = k2 & 1;
      u |= odd << i;
      k2 = (k2 >> 1) + (odd & (neg >> i));
    }
    const int nc = nbits / 2;
    TC_NOUNROLL for (int c = 0; c < nc; c++) {
      const uint32_t n0 = (uint32_t)(neg >> (2 * c)) & 1u, n1 = (uint32_t)(neg >> (2 * c + 1)) & 1u;
      const uint32_t u0 = (uint32_t)(u >> (2 * c)) & 1u, u1 = (uint32_t)(u >> (2 * c + 1)) & 1u;
      codes[(size_t)c * stride] = (uint8_t)(u0 | (u1 << 1) | ((n0 == n1 ? 1u : 0u) << 2) | (n1 << 3));
    }
    codes[(size_t)nc * stride] = (uint8_t)(k2 ? 3 : 0);
    return false;
  }
  if (fits) *fits = true;
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
  const tc_u128 neg = ~((k1 | 1) >> 1);  // bit i: s_i = -1 (i < 128)
  tc_u128 u = 0;
  TC_NOUNROLL for (int i = 0; i < 128; i++) {
    const tc_u128 odd = k2 & 1;
    u |= odd << i;
    k2 = (k2 >> 1) + (odd & (neg >> i));
  }
  TC_NOUNROLL for (int c = 0; c < 64; c++) {
    const uint32_t n0 = (uint32_t)(neg >> (2 * c)) & 1u, n1 = (uint32_t)(neg >> (2 * c + 1)) & 1u;
    const uint32_t u0 = (uint32_t)(u >> (2 * c)) & 1u, u1 = (uint32_t)(u >> (2 * c + 1)) & 1u;
    codes[(size_t)c * stride] = (uint8_t)(u0 | (u1 << 1) | ((n0 == n1 ? 1u : 0u) << 2) | (n1 << 3));
  }
  codes[(size_t)64 * stride] = (uint8_t)(k2 ? 3 : 0);  // u_128
  return flip;
}

// Stage T in G1 for the chunk `c` of one job: tbl = (4 * chunks) shares x 8 entries x 32 words; codes: 65 columns x
// (4 * chunks) shares, one byte each.
// build = false: codes and operand validity only -- the tables of a point set SHARED by every job (short-scalar mode: the
// recoding never negates a base, so the entries depend on the points alone) are written by one job for all.
TC_HD bool job_msm_tables_g1(size_t n, size_t c, const uint8_t* points, const uint32_t* scalars, int32_t* tbl, uint8_t* codes, int nbits = 128,
                             bool build = true) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G1Affine b0[kMsmChunk];
  G1Jac mult[2 * kMsmChunk];  // 2P, 3P
  bool ok = true;
  TC_NOUNROLL for (int k = 0; k < kMsmChunk; k++) {
    const size_t s = c * kMsmChunk + k;
    G1Affine p = G1Affine::infinity();
    uint32_t sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s < n) {
      ok &= g1_decode_uncompressed(points + s * 96, p);
      for (int w = 0; w < 8; w++) sc[w] = scalars[s * 8 + w];
      ok &= limbs_lt_p<FrParams>(sc);
    }
    if (!ok) p = G1Affine::infinity();
    if (s >= n && nbits < 128) sc[0] = 1;  // padding shares of the short-scalar mode: a valid short scalar (their point is the identity)
    bool fits = true;
    const bool flip = msm_g1_recode(sc, codes + s, shares4, nbits, &fits);
    ok &= fits;
    p.y = Fq::select(flip, -p.y, p.y).norm();
    b0[k] = p;
    mult[2 * k] = jac_dbl(G1Jac::from_affine(p));
    mult[2 * k + 1] = jac_add_mixed(mult[2 * k], p);
  }
  if (!wave_any(build)) return ok;
  // 2P and 3P of the four shares to affine with one inversion
  G1Affine ma[2 * kMsmChunk];
  jac_batch_to_affine<Fq, 2 * kMsmChunk>(mult, ma, 2 * kMsmChunk);
  G1Jac sums[6 * kMsmChunk];
  TC_NOUNROLL for (int k = 0; k < kMsmChunk; k++) {
    const G1Affine p = b0[k], p2 = ma[2 * k], p3 = ma[2 * k + 1];
    G1Affine f1 = g1_phi(p), f2 = g1_phi(p2), f3 = g1_phi(p3);  // phi' = -phi: negate y below
    G1Affine m1 = f1;                                            // -phi' (P) = phi(P)
    f1.y = (-f1.y).norm();
    f2.y = (-f2.y).norm();
    f3.y = (-f3.y).norm();
    sums[6 * k + 0] = jac_add_affine(p, m1);   // index 1: P - phi'
    sums[6 * k + 1] = jac_add_affine(p, f2);   // index 2: P + 2 phi'
    sums[6 * k + 2] = jac_add_affine(p, f1);   // index 3: P + phi'
    sums[6 * k + 3] = jac_add_affine(p3, f1);  // index 5: 3P + phi'
    sums[6 * k + 4] = jac_add_affine(p3, f2);  // index 6: 3P + 2 phi'
    sums[6 * k + 5] = jac_add_affine(p3, f3);  // index 7: 3P + 3 phi'
  }
  G1Affine aff[6 * kMsmChunk];
  jac_batch_to_affine<Fq, 6 * kMsmChunk>(sums, aff, 6 * kMsmChunk);
  TC_NOUNROLL for (int k = 0; k < kMsmChunk; k++) {
    if (!build) break;
    const size_t s = c * kMsmChunk + k;
    int32_t* t = tbl + s * 8 * kMsmEntryWordsG1;
    msm_store_entry_g1(t, b0[k]);
    msm_store_entry_g1(t + 4 * kMsmEntryWordsG1, ma[2 * k + 1]);
    TC_NOUNROLL for (int m = 0; m < 3; m++) {
      msm_store_entry_g1(t + (1 + m) * kMsmEntryWordsG1, aff[6 * k + m]);
      msm_store_entry_g1(t + (5 + m) * kMsmEntryWordsG1, aff[6 * k + 3 + m]);
    }
  }
  return ok;
}

// Stage L in G1 (one lane per part of a job): every special case of the addition handled -- the slow path and the
// g++ reference of the fast form below
TC_HD_NOINLINE G1Jac job_msm_ladder_g1_safe(size_t n, const int32_t* tbl, const uint8_t* codes, MsmPart part, int top = 64) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  G1Jac acc = G1Jac::infinity();
  TC_NOUNROLL for (int col = top; col >= 0; col--) {
    if (col != top) acc = jac_dbl(jac_dbl(acc));
    const uint8_t* cc = codes + (size_t)col * shares4;
    TC_NOUNROLL for (size_t t = 0; t < part.trips; t++) {
      const bool take = part.s0 + t < part.s1;
      const size_t s = take ? part.s0 + t : part.s0;
      const uint32_t code = cc[s];
      G1Affine e = msm_load_entry_g1(tbl + (s * 8 + (code & 7)) * kMsmEntryWordsG1);
      if (col != top) e.y = Fq::select((code >> 3) & 1, -e.y, e.y);
      acc = G1Jac::select(take, jac_add_mixed(acc, e), acc);
    }
  }
  return acc;
}
template <bool SPLIT>
TC_HD G1Jac job_msm_ladder_g1_part(size_t n, const int32_t* tbl, const uint8_t* codes, MsmPart part, int top = 64) {
  const size_t shares4 = msm_chunks(n) * kMsmChunk;
  bool exc = false;
  G1Jac acc;
  TC_NOUNROLL for (int col = top; col >= 0; col--) {
    tc_fair();
    const uint8_t* cc = codes + (size_t)col * shares4;
    size_t t = 0;
    if (col != top) {
      acc = jac_dbl(jac_dbl(acc));
    } else {
      const G1Affine e0 = msm_load_entry_g1(tbl + (part.s0 * 8 + (size_t)(cc[part.s0] & 7)) * kMsmEntryWordsG1);
      acc = G1Jac::from_affine(e0);
      exc = e0.inf;
      t = 1;
    }
    TC_NOUNROLL for (; t < part.trips; t++) {
      const bool take = !SPLIT || part.s0 + t < part.s1;
      const size_t s = take ? part.s0 + t : part.s0;
      const uint32_t code = cc[s];
      G1Affine e = msm_load_entry_g1(tbl + (s * 8 + (code & 7)) * kMsmEntryWordsG1);
      if (col != top) e.y = Fq::select((code >> 3) & 1, -e.y, e.y);
      if (SPLIT) {
        bool hit = false;
        const G1Jac sum = jac_add_mixed_generic(acc, e, hit);
        exc = exc || (take && hit);
        acc = G1Jac::select(take, sum, acc);
      } else {
        acc = jac_add_mixed_generic(acc, e, exc);
      }
    }
  }
  if (wave_any(exc)) acc = G1Jac::select(exc, job_msm_ladder_g1_safe(n, tbl, codes, part, top), acc);
  return acc;
}

}  // namespace tc
