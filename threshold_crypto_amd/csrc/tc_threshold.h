// Lagrange-at-zero coefficients and the Straus multi-scalar combiner (device side).
// Replaces, for batches, the private `interpolate` of /root/reference/src/lib.rs:719-767.
#pragma once
#include "tc_gls.h"

namespace tc {

// lambda_i for sample position i of one job, exactly as the reference builds it:
//   numerator   x_prod[i] = prod_{j != i (by POSITION)} x_j          (src/lib.rs:739-751)
//   denominator prod_{j : x_j != x_i (by VALUE)} (x_j - x_i)          (src/lib.rs:757-762)
//   x_j = idx_j + 1                                                   (src/lib.rs:769-773)
// Returns false when the denominator is not invertible (Error::DuplicateEntry, :763 --
// unreachable in practice because equal abscissae are filtered out by value).
TC_HD bool lagrange_coeff_at_zero(const uint64_t* idx, int t, int i, Fr& out) {
  const Fr one = Fr::one();
  const Fr xi = fr_from_u64(idx[i]) + one;
  Fr num = one;
  Fr den = one;
  TC_NOUNROLL for (int j = 0; j <= t; j++) {
    Fr xj = fr_from_u64(idx[j]) + one;
    if (j != i) num = num * xj;
    if (xj != xi) den = den * (xj - xi);
  }
  if (den.is_zero()) return false;
  out = num * den.inv();
  return true;
}

// The same coefficient for abscissae given as Fr values (`T: IntoFr` = Fr, negative i32 / i64: /root/reference/src/into_fr.rs:10-14,
// 28-56): xs = (t+1) x 8 canonical LE words, x_j = xs_j + 1.  Identical construction -- numerator by POSITION, denominator
// filtered by VALUE -- so repeated abscissae behave as in the reference.
TC_HD bool lagrange_coeff_at_zero_fr(const uint32_t* xs, int t, int i, Fr& out) {
  const Fr one = Fr::one();
  const Fr xi = Fr::from_canonical(xs + 8 * i) + one;
  Fr num = one;
  Fr den = one;
  TC_NOUNROLL for (int j = 0; j <= t; j++) {
    const Fr xj = Fr::from_canonical(xs + 8 * j) + one;
    if (j != i) num = num * xj;
    if (xj != xi) den = den * (xj - xi);
  }
  if (den.is_zero()) return false;
  out = num * den.inv();
  return true;
}

// All t+1 coefficients of one job with ONE field inversion (Montgomery's trick) instead of one Fermat power
// per coefficient -- what the reference does at src/lib.rs:763 costs ~320 Fr multiplications each, more than
// the t multiplications of the denominator itself; at t = 67 the per-coefficient kernel was 15 % of the whole
// combination.  Same values: lambda_i = (prod_{j != i by position} x_j) * (prod_{j: x_j != x_i} (x_j - x_i))^-1.
// ws: 4 (t+1) x 8 words of scratch (x, denominators, their prefix products, prefix products of x), Montgomery form.
// out: (t+1) x 8 canonical words.  Equal abscissae are filtered BY VALUE (src/lib.rs:758), so a denominator is
// never zero and DuplicateEntry stays unreachable, as in the reference.
TC_HD Fr lagrange_denominator(const uint64_t* idx, int n, int i);  // below
TC_HD uint8_t lagrange_all_at_zero(const uint64_t* idx, int t, uint32_t* out, uint32_t* ws) {
  const int n = t + 1;
  uint32_t* xm = ws;
  uint32_t* den = ws + (size_t)n * 8;
  uint32_t* pre = ws + (size_t)2 * n * 8;
  uint32_t* px = ws + (size_t)3 * n * 8;
  auto put = [](uint32_t* dst, const Fr& v) { TC_UNROLL for (int i = 0; i < 8; i++) dst[i] = v.v.l[i]; };
  auto get = [](const uint32_t* src) { Fr v; TC_UNROLL for (int i = 0; i < 8; i++) v.v.l[i] = src[i]; return v; };
  const Fr one = Fr::one();
  Fr accx = one;
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    const Fr x = fr_from_u64(idx[i]) + one;
    put(xm + (size_t)i * 8, x);
    put(px + (size_t)i * 8, accx);  // prod_{k < i} x_k
    accx = accx * x;
  }
  Fr accd = one;
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    const Fr d = lagrange_denominator(idx, n, i);  // x_j != x_i  <=>  idx_j != idx_i  (u64 + 1 is injective mod r)
    put(den + (size_t)i * 8, d);
    put(pre + (size_t)i * 8, accd);  // prod_{k < i} den_k
    accd = accd * d;
  }
  if (accd.is_zero()) {  // unreachable (see above); kept so that a zero can never be inverted silently
    TC_NOUNROLL for (int k = 0; k < n * 8; k++) out[k] = 0;
    return TC_JOB_DUPLICATE_ENTRY;
  }
  Fr inv = accd.inv();
  Fr sufx = one;  // prod_{k > i} x_k
  TC_NOUNROLL for (int i = n - 1; i >= 0; i--) {
    const Fr dinv = inv * get(pre + (size_t)i * 8);
    inv = inv * get(den + (size_t)i * 8);
    const Fr lam = get(px + (size_t)i * 8) * sufx * dinv;
    sufx = sufx * get(xm + (size_t)i * 8);
    lam.to_canonical(out + (size_t)i * 8);
  }
  return TC_JOB_OK;
}

// The same computation split for the device (k_combine.hip k_lagrange_den / k_lagrange_finish): the O(t^2)
// denominators are one lane per (job, i) with the job's abscissae staged in LDS; the prefix products, the single
// inversion and the coefficients are one lane per job.
//   den_i = prod_{j : idx_j != idx_i} (x_j - x_i)  in Montgomery form.
// x_j - x_i = idx_j - idx_i as INTEGERS (the + 1 of into_fr_plus_1 cancels; |difference| < 2^64), so the product is
// gathered in 64-bit chunks -- eight factors below 2^8 at N = 200 -- and only a full chunk costs field multiplications
// (its conversion and the product: 2 instead of 8); the sign is applied at the end.  A chunk is closed for the whole
// wave as soon as one lane's would overflow, so the lanes keep one control flow.
TC_HD Fr lagrange_denominator(const uint64_t* idx, int n, int i) {
  const uint64_t vi = idx[i];
  Fr d = Fr::one();
  uint64_t chunk = 1;
  bool neg = false;
  TC_NOUNROLL for (int j = 0; j < n; j++) {
    const uint64_t vj = idx[j];
    const bool lt = vj < vi;
    const uint64_t diff = lt ? vi - vj : vj - vi;
    const uint64_t m = diff ? diff : 1;  // equal abscissae are filtered by value (src/lib.rs:758): factor 1
    neg = neg != (lt && diff != 0);
    // bits(chunk) + bits(m) <= 64  =>  chunk * m < 2^64
    const bool full = __builtin_clzll(chunk) + __builtin_clzll(m) < 64;
    if (wave_any(full)) {
      d = d * fr_from_u64(chunk);
      chunk = m;
    } else {
      chunk *= m;
    }
  }
  d = d * fr_from_u64(chunk);
  return neg ? Fr::zero() - d : d;
}
//   xm, den: n x 8 words (Montgomery) as written by the denominator stage; pre: n x 8 words of scratch
TC_HD uint8_t lagrange_finish(int n, const uint32_t* xm, const uint32_t* den, uint32_t* pre, uint32_t* out) {
  auto put = [](uint32_t* dst, const Fr& v) { TC_UNROLL for (int i = 0; i < 8; i++) dst[i] = v.v.l[i]; };
  auto get = [](const uint32_t* src) { Fr v; TC_UNROLL for (int i = 0; i < 8; i++) v.v.l[i] = src[i]; return v; };
  const Fr one = Fr::one();
  Fr accd = one;
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    put(pre + (size_t)i * 8, accd);
    accd = accd * get(den + (size_t)i * 8);
  }
  if (accd.is_zero()) {
    TC_NOUNROLL for (int k = 0; k < n * 8; k++) out[k] = 0;
    return TC_JOB_DUPLICATE_ENTRY;
  }
  // lambda_i = (prod_{k < i} x_k) (prod_{k > i} x_k) / den_i: the prefix products of x are rebuilt on the way back
  // from the total (one more inversion-free pass: P_i = P_{i+1} / x_i is avoided by keeping them in `out`)
  Fr accx = one;
  TC_NOUNROLL for (int i = 0; i < n; i++) {
    put(out + (size_t)i * 8, accx);  // prod_{k < i} x_k, parked in the output buffer
    accx = accx * get(xm + (size_t)i * 8);
  }
  Fr inv = accd.inv();
  Fr sufx = one;
  TC_NOUNROLL for (int i = n - 1; i >= 0; i--) {
    const Fr dinv = inv * get(pre + (size_t)i * 8);
    inv = inv * get(den + (size_t)i * 8);
    const Fr lam = get(out + (size_t)i * 8) * sufx * dinv;
    sufx = sufx * get(xm + (size_t)i * 8);
    lam.to_canonical(out + (size_t)i * 8);
  }
  return TC_JOB_OK;
}

// sum_{k < K} s_k * P_k for K <= 4 points with per-lane 255-bit scalars: joint (Straus)
// double-and-add over a (2^K - 1)-entry subset-sum table held in the lane's scratch.
// One shared doubling chain for the K points; control flow is lane-uniform except the
// "all K bits are zero" skip.  sc[k] points at 8 little-endian u32 words (canonical, < r).
template <class F, int K>
TC_HD_NOINLINE Jac<F> straus_chunk_ladder_safe(const Affine<F>* tbl, const uint32_t (*sc)[8]) {
  Jac<F> acc = Jac<F>::infinity();
  TC_NOUNROLL for (int bit = 254; bit >= 0; bit--) {
    acc = jac_dbl(acc);
    uint32_t m = 0;
    TC_UNROLL for (int k = 0; k < K; k++) m |= ((sc[k][bit >> 5] >> (bit & 31)) & 1u) << k;
    if (m) acc = jac_add_mixed(acc, tbl[m]);
  }
  return acc;
}
template <class F, int K>
TC_HD Jac<F> straus_chunk(const Affine<F>* pts, const uint32_t (*sc)[8]) {
  // subset sums by mixed additions, then one common Z for the whole table (tc_curve.h
  // jac_batch_to_common_z, no inversion) so that the 255-step ladder also runs on mixed additions
  Affine<F> tbl[1 << K];
  F zc;
  {
    Jac<F> sums[(1 << K) - 1];
    Affine<F> sums_aff[(1 << K) - 1];
    int slot[1 << K];
    int ns = 0;
    TC_NOUNROLL for (int m = 1; m < (1 << K); m++) {
      const int low = __builtin_ctz((unsigned)m);
      const int rest = m & (m - 1);
      if (!rest) {
        slot[m] = -1;
      } else {
        const int rs = slot[rest];
        sums[ns] = (rs < 0) ? jac_add_affine(pts[__builtin_ctz((unsigned)rest)], pts[low]) : jac_add_mixed(sums[rs], pts[low]);
        slot[m] = ns++;
      }
    }
    zc = jac_batch_to_common_z(sums, sums_aff, ns);
    const F zc2 = zc.sqr();
    const F zc3 = zc2 * zc;
    TC_NOUNROLL for (int m = 1; m < (1 << K); m++)
      tbl[m] = (slot[m] >= 0) ? sums_aff[slot[m]] : affine_scale_z(pts[__builtin_ctz((unsigned)m)], zc2, zc3);
  }
  Jac<F> acc = Jac<F>::infinity();
  bool started = false, exc = false;
  TC_NOUNROLL for (int bit = 254; bit >= 0; bit--) {
    tc_fair();
    acc = jac_dbl(acc);  // the identity doubles to itself
    uint32_t m = 0;
    TC_UNROLL for (int k = 0; k < K; k++) m |= ((sc[k][bit >> 5] >> (bit & 31)) & 1u) << k;
    if (wave_any(m != 0)) {
      // generic-case addition + a `started` flag (tc_curve.h jac_add_mixed_generic); special cases go to the safe ladder
      const Affine<F> e = tbl[m ? m : 1];
      bool hit = false;
      const Jac<F> sum = jac_add_mixed_generic(acc, e, hit);
      const bool take = m != 0;
      exc = exc || (take && (started ? hit : e.inf));
      acc = Jac<F>::select(take, Jac<F>::select(started, sum, Jac<F>::from_affine(e)), acc);
      started = started || take;
    }
  }
  if (wave_any(exc)) acc = Jac<F>::select(exc, straus_chunk_ladder_safe<F, K>(tbl, sc), acc);
  acc.z = coord_norm(acc.z * zc);
  return acc;
}

// chunk of a G1 linear combination (G2 combinations run through tc_msm.h: per-share psi tables in HBM, one ladder)
TC_HD G1Jac lincomb_chunk4(const G1Affine* pts, const uint32_t (*sc)[8]) { return straus_chunk<Fq, 4>(pts, sc); }

// ---- small-index fast path ------------------------------------------------------------------
// Share indices are node numbers, so the abscissae x_i = idx_i + 1 are small integers and the
// Lagrange coefficients are ratios of SMALL integers:
//     lambda_i = num_i / den_i,  num_i = prod_{j != i} x_j,  den_i = prod_{j != i} (x_j - x_i).
// With D = lcm_i |den_i| and the integers c_i = num_i * (D / den_i), all divided by their common
// factor (a few bits to a few tens of bits),
//     sum_i lambda_i S_i = [D^-1 mod r] ( sum_i c_i S_i ),
// i.e. one short joint ladder over |c_i| followed by ONE full-size (GLS / GLV) multiplication,
// instead of t+1 full-size ones.  Same group element, same bytes.  Returns false -- caller
// uses the general path -- when indices repeat (the reference's by-value filtering then
// applies, src/lib.rs:758), are large, or a product leaves 63 bits.
TC_HD uint64_t gcd_u64(uint64_t a, uint64_t b) {
  TC_NOUNROLL while (wave_any(b != 0)) {
    if (b != 0) {
      const uint64_t t = a % b;
      a = b;
      b = t;
    }
  }
  return a;
}

template <int K>
TC_HD bool lagrange_small_coeffs(const uint64_t* idx, uint64_t* c_abs, bool* c_neg, uint64_t* d_abs, bool* d_neg) {
  int64_t x[K];
  TC_UNROLL for (int i = 0; i < K; i++) {
    if (idx[i] >= 65535) return false;
    x[i] = (int64_t)idx[i] + 1;
  }
  uint64_t den[K];
  bool den_neg[K];
  TC_UNROLL for (int i = 0; i < K; i++) {
    int64_t d = 1;
    TC_UNROLL for (int j = 0; j < K; j++) {
      if (j == i) continue;
      const int64_t diff = x[j] - x[i];
      if (diff == 0) return false;
      if (__builtin_mul_overflow(d, diff, &d)) return false;
    }
    den_neg[i] = d < 0;
    den[i] = (uint64_t)(d < 0 ? -d : d);
  }
  uint64_t D = 1;  // lcm of the denominators
  TC_UNROLL for (int i = 0; i < K; i++) {
    const uint64_t g = gcd_u64(D, den[i]);
    if (__builtin_mul_overflow(D / g, den[i], &D)) return false;
    if (D >> 62) return false;
  }
  TC_UNROLL for (int i = 0; i < K; i++) {
    uint64_t c = D / den[i];
    TC_UNROLL for (int j = 0; j < K; j++) {
      if (j == i) continue;
      if (__builtin_mul_overflow(c, (uint64_t)x[j], &c)) return false;
    }
    if (c >> 63) return false;
    c_neg[i] = den_neg[i];
    c_abs[i] = c;
  }
  // common factor of all numerators and the denominator: for 4 of 10 signers it shortens the
  // longest c_i from 14 to 9 bits (the ladder over the c_i runs to the longest one in the wave)
  uint64_t g = D;
  TC_UNROLL for (int i = 0; i < K; i++) g = gcd_u64(g, c_abs[i]);
  TC_UNROLL for (int i = 0; i < K; i++) c_abs[i] /= g;
  *d_neg = false;
  *d_abs = D / g;
  return true;
}

// D^-1 mod r as 8 canonical LE words, for 0 < D < 2^63, without a field exponentiation (an Fr
// Fermat inversion is ~230 k VALU instructions per lane on 32-bit saturated limbs, a tenth of the
// whole fast-path job):
//     k = -(r mod D)^-1 mod D        64-bit extended Euclid
//     D^-1 = (1 + r k) / D           exact, bitwise long division
TC_HD void fr_inverse_of_small(uint64_t d_abs, bool d_neg, uint32_t* out_words) {
  uint64_t rl[4];
  TC_UNROLL for (int i = 0; i < 4; i++) rl[i] = (uint64_t)FR_P[2 * i] | ((uint64_t)FR_P[2 * i + 1] << 32);
  uint64_t q[5] = {1, 0, 0, 0, 0};  // D = 1
  if (d_abs > 1) {
    // D below 2^32 in every lane of the wave (node numbers: always, in practice): both long divisions run on 32-bit
    // words with the 64-bit divide; otherwise bit by bit
    const bool small = !wave_any((d_abs >> 32) != 0);
    // r mod D
    uint64_t r1 = 0;
    if (small) {
      TC_UNROLL for (int i = 7; i >= 0; i--) r1 = ((r1 << 32) | FR_P[i]) % d_abs;
    } else {
      TC_NOUNROLL for (int bit = 254; bit >= 0; bit--) {
        r1 = (r1 << 1) | ((rl[bit >> 6] >> (bit & 63)) & 1ull);
        if (r1 >= d_abs) r1 -= d_abs;
      }
    }
    // r1^-1 mod D (gcd = 1: r is prime and D < r)
    int64_t t = 0, newt = 1;
    uint64_t a = d_abs, b = r1;
    TC_NOUNROLL while (wave_any(b != 0)) {
      if (b != 0) {
        const uint64_t quo = a / b;
        const uint64_t nb = a - quo * b;
        const int64_t nt = t - (int64_t)quo * newt;
        a = b;
        b = nb;
        t = newt;
        newt = nt;
      }
    }
    const uint64_t inv = t < 0 ? (uint64_t)(t + (int64_t)d_abs) : (uint64_t)t;
    const uint64_t k = d_abs - inv;  // -inv mod D, in [1, D)
    // N = 1 + r * k  (5 words)
    uint64_t n[5];
    unsigned __int128 carry = 1;
    TC_UNROLL for (int i = 0; i < 4; i++) {
      carry += (unsigned __int128)rl[i] * k;
      n[i] = (uint64_t)carry;
      carry >>= 64;
    }
    n[4] = (uint64_t)carry;
    // Q = N / D
    uint64_t rem = 0;
    TC_UNROLL for (int i = 0; i < 5; i++) q[i] = 0;
    if (small) {
      TC_UNROLL for (int i = 9; i >= 0; i--) {
        const uint64_t cur = (rem << 32) | (uint32_t)(n[i >> 1] >> (32 * (i & 1)));
        const uint64_t qq = cur / d_abs;  // < 2^32 because rem < D
        rem = cur - qq * d_abs;
        q[i >> 1] |= qq << (32 * (i & 1));
      }
    } else {
      TC_NOUNROLL for (int bit = 319; bit >= 0; bit--) {
        rem = (rem << 1) | ((n[bit >> 6] >> (bit & 63)) & 1ull);
        if (rem >= d_abs) {
          rem -= d_abs;
          q[bit >> 6] |= 1ull << (bit & 63);
        }
      }
    }
  }
  if (d_neg) {  // r - Q
    uint64_t borrow = 0;
    TC_UNROLL for (int i = 0; i < 4; i++) {
      const uint64_t x = rl[i], y = q[i];
      const uint64_t d1 = x - y;
      const uint64_t d2 = d1 - borrow;
      borrow = (uint64_t)(x < y) | (uint64_t)(d1 < borrow);
      q[i] = d2;
    }
  }
  TC_UNROLL for (int i = 0; i < 4; i++) {
    out_words[2 * i] = (uint32_t)q[i];
    out_words[2 * i + 1] = (uint32_t)(q[i] >> 32);
  }
}

// the ladder below starts at the highest set bit of any scalar in the WAVE so that all lanes
// run the same trip count
TC_HD uint32_t wave_max_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  TC_UNROLL for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
#endif
  return v;
}

// sum_k c_k * P_k for K <= 4 points and 64-bit scalars: the joint ladder over the subset-sum table, with every special
// case of the addition handled (the slow path of straus_small below)
template <class F, int K>
TC_HD Jac<F> straus_small_safe_body(const Jac<F>* tbl, const uint64_t* c, uint32_t bits) {
  Jac<F> acc = Jac<F>::infinity();
  TC_NOUNROLL for (int bit = (int)bits - 1; bit >= 0; bit--) {
    acc = jac_dbl(acc);
    uint32_t m = 0;
    TC_UNROLL for (int k = 0; k < K; k++) m |= (uint32_t)((c[k] >> bit) & 1ull) << k;
    if (m) acc = jac_add(acc, tbl[m]);
  }
  return acc;
}
template <class F, int K>
TC_HD_NOINLINE Jac<F> straus_small_safe(const Jac<F>* tbl, const uint64_t* c, uint32_t bits) { return straus_small_safe_body<F, K>(tbl, c, bits); }
// The fast form: a `started` flag instead of tests for the identity, the branch-free generic addition
// (tc_curve.h jac_add_generic), and a second pass through straus_small_safe for the lanes that may have met a special
// case (an input or a subset sum at infinity, P = +-Q).  INLINE_SAFE: the fallback inlined too, so that the CALLING KERNEL's launch
// bounds govern its registers -- an out-of-line device routine has none of its own, and the fallback's 38 accumulation registers
// dropped the two-waves-per-SIMD build of the G1 combine kernel to one ("failed to meet occupancy target").
template <class F, int K, bool INLINE_SAFE = false>
TC_HD Jac<F> straus_small(const Affine<F>* pts, const uint64_t* c) {
  Jac<F> tbl[1 << K];
  tbl[0] = Jac<F>::infinity();
  TC_NOUNROLL for (int m = 1; m < (1 << K); m++) {
    const int low = __builtin_ctz((unsigned)m);
    const int rest = m & (m - 1);
    if (!rest) tbl[m] = Jac<F>::from_affine(pts[low]);
    else if (!(rest & (rest - 1))) tbl[m] = jac_add_affine(pts[__builtin_ctz((unsigned)rest)], pts[low]);
    else tbl[m] = jac_add_mixed(tbl[rest], pts[low]);
  }
  uint64_t any = 0;
  TC_UNROLL for (int k = 0; k < K; k++) any |= c[k];
  uint32_t bits = any ? 64u - (uint32_t)__builtin_clzll(any) : 0u;
  bits = wave_max_u32(bits);
  if (bits > 64) bits = 64;  // lanes that left for the general path contribute undefined values
  Jac<F> acc = Jac<F>::infinity();
  bool started = false, exc = false;
  TC_NOUNROLL for (int bit = (int)bits - 1; bit >= 0; bit--) {
    tc_fair();
    acc = jac_dbl(acc);  // the identity (0 : 1 : 0) doubles to itself
    uint32_t m = 0;
    TC_UNROLL for (int k = 0; k < K; k++) m |= (uint32_t)((c[k] >> bit) & 1ull) << k;
    if (wave_any(m != 0)) {
      const Jac<F> e = tbl[m];
      bool hit = false;
      const Jac<F> sum = jac_add_generic(acc, e, hit);
      const bool take = m != 0;
      exc = exc || (take && (started ? hit : maybe_zero56(e.z)));
      acc = Jac<F>::select(take, Jac<F>::select(started, sum, e), acc);
      started = started || take;
    }
  }
  if (wave_any(exc)) acc = Jac<F>::select(exc, INLINE_SAFE ? straus_small_safe_body<F, K>(tbl, c, bits) : straus_small_safe<F, K>(tbl, c, bits), acc);
  return acc;
}

// out-of-line form: the short ladder gets its own register allocation instead of sharing the caller's
template <class F, int K>
TC_HD_NOINLINE Jac<F> straus_small_call(const Affine<F>* pts, const uint64_t* c) { return straus_small<F, K>(pts, c); }

}  // namespace tc
