// Per-lane job bodies: one independent (message, share-set) job or one curve op per lane.
// The __global__ kernels in tc_kernels.hip call these with the lane's slice of the batch;
// tests/hostsim calls the same bodies from a host loop (test harness only).
#pragma once
#include "tc_codec.h"
#include "tc_gls.h"
#include "tc_hash.h"
#include "tc_pairing.h"
#include "tc_threshold.h"

namespace tc {

template <class F>
struct PointIO;

// lanes that work on one job: Fq2 jobs are spread over a lane pair in the hipcc build (tc_common.h)
template <class F>
struct JobLanes {
  static constexpr int N = 1;
};
template <>
struct JobLanes<Fq2> {
  static constexpr int N = kG2Lanes;
};

template <>
struct PointIO<Fq> {
  static constexpr int BYTES = 96;
  static constexpr int CBYTES = 48;
  TC_HD static bool decode(const uint8_t* b, Affine<Fq>& p) { return g1_decode_uncompressed(b, p); }
  TC_HD static void encode(const Affine<Fq>& p, uint8_t* b) { g1_encode_uncompressed(p, b); }
};
template <>
struct PointIO<Fq2> {
  static constexpr int BYTES = 192;
  static constexpr int CBYTES = 96;
  TC_HD static bool decode(const uint8_t* b, Affine<Fq2>& p) { return g2_decode_uncompressed(b, p); }
  TC_HD static void encode(const Affine<Fq2>& p, uint8_t* b) { g2_encode_uncompressed(p, b); }
};

// [k] p.  G1: 2-dimensional GLV through phi (128 doublings); G2: 4-dimensional GLS through psi
// (64 doublings) -- tc_gls.h -- instead of the reference's 255-bit ladder.  Operands must lie in
// the prime-order subgroup, as every G1/G2 value the reference holds does.
TC_HD Jac<Fq> point_mul_scalar(const Affine<Fq>& p, const uint32_t* k) { return g1_mul_glv(p, k); }
TC_HD Jac<Fq> point_mul_scalar(const Jac<Fq>& p, const uint32_t* k) { return g1_mul_glv(p, k); }
TC_HD Jac<Fq2> point_mul_scalar(const Jac<Fq2>& p, const uint32_t* k) { return g2_mul_gls(p, k); }
TC_HD Jac<Fq2> point_mul_scalar(const Affine<Fq2>& p, const uint32_t* k) {
  return g2_mul_gls(p, k);
}

// order-r membership (tc_sqrt.h: Scott's endomorphism tests, one / two 64-bit ladders)
TC_HD bool point_in_subgroup(const Affine<Fq>& p) { return g1_in_subgroup(p); }
TC_HD bool point_in_subgroup(const Affine<Fq2>& p) { return g2_in_subgroup(p); }

// out = fr * pt           (CurveAffine::mul: sign_g2 src/lib.rs:373, decrypt_share :461)
// The scalar is shared by all lanes of a wave (one secret key share per wave), so the
// bit-serial double-and-add below has wave-uniform control flow.
template <class F>
TC_HD uint8_t job_point_mul(const uint8_t* fr_le32, const uint8_t* pt, uint8_t* out) {
  uint32_t k[8];
  Affine<F> p;
  bool ok = fr_from_le32(fr_le32, k);
  ok &= PointIO<F>::decode(pt, p);
  if (!ok) {
    PointIO<F>::encode(Affine<F>::infinity(), out);
    return TC_JOB_INVALID_ENCODING;
  }
  PointIO<F>::encode(jac_to_affine(point_mul_scalar(p, k)), out);
  return TC_JOB_OK;
}

// the G1 multiplication with its ladder table in the wave's arena slot (tc_gls.h g1_mul_glv_arena): the body of the kernel built
// for two waves per SIMD.  Every lane of the wave runs the ladder (a failed decode multiplies the identity's stand-in).
TC_HD uint8_t job_g1_mul_arena(const uint8_t* fr_le32, const uint8_t* pt, uint8_t* out) {
  uint32_t k[8];
  G1Affine p;
  bool ok = fr_from_le32(fr_le32, k);
  ok &= g1_decode_uncompressed(pt, p);
  if (!ok) {
    p = G1Affine{g1_generator().x, g1_generator().y, true};
    TC_UNROLL for (int i = 0; i < 8; i++) k[i] = (i == 0) ? 1u : 0u;
  }
  const G1Affine r = jac_to_affine(g1_mul_glv_arena(p, k));
  g1_encode_uncompressed(ok ? r : G1Affine::infinity(), out);
  return ok ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}

// out[s] = fr[s] * pt for n <= C scalars and ONE G2 point (the S signers of tc_g2_mul_batch over the
// same hash point): one decode, one table of psi-images (tc_gls.h g2_sac_table) and one inversion
// for the n results.  A bad point fails all n outputs, a bad scalar only its own.
constexpr int kMulShare = 4;
TC_HD void job_g2_mul_shared(const uint8_t* fr_le32, int n, const uint8_t* pt, uint8_t* out, uint8_t* status, bool leader) {
  G2Affine p;
  const bool pok = g2_decode_uncompressed(pt, p);
  if (!pok) p = G2Affine::infinity();
  G2Affine base[4];
  g2_gls_bases(p, base);
  G2SacTable tb;
  g2_sac_table_call(base, tb);
  G2Jac res[kMulShare];
  bool ok[kMulShare];
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    uint32_t k[8];
    ok[s] = fr_from_le32(fr_le32 + 32 * s, k) && pok;
    uint64_t d[4];
    const bool flip = gls_decompose_odd(k, d);
    G2Jac r = g2_sac_ladder_call(tb, d);
    r.y = Fq2::select(flip, -r.y, r.y);
    res[s] = G2Jac::select(ok[s], r, G2Jac::infinity());
  }
  G2Affine aff[kMulShare];
  jac_batch_to_affine<Fq2, kMulShare>(res, aff, n);
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    g2_encode_uncompressed(aff[s], out + (size_t)s * 192);
    if (status && leader) status[s] = ok[s] ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  }
}

constexpr int kGatherShare = 8;  // scalars per table in the gather kernel (68 signers: 9 tables per message instead of 17)
// out[s] = sk[idx[s]] * pt for n <= kGatherShare signer indices into a table of N secret key shares: the shares of
// ONE message by the signers of its subset (SecretKeyShare::sign_g2, src/lib.rs:442-444, for every selected
// signer), generated on the device so that a (t, N, batch) workload needs only the key set and the hash points.
// An index >= N fails its own output.
TC_HD void job_g2_mul_gather(const uint8_t* sk_table, size_t N, const uint64_t* idx, int n, const uint8_t* pt, uint8_t* out,
                             uint8_t* status, bool leader) {
  G2Affine p;
  const bool pok = g2_decode_uncompressed(pt, p);
  if (!pok) p = G2Affine::infinity();
  G2Affine base[4];
  g2_gls_bases(p, base);
  G2SacTable tb;
  g2_sac_table_call(base, tb);
  G2Jac res[kGatherShare];
  bool ok[kGatherShare];
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t i = idx[s];
    ok[s] = pok && i < N;
    if (i < N) ok[s] = fr_from_le32(sk_table + 32 * i, k) && ok[s];
    uint64_t d[4];
    const bool flip = gls_decompose_odd(k, d);
    G2Jac r = g2_sac_ladder_call(tb, d);
    r.y = Fq2::select(flip, -r.y, r.y);
    res[s] = G2Jac::select(ok[s], r, G2Jac::infinity());
  }
  G2Affine aff[kGatherShare];
  jac_batch_to_affine<Fq2, kGatherShare>(res, aff, n);
  TC_NOUNROLL for (int s = 0; s < n; s++) {
    g2_encode_uncompressed(aff[s], out + (size_t)s * 192);
    if (status && leader) status[s] = ok[s] ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  }
}

// k * c mod r for c = FR_COFACTOR_FIX (canonical words in and out)
TC_HD void fr_mul_cofactor_fix(const uint32_t* k, uint32_t* out) {
  (Fr::from_canonical(k) * Fr::from_canonical(FR_COFACTOR_FIX)).to_canonical(out);
}
// scalar of a signer folded with the hash's constant: out = fr * c mod r; a non-canonical scalar stays
// non-canonical (all ones) so that the multiplication kernel flags it as before
TC_HD void job_fr_scale_cofactor_fix(const uint8_t* fr_le32, uint8_t* out_le32) {
  uint32_t k[8], kc[8];
  const bool ok = fr_from_le32(fr_le32, k);
  fr_mul_cofactor_fix(k, kc);
  TC_UNROLL for (int i = 0; i < 8; i++) {
    const uint32_t w = ok ? kc[i] : 0xffffffffu;
    out_le32[4 * i] = (uint8_t)w;
    out_le32[4 * i + 1] = (uint8_t)(w >> 8);
    out_le32[4 * i + 2] = (uint8_t)(w >> 16);
    out_le32[4 * i + 3] = (uint8_t)(w >> 24);
  }
}
// G1 operand of a pairing against a hash point, folded with the hash's constant: out = [c] P.
// An encoding that does not decode is passed through unchanged, so the pairing kernel rejects it
// exactly as it would have rejected the original.
TC_HD void job_g1_scale_cofactor_fix(const uint8_t* in96, uint8_t* out96) {
  G1Affine p;
  if (!g1_decode_uncompressed(in96, p)) {
    for (int i = 0; i < 96; i++) out96[i] = in96[i];
    return;
  }
  g1_encode_uncompressed(jac_to_affine(g1_mul_glv(p, FR_COFACTOR_FIX)), out96);
}

// Lagrange coefficient (job, position i) -> 8 canonical LE words
TC_HD uint8_t job_lagrange(const uint64_t* idx, int t, int i, uint32_t* out_words) {
  Fr lam;
  if (!lagrange_coeff_at_zero(idx, t, i, lam)) {
    for (int w = 0; w < 8; w++) out_words[w] = 0;
    return TC_JOB_DUPLICATE_ENTRY;
  }
  lam.to_canonical(out_words);
  return TC_JOB_OK;
}

// out = sum_{i < n} scalar_i * point_i   (scalars: n x 8 canonical LE words, < r)
// Used by interpolate (below) and by Commitment::evaluate (src/poly.rs:497-508), which is the
// linear combination sum_k x^k * commit[k].
template <class F>
TC_HD uint8_t job_lincomb(int n, const uint8_t* points, const uint32_t* scalars, uint8_t* out) {
  constexpr int PB = PointIO<F>::BYTES;
  Jac<F> total = Jac<F>::infinity();
  bool ok = true;
  TC_NOUNROLL for (int base = 0; base < n; base += 4) {
    const int k = (n - base) < 4 ? (n - base) : 4;
    Affine<F> pts[4];
    uint32_t sc[4][8];
    TC_NOUNROLL for (int j = 0; j < 4; j++) {
      if (j < k) {
        ok &= PointIO<F>::decode(points + (size_t)(base + j) * PB, pts[j]);
        for (int w = 0; w < 8; w++) sc[j][w] = scalars[(size_t)(base + j) * 8 + w];
        ok &= limbs_lt_p<FrParams>(sc[j]);
      } else {
        pts[j] = Affine<F>::infinity();
        for (int w = 0; w < 8; w++) sc[j][w] = 0;
      }
    }
    Jac<F> part = lincomb_chunk4(pts, sc);
    total = jac_add(total, part);
  }
  if (!ok) {
    PointIO<F>::encode(Affine<F>::infinity(), out);
    return TC_JOB_INVALID_ENCODING;
  }
  PointIO<F>::encode(jac_to_affine(total), out);
  return TC_JOB_OK;
}

// Classes of the common denominator D of the small-index fast path (below).  The last step of that
// path is [1/D] Q, in general one full GLS multiplication; two classes are much cheaper, and the
// combine kernels group jobs by class so that a wave takes ONE of the branches (k_combine.hip):
//   D = 1            nothing to do
//   D = 2^a, a <= 16 |x| = 2^16 * 0xd20100000001 and x^2 - x^4 = 1 (mod r), so on G2
//                    [1/D] Q = [|x| / D] (psi^3(Q) - psi(Q)):  16 - a doublings, then the 48-bit ladder of
//                    |x| >> 16 (Hamming weight 6): 62 doublings + 5 additions instead of 64 + 64 + a table
constexpr int kCombineClassGeneric = 0, kCombineClassOne = 1, kCombineClassPow2 = 2, kCombineClasses = 3;
TC_HD int combine_denominator_class(uint64_t d_abs) {
  if (d_abs == 1) return kCombineClassOne;
  if ((d_abs & (d_abs - 1)) == 0 && d_abs <= (1ull << 16)) return kCombineClassPow2;
  return kCombineClassGeneric;
}
// [sign / d_abs] q
TC_HD G1Jac combine_divide(const G1Jac& q, uint64_t d_abs, bool d_neg) {
  uint32_t dinv[8];
  fr_inverse_of_small(d_abs, d_neg, dinv);
  return point_mul_scalar(q, dinv);
}
// the same with the ladder's table in the lane's arena entries (the two-waves-per-SIMD build of k_combine_fast<Fq>)
TC_HD G1Jac combine_divide_arena(const G1Jac& q, uint64_t d_abs, bool d_neg) {
  // a wave whose jobs all have D = 1 (the combine kernels group jobs by class from kG1GroupMinJobs on) has nothing to divide by;
  // G1 has no cheap form for D = 2^a (the G2 shortcut rests on psi = [x])
  if (!wave_any(d_abs != 1)) {
    G1Jac r = q;
    r.y = Fq::select(d_neg, -r.y, r.y);
    return r;
  }
  uint32_t dinv[8];
  fr_inverse_of_small(d_abs, d_neg, dinv);
  G1Jac r = g1_mul_glv_arena(G1Affine{q.x, q.y, q.is_inf()}, dinv);
  r.z = coord_norm(r.z * q.z);
  return r;
}

TC_HD G2Jac combine_divide(const G2Jac& q, uint64_t d_abs, bool d_neg) {
  const int cls = combine_denominator_class(d_abs);
  G2Jac r = q;
  if (wave_any(cls == kCombineClassGeneric)) {
    uint32_t dinv[8];
    fr_inverse_of_small(d_abs, false, dinv);
    r = G2Jac::select(cls == kCombineClassGeneric, point_mul_scalar(q, dinv), r);
  }
  if (wave_any(cls == kCombineClassPow2)) {
    const G2Jac p1 = g2_psi(q);
    const G2Jac p3 = g2_psi(g2_psi(p1));
    G2Jac base = jac_add(p3, jac_neg(p1));  // psi^3(Q) - psi(Q)
    const int pre = 16 - (int)__builtin_ctzll(d_abs | (1ull << 16));  // 16 - a doublings first
    TC_NOUNROLL for (int i = 0; i < 15; i++) base = G2Jac::select(i < pre, jac_dbl(base), base);
    const uint64_t s = BLS_X_ABS >> 16;
    const G2Affine ba{base.x, base.y, base.is_inf()};  // affine on the curve scaled by base.z (tc_gls.h)
    G2Jac acc = jac_ladder_uniform(ba, s, 47);  // bit 47 is the leading one
    acc.z = coord_norm(acc.z * base.z);
    r = G2Jac::select(cls == kCombineClassPow2, acc, r);
  }
  r.y = Fq2::select(d_neg, -r.y, r.y);
  return r;
}
TC_HD G2Jac combine_divide_arena(const G2Jac& q, uint64_t d_abs, bool d_neg) { return combine_divide(q, d_abs, d_neg); }  // (G2: always in the arena)

// Where a job body finds the encodings of its operands and where its result goes.  DirectIO addresses global
// memory as it is (one job per lane: 96/192-byte records `stride` apart); the kernels pass tc_stage.h's WaveRowIO
// instead, which moves whole-wave rows through LDS with coalesced 8-byte loads and stores.  Every lane of a wave
// -- live or not -- must make the same sequence of operand() / result() / commit(wrote) calls.
struct DirectIO {
  const uint8_t* in;
  size_t stride;
  uint8_t* out;
  TC_HD const uint8_t* operand(int k) const { return in + (size_t)k * stride; }
  TC_HD uint8_t* result() const { return out; }
  TC_HD void commit(bool) const {}
};

// combination through the small-index fast path (tc_threshold.h); false => not applicable (or !live)
template <class F, int K, class IO, bool ARENA = false>
TC_HD bool job_combine_small_io(const uint64_t* idx, bool live, IO& io, uint8_t* status) {
  uint64_t c_abs[K], d_abs = 1;
  bool c_neg[K], d_neg = false;
  TC_UNROLL for (int k = 0; k < K; k++) {
    c_abs[k] = 0;
    c_neg[k] = false;
  }
  TC_MARK(0);
  TC_MARK_WALL(8);
  const bool applies = live && lagrange_small_coeffs<K>(idx, c_abs, c_neg, &d_abs, &d_neg);
  Affine<F> pts[K];
  bool ok = true;
  TC_NOUNROLL for (int k = 0; k < K; k++) {
    const uint8_t* enc = io.operand(k);  // wave-uniform: stages share k of every job of the wave
    if (applies) {
      ok &= PointIO<F>::decode(enc, pts[k]);
      if (c_neg[k]) pts[k].y = (-pts[k].y).norm();
    }
  }
  uint8_t* dst = io.result();
  TC_MARK(1);
  if (applies) {
    if (!ok) {
      PointIO<F>::encode(Affine<F>::infinity(), dst);
      *status = TC_JOB_INVALID_ENCODING;
    } else {
      // (G2: out of line -- its own register allocation: the kernel's private segment drops from 8.0 to 5.6 KB at the
      // same speed; the one-lane G1 kernel measured 3 % slower that way)
      Jac<F> a = (JobLanes<F>::N > 1) ? straus_small_call<F, K>(pts, c_abs) : straus_small<F, K, ARENA>(pts, c_abs);
      TC_MARK(2);
      const Jac<F> q = ARENA ? combine_divide_arena(a, d_abs, d_neg) : combine_divide(a, d_abs, d_neg);
      TC_MARK(4);
      const Affine<F> qa = jac_to_affine(q);
      TC_MARK(5);
      PointIO<F>::encode(qa, dst);
      *status = TC_JOB_OK;
    }
  }
  io.commit(applies);
  TC_MARK(6);
  TC_MARK_WALL(9);
  return applies;
}
template <class F, int K>
TC_HD bool job_combine_small(const uint64_t* idx, const uint8_t* shares, uint8_t* out, uint8_t* status) {
  DirectIO io{shares, (size_t)PointIO<F>::BYTES, out};
  return job_combine_small_io<F, K>(idx, true, io, status);
}

// class of the job's denominator (kCombineClass*); generic when the fast path does not apply
TC_HD int combine_job_class(const uint64_t* idx, int t) {
  uint64_t c_abs[4], d_abs = 0;
  bool c_neg[4], d_neg;
  bool applies = false;
  if (t == 1) applies = lagrange_small_coeffs<2>(idx, c_abs, c_neg, &d_abs, &d_neg);
  if (t == 2) applies = lagrange_small_coeffs<3>(idx, c_abs, c_neg, &d_abs, &d_neg);
  if (t == 3) applies = lagrange_small_coeffs<4>(idx, c_abs, c_neg, &d_abs, &d_neg);
  return applies ? combine_denominator_class(d_abs) : kCombineClassGeneric;
}

// true when job_combine_small will handle the job (so the Lagrange kernel can skip it)
TC_HD bool combine_small_applies(const uint64_t* idx, int t) {
  uint64_t c_abs[4], d_abs;
  bool c_neg[4], d_neg;
  if (t == 1) return lagrange_small_coeffs<2>(idx, c_abs, c_neg, &d_abs, &d_neg);
  if (t == 2) return lagrange_small_coeffs<3>(idx, c_abs, c_neg, &d_abs, &d_neg);
  if (t == 3) return lagrange_small_coeffs<4>(idx, c_abs, c_neg, &d_abs, &d_neg);
  return false;
}

// out = sum_{i <= t} lambda_i * share_i over the FIRST t+1 samples of the job
// (interpolate, src/lib.rs:719-767).  lam: (t+1) x 8 canonical words from job_lagrange.
// t == 0: the first sample is returned unchanged (src/lib.rs:735-737) -- decoded and encoded again, so a bad encoding fails
template <class F>
TC_HD uint8_t job_first_sample(const uint8_t* shares, uint8_t* out) {
  Affine<F> p;
  if (!PointIO<F>::decode(shares, p)) {
    PointIO<F>::encode(Affine<F>::infinity(), out);
    return TC_JOB_INVALID_ENCODING;
  }
  PointIO<F>::encode(p, out);
  return TC_JOB_OK;
}
// (G1; the G2 general jobs with t >= 1 run through the two-stage kernels of k_msm.hip)
template <class F>
TC_HD uint8_t job_combine(int t, const uint8_t* shares, const uint32_t* lam, uint8_t* out) {
  if (t == 0) return job_first_sample<F>(shares, out);
  return job_lincomb<F>(t + 1, shares, lam, out);
}

// ok = e(a, b) == e(c, d)     (src/lib.rs:109, :185, :511)
TC_HD uint8_t job_pairing_check(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d) {
  G1Affine pa, pc;
  G2Affine qb, qd;
  bool ok = g1_decode_uncompressed(a, pa);
  ok &= g2_decode_uncompressed(b, qb);
  ok &= g1_decode_uncompressed(c, pc);
  ok &= g2_decode_uncompressed(d, qd);
  if (!ok) return 0;
  return pairing_check(pa, qb, pc, qd) ? 1 : 0;
}

// the same with the four operands fetched through IO objects (tc_stage.h: whole-wave rows through LDS); every
// lane of the wave calls the four operand() functions, live or not
template <class IOA, class IOB, class IOC, class IOD>
TC_HD uint8_t job_pairing_check_io(bool live, IOA& a, IOB& b, IOC& c, IOD& d) {
  G1Affine pa = G1Affine::infinity(), pc = G1Affine::infinity();
  G2Affine qb = G2Affine::infinity(), qd = G2Affine::infinity();
  bool ok = live;
  const uint8_t* e = a.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pa);
  e = b.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qb);
  e = c.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pc);
  e = d.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qd);
  if (!ok) return 0;
  return pairing_check(pa, qb, pc, qd) ? 1 : 0;
}

// The check in two halves (k_pairing.hip runs them as two kernels): the product Miller loop of the two pairs ...
template <class IOA, class IOB, class IOC, class IOD>
TC_HD bool job_miller_io(bool live, IOA& a, IOB& b, IOC& c, IOD& d, Fq12& f) {
  G1Affine pa = G1Affine::infinity(), pc = G1Affine::infinity();
  G2Affine qb = G2Affine::infinity(), qd = G2Affine::infinity();
  bool ok = live;
  const uint8_t* e = a.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pa);
  e = b.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qb);
  e = c.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pc);
  e = d.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qd);
  if (!ok) {  // the value of an empty product; the caller reports the job as failed
    pa = pc = G1Affine::infinity();
    qb = qd = G2Affine::infinity();
  }
  G1Affine ps[2] = {pa, G1Affine{pc.x, (-pc.y), pc.inf}};
  G2Affine qs[2] = {qb, qd};
  f = miller_loop<2>(ps, qs);
  return ok;
}
// The Miller loop itself in two stages that meet in memory (tc_pairing.h: prepared line products): stage P ...
template <class IOA, class IOB, class IOC, class IOD>
TC_HD bool job_miller_lines_io(bool live, IOA& a, IOB& b, IOC& c, IOD& d, const Fq2Rows& rows) {
  G1Affine pa = G1Affine::infinity(), pc = G1Affine::infinity();
  G2Affine qb = G2Affine::infinity(), qd = G2Affine::infinity();
  bool ok = live;
  const uint8_t* e = a.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pa);
  e = b.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qb);
  e = c.operand(0);
  if (live) ok &= g1_decode_uncompressed(e, pc);
  e = d.operand(0);
  if (live) ok &= g2_decode_uncompressed(e, qd);
  if (!ok) {  // the value of an empty product; the caller reports the job as failed
    pa = pc = G1Affine::infinity();
    qb = qd = G2Affine::infinity();
  }
  const G1Affine ps[2] = {pa, G1Affine{pc.x, (-pc.y), pc.inf}};
  const G2Affine qs[2] = {qb, qd};
  const bool skip[2] = {ps[0].inf || qs[0].inf, ps[1].inf || qs[1].inf};
  miller_prepare_lines(ps, qs, skip, rows);
  return ok;
}
// ... stage M: miller_accumulate(rows) ...
// ... and the final exponentiation with the comparison
TC_HD uint8_t job_final_exp_is_one(const Fq12& f) { return final_exponentiation(f) == Fq12::one() ? 1 : 0; }

// hash_g2(msg)      (src/lib.rs:691-694)
// fix = false: the point Q' with hash_g2(msg) = [FR_COFACTOR_FIX] Q' (tc_gls.h g2_clear_cofactor), for the
// composed entry points that fold the constant into a scalar or into the G1 operand of a pairing.
TC_HD G2Jac hash_g2_point(const uint8_t* msg, size_t len, bool fix = true) {
  uint32_t seed[8];
  sha3_256_words(msg, len, seed);
  return g2_random_from_seed(seed, fix);
}
// the same over a per-lane byte buffer (hash_g1_g2): SHA3 out of line (tc_hash.h sha3_256_words_call)
TC_HD G2Jac hash_g2_point_of_buffer(const uint8_t* buf, size_t len, bool fix) {
  uint32_t seed[8];
  sha3_256_words_call(buf, len, seed);
  return g2_random_from_seed(seed, fix);
}
TC_HD void job_hash_g2(const uint8_t* msg, size_t len, uint8_t* out_g2, bool fix = true) {
  g2_encode_uncompressed(jac_to_affine(hash_g2_point(msg, len, fix)), out_g2);
}

// hash_g1_g2(g1, msg)   (src/lib.rs:697-707): the ChaCha20 seed -- sha3_256 of (msg, or its digest beyond 64 bytes) || compress(g1)
TC_HD void hash_g1_g2_seed(const G1Affine& p, const uint8_t* msg, size_t len, uint32_t* seed) {
  uint8_t buf[64 + 48];
  size_t n;
  if (len > 64) {
    uint32_t d[8];
    sha3_256_words_call(msg, len, d);
    for (int i = 0; i < 8; i++) {
      buf[4 * i] = (uint8_t)d[i];
      buf[4 * i + 1] = (uint8_t)(d[i] >> 8);
      buf[4 * i + 2] = (uint8_t)(d[i] >> 16);
      buf[4 * i + 3] = (uint8_t)(d[i] >> 24);
    }
    n = 32;
  } else {
    for (size_t i = 0; i < 64; i++)  // fixed trip count (tc_common.h wave_any)
      if (i < len) buf[i] = msg[i];
    n = len;
  }
  g1_encode_compressed(p, buf + n);
  sha3_256_words_call(buf, n + 48, seed);
}
TC_HD G2Jac hash_g1_g2_point(const G1Affine& p, const uint8_t* msg, size_t len, bool fix = true) {
  uint32_t seed[8];
  hash_g1_g2_seed(p, msg, len, seed);
  return g2_random_from_seed(seed, fix);
}
TC_HD uint8_t job_hash_g1_g2(const uint8_t* g1, const uint8_t* msg, size_t len, uint8_t* out_g2, bool fix = true) {
  G1Affine p;
  if (!g1_decode_uncompressed(g1, p)) {
    // public form: infinity + status.  Internal (fix = false) form, consumed by a pairing check: an
    // encoding that does not decode, so the check fails instead of skipping the pair
    if (fix) g2_encode_uncompressed(G2Affine::infinity(), out_g2);
    else for (int i = 0; i < 192; i++) out_g2[i] = 0xff;
    return TC_JOB_INVALID_ENCODING;
  }
  g2_encode_uncompressed(jac_to_affine(hash_g1_g2_point(p, msg, len, fix)), out_g2);
  return TC_JOB_OK;
}

// ---- two messages per lane pair (tc_duo.h, tc_hash.h g2_random_from_seed_x2): batches from 65 536 messages on ----------------
// SHA3, the streams, the sampling and the Fq exponentiations of message A on lane 0 and of message B on lane 1; out_b may be null
// (an odd batch: the last pair's second slot repeats the first)
TC_HD void hash_g2_x2_finish(const Duo<Seed8>& seed, bool fix, uint8_t* out_a, uint8_t* out_b) {
  G2Jac ra, rb;
  g2_random_from_seed_x2(seed, fix, ra, rb);
  G2Affine pa, pb;
  jac_to_affine_x2(ra, rb, pa, pb);
  g2_encode_uncompressed(pa, out_a);
  if (out_b) g2_encode_uncompressed(pb, out_b);
}
TC_HD void job_hash_g2_x2(const uint8_t* msg_a, size_t len_a, const uint8_t* msg_b, size_t len_b, uint8_t* out_a, uint8_t* out_b,
                          bool fix = true) {
  const Duo<const uint8_t*> msg = duo_pick(msg_a, msg_b);
  const Duo<size_t> len = duo_pick(len_a, len_b);
  Duo<Seed8> seed;
  duo_each([&](int s) { sha3_256_words_call(msg.at(s), len.at(s), seed.at(s).w); });
  hash_g2_x2_finish(seed, fix, out_a, out_b);
}
TC_HD void job_hash_g1_g2_x2(const uint8_t* g1_a, const uint8_t* msg_a, size_t len_a, const uint8_t* g1_b, const uint8_t* msg_b,
                             size_t len_b, uint8_t* out_a, uint8_t* out_b, bool fix, uint8_t& st_a, uint8_t& st_b) {
  const Duo<const uint8_t*> g1 = duo_pick(g1_a, g1_b), msg = duo_pick(msg_a, msg_b);
  const Duo<size_t> len = duo_pick(len_a, len_b);
  Duo<Seed8> seed;
  Duo<bool> decoded;
  duo_each([&](int s) {
    G1Affine p;
    decoded.at(s) = g1_decode_uncompressed(g1.at(s), p);
    if (!decoded.at(s)) p = G1Affine::infinity();  // stand-in: the slot's result is replaced below
    hash_g1_g2_seed(p, msg.at(s), len.at(s), seed.at(s).w);
  });
  hash_g2_x2_finish(seed, fix, out_a, out_b);
  bool oka, okb;
  duo_to(decoded, oka, okb);
  // an undecodable G1 operand: as job_hash_g1_g2 -- infinity + status (public form) / an encoding that does not decode (internal form)
  if (!oka) {
    if (fix) g2_encode_uncompressed(G2Affine::infinity(), out_a);
    else for (int i = 0; i < 192; i++) out_a[i] = 0xff;
  }
  if (!okb && out_b) {
    if (fix) g2_encode_uncompressed(G2Affine::infinity(), out_b);
    else for (int i = 0; i < 192; i++) out_b[i] = 0xff;
  }
  st_a = oka ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  st_b = okb ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}

// out[i] = data[i] ^ (u8) ChaCha20(sha3_256(compress(g1))).next_u32()   (src/lib.rs:710-715)
TC_HD uint8_t job_xor_with_hash(const uint8_t* g1, const uint8_t* data, size_t len, uint8_t* out) {
  G1Affine p;
  if (!g1_decode_uncompressed(g1, p)) return TC_JOB_INVALID_ENCODING;
  uint8_t comp[48];
  g1_encode_compressed(p, comp);
  uint32_t seed[8];
  sha3_256_words(comp, 48, seed);
  ChaChaRng rng;
  rng.init(seed);
  KeystreamBytes ks;
  size_t i = 0;
  TC_NOUNROLL while (wave_any(i < len)) {
    if (i < len) {
      out[i] = data[i] ^ ks.next(rng);
      i++;
    }
  }
  return TC_JOB_OK;
}

// uncompressed -> compressed (to_bytes, src/lib.rs:149-153, 255-259)
template <class F>
TC_HD uint8_t job_compress(const uint8_t* in, uint8_t* out);
template <>
TC_HD uint8_t job_compress<Fq>(const uint8_t* in, uint8_t* out) {
  G1Affine p;
  bool ok = g1_decode_uncompressed(in, p);
  if (!ok) p = G1Affine::infinity();
  g1_encode_compressed(p, out);
  return ok ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}
template <>
TC_HD uint8_t job_compress<Fq2>(const uint8_t* in, uint8_t* out) {
  G2Affine p;
  bool ok = g2_decode_uncompressed(in, p);
  if (!ok) p = G2Affine::infinity();
  g2_encode_compressed(p, out);
  return ok ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}

// (u, v, w) = pk.encrypt_with_rng(rng, msg) with the rng's Fr draw r supplied by the caller
// (src/lib.rs:128-137):  u = r g1;  v = msg ^ keystream(r pk);  w = r hash_g1_g2(u, v)
TC_HD uint8_t job_encrypt(const uint8_t* pk96, const uint8_t* r_le32, const uint8_t* msg, size_t len, uint8_t* out_u,
                          uint8_t* out_v, uint8_t* out_w) {
  uint32_t k[8];
  G1Affine pk;
  bool ok = fr_from_le32(r_le32, k);
  ok &= g1_decode_uncompressed(pk96, pk);
  if (!ok) {
    g1_encode_uncompressed(G1Affine::infinity(), out_u);
    g2_encode_uncompressed(G2Affine::infinity(), out_w);
    size_t i = 0;
    TC_NOUNROLL while (wave_any(i < len)) {
      if (i < len) out_v[i++] = 0;
    }
    return TC_JOB_INVALID_ENCODING;
  }
  const G1Affine u = jac_to_affine(g1_mul_glv(g1_generator(), k));
  g1_encode_uncompressed(u, out_u);
  uint8_t g[96];
  g1_encode_uncompressed(jac_to_affine(g1_mul_glv(pk, k)), g);
  job_xor_with_hash(g, msg, len, out_v);
  // (G2 values never round-trip through a per-lane byte buffer: in the lane-pair build each
  // lane only holds half of an encoding)
  // w = [r] [c] Q' = [r c mod r] Q': the cofactor-fix multiplication of the hash folds into this one
  uint32_t kc[8];
  fr_mul_cofactor_fix(k, kc);
  g2_encode_uncompressed(jac_to_affine(g2_mul_gls(hash_g1_g2_point(u, out_v, len, false), kc)), out_w);
  return TC_JOB_OK;
}

// Commitment::evaluate(i + 1) (src/poly.rs:497-508) = PublicKeySet::public_key_share(i)
// (src/lib.rs:570-573): Horner in G1, result = result * x + c_k from the top coefficient down,
// x = idx + 1 as a 64-bit ladder (x = 2^64 is handled by the general scalar path of the caller).
TC_HD uint8_t job_commitment_evaluate(const uint8_t* commit, int t, uint64_t idx, uint8_t* out96) {
  const uint64_t x = idx + 1;
  G1Affine c;
  if (x == 0 || !g1_decode_uncompressed(commit + (size_t)t * 96, c)) {
    g1_encode_uncompressed(G1Affine::infinity(), out96);
    return TC_JOB_INVALID_ENCODING;
  }
  G1Jac res = G1Jac::from_affine(c);
  bool ok = true;
  TC_NOUNROLL for (int kk = t - 1; kk >= 0; kk--) {
    // res = res * x
    G1Jac acc = G1Jac::infinity();
    bool started = false;
    TC_NOUNROLL for (int bit = 63; bit >= 0; bit--) {
      if (started) acc = jac_dbl(acc);
      if ((x >> bit) & 1ull) {
        acc = started ? jac_add(acc, res) : res;
        started = true;
      }
    }
    ok &= g1_decode_uncompressed(commit + (size_t)kk * 96, c);
    res = jac_add_mixed(acc, c);
  }
  if (!ok) {
    g1_encode_uncompressed(G1Affine::infinity(), out96);
    return TC_JOB_INVALID_ENCODING;
  }
  g1_encode_uncompressed(jac_to_affine(res), out96);
  return TC_JOB_OK;
}

// compressed -> uncompressed with the reference's CHECKED decode (from_bytes,
// src/lib.rs:140-146, 246-252): on the curve and in the order-r subgroup, else Invalid
template <class F>
TC_HD uint8_t job_decompress(const uint8_t* in, uint8_t* out);
template <>
TC_HD uint8_t job_decompress<Fq>(const uint8_t* in, uint8_t* out) {
  G1Affine p;
  bool ok = g1_decode_compressed(in, p);
  if (!ok) p = G1Affine::infinity();
  g1_encode_uncompressed(p, out);
  return ok ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}
template <>
TC_HD uint8_t job_decompress<Fq2>(const uint8_t* in, uint8_t* out) {
  G2Affine p;
  bool ok = g2_decode_compressed(in, p);
  if (!ok) p = G2Affine::infinity();
  g2_encode_uncompressed(p, out);
  return ok ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}

// the same for TWO compressed G2 points on one lane pair (tc_duo.h: the Fq exponentiations of the two square roots run on one
// lane each); out_b may be null (an odd point count: the pair's second slot repeats the first)
TC_HD void job_decompress_g2_x2(const uint8_t* in_a, const uint8_t* in_b, uint8_t* out_a, uint8_t* out_b, uint8_t& st_a, uint8_t& st_b) {
  G2Affine pa, pb;
  bool oka, okb;
  g2_decode_compressed_x2(in_a, in_b, pa, pb, oka, okb);
  g2_encode_uncompressed(pa, out_a);
  if (out_b) g2_encode_uncompressed(pb, out_b);
  st_a = oka ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
  st_b = okb ? TC_JOB_OK : TC_JOB_INVALID_ENCODING;
}

}  // namespace tc
