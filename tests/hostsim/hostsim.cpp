// TEST HARNESS ONLY.  Compiles the device headers of threshold_crypto_amd/csrc with g++ so
// the per-lane job bodies can be differential-tested against the oracle in a container
// without a GPU.  Never linked into, loaded by, or used as a fallback for libtc_amd.so.
#include "tc_jobs.h"
#include "tc_dkg.h"
#include "tc_msm.h"
#include "tc_quad.h"
#include <thread>
#include "tc_comb.h"
#include <vector>
#include <string.h>
using namespace tc;

extern "C" {
#if defined(TC_COUNT_OPS)
void hs_op_counts(uint64_t* mul, uint64_t* sqr, int reset) {
  *mul = g_tc_mul_count;
  *sqr = g_tc_sqr_count;
  if (reset) g_tc_mul_count = g_tc_sqr_count = 0;
}
// out[0..4] = mul2 (Fq2 products' coefficient formulas), split mul, split sqr, all mul, all sqr
void hs_op_counts5(uint64_t* out, int reset) {
  out[0] = g_tc_mul2_count;
  out[1] = g_tc_split_mul_count;
  out[2] = g_tc_split_sqr_count;
  out[3] = g_tc_mul_count;
  out[4] = g_tc_sqr_count;
  if (reset) g_tc_mul2_count = g_tc_split_mul_count = g_tc_split_sqr_count = g_tc_mul_count = g_tc_sqr_count = 0;
}
#endif
int hs_fq_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fq x, y;
  if (!fq_from_be48(a, false, x) || !fq_from_be48(b, false, y)) return -1;
  fq_to_be48(x * y, out);
  return 0;
}
int hs_fq_inv(const uint8_t* a, uint8_t* out) {
  Fq x;
  if (!fq_from_be48(a, false, x)) return -1;
  fq_to_be48(x.inv(), out);
  return 0;
}
int hs_fq_inv_both(const uint8_t* a, uint8_t* out_gcd, uint8_t* out_fermat) {
  Fq x;
  if (!fq_from_be48(a, false, x)) return -1;
  fq_to_be48(x.inv(), out_gcd);
  fq_to_be48(fq_inv_fermat(x), out_fermat);
  return 0;
}
int hs_fq_addsub(const uint8_t* a, const uint8_t* b, uint8_t* sum, uint8_t* diff, uint8_t* neg) {
  Fq x, y;
  if (!fq_from_be48(a, false, x) || !fq_from_be48(b, false, y)) return -1;
  fq_to_be48(x + y, sum);
  fq_to_be48(x - y, diff);
  fq_to_be48(-x, neg);
  return 0;
}
// zero test on k*p + delta built as a LAZY value: x - y with x = a + (k+1) p-ish pieces.  Returns
// bit0 = maybe_zero(), bit1 = is_zero_full(), bit2 = is_zero(), bit3 = maybe_zero56() (the two-limb hint of the
// branch-free ladder additions), for value  a - b + k*p  (a, b canonical inputs).
int hs_fq_zero_probe(const uint8_t* a, const uint8_t* b, int k) {
  Fq x, y;
  fq_from_be48(a, false, x);
  fq_from_be48(b, false, y);
  Fq v = x - y;
  const Fq p = Fq::from_limbs(FQL_P);
  for (int i = 0; i < (k < 0 ? -k : k); i++) {
    v = (k < 0) ? v - p : v + p;
    if ((i & 3) == 3) v = v.norm();  // keep the lazy limbs inside int32
  }
  return (v.maybe_zero() ? 1 : 0) | (v.is_zero_full() ? 2 : 0) | (v.is_zero() ? 4 : 0) | (v.maybe_zero56() ? 8 : 0);
}
int hs_fq_legendre(const uint8_t* a) {
  Fq x;
  fq_from_be48(a, false, x);
  return fq_legendre(x);
}
int hs_fq2_sqrt(const uint8_t* a /*c0||c1 be48*/, uint8_t* out) {
  Fq2 x, y;
  fq_from_be48(a, false, x.c0);
  fq_from_be48(a + 48, false, x.c1);
  if (!fq2_sqrt(x, y)) return 0;
  fq_to_be48(y.c0, out);
  fq_to_be48(y.c1, out + 48);
  return 1;
}
int hs_g1_mul(const uint8_t* fr, const uint8_t* pt, uint8_t* out) { return job_point_mul<Fq>(fr, pt, out); }
int hs_g2_mul(const uint8_t* fr, const uint8_t* pt, uint8_t* out) { return job_point_mul<Fq2>(fr, pt, out); }
// the body of k_g1_mul_arena: the ladder's table in the lane's arena entries (tc_gls.h g1_mul_glv_arena)
int hs_g1_mul_arena(const uint8_t* fr, const uint8_t* pt, uint8_t* out) { return job_g1_mul_arena(fr, pt, out); }
int hs_combine_job_class(const uint64_t* idx, int t) { return combine_job_class(idx, t); }
// base-|x| digits of a scalar (8 LE u32 words)
void hs_gls_decompose(const uint32_t* k, uint64_t* d) { gls_decompose(k, d); }
void hs_g2_mul_shared(const uint8_t* fr, int n, const uint8_t* pt, uint8_t* out, uint8_t* status) {
  job_g2_mul_shared(fr, n, pt, out, status, true);
}
// out[s] = sk[idx[s]] * pt, n <= kGatherShare signers over one table (the share-generation kernel's job body)
void hs_g2_mul_gather(const uint8_t* sk, size_t N, const uint64_t* idx, int n, const uint8_t* pt, uint8_t* out, uint8_t* status) {
  job_g2_mul_gather(sk, N, idx, n, pt, out, status, true);
}
int hs_gather_share() { return kGatherShare; }
// the same shares through the per-message comb (tc_comb.h): stage T once, stage S in chunks of kCombShare signers
void hs_comb_sign(const uint8_t* sk, size_t N, const uint64_t* idx, int n, const uint8_t* pt, uint8_t* out, uint8_t* status) {
  std::vector<int32_t> tbl(kCombTableWords);
  const bool ok = job_comb_tables(pt, (tbl_word*)tbl.data());
  for (int s0 = 0; s0 < n; s0 += kCombShare) {
    const int cnt = (n - s0 < kCombShare) ? n - s0 : kCombShare;
    job_comb_sign(sk, N, idx + s0, cnt, (const tbl_word*)tbl.data(), ok, out + (size_t)s0 * 192, status + s0, true);
  }
}
int hs_lagrange(const uint64_t* idx, int t, int i, uint8_t* out32le) {
  uint32_t w[8];
  int st = job_lagrange(idx, t, i, w);
  memcpy(out32le, w, 32);
  return st;
}
// lambda_i from Fr abscissae (`T: IntoFr` beyond u64: tc_threshold.h lagrange_coeff_at_zero_fr, the body of k_lagrange_fr)
int hs_lagrange_fr(const uint8_t* xs32le, int t, int i, uint8_t* out32le) {
  std::vector<uint32_t> xs((size_t)(t + 1) * 8);
  memcpy(xs.data(), xs32le, xs.size() * 4);
  Fr l;
  if (!lagrange_coeff_at_zero_fr(xs.data(), t, i, l)) return TC_JOB_DUPLICATE_ENTRY;
  uint32_t w[8];
  l.to_canonical(w);
  memcpy(out32le, w, 32);
  return 0;
}
void hs_fr_inverse_of_small(uint64_t d_abs, int d_neg, uint8_t* out32le) {
  uint32_t w[8];
  fr_inverse_of_small(d_abs, d_neg != 0, w);
  memcpy(out32le, w, 32);
}
static int hs_combine_g2_large(int t, const uint64_t* idx, const uint8_t* shares, uint32_t* lam, uint8_t* out);
int hs_msm_g2(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out192);
static int g_force_general = 0;
void hs_force_general_combine(int on) { g_force_general = on; }
static int combine(int g2, int t, const uint64_t* idx, const uint8_t* shares, uint8_t* out) {
  uint32_t lam[8 * 256];
  if (t + 1 > 256) return -1;
  if (!g_force_general) {  // same dispatch as k_combine
    uint8_t st = 0;
    bool done = false;
    if (g2) {
      if (t == 1) done = job_combine_small<Fq2, 2>(idx, shares, out, &st);
      if (t == 2) done = job_combine_small<Fq2, 3>(idx, shares, out, &st);
      if (t == 3) done = job_combine_small<Fq2, 4>(idx, shares, out, &st);
    } else {
      if (t == 1) done = job_combine_small<Fq, 2>(idx, shares, out, &st);
      if (t == 2) done = job_combine_small<Fq, 3>(idx, shares, out, &st);
      if (t == 3) done = job_combine_small<Fq, 4>(idx, shares, out, &st);
    }
    if (done) return st;
  }
  if (g2 && (size_t)(t + 1) >= 8 && !g_force_general) {  // the large-threshold dispatch of tc_api.hip: k_lagrange_all + k_msm_*
    return hs_combine_g2_large(t, idx, shares, lam, out);
  }
  for (int i = 0; i <= t; i++) {
    int st = job_lagrange(idx, t, i, lam + 8 * i);
    if (st) return st;
  }
  // G2, t >= 1: the general jobs go through the two-stage kernels (tc_api.hip combine()); t = 0: the first sample
  if (g2) return t >= 1 ? hs_msm_g2((size_t)t + 1, shares, lam, out) : job_first_sample<Fq2>(shares, out);
  return job_combine<Fq>(t, shares, lam, out);
}
int hs_combine_g2(int t, const uint64_t* idx, const uint8_t* shares, uint8_t* out) { return combine(1, t, idx, shares, out); }
int hs_combine_g1(int t, const uint64_t* idx, const uint8_t* shares, uint8_t* out) { return combine(0, t, idx, shares, out); }
int hs_pairing_check(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d) {
  return job_pairing_check(a, b, c, d);
}
// the prepared form (tc_pairing.h): stage P writes the 68 x 5 line-product coefficients, stage M reads them back, then the
// final exponentiation -- the three kernels of k_pairing.hip one after the other
int hs_pairing_check_prepared(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d) {
  std::vector<Fq2> mem(kMillerRowSlots);
  const Fq2Rows rows = Fq2Rows::at(mem.data());
  DirectIO ia{a, 0, nullptr}, ib{b, 0, nullptr}, ic{c, 0, nullptr}, id{d, 0, nullptr};
  if (!job_miller_lines_io(true, ia, ib, ic, id, rows)) return 0;
  return job_final_exp_is_one(miller_accumulate(rows));
}
// GT value of FE(ML(a,b)) : 12 x 48 B big-endian in tower order c0.c0.c0, c0.c0.c1, c0.c1.c0 ...
// the four-lanes-per-check form (tc_quad.h): the two pairs of the quad as two host threads
int hs_pairing_check_quad(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d) {
  QuadSim sim;
  int res[2] = {-1, -1};
  auto run = [&](int hi) {
    tl_quad_sim = &sim;
    tl_quad_hi = hi;
    DirectIO g1{hi ? c : a, 0, nullptr}, g2{hi ? d : b, 0, nullptr};
    res[hi] = job_pairing_check_quad_io(true, g1, g2);
  };
  std::thread tb(run, 1);
  run(0);
  tb.join();
  return res[0] == res[1] ? res[0] : -1;
}
int hs_pairing_gt(const uint8_t* a, const uint8_t* b, uint8_t* out576) {
  G1Affine p;
  G2Affine q;
  if (!g1_decode_uncompressed(a, p) || !g2_decode_uncompressed(b, q)) return -1;
  G1Affine ps[1] = {p};
  G2Affine qs[1] = {q};
  Fq12 f = final_exponentiation(miller_loop<1>(ps, qs));
  const Fq* e = reinterpret_cast<const Fq*>(&f);
  for (int i = 0; i < 12; i++) fq_to_be48(e[i], out576 + 48 * i);
  return 0;
}
int hs_cyclo_check(const uint8_t* a, const uint8_t* b) {
  // cyclotomic_sqr == generic sqr on an element of the cyclotomic subgroup
  G1Affine p;
  G2Affine q;
  if (!g1_decode_uncompressed(a, p) || !g2_decode_uncompressed(b, q)) return -1;
  G1Affine ps[1] = {p};
  G2Affine qs[1] = {q};
  Fq12 f = miller_loop<1>(ps, qs);
  Fq12 r = f.conj() * f.inv();
  r = r.frobenius(2) * r;
  // Karabina form: compressed squaring and the shared-inversion decompression, identity included
  const Fq12 r2 = r.cyclotomic_sqr();
  const CycloCompressed c[3] = {CycloCompressed::from(r).sqr(), CycloCompressed::from(Fq12::one()), CycloCompressed::from(r2).sqr()};
  Fq12 back[3];
  cyclotomic_decompress3(c, back);
  const bool karabina = back[0] == r2 && back[1] == Fq12::one() && back[2] == r2.cyclotomic_sqr();
  return karabina && (r2 == r.sqr()) && (r.sqr() == r * r) ? 1 : 0;
}
// ---- Fq12-level probes (12 x 48 B big-endian, tower order c0.c0.c0, c0.c0.c1, c0.c1.c0, ...) ----
static void fq12_read(const uint8_t* in, Fq12& f) {
  Fq* e = reinterpret_cast<Fq*>(&f);
  for (int i = 0; i < 12; i++) fq_from_be48(in + 48 * i, false, e[i]);
}
static void fq12_write(const Fq12& f, uint8_t* out) {
  const Fq* e = reinterpret_cast<const Fq*>(&f);
  for (int i = 0; i < 12; i++) fq_to_be48(e[i], out + 48 * i);
}
void hs_fq12_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fq12 x, y;
  fq12_read(a, x);
  fq12_read(b, y);
  fq12_write(x * y, out);
}
void hs_fq12_sqr(const uint8_t* a, uint8_t* out) {
  Fq12 x;
  fq12_read(a, x);
  fq12_write(x.sqr(), out);
}
void hs_fq12_inv(const uint8_t* a, uint8_t* out) {
  Fq12 x;
  fq12_read(a, x);
  fq12_write(x.inv(), out);
}
void hs_fq12_frobenius(const uint8_t* a, int k, uint8_t* out) {
  Fq12 x;
  fq12_read(a, x);
  fq12_write(x.frobenius(k), out);
}
void hs_fq12_cyclotomic_sqr(const uint8_t* a, uint8_t* out) {
  Fq12 x;
  fq12_read(a, x);
  fq12_write(x.cyclotomic_sqr(), out);
}
void hs_fq12_exp_by_x(const uint8_t* a, uint64_t x, uint8_t* out) {
  Fq12 f;
  fq12_read(a, f);
  fq12_write(cyclotomic_exp_by_x(f, x), out);
}
void hs_fq12_final_exp(const uint8_t* a, uint8_t* out) {
  Fq12 x;
  fq12_read(a, x);
  fq12_write(final_exponentiation(x), out);
}
int hs_miller_loop(const uint8_t* a, const uint8_t* b, uint8_t* out576) {
  G1Affine p;
  G2Affine q;
  if (!g1_decode_uncompressed(a, p) || !g2_decode_uncompressed(b, q)) return -1;
  G1Affine ps[1] = {p};
  G2Affine qs[1] = {q};
  fq12_write(miller_loop<1>(ps, qs), out576);
  return 0;
}

void hs_sha3(const uint8_t* msg, size_t len, uint8_t* out32) {
  uint32_t w[8];
  sha3_256_words(msg, len, w);
  memcpy(out32, w, 32);
}
void hs_chacha_words(const uint8_t* seed32, int n, uint32_t* out) {
  uint32_t k[8];
  memcpy(k, seed32, 32);
  ChaChaRng rng;
  rng.init(k);
  for (int i = 0; i < n; i++) out[i] = rng.next_u32();
}
void hs_hash_g2(const uint8_t* msg, size_t len, uint8_t* out192) { job_hash_g2(msg, len, out192); }
// the hash point before its last constant multiplication, and the two ways of folding that constant
void hs_hash_g2_unfixed(const uint8_t* msg, size_t len, uint8_t* out192) { job_hash_g2(msg, len, out192, false); }
void hs_fr_scale_cofactor_fix(const uint8_t* fr, uint8_t* out) { job_fr_scale_cofactor_fix(fr, out); }
void hs_g1_scale_cofactor_fix(const uint8_t* in96, uint8_t* out96) { job_g1_scale_cofactor_fix(in96, out96); }
#if defined(TC_TEST_HOOKS)
void hs_force_extra_hash_rounds(int n) { g_tc_force_extra_rounds = n; }
#endif
int hs_hash_g1_g2(const uint8_t* g1, const uint8_t* msg, size_t len, uint8_t* out192) {
  return job_hash_g1_g2(g1, msg, len, out192);
}
int hs_xor_with_hash(const uint8_t* g1, const uint8_t* data, size_t len, uint8_t* out) {
  return job_xor_with_hash(g1, data, len, out);
}
int hs_encrypt(const uint8_t* pk, const uint8_t* r, const uint8_t* msg, size_t len, uint8_t* u, uint8_t* v, uint8_t* w) {
  return job_encrypt(pk, r, msg, len, u, v, w);
}
int hs_commitment_evaluate(const uint8_t* commit, int t, uint64_t idx, uint8_t* out) {
  return job_commitment_evaluate(commit, t, idx, out);
}
int hs_decompress_g1(const uint8_t* in, uint8_t* out) { return job_decompress<Fq>(in, out); }
int hs_decompress_g2(const uint8_t* in, uint8_t* out) { return job_decompress<Fq2>(in, out); }
// two jobs per lane pair (tc_duo.h): on the host one thread runs both slots of every lane-split phase
int hs_decompress_g2_x2(const uint8_t* in_a, const uint8_t* in_b, uint8_t* out_a, uint8_t* out_b) {
  uint8_t sa, sb;
  job_decompress_g2_x2(in_a, in_b, out_a, out_b, sa, sb);
  return sa | (sb << 8);
}
void hs_hash_g2_x2(const uint8_t* msg_a, size_t len_a, const uint8_t* msg_b, size_t len_b, uint8_t* out_a, uint8_t* out_b, int fix) {
  job_hash_g2_x2(msg_a, len_a, msg_b, len_b, out_a, out_b, fix != 0);
}
int hs_hash_g1_g2_x2(const uint8_t* g1_a, const uint8_t* msg_a, size_t len_a, const uint8_t* g1_b, const uint8_t* msg_b, size_t len_b,
                     uint8_t* out_a, uint8_t* out_b, int fix) {
  uint8_t sa, sb;
  job_hash_g1_g2_x2(g1_a, msg_a, len_a, g1_b, msg_b, len_b, out_a, out_b, fix != 0, sa, sb);
  return sa | (sb << 8);
}
int hs_fq2_sqrt_x2(const uint8_t* a, const uint8_t* b, uint8_t* out_a, uint8_t* out_b) {
  Fq2 x, z, y, w;
  fq_from_be48(a, false, x.c0);
  fq_from_be48(a + 48, false, x.c1);
  fq_from_be48(b, false, z.c0);
  fq_from_be48(b + 48, false, z.c1);
  bool oka, okb;
  fq2_sqrt_x2(x, z, y, w, oka, okb);
  fq_to_be48(y.c0, out_a);
  fq_to_be48(y.c1, out_a + 48);
  fq_to_be48(w.c0, out_b);
  fq_to_be48(w.c1, out_b + 48);
  return (oka ? 1 : 0) | (okb ? 2 : 0);
}
int hs_compress_g1(const uint8_t* in, uint8_t* out) { return job_compress<Fq>(in, out); }
int hs_compress_g2(const uint8_t* in, uint8_t* out) { return job_compress<Fq2>(in, out); }

// ---- DKG algebra (tc_dkg.h) ----------------------------------------------------------------------
static std::vector<int32_t>& hs_fb_table() {
  static std::vector<int32_t> t;
  if (t.empty()) {
    t.resize(kFbTableWords);
    for (int e = 0; e < kFbWindows * kFbEntries; e++) fixed_base_table_entry(e, t.data() + (size_t)e * kFbPointWords);
  }
  return t;
}
int hs_g1_fixed_base_mul(const uint8_t* fr, uint8_t* out96) {
  return job_g1_fixed_base_mul((const int32_t*)hs_fb_table().data(), fr, out96);
}
int hs_bivar_commitment_row(const uint8_t* commit, size_t degree, size_t i, uint64_t x, uint8_t* out96) {
  return job_bivar_commitment_row(commit, degree, i, x, out96);
}
int hs_fr_interpolate(size_t n, const uint32_t* xs, const uint32_t* ys, uint32_t* out) {
  std::vector<uint32_t> ws(2 * (n + 1) * 8);
  return job_fr_interpolate(n, xs, ys, out, ws.data());
}

// ---- two-stage G2 linear combination (tc_msm.h): stage T for every chunk, then stage L ----------------------
int hs_msm_g2_nbits(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out192, int nbits);
int hs_msm_g2(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out192) {
  return hs_msm_g2_nbits(n, points, scalars, out192, 64);
}
int hs_msm_g2_nbits(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out192, int nbits) {
  const size_t chunks = msm_chunks(n), shares4 = chunks * kMsmChunk;
  std::vector<int32_t> tbl(shares4 * 8 * kMsmEntryWords);
  std::vector<uint8_t> codes(kMsmColumns * shares4);
  bool ok = true;
  for (size_t c = 0; c < chunks; c++) ok &= job_msm_tables(n, c, points, scalars, tbl.data(), codes.data(), true, nbits);
  if (!ok) {
    g2_encode_uncompressed(G2Affine::infinity(), out192);
    return TC_JOB_INVALID_ENCODING;
  }
  g2_encode_uncompressed(jac_to_affine(job_msm_ladder(n, tbl.data(), codes.data(), nbits)), out192);
  return TC_JOB_OK;
}
// stage L split over `parts` lane pairs (k_msm_ladder_split): the parts one after the other, then the same
// butterfly of additions the lane pairs of a wave run
int hs_msm_g2_split(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out192, size_t parts) {
  const size_t chunks = msm_chunks(n), shares4 = chunks * kMsmChunk;
  std::vector<int32_t> tbl(shares4 * 8 * kMsmEntryWords);
  std::vector<uint8_t> codes(kMsmColumns * shares4);
  bool ok = true;
  for (size_t c = 0; c < chunks; c++) ok &= job_msm_tables(n, c, points, scalars, tbl.data(), codes.data(), true, 64);
  if (!ok) return TC_JOB_INVALID_ENCODING;
  std::vector<G2Jac> r(parts);
  for (size_t g = 0; g < parts; g++) r[g] = job_msm_ladder_part<true>(n, tbl.data(), codes.data(), 64, msm_part(n, g, parts));
  for (size_t d = 1; d < parts; d <<= 1) {
    std::vector<G2Jac> nx(parts);
    for (size_t g = 0; g < parts; g++) nx[g] = jac_add(r[g], r[g ^ d]);
    r = nx;
  }
  g2_encode_uncompressed(jac_to_affine(r[0]), out192);
  return TC_JOB_OK;
}

// the same two stages in G1 (tc_msm.h job_msm_tables_g1 / job_msm_ladder_g1_part); parts > 1: the split form with the
// butterfly of additions the lanes of a job run
int hs_msm_g1_nbits(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out96, size_t parts, int nbits);
int hs_msm_g1(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out96, size_t parts) {
  return hs_msm_g1_nbits(n, points, scalars, out96, parts, 128);
}
int hs_msm_g1_nbits(size_t n, const uint8_t* points, const uint32_t* scalars, uint8_t* out96, size_t parts, int nbits) {
  const size_t chunks = msm_chunks(n), shares4 = chunks * kMsmChunk;
  const int top = nbits / 2;
  std::vector<int32_t> tbl(shares4 * 8 * kMsmEntryWordsG1);
  std::vector<uint8_t> codes(kMsmColumns * shares4);
  bool ok = true;
  for (size_t c = 0; c < chunks; c++) ok &= job_msm_tables_g1(n, c, points, scalars, tbl.data(), codes.data(), nbits);
  if (!ok) {
    g1_encode_uncompressed(G1Affine::infinity(), out96);
    return TC_JOB_INVALID_ENCODING;
  }
  if (parts <= 1) {
    g1_encode_uncompressed(jac_to_affine(job_msm_ladder_g1_part<false>(n, tbl.data(), codes.data(), msm_part(n), top)), out96);
    return TC_JOB_OK;
  }
  std::vector<G1Jac> r(parts);
  for (size_t g = 0; g < parts; g++) r[g] = job_msm_ladder_g1_part<true>(n, tbl.data(), codes.data(), msm_part(n, g, parts), top);
  for (size_t d = 1; d < parts; d <<= 1) {
    std::vector<G1Jac> nx(parts);
    for (size_t g = 0; g < parts; g++) nx[g] = jac_add(r[g], r[g ^ d]);
    r = nx;
  }
  g1_encode_uncompressed(jac_to_affine(r[0]), out96);
  return TC_JOB_OK;
}
void hs_msm_g1_recode(const uint32_t* k, uint8_t* codes65, int* flip) { *flip = msm_g1_recode(k, codes65, 1) ? 1 : 0; }

int hs_lagrange_split(const uint64_t* idx, int t, uint32_t* out) {  // k_lagrange_den + k_lagrange_finish, lane by lane
  const int n = t + 1;
  std::vector<uint32_t> xm(n * 8), den(n * 8), pre(n * 8);
  for (int i = 0; i < n; i++) {
    const Fr x = fr_from_u64(idx[i]) + Fr::one();
    for (int w = 0; w < 8; w++) xm[i * 8 + w] = x.v.l[w];
  }
  for (int i = 0; i < n; i++) {
    const Fr d = lagrange_denominator(idx, n, i);
    for (int w = 0; w < 8; w++) den[i * 8 + w] = d.v.l[w];
  }
  return lagrange_finish(n, xm.data(), den.data(), pre.data(), out);
}
int hs_lagrange_all(const uint64_t* idx, int t, uint32_t* out) {
  std::vector<uint32_t> ws(4 * (size_t)(t + 1) * 8);
  return lagrange_all_at_zero(idx, t, out, ws.data());
}

}  // extern "C"
static int hs_combine_g2_large(int t, const uint64_t* idx, const uint8_t* shares, uint32_t* lam, uint8_t* out) {
  std::vector<uint32_t> ws(4 * (size_t)(t + 1) * 8);
  int st = lagrange_all_at_zero(idx, t, lam, ws.data());
  if (st) {
    g2_encode_uncompressed(G2Affine::infinity(), out);
    return st;
  }
  return hs_msm_g2((size_t)t + 1, shares, lam, out);
}
extern "C" {
}
