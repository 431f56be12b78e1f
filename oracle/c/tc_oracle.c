/* Oracle B -- plain-C CPU restatement of the threshold_crypto 0.4.0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- never by the product package.  It shares no code with the HIP
 * kernels (64-bit limbs + unsigned __int128, in-place "assign" style, binary-Euclid
 * inversion, per-share inversion and two full pairings per check, as the reference's
 * dependencies do), so agreement between the two is evidence, not tautology.
 *
 * PARITY STATUS: "parity unpinned" against the Rust crate for the implementation-defined
 * hash_g2 sampling order (H-spec): the reference's tests hold no BLS12-381
 * known-answer vectors and neither cargo nor the pairing/ff/rand_chacha crates exist here.
 * Pinned instead against Oracle A (oracle/tc_oracle.py, textbook affine arithmetic) and the
 * public anchors listed there; see tests/test_oracle.py.
 *
 * Algorithm-faithful to (restated from the published sources, versions per Cargo.toml:22-33):
 *   ff_derive 0.6    Fq/Fr: schoolbook product + Montgomery reduction, dedicated squaring,
 *                    inverse by binary extended Euclid, random() by masked rejection sampling
 *   pairing 0.16     Fq2 (Karatsuba / complex squaring), Fq6, Fq12 (generic squaring in
 *                    exp_by_x), Jacobian dbl-2009-l / madd-2007-bl / add-2007-bl,
 *                    CurveAffine::mul = MSB-first double-and-add over all 256 bits,
 *                    G2Prepared (68 coefficient triples) + miller_loop + final_exponentiation,
 *                    G2::random, scale_by_cofactor, Zcash encodings
 *   rand_chacha 0.2  ChaCha20 word stream;  tiny-keccak 2.0  SHA3-256
 * Reference call sites followed (relative to /root/reference):
 *   src/lib.rs:691-694 hash_g2; :697-707 hash_g1_g2; :710-715 xor_with_hash;
 *   :719-773 interpolate/into_fr_plus_1; :108-117 verify_g2/verify; :182-186
 *   verify_decryption_share; :372-381 sign_g2/sign; :460-462 decrypt_share_no_verify;
 *   :508-512 Ciphertext::verify; :608-626 combine_signatures/decrypt; src/into_fr.rs:16-20.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------ */
/* Fq: 6 x u64 Montgomery, R = 2^384                                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct { u64 l[6]; } fq;

static const u64 FQ_MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                              0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 FQ_R1[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                             0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
static const u64 FQ_R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                             0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
#define FQ_INV 0x89f3fffcfffcfffdull

static __thread u64 g_fq_mul_count = 0; /* Fq multiplications + squarings by this thread (work constants) */

static inline u64 adc(u64 a, u64 b, u64 *carry) {
  u128 t = (u128)a + b + *carry;
  *carry = (u64)(t >> 64);
  return (u64)t;
}
static inline u64 sbb(u64 a, u64 b, u64 *borrow) {
  u128 t = (u128)a - b - *borrow;
  *borrow = (u64)(t >> 64) & 1;
  return (u64)t;
}
static inline u64 mac(u64 a, u64 b, u64 c, u64 *carry) {
  u128 t = (u128)b * c + a + *carry;
  *carry = (u64)(t >> 64);
  return (u64)t;
}

static int big_geq(const u64 *a, const u64 *b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static int big_is_zero(const u64 *a, int n) {
  u64 o = 0;
  for (int i = 0; i < n; i++) o |= a[i];
  return o == 0;
}
static void big_sub(u64 *a, const u64 *b, int n) {
  u64 br = 0;
  for (int i = 0; i < n; i++) a[i] = sbb(a[i], b[i], &br);
}
static void big_add(u64 *a, const u64 *b, int n) {
  u64 c = 0;
  for (int i = 0; i < n; i++) a[i] = adc(a[i], b[i], &c);
}
static void big_div2(u64 *a, int n) {
  u64 t = 0;
  for (int i = n - 1; i >= 0; i--) {
    u64 t2 = a[i] << 63;
    a[i] = (a[i] >> 1) | t;
    t = t2;
  }
}

static void fq_reduce(fq *a) {
  if (big_geq(a->l, FQ_MOD, 6)) big_sub(a->l, FQ_MOD, 6);
}
static void fq_add(fq *a, const fq *b) {
  big_add(a->l, b->l, 6);
  fq_reduce(a);
}
static void fq_sub(fq *a, const fq *b) {
  if (!big_geq(a->l, b->l, 6)) big_add(a->l, FQ_MOD, 6);
  big_sub(a->l, b->l, 6);
}
static void fq_dbl(fq *a) {
  fq t = *a;
  fq_add(a, &t);
}
static void fq_neg(fq *a) {
  if (!big_is_zero(a->l, 6)) {
    fq t;
    memcpy(t.l, FQ_MOD, 48);
    big_sub(t.l, a->l, 6);
    *a = t;
  }
}
static int fq_is_zero(const fq *a) { return big_is_zero(a->l, 6); }
static int fq_eq(const fq *a, const fq *b) { return memcmp(a->l, b->l, 48) == 0; }

static void fq_mont_reduce(fq *out, u64 *r /* 12 limbs */) {
  /* ff_derive mont_reduce: one round per low limb */
  u64 carry2 = 0;
  for (int i = 0; i < 6; i++) {
    u64 k = r[i] * FQ_INV;
    u64 carry = 0;
    (void)mac(r[i], k, FQ_MOD[0], &carry);
    for (int j = 1; j < 6; j++) r[i + j] = mac(r[i + j], k, FQ_MOD[j], &carry);
    r[i + 6] = adc(r[i + 6], carry2, &carry);
    carry2 = carry;
  }
  memcpy(out->l, r + 6, 48);
  fq_reduce(out);
}
static void fq_mul(fq *a, const fq *b) {
  u64 r[12] = {0};
  for (int i = 0; i < 6; i++) {
    u64 carry = 0;
    for (int j = 0; j < 6; j++) r[i + j] = mac(r[i + j], a->l[i], b->l[j], &carry);
    r[i + 6] = carry;
  }
  g_fq_mul_count++;
  fq_mont_reduce(a, r);
}
static void fq_sqr(fq *a) {
  /* off-diagonal products once, doubled, plus the diagonal */
  u64 r[12] = {0};
  for (int i = 0; i < 5; i++) {
    u64 carry = 0;
    for (int j = i + 1; j < 6; j++) r[i + j] = mac(r[i + j], a->l[i], a->l[j], &carry);
    r[i + 6] = carry;
  }
  r[11] = r[10] >> 63;
  for (int i = 10; i >= 2; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
  r[1] = r[1] << 1;
  u64 carry = 0;
  for (int i = 0; i < 6; i++) {
    r[2 * i] = mac(r[2 * i], a->l[i], a->l[i], &carry);
    r[2 * i + 1] = adc(r[2 * i + 1], 0, &carry);
  }
  g_fq_mul_count++;
  fq_mont_reduce(a, r);
}

/* binary extended Euclid on Montgomery values (ff_derive Field::inverse): returns 0 for 0 */
static int fq_inv(fq *a) {
  if (fq_is_zero(a)) return 0;
  u64 u[6], v[6];
  fq b, c;
  memcpy(u, a->l, 48);
  memcpy(v, FQ_MOD, 48);
  memcpy(b.l, FQ_R2, 48); /* b = R^2 so the result stays in Montgomery form */
  memset(c.l, 0, 48);
  static const u64 ONE[6] = {1, 0, 0, 0, 0, 0};
  while (memcmp(u, ONE, 48) != 0 && memcmp(v, ONE, 48) != 0) {
    while ((u[0] & 1) == 0) {
      big_div2(u, 6);
      if (b.l[0] & 1) {
        u64 cy = 0;
        for (int i = 0; i < 6; i++) b.l[i] = adc(b.l[i], FQ_MOD[i], &cy);
        big_div2(b.l, 6);
        b.l[5] |= cy << 63;
      } else big_div2(b.l, 6);
    }
    while ((v[0] & 1) == 0) {
      big_div2(v, 6);
      if (c.l[0] & 1) {
        u64 cy = 0;
        for (int i = 0; i < 6; i++) c.l[i] = adc(c.l[i], FQ_MOD[i], &cy);
        big_div2(c.l, 6);
        c.l[5] |= cy << 63;
      } else big_div2(c.l, 6);
    }
    if (big_geq(u, v, 6)) {
      big_sub(u, v, 6);
      fq_sub(&b, &c);
    } else {
      big_sub(v, u, 6);
      fq_sub(&c, &b);
    }
  }
  *a = (memcmp(u, ONE, 48) == 0) ? b : c;
  return 1;
}

static void fq_from_raw(fq *out, const u64 *canon) { /* canonical integer -> Montgomery */
  fq t, r2;
  memcpy(t.l, canon, 48);
  memcpy(r2.l, FQ_R2, 48);
  fq_mul(&t, &r2);
  *out = t;
}
static void fq_to_raw(const fq *a, u64 *canon) {
  u64 r[12] = {0};
  memcpy(r, a->l, 48);
  fq t;
  fq_mont_reduce(&t, r);
  memcpy(canon, t.l, 48);
}
static void fq_one(fq *a) { memcpy(a->l, FQ_R1, 48); }
static void fq_zero(fq *a) { memset(a->l, 0, 48); }

/* canonical-value comparison (Fq Ord) */
static int fq_cmp(const fq *a, const fq *b) {
  u64 x[6], y[6];
  fq_to_raw(a, x);
  fq_to_raw(b, y);
  for (int i = 5; i >= 0; i--) {
    if (x[i] > y[i]) return 1;
    if (x[i] < y[i]) return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Fr: 4 x u64 Montgomery, R = 2^256                                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct { u64 l[4]; } fr;
static const u64 FR_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 FR_R1[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const u64 FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
#define FR_INV 0xfffffffeffffffffull

static void fr_reduce(fr *a) {
  if (big_geq(a->l, FR_MOD, 4)) big_sub(a->l, FR_MOD, 4);
}
static void fr_add(fr *a, const fr *b) {
  big_add(a->l, b->l, 4);
  fr_reduce(a);
}
static void fr_sub(fr *a, const fr *b) {
  if (!big_geq(a->l, b->l, 4)) big_add(a->l, FR_MOD, 4);
  big_sub(a->l, b->l, 4);
}
static void fr_mont_reduce(fr *out, u64 *r) {
  u64 carry2 = 0;
  for (int i = 0; i < 4; i++) {
    u64 k = r[i] * FR_INV;
    u64 carry = 0;
    (void)mac(r[i], k, FR_MOD[0], &carry);
    for (int j = 1; j < 4; j++) r[i + j] = mac(r[i + j], k, FR_MOD[j], &carry);
    r[i + 4] = adc(r[i + 4], carry2, &carry);
    carry2 = carry;
  }
  memcpy(out->l, r + 4, 32);
  fr_reduce(out);
}
static void fr_mul(fr *a, const fr *b) {
  u64 r[8] = {0};
  for (int i = 0; i < 4; i++) {
    u64 carry = 0;
    for (int j = 0; j < 4; j++) r[i + j] = mac(r[i + j], a->l[i], b->l[j], &carry);
    r[i + 4] = carry;
  }
  fr_mont_reduce(a, r);
}
static int fr_is_zero(const fr *a) { return big_is_zero(a->l, 4); }
static int fr_eq(const fr *a, const fr *b) { return memcmp(a->l, b->l, 32) == 0; }
static void fr_one(fr *a) { memcpy(a->l, FR_R1, 32); }
static int fr_inv(fr *a) {
  if (fr_is_zero(a)) return 0;
  u64 u[4], v[4];
  fr b, c;
  memcpy(u, a->l, 32);
  memcpy(v, FR_MOD, 32);
  memcpy(b.l, FR_R2, 32);
  memset(c.l, 0, 32);
  static const u64 ONE[4] = {1, 0, 0, 0};
  while (memcmp(u, ONE, 32) != 0 && memcmp(v, ONE, 32) != 0) {
    while ((u[0] & 1) == 0) {
      big_div2(u, 4);
      if (b.l[0] & 1) {
        u64 cy = 0;
        for (int i = 0; i < 4; i++) b.l[i] = adc(b.l[i], FR_MOD[i], &cy);
        big_div2(b.l, 4);
        b.l[3] |= cy << 63;
      } else big_div2(b.l, 4);
    }
    while ((v[0] & 1) == 0) {
      big_div2(v, 4);
      if (c.l[0] & 1) {
        u64 cy = 0;
        for (int i = 0; i < 4; i++) c.l[i] = adc(c.l[i], FR_MOD[i], &cy);
        big_div2(c.l, 4);
        c.l[3] |= cy << 63;
      } else big_div2(c.l, 4);
    }
    if (big_geq(u, v, 4)) {
      big_sub(u, v, 4);
      fr_sub(&b, &c);
    } else {
      big_sub(v, u, 4);
      fr_sub(&c, &b);
    }
  }
  *a = (memcmp(u, ONE, 32) == 0) ? b : c;
  return 1;
}
static void fr_from_raw(fr *out, const u64 *canon) {
  fr t, r2;
  memcpy(t.l, canon, 32);
  memcpy(r2.l, FR_R2, 32);
  fr_mul(&t, &r2);
  *out = t;
}
static void fr_to_raw(const fr *a, u64 *canon) {
  u64 r[8] = {0};
  memcpy(r, a->l, 32);
  fr t;
  fr_mont_reduce(&t, r);
  memcpy(canon, t.l, 32);
}
/* IntoFr for u64 (src/into_fr.rs:16-20) */
static void fr_from_u64(fr *out, u64 x) {
  u64 c[4] = {x, 0, 0, 0};
  fr_from_raw(out, c);
}

/* ------------------------------------------------------------------------------------ */
/* Fq2, Fq6, Fq12                                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct { fq c0, c1; } fq2;
typedef struct { fq2 c0, c1, c2; } fq6;
typedef struct { fq6 c0, c1; } fq12;

static void fq2_add(fq2 *a, const fq2 *b) { fq_add(&a->c0, &b->c0); fq_add(&a->c1, &b->c1); }
static void fq2_sub(fq2 *a, const fq2 *b) { fq_sub(&a->c0, &b->c0); fq_sub(&a->c1, &b->c1); }
static void fq2_dbl(fq2 *a) { fq_dbl(&a->c0); fq_dbl(&a->c1); }
static void fq2_neg(fq2 *a) { fq_neg(&a->c0); fq_neg(&a->c1); }
static int fq2_is_zero(const fq2 *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static int fq2_eq(const fq2 *a, const fq2 *b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static void fq2_zero(fq2 *a) { fq_zero(&a->c0); fq_zero(&a->c1); }
static void fq2_one(fq2 *a) { fq_one(&a->c0); fq_zero(&a->c1); }
static void fq2_mul(fq2 *a, const fq2 *b) {
  fq aa = a->c0, bb = a->c1, o = b->c0;
  fq_mul(&aa, &b->c0);
  fq_mul(&bb, &b->c1);
  fq_add(&o, &b->c1);
  fq_add(&a->c1, &a->c0);
  fq_mul(&a->c1, &o);
  fq_sub(&a->c1, &aa);
  fq_sub(&a->c1, &bb);
  a->c0 = aa;
  fq_sub(&a->c0, &bb);
}
static void fq2_sqr(fq2 *a) {
  fq ab = a->c0, c0c1 = a->c0, t = a->c1;
  fq_mul(&ab, &a->c1);
  fq_add(&c0c1, &a->c1);
  fq_neg(&t);
  fq_add(&t, &a->c0); /* c0 - c1 */
  fq_mul(&c0c1, &t);
  a->c0 = c0c1;
  a->c1 = ab;
  fq_dbl(&a->c1);
}
static void fq2_mul_by_nonresidue(fq2 *a) { /* times (1 + u) */
  fq t0 = a->c0;
  fq_sub(&a->c0, &a->c1);
  fq_add(&a->c1, &t0);
}
static void fq2_frobenius(fq2 *a, int power) {
  if (power & 1) fq_neg(&a->c1);
}
static int fq2_inv(fq2 *a) {
  fq t1 = a->c1, t0 = a->c0;
  fq_sqr(&t1);
  fq_sqr(&t0);
  fq_add(&t1, &t0);
  if (!fq_inv(&t1)) return 0;
  fq_mul(&a->c0, &t1);
  fq_mul(&a->c1, &t1);
  fq_neg(&a->c1);
  return 1;
}
static void fq2_pow(fq2 *a, const u64 *e, int nlimbs) {
  fq2 res;
  fq2_one(&res);
  int found = 0;
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    int bit = (e[i / 64] >> (i % 64)) & 1;
    if (found) fq2_sqr(&res); else found = bit;
    if (bit) fq2_mul(&res, a);
  }
  *a = res;
}
/* Fq2 Ord: c1 first, then c0 */
static int fq2_cmp(const fq2 *a, const fq2 *b) {
  int c = fq_cmp(&a->c1, &b->c1);
  return c ? c : fq_cmp(&a->c0, &b->c0);
}
/* Fq2::sqrt, Algorithm 9 of eprint 2012/685 */
static const u64 Q_M3_D4[6] = {0xee7fbfffffffeaaaull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                               0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
static const u64 Q_M1_D2[6] = {0xdcff7fffffffd555ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull,
                               0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};
static int fq2_sqrt(fq2 *out, const fq2 *a) {
  if (fq2_is_zero(a)) { fq2_zero(out); return 1; }
  fq2 a1 = *a;
  fq2_pow(&a1, Q_M3_D4, 6);
  fq2 alpha = a1;
  fq2_sqr(&alpha);
  fq2_mul(&alpha, a);
  fq2 a0 = alpha;
  fq2_frobenius(&a0, 1);
  fq2_mul(&a0, &alpha);
  fq2 neg1;
  fq2_one(&neg1);
  fq2_neg(&neg1);
  if (fq2_eq(&a0, &neg1)) return 0;
  fq2_mul(&a1, a);
  if (fq2_eq(&alpha, &neg1)) {
    fq2 u;
    fq_zero(&u.c0);
    fq_one(&u.c1);
    fq2_mul(&a1, &u);
  } else {
    fq2 one;
    fq2_one(&one);
    fq2_add(&alpha, &one);
    fq2_pow(&alpha, Q_M1_D2, 6);
    fq2_mul(&a1, &alpha);
  }
  *out = a1;
  return 1;
}

static void fq6_add(fq6 *a, const fq6 *b) { fq2_add(&a->c0, &b->c0); fq2_add(&a->c1, &b->c1); fq2_add(&a->c2, &b->c2); }
static void fq6_sub(fq6 *a, const fq6 *b) { fq2_sub(&a->c0, &b->c0); fq2_sub(&a->c1, &b->c1); fq2_sub(&a->c2, &b->c2); }
static void fq6_neg(fq6 *a) { fq2_neg(&a->c0); fq2_neg(&a->c1); fq2_neg(&a->c2); }
static void fq6_zero(fq6 *a) { fq2_zero(&a->c0); fq2_zero(&a->c1); fq2_zero(&a->c2); }
static void fq6_one(fq6 *a) { fq2_one(&a->c0); fq2_zero(&a->c1); fq2_zero(&a->c2); }
static int fq6_eq(const fq6 *a, const fq6 *b) { return fq2_eq(&a->c0, &b->c0) && fq2_eq(&a->c1, &b->c1) && fq2_eq(&a->c2, &b->c2); }
static void fq6_mul_by_nonresidue(fq6 *a) { /* times v */
  fq2 t = a->c2;
  a->c2 = a->c1;
  a->c1 = a->c0;
  fq2_mul_by_nonresidue(&t);
  a->c0 = t;
}
static void fq6_mul(fq6 *a, const fq6 *b) {
  fq2 aa = a->c0, bb = a->c1, cc = a->c2;
  fq2_mul(&aa, &b->c0);
  fq2_mul(&bb, &b->c1);
  fq2_mul(&cc, &b->c2);
  fq2 t1 = b->c1, tmp = a->c1;
  fq2_add(&t1, &b->c2);
  fq2_add(&tmp, &a->c2);
  fq2_mul(&t1, &tmp);
  fq2_sub(&t1, &bb);
  fq2_sub(&t1, &cc);
  fq2_mul_by_nonresidue(&t1);
  fq2_add(&t1, &aa);
  fq2 t3 = b->c0;
  tmp = a->c0;
  fq2_add(&t3, &b->c2);
  fq2_add(&tmp, &a->c2);
  fq2_mul(&t3, &tmp);
  fq2_sub(&t3, &aa);
  fq2_add(&t3, &bb);
  fq2_sub(&t3, &cc);
  fq2 t2 = b->c0;
  tmp = a->c0;
  fq2_add(&t2, &b->c1);
  fq2_add(&tmp, &a->c1);
  fq2_mul(&t2, &tmp);
  fq2_sub(&t2, &aa);
  fq2_sub(&t2, &bb);
  fq2_mul_by_nonresidue(&cc);
  fq2_add(&t2, &cc);
  a->c0 = t1;
  a->c1 = t2;
  a->c2 = t3;
}
static void fq6_sqr(fq6 *a) {
  fq2 s0 = a->c0, ab = a->c0, s1, s2 = a->c0, bc = a->c1, s3, s4 = a->c2;
  fq2_sqr(&s0);
  fq2_mul(&ab, &a->c1);
  s1 = ab;
  fq2_dbl(&s1);
  fq2_sub(&s2, &a->c1);
  fq2_add(&s2, &a->c2);
  fq2_sqr(&s2);
  fq2_mul(&bc, &a->c2);
  s3 = bc;
  fq2_dbl(&s3);
  fq2_sqr(&s4);
  a->c0 = s3;
  fq2_mul_by_nonresidue(&a->c0);
  fq2_add(&a->c0, &s0);
  a->c1 = s4;
  fq2_mul_by_nonresidue(&a->c1);
  fq2_add(&a->c1, &s1);
  a->c2 = s1;
  fq2_add(&a->c2, &s2);
  fq2_add(&a->c2, &s3);
  fq2_sub(&a->c2, &s0);
  fq2_sub(&a->c2, &s4);
}
static void fq6_mul_by_1(fq6 *a, const fq2 *c1) {
  fq2 bb = a->c1, t1 = *c1, tmp = a->c1, t2 = *c1;
  fq2_mul(&bb, c1);
  fq2_add(&tmp, &a->c2);
  fq2_mul(&t1, &tmp);
  fq2_sub(&t1, &bb);
  fq2_mul_by_nonresidue(&t1);
  tmp = a->c0;
  fq2_add(&tmp, &a->c1);
  fq2_mul(&t2, &tmp);
  fq2_sub(&t2, &bb);
  a->c0 = t1;
  a->c1 = t2;
  a->c2 = bb;
}
static void fq6_mul_by_01(fq6 *a, const fq2 *c0, const fq2 *c1) {
  fq2 aa = a->c0, bb = a->c1, t1 = *c1, tmp = a->c1, t3 = *c0, t2 = *c0;
  fq2_mul(&aa, c0);
  fq2_mul(&bb, c1);
  fq2_add(&tmp, &a->c2);
  fq2_mul(&t1, &tmp);
  fq2_sub(&t1, &bb);
  fq2_mul_by_nonresidue(&t1);
  fq2_add(&t1, &aa);
  tmp = a->c0;
  fq2_add(&tmp, &a->c2);
  fq2_mul(&t3, &tmp);
  fq2_sub(&t3, &aa);
  fq2_add(&t3, &bb);
  fq2_add(&t2, c1);
  tmp = a->c0;
  fq2_add(&tmp, &a->c1);
  fq2_mul(&t2, &tmp);
  fq2_sub(&t2, &aa);
  fq2_sub(&t2, &bb);
  a->c0 = t1;
  a->c1 = t2;
  a->c2 = t3;
}
static int fq6_inv(fq6 *a) {
  fq2 c0 = a->c2, c1 = a->c2, c2 = a->c1, t;
  fq2_mul_by_nonresidue(&c0);
  fq2_mul(&c0, &a->c1);
  fq2_neg(&c0);
  t = a->c0;
  fq2_sqr(&t);
  fq2_add(&c0, &t);
  fq2_sqr(&c1);
  fq2_mul_by_nonresidue(&c1);
  t = a->c0;
  fq2_mul(&t, &a->c1);
  fq2_sub(&c1, &t);
  fq2_sqr(&c2);
  t = a->c0;
  fq2_mul(&t, &a->c2);
  fq2_sub(&c2, &t);
  fq2 tmp1 = a->c2, tmp2 = a->c1;
  fq2_mul(&tmp1, &c1);
  fq2_mul(&tmp2, &c2);
  fq2_add(&tmp1, &tmp2);
  fq2_mul_by_nonresidue(&tmp1);
  tmp2 = a->c0;
  fq2_mul(&tmp2, &c0);
  fq2_add(&tmp1, &tmp2);
  if (!fq2_inv(&tmp1)) return 0;
  a->c0 = tmp1; fq2_mul(&a->c0, &c0);
  a->c1 = tmp1; fq2_mul(&a->c1, &c1);
  a->c2 = tmp1; fq2_mul(&a->c2, &c2);
  return 1;
}

/* Frobenius coefficients are derived at start-up from xi = 1 + u by exponentiation */
static fq2 FROB6_C1[4], FROB6_C2[4], FROB12_C1[4]; /* powers 0..3 are all the path uses */
static int g_init_done = 0;

static void fq12_one(fq12 *a) { fq6_one(&a->c0); fq6_zero(&a->c1); }
static int fq12_eq(const fq12 *a, const fq12 *b) { return fq6_eq(&a->c0, &b->c0) && fq6_eq(&a->c1, &b->c1); }
static void fq12_conjugate(fq12 *a) { fq6_neg(&a->c1); }
static void fq12_mul(fq12 *a, const fq12 *b) {
  fq6 aa = a->c0, bb = a->c1, o = b->c0;
  fq6_mul(&aa, &b->c0);
  fq6_mul(&bb, &b->c1);
  fq6_add(&o, &b->c1);
  fq6_add(&a->c1, &a->c0);
  fq6_mul(&a->c1, &o);
  fq6_sub(&a->c1, &aa);
  fq6_sub(&a->c1, &bb);
  a->c0 = bb;
  fq6_mul_by_nonresidue(&a->c0);
  fq6_add(&a->c0, &aa);
}
static void fq12_sqr(fq12 *a) {
  fq6 ab = a->c0, c0c1 = a->c0, c0 = a->c1;
  fq6_mul(&ab, &a->c1);
  fq6_add(&c0c1, &a->c1);
  fq6_mul_by_nonresidue(&c0);
  fq6_add(&c0, &a->c0);
  fq6_mul(&c0, &c0c1);
  fq6_sub(&c0, &ab);
  a->c1 = ab;
  fq6_add(&a->c1, &ab);
  fq6_mul_by_nonresidue(&ab);
  fq6_sub(&c0, &ab);
  a->c0 = c0;
}
static int fq12_inv(fq12 *a) {
  fq6 c0s = a->c0, c1s = a->c1;
  fq6_sqr(&c0s);
  fq6_sqr(&c1s);
  fq6_mul_by_nonresidue(&c1s);
  fq6_sub(&c0s, &c1s);
  if (!fq6_inv(&c0s)) return 0;
  fq6_mul(&a->c0, &c0s);
  fq6_mul(&a->c1, &c0s);
  fq6_neg(&a->c1);
  return 1;
}
static void fq12_mul_by_014(fq12 *a, const fq2 *c0, const fq2 *c1, const fq2 *c4) {
  fq6 aa = a->c0, bb = a->c1;
  fq6_mul_by_01(&aa, c0, c1);
  fq6_mul_by_1(&bb, c4);
  fq2 o = *c1;
  fq2_add(&o, c4);
  fq6_add(&a->c1, &a->c0);
  fq6_mul_by_01(&a->c1, c0, &o);
  fq6_sub(&a->c1, &aa);
  fq6_sub(&a->c1, &bb);
  a->c0 = bb;
  fq6_mul_by_nonresidue(&a->c0);
  fq6_add(&a->c0, &aa);
}
static void fq6_frobenius(fq6 *a, int power) {
  fq2_frobenius(&a->c0, power);
  fq2_frobenius(&a->c1, power);
  fq2_frobenius(&a->c2, power);
  fq2_mul(&a->c1, &FROB6_C1[power]);
  fq2_mul(&a->c2, &FROB6_C2[power]);
}
static void fq12_frobenius(fq12 *a, int power) {
  fq6_frobenius(&a->c0, power);
  fq6_frobenius(&a->c1, power);
  fq2_mul(&a->c1.c0, &FROB12_C1[power]);
  fq2_mul(&a->c1.c1, &FROB12_C1[power]);
  fq2_mul(&a->c1.c2, &FROB12_C1[power]);
}
static void fq12_pow_u64(fq12 *a, u64 e) { /* Field::pow: generic square-and-multiply */
  fq12 res;
  fq12_one(&res);
  int found = 0;
  for (int i = 63; i >= 0; i--) {
    int bit = (e >> i) & 1;
    if (found) fq12_sqr(&res); else found = bit;
    if (bit) fq12_mul(&res, a);
  }
  *a = res;
}

/* big-integer helpers for start-up constant derivation: (q^k - 1) / d for small k, d */
static void bn_mul_small(u64 *a, int *n, const u64 *b, int nb) { /* a (n limbs) *= b */
  u64 r[40] = {0};
  for (int i = 0; i < *n; i++) {
    u64 c = 0;
    for (int j = 0; j < nb; j++) r[i + j] = mac(r[i + j], a[i], b[j], &c);
    r[i + nb] += c;
  }
  *n += nb;
  memcpy(a, r, (size_t)(*n) * 8);
}
static void bn_div_small(u64 *a, int n, u64 d) {
  u128 rem = 0;
  for (int i = n - 1; i >= 0; i--) {
    u128 cur = (rem << 64) | a[i];
    a[i] = (u64)(cur / d);
    rem = cur % d;
  }
}
static void tc_init(void) {
  if (g_init_done) return;
  fq2 xi;
  fq_one(&xi.c0);
  fq_one(&xi.c1);
  for (int k = 0; k < 4; k++) {
    /* e = q^k - 1 */
    u64 e[40] = {1};
    int n = 1;
    for (int i = 0; i < k; i++) bn_mul_small(e, &n, FQ_MOD, 6);
    u64 one[40] = {1};
    big_sub(e, one, n);
    u64 e3[40], e6[40], e23[40];
    memcpy(e3, e, sizeof e); bn_div_small(e3, n, 3);
    memcpy(e6, e, sizeof e); bn_div_small(e6, n, 6);
    memcpy(e23, e3, sizeof e3); { u64 c = 0; for (int i = 0; i < n; i++) e23[i] = adc(e23[i], e3[i], &c); }
    FROB6_C1[k] = xi; fq2_pow(&FROB6_C1[k], e3, n);
    FROB6_C2[k] = xi; fq2_pow(&FROB6_C2[k], e23, n);
    FROB12_C1[k] = xi; fq2_pow(&FROB12_C1[k], e6, n);
  }
  g_init_done = 1;
}

/* ------------------------------------------------------------------------------------ */
/* Curves: affine + Jacobian over Fq (G1) and Fq2 (G2); written twice, as the crate does   */
/* (its curve_impl! macro expands once per group)                                        */
/* ------------------------------------------------------------------------------------ */
#define CURVE_IMPL(P, F)                                                                              \
  typedef struct { F x, y; int inf; } P##_aff;                                                        \
  typedef struct { F x, y, z; } P##_jac;                                                              \
  static void P##_jac_zero(P##_jac *r) { F##_zero(&r->x); F##_one(&r->y); F##_zero(&r->z); }          \
  static int P##_jac_is_zero(const P##_jac *p) { return F##_is_zero(&p->z); }                         \
  static void P##_double(P##_jac *p) {                                                                \
    if (P##_jac_is_zero(p)) return;                                                                   \
    F a = p->x, b = p->y, c, d = p->x, e, f, t;                                                       \
    F##_sqr(&a); F##_sqr(&b); c = b; F##_sqr(&c);                                                     \
    F##_add(&d, &b); F##_sqr(&d); F##_sub(&d, &a); F##_sub(&d, &c); F##_dbl(&d);                      \
    e = a; F##_dbl(&e); F##_add(&e, &a);                                                              \
    f = e; F##_sqr(&f);                                                                               \
    F##_mul(&p->z, &p->y); F##_dbl(&p->z);                                                            \
    p->x = f; F##_sub(&p->x, &d); F##_sub(&p->x, &d);                                                 \
    p->y = d; F##_sub(&p->y, &p->x); F##_mul(&p->y, &e);                                              \
    t = c; F##_dbl(&t); F##_dbl(&t); F##_dbl(&t); F##_sub(&p->y, &t);                                 \
  }                                                                                                   \
  static void P##_add_mixed(P##_jac *p, const P##_aff *q) {                                           \
    if (q->inf) return;                                                                               \
    if (P##_jac_is_zero(p)) { p->x = q->x; p->y = q->y; F##_one(&p->z); return; }                     \
    F z1z1 = p->z, u2 = q->x, s2 = q->y;                                                              \
    F##_sqr(&z1z1); F##_mul(&u2, &z1z1); F##_mul(&s2, &p->z); F##_mul(&s2, &z1z1);                    \
    if (F##_eq(&p->x, &u2) && F##_eq(&p->y, &s2)) { P##_double(p); return; }                          \
    F h = u2, hh, i, j, r = s2, v = p->x, t;                                                          \
    F##_sub(&h, &p->x); hh = h; F##_sqr(&hh); i = hh; F##_dbl(&i); F##_dbl(&i);                       \
    j = h; F##_mul(&j, &i); F##_sub(&r, &p->y); F##_dbl(&r); F##_mul(&v, &i);                         \
    p->x = r; F##_sqr(&p->x); F##_sub(&p->x, &j); F##_sub(&p->x, &v); F##_sub(&p->x, &v);             \
    F##_mul(&j, &p->y); F##_dbl(&j);                                                                  \
    p->y = v; F##_sub(&p->y, &p->x); F##_mul(&p->y, &r); F##_sub(&p->y, &j);                          \
    t = p->z; F##_add(&t, &h); F##_sqr(&t); F##_sub(&t, &z1z1); F##_sub(&t, &hh); p->z = t;           \
  }                                                                                                   \
  static void P##_add(P##_jac *p, const P##_jac *q) {                                                 \
    if (P##_jac_is_zero(p)) { *p = *q; return; }                                                      \
    if (P##_jac_is_zero(q)) return;                                                                   \
    F z1z1 = p->z, z2z2 = q->z, u1 = p->x, u2 = q->x, s1 = p->y, s2 = q->y;                           \
    F##_sqr(&z1z1); F##_sqr(&z2z2); F##_mul(&u1, &z2z2); F##_mul(&u2, &z1z1);                         \
    F##_mul(&s1, &q->z); F##_mul(&s1, &z2z2); F##_mul(&s2, &p->z); F##_mul(&s2, &z1z1);               \
    if (F##_eq(&u1, &u2) && F##_eq(&s1, &s2)) { P##_double(p); return; }                              \
    F h = u2, i, j, r = s2, v = u1, t;                                                                \
    F##_sub(&h, &u1); i = h; F##_dbl(&i); F##_sqr(&i); j = h; F##_mul(&j, &i);                        \
    F##_sub(&r, &s1); F##_dbl(&r); F##_mul(&v, &i);                                                   \
    p->x = r; F##_sqr(&p->x); F##_sub(&p->x, &j); F##_sub(&p->x, &v); F##_sub(&p->x, &v);             \
    p->y = v; F##_sub(&p->y, &p->x); F##_mul(&p->y, &r); F##_mul(&s1, &j); F##_dbl(&s1);              \
    F##_sub(&p->y, &s1);                                                                              \
    t = p->z; F##_add(&t, &q->z); F##_sqr(&t); F##_sub(&t, &z1z1); F##_sub(&t, &z2z2);                \
    F##_mul(&t, &h); p->z = t;                                                                        \
  }                                                                                                   \
  static void P##_into_affine(P##_aff *r, const P##_jac *p) {                                         \
    if (P##_jac_is_zero(p)) { F##_zero(&r->x); F##_one(&r->y); r->inf = 1; return; }                  \
    F zi = p->z, zi2;                                                                                 \
    F##_inv(&zi); zi2 = zi; F##_sqr(&zi2);                                                            \
    r->x = p->x; F##_mul(&r->x, &zi2);                                                                \
    r->y = p->y; F##_mul(&r->y, &zi2); F##_mul(&r->y, &zi); r->inf = 0;                               \
  }                                                                                                   \
  /* CurveAffine::mul -> mul_bits over ALL bits of the representation, MSB first */                   \
  static void P##_mul_bits(P##_jac *res, const P##_aff *p, const u64 *k, int nlimbs) {                \
    P##_jac_zero(res);                                                                                \
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {                                                      \
      P##_double(res);                                                                                \
      if ((k[i / 64] >> (i % 64)) & 1) P##_add_mixed(res, p);                                         \
    }                                                                                                 \
  }

CURVE_IMPL(g1, fq)
CURVE_IMPL(g2, fq2)

static void fq_b_g1(fq *b) { u64 four[6] = {4, 0, 0, 0, 0, 0}; fq_from_raw(b, four); }
static void fq2_b_g2(fq2 *b) { fq_b_g1(&b->c0); b->c1 = b->c0; }
static int g1_on_curve(const g1_aff *p) {
  if (p->inf) return 1;
  fq y2 = p->y, x3 = p->x, b;
  fq_sqr(&y2); fq_sqr(&x3); fq_mul(&x3, &p->x); fq_b_g1(&b); fq_add(&x3, &b);
  return fq_eq(&y2, &x3);
}
static int g2_on_curve(const g2_aff *p) {
  if (p->inf) return 1;
  fq2 y2 = p->y, x3 = p->x, b;
  fq2_sqr(&y2); fq2_sqr(&x3); fq2_mul(&x3, &p->x); fq2_b_g2(&b); fq2_add(&x3, &b);
  return fq2_eq(&y2, &x3);
}
static void g1_generator(g1_aff *g) {
  static const u64 X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                           0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
  static const u64 Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                           0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
  fq_from_raw(&g->x, X);
  fq_from_raw(&g->y, Y);
  g->inf = 0;
}

/* ------------------------------------------------------------------------------------ */
/* encodings                                                                             */
/* ------------------------------------------------------------------------------------ */
static int fq_read_be(fq *out, const uint8_t *b, int mask) {
  u64 c[6];
  for (int i = 0; i < 6; i++) {
    u64 v = 0;
    for (int k = 0; k < 8; k++) v = (v << 8) | b[(5 - i) * 8 + k];
    c[i] = v;
  }
  if (mask) c[5] &= 0x1fffffffffffffffull;
  if (big_geq(c, FQ_MOD, 6)) return 0;
  fq_from_raw(out, c);
  return 1;
}
static void fq_write_be(const fq *a, uint8_t *b) {
  u64 c[6];
  fq_to_raw(a, c);
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < 8; k++) b[(5 - i) * 8 + k] = (uint8_t)(c[i] >> (56 - 8 * k));
}
static int g1_read(g1_aff *p, const uint8_t *b) {
  if (b[0] & 0x80) return 0;
  if (b[0] & 0x40) {
    if (b[0] & 0x3f) return 0;
    for (int i = 1; i < 96; i++) if (b[i]) return 0;
    fq_zero(&p->x); fq_one(&p->y); p->inf = 1;
    return 1;
  }
  if (b[0] & 0x20) return 0;
  p->inf = 0;
  if (!fq_read_be(&p->x, b, 1) || !fq_read_be(&p->y, b + 48, 0)) return 0;
  return g1_on_curve(p);
}
static int g2_read(g2_aff *p, const uint8_t *b) {
  if (b[0] & 0x80) return 0;
  if (b[0] & 0x40) {
    if (b[0] & 0x3f) return 0;
    for (int i = 1; i < 192; i++) if (b[i]) return 0;
    fq2_zero(&p->x); fq2_one(&p->y); p->inf = 1;
    return 1;
  }
  if (b[0] & 0x20) return 0;
  p->inf = 0;
  if (!fq_read_be(&p->x.c1, b, 1) || !fq_read_be(&p->x.c0, b + 48, 0) || !fq_read_be(&p->y.c1, b + 96, 0) ||
      !fq_read_be(&p->y.c0, b + 144, 0))
    return 0;
  return g2_on_curve(p);
}
static void g1_write(const g1_aff *p, uint8_t *b) {
  if (p->inf) { memset(b, 0, 96); b[0] = 0x40; return; }
  fq_write_be(&p->x, b);
  fq_write_be(&p->y, b + 48);
}
static void g2_write(const g2_aff *p, uint8_t *b) {
  if (p->inf) { memset(b, 0, 192); b[0] = 0x40; return; }
  fq_write_be(&p->x.c1, b);
  fq_write_be(&p->x.c0, b + 48);
  fq_write_be(&p->y.c1, b + 96);
  fq_write_be(&p->y.c0, b + 144);
}
static void g1_write_compressed(const g1_aff *p, uint8_t *b) {
  if (p->inf) { memset(b, 0, 48); b[0] = 0xc0; return; }
  fq_write_be(&p->x, b);
  fq negy = p->y;
  fq_neg(&negy);
  b[0] |= 0x80;
  if (fq_cmp(&p->y, &negy) > 0) b[0] |= 0x20;
}
static void g2_write_compressed(const g2_aff *p, uint8_t *b) {
  if (p->inf) { memset(b, 0, 96); b[0] = 0xc0; return; }
  fq_write_be(&p->x.c1, b);
  fq_write_be(&p->x.c0, b + 48);
  fq2 negy = p->y;
  fq2_neg(&negy);
  b[0] |= 0x80;
  if (fq2_cmp(&p->y, &negy) > 0) b[0] |= 0x20;
}
static int fr_read_le(u64 *canon, const uint8_t *b) {
  for (int i = 0; i < 4; i++) {
    u64 v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | b[i * 8 + k];
    canon[i] = v;
  }
  return !big_geq(canon, FR_MOD, 4);
}

/* ------------------------------------------------------------------------------------ */
/* pairing (pairing 0.16 bls12_381/mod.rs shape)                                         */
/* ------------------------------------------------------------------------------------ */
#define BLS_X 0xd201000000010000ull
typedef struct { fq2 c0, c1, c2; } coeffs_t;
typedef struct { coeffs_t c[68]; int infinity; } g2_prepared;

static void doubling_step(coeffs_t *o, g2_jac *r) {
  fq2 tmp0 = r->x, tmp1 = r->y, tmp2, tmp3, tmp4, tmp5, tmp6, zsq = r->z;
  fq2_sqr(&tmp0); fq2_sqr(&tmp1); tmp2 = tmp1; fq2_sqr(&tmp2);
  tmp3 = tmp1; fq2_add(&tmp3, &r->x); fq2_sqr(&tmp3); fq2_sub(&tmp3, &tmp0); fq2_sub(&tmp3, &tmp2); fq2_dbl(&tmp3);
  tmp4 = tmp0; fq2_dbl(&tmp4); fq2_add(&tmp4, &tmp0);
  tmp6 = r->x; fq2_add(&tmp6, &tmp4);
  tmp5 = tmp4; fq2_sqr(&tmp5);
  fq2_sqr(&zsq);
  r->x = tmp5; fq2_sub(&r->x, &tmp3); fq2_sub(&r->x, &tmp3);
  fq2_add(&r->z, &r->y); fq2_sqr(&r->z); fq2_sub(&r->z, &tmp1); fq2_sub(&r->z, &zsq);
  r->y = tmp3; fq2_sub(&r->y, &r->x); fq2_mul(&r->y, &tmp4);
  fq2_dbl(&tmp2); fq2_dbl(&tmp2); fq2_dbl(&tmp2);
  fq2_sub(&r->y, &tmp2);
  tmp3 = tmp4; fq2_mul(&tmp3, &zsq); fq2_dbl(&tmp3); fq2_neg(&tmp3);
  fq2_sqr(&tmp6); fq2_sub(&tmp6, &tmp0); fq2_sub(&tmp6, &tmp5);
  fq2_dbl(&tmp1); fq2_dbl(&tmp1);
  fq2_sub(&tmp6, &tmp1);
  tmp0 = r->z; fq2_mul(&tmp0, &zsq); fq2_dbl(&tmp0);
  o->c0 = tmp0; o->c1 = tmp3; o->c2 = tmp6;
}
static void addition_step(coeffs_t *o, g2_jac *r, const g2_aff *q) {
  fq2 zsq = r->z, ysq = q->y, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, ztsq;
  fq2_sqr(&zsq); fq2_sqr(&ysq);
  t0 = zsq; fq2_mul(&t0, &q->x);
  t1 = q->y; fq2_add(&t1, &r->z); fq2_sqr(&t1); fq2_sub(&t1, &ysq); fq2_sub(&t1, &zsq); fq2_mul(&t1, &zsq);
  t2 = t0; fq2_sub(&t2, &r->x);
  t3 = t2; fq2_sqr(&t3);
  t4 = t3; fq2_dbl(&t4); fq2_dbl(&t4);
  t5 = t4; fq2_mul(&t5, &t2);
  t6 = t1; fq2_sub(&t6, &r->y); fq2_sub(&t6, &r->y);
  t9 = t6; fq2_mul(&t9, &q->x);
  t7 = t4; fq2_mul(&t7, &r->x);
  r->x = t6; fq2_sqr(&r->x); fq2_sub(&r->x, &t5); fq2_sub(&r->x, &t7); fq2_sub(&r->x, &t7);
  fq2_add(&r->z, &t2); fq2_sqr(&r->z); fq2_sub(&r->z, &zsq); fq2_sub(&r->z, &t3);
  t10 = q->y; fq2_add(&t10, &r->z);
  t8 = t7; fq2_sub(&t8, &r->x); fq2_mul(&t8, &t6);
  t0 = r->y; fq2_mul(&t0, &t5); fq2_dbl(&t0);
  r->y = t8; fq2_sub(&r->y, &t0);
  fq2_sqr(&t10); fq2_sub(&t10, &ysq);
  ztsq = r->z; fq2_sqr(&ztsq);
  fq2_sub(&t10, &ztsq);
  fq2_dbl(&t9); fq2_sub(&t9, &t10);
  t10 = r->z; fq2_dbl(&t10);
  fq2_neg(&t6);
  t1 = t6; fq2_dbl(&t1);
  o->c0 = t10; o->c1 = t1; o->c2 = t9;
}
static void g2_prepare(g2_prepared *out, const g2_aff *q) {
  if (q->inf) { out->infinity = 1; return; }
  out->infinity = 0;
  g2_jac r;
  r.x = q->x; r.y = q->y; fq2_one(&r.z);
  int n = 0, found = 0;
  u64 xs = BLS_X >> 1;
  for (int i = 63; i >= 0; i--) {
    int bit = (xs >> i) & 1;
    if (!found) { found = bit; continue; }
    doubling_step(&out->c[n++], &r);
    if (bit) addition_step(&out->c[n++], &r, q);
  }
  doubling_step(&out->c[n++], &r);
}
static void ell(fq12 *f, const coeffs_t *c, const g1_aff *p) {
  fq2 c0 = c->c0, c1 = c->c1;
  fq_mul(&c0.c0, &p->y); fq_mul(&c0.c1, &p->y);
  fq_mul(&c1.c0, &p->x); fq_mul(&c1.c1, &p->x);
  fq12_mul_by_014(f, &c->c2, &c1, &c0);
}
static void miller_loop(fq12 *f, const g1_aff *ps, const g2_prepared *qs, int n) {
  fq12_one(f);
  int idx = 0, found = 0;
  u64 xs = BLS_X >> 1;
  for (int i = 63; i >= 0; i--) {
    int bit = (xs >> i) & 1;
    if (!found) { found = bit; continue; }
    for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].infinity) ell(f, &qs[k].c[idx], &ps[k]);
    idx++;
    if (bit) {
      for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].infinity) ell(f, &qs[k].c[idx], &ps[k]);
      idx++;
    }
    fq12_sqr(f);
  }
  for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].infinity) ell(f, &qs[k].c[idx], &ps[k]);
  fq12_conjugate(f); /* BLS_X_IS_NEGATIVE */
}
static void exp_by_x(fq12 *f, u64 x) {
  fq12_pow_u64(f, x);
  fq12_conjugate(f);
}
static int final_exponentiation(fq12 *out, const fq12 *rin) {
  fq12 f1 = *rin, f2 = *rin;
  fq12_conjugate(&f1);
  if (!fq12_inv(&f2)) return 0;
  fq12 r = f1;
  fq12_mul(&r, &f2);
  f2 = r;
  fq12_frobenius(&r, 2);
  fq12_mul(&r, &f2);
  u64 x = BLS_X;
  fq12 y0 = r, y1, y2, y3;
  fq12_sqr(&y0);
  y1 = y0; exp_by_x(&y1, x);
  x >>= 1;
  y2 = y1; exp_by_x(&y2, x);
  x <<= 1;
  y3 = r; fq12_conjugate(&y3);
  fq12_mul(&y1, &y3);
  fq12_conjugate(&y1);
  fq12_mul(&y1, &y2);
  y2 = y1; exp_by_x(&y2, x);
  y3 = y2; exp_by_x(&y3, x);
  fq12_conjugate(&y1);
  fq12_mul(&y3, &y1);
  fq12_conjugate(&y1);
  fq12_frobenius(&y1, 3);
  fq12_frobenius(&y2, 2);
  fq12_mul(&y1, &y2);
  y2 = y3; exp_by_x(&y2, x);
  fq12_mul(&y2, &y0);
  fq12_mul(&y2, &r);
  fq12_mul(&y1, &y2);
  y2 = y3; fq12_frobenius(&y2, 1);
  fq12_mul(&y1, &y2);
  *out = y1;
  return 1;
}
/* Engine::pairing(p, q) */
static void pairing(fq12 *out, const g1_aff *p, const g2_aff *q) {
  static g2_prepared prep; /* not re-entrant across threads: callers below use their own */
  g2_prepared *pp = (g2_prepared *)malloc(sizeof(g2_prepared));
  (void)prep;
  g2_prepare(pp, q);
  fq12 f;
  miller_loop(&f, p, pp, 1);
  final_exponentiation(out, &f);
  free(pp);
}

/* ------------------------------------------------------------------------------------ */
/* SHA3-256 (table/loop form) and ChaCha20                                               */
/* ------------------------------------------------------------------------------------ */
static const u64 KRC[24] = {0x1ull, 0x8082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x808bull, 0x80000001ull,
                            0x8000000080008081ull, 0x8000000000008009ull, 0x8aull, 0x88ull, 0x80008009ull, 0x8000000aull,
                            0x8000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
                            0x8000000000008002ull, 0x8000000000000080ull, 0x800aull, 0x800000008000000aull,
                            0x8000000080008081ull, 0x8000000000008080ull, 0x80000001ull, 0x8000000080008008ull};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static void keccakf(u64 st[25]) {
  for (int round = 0; round < 24; round++) {
    u64 bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      u64 t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    u64 t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = KPIL[i];
      u64 b = st[j];
      st[j] = (t << KROT[i]) | (t >> (64 - KROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[round];
  }
}
static void sha3_256(const uint8_t *in, size_t len, uint8_t out[32]) {
  u64 st[25] = {0};
  uint8_t *sb = (uint8_t *)st; /* little-endian host assumed (x86-64) */
  size_t pos = 0;
  for (size_t i = 0; i < len; i++) {
    sb[pos++] ^= in[i];
    if (pos == 136) { keccakf(st); pos = 0; }
  }
  sb[pos] ^= 0x06;
  sb[135] ^= 0x80;
  keccakf(st);
  memcpy(out, sb, 32);
}

typedef struct { uint32_t key[8]; u64 counter; uint32_t buf[16]; int idx; } chacha_rng;
#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
static void chacha_block(chacha_rng *g) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; i++) s[4 + i] = g->key[i];
  s[12] = (uint32_t)g->counter; s[13] = (uint32_t)(g->counter >> 32); s[14] = 0; s[15] = 0;
  uint32_t w[16];
  memcpy(w, s, 64);
  static const int QR[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15},
                               {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
  for (int r = 0; r < 10; r++)
    for (int q = 0; q < 8; q++) {
      uint32_t *a = &w[QR[q][0]], *b = &w[QR[q][1]], *c = &w[QR[q][2]], *d = &w[QR[q][3]];
      *a += *b; *d ^= *a; *d = ROTL32(*d, 16);
      *c += *d; *b ^= *c; *b = ROTL32(*b, 12);
      *a += *b; *d ^= *a; *d = ROTL32(*d, 8);
      *c += *d; *b ^= *c; *b = ROTL32(*b, 7);
    }
  for (int i = 0; i < 16; i++) g->buf[i] = w[i] + s[i];
  g->counter++;
  g->idx = 0;
}
static void chacha_seed(chacha_rng *g, const uint8_t seed[32]) {
  for (int i = 0; i < 8; i++)
    g->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                ((uint32_t)seed[4 * i + 3] << 24);
  g->counter = 0;
  g->idx = 16;
}
static uint32_t chacha_u32(chacha_rng *g) {
  if (g->idx >= 16) chacha_block(g);
  return g->buf[g->idx++];
}
/* H-spec alternatives (the implementation-defined sampling order behind hash_g2 / xor_with_hash that no
 * vector of the real crate pins yet).  0 = the recalled behaviour of rand_chacha 0.2 / ff_derive 0.6 / pairing 0.16; each
 * bit switches ONE item to its documented alternative, the same bits as TC_HSPEC in threshold_crypto_amd/csrc/tc_hash.h and
 * HSPEC in oracle/tc_oracle.py.  tests/ref_fixtures.py diagnose() names the setting that reproduces reference vectors. */
#define HSPEC_U64_HI_FIRST 1      /* next_u64 = high word then low word */
#define HSPEC_COMPARE_THEN_MASK 2 /* Fq::random accepts iff the UNMASKED draw is below q */
#define HSPEC_GREATEST_MSB 4      /* greatest = top bit of next_u32 (rand's bool sampling) instead of next_u32 % 2 */
#define HSPEC_KEYSTREAM_BYTES 8   /* xor_with_hash uses consecutive keystream bytes (fill_bytes), not one word per byte */
#define HSPEC_CANONICAL_DRAW 16   /* the accepted pattern is the canonical value, not the Montgomery representation */
static int g_hspec = 0;
__attribute__((visibility("default"))) void or_set_hspec(int v) { g_hspec = v; }
__attribute__((visibility("default"))) int or_get_hspec(void) { return g_hspec; }

static u64 chacha_u64(chacha_rng *g) {
  u64 a = chacha_u32(g);
  u64 b = chacha_u32(g);
  return (g_hspec & HSPEC_U64_HI_FIRST) ? (b | (a << 32)) : (a | (b << 32));
}

/* ff_derive random(): raw limbs ARE the Montgomery representation */
static void fq_random(fq *out, chacha_rng *g) {
  for (;;) {
    for (int i = 0; i < 6; i++) out->l[i] = chacha_u64(g);
    if (!(g_hspec & HSPEC_COMPARE_THEN_MASK)) out->l[5] &= 0xffffffffffffffffull >> 3;
    if (big_geq(out->l, FQ_MOD, 6)) continue;
    if (g_hspec & HSPEC_CANONICAL_DRAW) {  /* value = the pattern itself: Montgomery form = pattern * R */
      fq r2;
      memcpy(r2.l, FQ_R2, sizeof r2.l);
      fq_mul(out, &r2);
    }
    return;
  }
}
static const u64 G2_COFACTOR[8] = {0xcf1c38e31c7238e5ull, 0x1616ec6e786f0c70ull, 0x21537e293a6691aeull,
                                   0xa628f1cb4d9e82efull, 0xa68a205b2e5a7ddfull, 0xcd91de4547085abaull,
                                   0x091d50792876a202ull, 0x05d543a95414e7f1ull};
/* G2::random(rng) */
static void g2_random(g2_jac *out, chacha_rng *g) {
  for (;;) {
    fq2 x;
    fq_random(&x.c0, g);
    fq_random(&x.c1, g);
    const uint32_t gw = chacha_u32(g);
    int greatest = (g_hspec & HSPEC_GREATEST_MSB) ? (int)(gw >> 31) : (gw % 2) != 0;
    fq2 x3b = x, b, y;
    fq2_sqr(&x3b); fq2_mul(&x3b, &x); fq2_b_g2(&b); fq2_add(&x3b, &b);
    if (!fq2_sqrt(&y, &x3b)) continue;
    fq2 negy = y;
    fq2_neg(&negy);
    g2_aff p;
    p.x = x;
    p.y = ((fq2_cmp(&y, &negy) < 0) ^ greatest) ? y : negy;
    p.inf = 0;
    g2_mul_bits(out, &p, G2_COFACTOR, 8);
    if (!g2_jac_is_zero(out)) return;
  }
}

/* ------------------------------------------------------------------------------------ */
/* exported entry points (ctypes)                                                        */
/* ------------------------------------------------------------------------------------ */
#define EXPORT __attribute__((visibility("default")))

EXPORT u64 or_fq_mul_count(void) { return g_fq_mul_count; }
EXPORT void or_fq_mul_count_reset(void) { g_fq_mul_count = 0; }

/* hash_g2 (src/lib.rs:691-694) -> uncompressed affine */
static void hash_g2_point(g2_jac *out, const uint8_t *msg, size_t len) {
  uint8_t digest[32];
  sha3_256(msg, len, digest);
  chacha_rng g;
  chacha_seed(&g, digest);
  g2_random(out, &g);
}
EXPORT void or_hash_g2(const uint8_t *msg, size_t len, uint8_t *out192) {
  tc_init();
  g2_jac h;
  g2_aff a;
  hash_g2_point(&h, msg, len);
  g2_into_affine(&a, &h);
  g2_write(&a, out192);
}
/* hash_g1_g2 (src/lib.rs:697-707) */
static void hash_g1_g2_point(g2_jac *out, const g1_aff *g1, const uint8_t *msg, size_t len) {
  uint8_t buf[64 + 48];
  size_t n;
  if (len > 64) { sha3_256(msg, len, buf); n = 32; } else { memcpy(buf, msg, len); n = len; }
  g1_write_compressed(g1, buf + n);
  hash_g2_point(out, buf, n + 48);
}
EXPORT int or_hash_g1_g2(const uint8_t *g1, const uint8_t *msg, size_t len, uint8_t *out192) {
  tc_init();
  g1_aff p;
  if (!g1_read(&p, g1)) return 3;
  g2_jac h;
  g2_aff a;
  hash_g1_g2_point(&h, &p, msg, len);
  g2_into_affine(&a, &h);
  g2_write(&a, out192);
  return 0;
}
/* xor_with_hash (src/lib.rs:710-715) */
static void xor_with_hash(const g1_aff *g, const uint8_t *data, size_t len, uint8_t *out) {
  uint8_t comp[48], digest[32];
  g1_write_compressed(g, comp);
  sha3_256(comp, 48, digest);
  chacha_rng rng;
  chacha_seed(&rng, digest);
  if (g_hspec & HSPEC_KEYSTREAM_BYTES) {
    uint32_t w = 0;
    for (size_t i = 0; i < len; i++) {
      if ((i & 3) == 0) w = chacha_u32(&rng);
      out[i] = data[i] ^ (uint8_t)(w >> (8 * (i & 3)));
    }
    return;
  }
  for (size_t i = 0; i < len; i++) out[i] = data[i] ^ (uint8_t)chacha_u32(&rng);
}
EXPORT int or_xor_with_hash(const uint8_t *g1, const uint8_t *data, size_t len, uint8_t *out) {
  tc_init();
  g1_aff p;
  if (!g1_read(&p, g1)) return 3;
  xor_with_hash(&p, data, len, out);
  return 0;
}
/* SecretKey::sign_g2 (src/lib.rs:372-374): affine.mul(fr) */
EXPORT int or_g2_mul(const uint8_t *fr32, const uint8_t *pt192, uint8_t *out192) {
  tc_init();
  u64 k[4];
  g2_aff p, a;
  if (!fr_read_le(k, fr32) || !g2_read(&p, pt192)) return 3;
  g2_jac r;
  g2_mul_bits(&r, &p, k, 4);
  g2_into_affine(&a, &r);
  g2_write(&a, out192);
  return 0;
}
EXPORT int or_g1_mul(const uint8_t *fr32, const uint8_t *pt96, uint8_t *out96) {
  tc_init();
  u64 k[4];
  g1_aff p, a;
  if (!fr_read_le(k, fr32) || !g1_read(&p, pt96)) return 3;
  g1_jac r;
  g1_mul_bits(&r, &p, k, 4);
  g1_into_affine(&a, &r);
  g1_write(&a, out96);
  return 0;
}
/* SecretKey::sign (src/lib.rs:379-381) */
EXPORT int or_sign(const uint8_t *fr32, const uint8_t *msg, size_t len, uint8_t *out192) {
  tc_init();
  u64 k[4];
  if (!fr_read_le(k, fr32)) return 3;
  g2_jac h, r;
  g2_aff ha, a;
  hash_g2_point(&h, msg, len);
  g2_into_affine(&ha, &h);
  g2_mul_bits(&r, &ha, k, 4);
  g2_into_affine(&a, &r);
  g2_write(&a, out192);
  return 0;
}

/* interpolate (src/lib.rs:719-767): n samples supplied, first t+1 taken */
static int lagrange_l0(fr *l0s, const u64 *idx, size_t t) {
  size_t n = t + 1;
  fr *x = (fr *)malloc(sizeof(fr) * n);
  fr *x_prod = (fr *)malloc(sizeof(fr) * n);
  fr one, tmp;
  fr_one(&one);
  for (size_t i = 0; i < n; i++) { fr_from_u64(&x[i], idx[i]); fr_add(&x[i], &one); } /* into_fr_plus_1 */
  tmp = one;
  x_prod[0] = tmp;
  for (size_t i = 0; i < t; i++) { fr_mul(&tmp, &x[i]); x_prod[i + 1] = tmp; }
  tmp = one;
  for (size_t i = t; i-- > 0;) { fr_mul(&tmp, &x[i + 1]); fr_mul(&x_prod[i], &tmp); }
  int rc = 0;
  for (size_t i = 0; i < n; i++) {
    fr denom = one;
    for (size_t j = 0; j < n; j++) {
      if (!fr_eq(&x[j], &x[i])) {
        fr diff = x[j];
        fr_sub(&diff, &x[i]);
        fr_mul(&denom, &diff);
      }
    }
    if (!fr_inv(&denom)) { rc = 2; break; } /* DuplicateEntry */
    l0s[i] = x_prod[i];
    fr_mul(&l0s[i], &denom);
  }
  free(x);
  free(x_prod);
  return rc;
}
EXPORT int or_lagrange(size_t t, const u64 *idx, uint8_t *out /* (t+1) x 32 LE */) {
  fr *l0 = (fr *)malloc(sizeof(fr) * (t + 1));
  int rc = lagrange_l0(l0, idx, t);
  if (!rc)
    for (size_t i = 0; i <= t; i++) {
      u64 c[4];
      fr_to_raw(&l0[i], c);
      memcpy(out + 32 * i, c, 32);
    }
  free(l0);
  return rc;
}
EXPORT int or_combine_g2(size_t t, size_t n, const u64 *idx, const uint8_t *shares, uint8_t *out192) {
  tc_init();
  if (n <= t) return 1; /* NotEnoughShares */
  g2_aff p, a;
  if (t == 0) {
    if (!g2_read(&p, shares)) return 3;
    g2_write(&p, out192);
    return 0;
  }
  fr *l0 = (fr *)malloc(sizeof(fr) * (t + 1));
  int rc = lagrange_l0(l0, idx, t);
  g2_jac result;
  g2_jac_zero(&result);
  for (size_t i = 0; i <= t && !rc; i++) {
    if (!g2_read(&p, shares + 192 * i)) { rc = 3; break; }
    u64 k[4];
    fr_to_raw(&l0[i], k);
    g2_jac term;
    g2_mul_bits(&term, &p, k, 4); /* sample.into_affine().mul(l0) */
    g2_add(&result, &term);
  }
  free(l0);
  if (rc) return rc;
  g2_into_affine(&a, &result);
  g2_write(&a, out192);
  return 0;
}
static int combine_g1_point(g1_aff *a, size_t t, size_t n, const u64 *idx, const uint8_t *shares) {
  if (n <= t) return 1;
  g1_aff p;
  if (t == 0) return g1_read(a, shares) ? 0 : 3;
  fr *l0 = (fr *)malloc(sizeof(fr) * (t + 1));
  int rc = lagrange_l0(l0, idx, t);
  g1_jac result;
  g1_jac_zero(&result);
  for (size_t i = 0; i <= t && !rc; i++) {
    if (!g1_read(&p, shares + 96 * i)) { rc = 3; break; }
    u64 k[4];
    fr_to_raw(&l0[i], k);
    g1_jac term;
    g1_mul_bits(&term, &p, k, 4);
    g1_add(&result, &term);
  }
  free(l0);
  if (rc) return rc;
  g1_into_affine(a, &result);
  return 0;
}
EXPORT int or_combine_g1(size_t t, size_t n, const u64 *idx, const uint8_t *shares, uint8_t *out96) {
  tc_init();
  g1_aff a;
  int rc = combine_g1_point(&a, t, n, idx, shares);
  if (rc) return rc;
  g1_write(&a, out96);
  return 0;
}
/* PublicKeySet::decrypt (src/lib.rs:618-626) */
EXPORT int or_threshold_decrypt(size_t t, size_t n, const u64 *idx, const uint8_t *shares, const uint8_t *v, size_t len,
                                uint8_t *out) {
  tc_init();
  g1_aff a;
  int rc = combine_g1_point(&a, t, n, idx, shares);
  if (rc) return rc;
  xor_with_hash(&a, v, len, out);
  return 0;
}

/* e(a,b) == e(c,d) exactly as the reference: two full pairings, compare Fq12 */
static int pairing_eq(const g1_aff *a, const g2_aff *b, const g1_aff *c, const g2_aff *d) {
  fq12 l, r;
  pairing(&l, a, b);
  pairing(&r, c, d);
  return fq12_eq(&l, &r);
}
EXPORT int or_pairing_check(const uint8_t *a, const uint8_t *b, const uint8_t *c, const uint8_t *d) {
  tc_init();
  g1_aff pa, pc;
  g2_aff qb, qd;
  if (!g1_read(&pa, a) || !g2_read(&qb, b) || !g1_read(&pc, c) || !g2_read(&qd, d)) return 0;
  return pairing_eq(&pa, &qb, &pc, &qd);
}
/* GT = pairing(a, b) as 12 x 48 B big-endian (tower order) -- debugging aid */
EXPORT int or_pairing_gt(const uint8_t *a, const uint8_t *b, uint8_t *out576) {
  tc_init();
  g1_aff pa;
  g2_aff qb;
  if (!g1_read(&pa, a) || !g2_read(&qb, b)) return -1;
  fq12 f;
  pairing(&f, &pa, &qb);
  const fq *e = (const fq *)&f;
  for (int i = 0; i < 12; i++) fq_write_be(&e[i], out576 + 48 * i);
  return 0;
}
/* PublicKey::verify_g2 (src/lib.rs:108-110) */
EXPORT int or_verify_g2(const uint8_t *pk, const uint8_t *sig, const uint8_t *hash) {
  tc_init();
  g1_aff p, g;
  g2_aff s, h;
  if (!g1_read(&p, pk) || !g2_read(&s, sig) || !g2_read(&h, hash)) return 0;
  g1_generator(&g);
  return pairing_eq(&p, &h, &g, &s);
}
/* PublicKey::verify (src/lib.rs:115-117) */
EXPORT int or_verify(const uint8_t *pk, const uint8_t *sig, const uint8_t *msg, size_t len) {
  tc_init();
  g1_aff p, g;
  g2_aff s, h;
  if (!g1_read(&p, pk) || !g2_read(&s, sig)) return 0;
  g2_jac hj;
  hash_g2_point(&hj, msg, len);
  g2_into_affine(&h, &hj);
  g1_generator(&g);
  return pairing_eq(&p, &h, &g, &s);
}
/* Ciphertext::verify (src/lib.rs:508-512) */
EXPORT int or_ciphertext_verify(const uint8_t *u, const uint8_t *v, size_t len, const uint8_t *w) {
  tc_init();
  g1_aff pu, g;
  g2_aff qw, h;
  if (!g1_read(&pu, u) || !g2_read(&qw, w)) return 0;
  g2_jac hj;
  hash_g1_g2_point(&hj, &pu, v, len);
  g2_into_affine(&h, &hj);
  g1_generator(&g);
  return pairing_eq(&g, &qw, &pu, &h);
}
/* PublicKeyShare::verify_decryption_share (src/lib.rs:182-186) */
EXPORT int or_verify_decryption_share(const uint8_t *pk_share, const uint8_t *share, const uint8_t *u, const uint8_t *v,
                                      size_t len, const uint8_t *w) {
  tc_init();
  g1_aff pk, sh, pu;
  g2_aff qw, h;
  if (!g1_read(&pk, pk_share) || !g1_read(&sh, share) || !g1_read(&pu, u) || !g2_read(&qw, w)) return 0;
  g2_jac hj;
  hash_g1_g2_point(&hj, &pu, v, len);
  g2_into_affine(&h, &hj);
  return pairing_eq(&sh, &h, &pk, &qw);
}
EXPORT int or_g1_compress(const uint8_t *in96, uint8_t *out48) {
  tc_init();
  g1_aff p;
  if (!g1_read(&p, in96)) return 3;
  g1_write_compressed(&p, out48);
  return 0;
}
EXPORT int or_g2_compress(const uint8_t *in192, uint8_t *out96) {
  tc_init();
  g2_aff p;
  if (!g2_read(&p, in192)) return 3;
  g2_write_compressed(&p, out96);
  return 0;
}
/* ---- multi-threaded batch drivers for the timed CPU baseline (bench.py cpu_baseline) ---------- */
#include <pthread.h>
typedef struct {
  int kind; /* 0 = combine_g2, 1 = verify_g2, 2 = ciphertext_verify, 3 = threshold_decrypt (fixed-length v) */
  size_t t, n, lo, hi, len;
  const u64 *idx;
  const uint8_t *a, *b, *c;
  uint8_t *out;
  int *rc;
} batch_job;
static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  for (size_t k = j->lo; k < j->hi; k++) {
    if (j->kind == 0)
      j->rc[k] = or_combine_g2(j->t, j->n, j->idx + k * j->n, j->a + k * j->n * 192, j->out + k * 192);
    else if (j->kind == 1)
      j->rc[k] = or_verify_g2(j->a, j->b + k * 192, j->c + k * 192);
    else if (j->kind == 2)
      j->rc[k] = or_ciphertext_verify(j->a + k * 96, j->b + k * j->len, j->len, j->c + k * 192);
    else
      j->rc[k] = or_threshold_decrypt(j->t, j->n, j->idx + k * j->n, j->a + k * j->n * 96, j->b + k * j->len, j->len,
                                      j->out + k * j->len);
  }
  return NULL;
}
static void run_batch(batch_job proto, size_t B, int nthreads) {
  tc_init();
  if (nthreads < 1) nthreads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  batch_job *jobs = (batch_job *)malloc(sizeof(batch_job) * (size_t)nthreads);
  for (int i = 0; i < nthreads; i++) {
    jobs[i] = proto;
    jobs[i].lo = B * (size_t)i / (size_t)nthreads;
    jobs[i].hi = B * (size_t)(i + 1) / (size_t)nthreads;
    pthread_create(&th[i], NULL, batch_worker, &jobs[i]);
  }
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
}
/* B combine_signatures jobs (n samples each), nthreads host threads; rc[k] per job */
EXPORT void or_combine_g2_batch(size_t t, size_t n, const u64 *idx, const uint8_t *shares, size_t B, uint8_t *out, int *rc,
                                int nthreads) {
  batch_job p = {0, t, n, 0, 0, 0, idx, shares, NULL, NULL, out, rc};
  run_batch(p, B, nthreads);
}
/* B verify_g2 checks under one public key; rc[k] = 1 if valid */
EXPORT void or_verify_g2_batch(const uint8_t *pk, const uint8_t *sigs, const uint8_t *hashes, size_t B, int *rc, int nthreads) {
  batch_job p = {1, 0, 0, 0, 0, 0, NULL, pk, sigs, hashes, NULL, rc};
  run_batch(p, B, nthreads);
}
/* B Ciphertext::verify checks (src/lib.rs:508-512), every v of the same length len; rc[k] = 1 if valid */
EXPORT void or_ciphertext_verify_batch(const uint8_t *u, const uint8_t *v, size_t len, const uint8_t *w, size_t B, int *rc,
                                       int nthreads) {
  batch_job p = {2, 0, 0, 0, 0, len, NULL, u, v, w, NULL, rc};
  run_batch(p, B, nthreads);
}
/* B PublicKeySet::decrypt jobs (src/lib.rs:618-626): n decryption shares each, every v of length len */
EXPORT void or_threshold_decrypt_batch(size_t t, size_t n, const u64 *idx, const uint8_t *shares, const uint8_t *v,
                                       size_t len, size_t B, uint8_t *out, int *rc, int nthreads) {
  batch_job p = {3, t, n, 0, 0, len, idx, shares, v, NULL, out, rc};
  run_batch(p, B, nthreads);
}

/* ---- threaded drivers for the large-threshold tests (round 4): more jobs per test than a Python loop over the single-job
 * functions can afford.  Each job is the composition of the single-job functions above, nothing else. ---- */
typedef struct {
  int kind; /* 0 = sign the n shares of a job and combine them, 1 = combine_g1, 2 = the n signature shares of a message */
  size_t t, n, N, lo, hi;
  const u64 *idx;
  const uint8_t *a, *b;
  uint8_t *out;
  int *rc;
} big_job;
static void *big_worker(void *arg) {
  big_job *j = (big_job *)arg;
  uint8_t *sh = (uint8_t *)malloc(j->n * 192);
  for (size_t k = j->lo; k < j->hi; k++) {
    if (j->kind == 1) {
      j->rc[k] = or_combine_g1(j->t, j->n, j->idx + k * j->n, j->a + k * j->n * 96, j->out + k * 96);
      continue;
    }
    /* SecretKeyShare::sign_g2 (src/lib.rs:442-444) for the signers idx[k][0..n) of message k: a = the N x 32 B table of
     * secret key shares, b = the hash point of every message */
    int rc = 0;
    uint8_t *dst = j->kind == 2 ? j->out + k * j->n * 192 : sh;
    for (size_t s = 0; s < j->n && !rc; s++) {
      const u64 who = j->idx[k * j->n + s];
      rc = who < j->N ? or_g2_mul(j->a + who * 32, j->b + k * 192, dst + s * 192) : 3;
    }
    if (!rc && j->kind == 0) rc = or_combine_g2(j->t, j->n, j->idx + k * j->n, sh, j->out + k * 192);   /* src/lib.rs:608-615 */
    j->rc[k] = rc;
  }
  free(sh);
  return NULL;
}
static void run_big(big_job proto, size_t B, int nthreads) {
  tc_init();
  if (nthreads < 1) nthreads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  big_job *jobs = (big_job *)malloc(sizeof(big_job) * (size_t)nthreads);
  for (int i = 0; i < nthreads; i++) {
    jobs[i] = proto;
    jobs[i].lo = B * (size_t)i / (size_t)nthreads;
    jobs[i].hi = B * (size_t)(i + 1) / (size_t)nthreads;
    pthread_create(&th[i], NULL, big_worker, &jobs[i]);
  }
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
}
/* B threshold signatures from scratch: job k signs hash[k] with the key shares sk_table[idx[k][*]] and combines the n shares */
EXPORT void or_sign_combine_batch(size_t t, size_t n, size_t N, const uint8_t *sk_table, const u64 *idx, const uint8_t *hashes, size_t B,
                                  uint8_t *out, int *rc, int nthreads) {
  big_job p = {0, t, n, N, 0, 0, idx, sk_table, hashes, out, rc};
  run_big(p, B, nthreads);
}
/* B combinations in G1 (PublicKeySet::decrypt's interpolate, src/lib.rs:618-626) */
EXPORT void or_combine_g1_batch(size_t t, size_t n, const u64 *idx, const uint8_t *shares, size_t B, uint8_t *out, int *rc, int nthreads) {
  big_job p = {1, t, n, 0, 0, 0, idx, shares, NULL, out, rc};
  run_big(p, B, nthreads);
}
/* the n signature shares of each of B messages: out[k][s] = [sk_table[idx[k][s]]] hash[k]; an index >= N fails the message */
EXPORT void or_sign_shares_batch(size_t n, size_t N, const uint8_t *sk_table, const u64 *idx, const uint8_t *hashes, size_t B, uint8_t *out,
                                 int *rc, int nthreads) {
  big_job p = {2, 0, n, N, 0, 0, idx, sk_table, hashes, out, rc};
  run_big(p, B, nthreads);
}

/* checked decode of the compressed forms: EncodedPoint::into_affine (on-curve + [r]P == 0), the
 * check behind PublicKey::from_bytes (src/lib.rs:140-146) / Signature::from_bytes (:246-252) */
static const u64 Q_P1_D4[6] = {0xee7fbfffffffeaabull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                               0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
static void fq_pow(fq *a, const u64 *e, int nlimbs) {
  fq res;
  fq_one(&res);
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    fq_sqr(&res);
    if ((e[i / 64] >> (i % 64)) & 1) fq_mul(&res, a);
  }
  *a = res;
}
EXPORT int or_g1_decompress(const uint8_t *in48, uint8_t *out96) {
  tc_init();
  g1_aff p;
  if (!(in48[0] & 0x80)) return 3;
  if (in48[0] & 0x40) {
    if (in48[0] & 0x3f) return 3;
    for (int i = 1; i < 48; i++) if (in48[i]) return 3;
    p.inf = 1; fq_zero(&p.x); fq_one(&p.y);
    g1_write(&p, out96);
    return 0;
  }
  int greatest = (in48[0] & 0x20) != 0;
  if (!fq_read_be(&p.x, in48, 1)) return 3;
  fq rhs = p.x, b, y, negy;
  fq_sqr(&rhs); fq_mul(&rhs, &p.x); fq_b_g1(&b); fq_add(&rhs, &b);
  y = rhs;
  fq_pow(&y, Q_P1_D4, 6);
  fq y2 = y;
  fq_sqr(&y2);
  if (!fq_eq(&y2, &rhs)) return 3;
  negy = y; fq_neg(&negy);
  p.y = ((fq_cmp(&y, &negy) > 0) == greatest) ? y : negy;
  p.inf = 0;
  g1_jac t;
  g1_mul_bits(&t, &p, FR_MOD, 4);
  if (!g1_jac_is_zero(&t)) return 3;
  g1_write(&p, out96);
  return 0;
}
EXPORT int or_g2_decompress(const uint8_t *in96, uint8_t *out192) {
  tc_init();
  g2_aff p;
  if (!(in96[0] & 0x80)) return 3;
  if (in96[0] & 0x40) {
    if (in96[0] & 0x3f) return 3;
    for (int i = 1; i < 96; i++) if (in96[i]) return 3;
    p.inf = 1; fq2_zero(&p.x); fq2_one(&p.y);
    g2_write(&p, out192);
    return 0;
  }
  int greatest = (in96[0] & 0x20) != 0;
  if (!fq_read_be(&p.x.c1, in96, 1) || !fq_read_be(&p.x.c0, in96 + 48, 0)) return 3;
  fq2 rhs = p.x, b, y, negy;
  fq2_sqr(&rhs); fq2_mul(&rhs, &p.x); fq2_b_g2(&b); fq2_add(&rhs, &b);
  if (!fq2_sqrt(&y, &rhs)) return 3;
  negy = y; fq2_neg(&negy);
  p.y = ((fq2_cmp(&y, &negy) > 0) == greatest) ? y : negy;
  p.inf = 0;
  g2_jac t;
  g2_mul_bits(&t, &p, FR_MOD, 4);
  if (!g2_jac_is_zero(&t)) return 3;
  g2_write(&p, out192);
  return 0;
}

/* Wire-level PublicKeySet::combine_signatures (what a caller of the reference does with bytes off the network): each of the
 * first t+1 shares through Signature::from_bytes (src/lib.rs:246-252: checked decode above), interpolate (src/lib.rs:719-767),
 * Signature::to_bytes (src/lib.rs:255-259).  rc: 0, 1 = NotEnoughShares, 3 = FromBytesError::Invalid (out = the identity). */
EXPORT int or_combine_signatures_wire(size_t t, size_t n, const u64 *idx, const uint8_t *shares96, uint8_t *out96) {
  tc_init();
  memset(out96, 0, 96);
  out96[0] = 0xc0;
  if (n <= t) return 1;
  uint8_t *dec = (uint8_t *)malloc((t + 1) * 192);
  int rc = 0;
  for (size_t k = 0; k <= t && !rc; k++) rc = or_g2_decompress(shares96 + k * 96, dec + k * 192);
  uint8_t sig[192];
  if (!rc) rc = or_combine_g2(t, t + 1, idx, dec, sig);
  if (!rc) rc = or_g2_compress(sig, out96);
  free(dec);
  return rc;
}
/* Wire-level PublicKeySet::decrypt: the decryption shares in their 48-byte compressed form (src/serde_impl.rs:174-218) */
EXPORT int or_decrypt_wire(size_t t, size_t n, const u64 *idx, const uint8_t *shares48, const uint8_t *v, size_t len, uint8_t *out) {
  tc_init();
  memset(out, 0, len);
  if (n <= t) return 1;
  uint8_t *dec = (uint8_t *)malloc((t + 1) * 96);
  int rc = 0;
  for (size_t k = 0; k <= t && !rc; k++) rc = or_g1_decompress(shares48 + k * 48, dec + k * 96);
  if (!rc) rc = or_threshold_decrypt(t, t + 1, idx, dec, v, len, out);
  free(dec);
  return rc;
}
typedef struct {
  size_t t, n, lo, hi;
  const u64 *idx;
  const uint8_t *shares;
  uint8_t *out;
  int *rc;
} wire_job;
static void *wire_worker(void *arg) {
  wire_job *j = (wire_job *)arg;
  for (size_t k = j->lo; k < j->hi; k++)
    j->rc[k] = or_combine_signatures_wire(j->t, j->n, j->idx + k * j->n, j->shares + k * j->n * 96, j->out + k * 96);
  return NULL;
}
EXPORT void or_combine_signatures_wire_batch(size_t t, size_t n, const u64 *idx, const uint8_t *shares96, size_t B, uint8_t *out96, int *rc,
                                             int nthreads) {
  tc_init();
  if (nthreads < 1) nthreads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  wire_job *jobs = (wire_job *)malloc(sizeof(wire_job) * (size_t)nthreads);
  for (int i = 0; i < nthreads; i++) {
    wire_job w = {t, n, B * (size_t)i / (size_t)nthreads, B * (size_t)(i + 1) / (size_t)nthreads, idx, shares96, out96, rc};
    jobs[i] = w;
    pthread_create(&th[i], NULL, wire_worker, &jobs[i]);
  }
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
}

EXPORT void or_sha3_256(const uint8_t *msg, size_t len, uint8_t *out32) { sha3_256(msg, len, out32); }
