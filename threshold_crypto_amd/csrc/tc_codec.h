// Zcash BLS12-381 wire encodings <-> Montgomery limbs (device side of the C-ABI I/O).
//   G1 uncompressed 96 B  = x || y                big-endian, flags in byte 0 bits 7..5
//   G2 uncompressed 192 B = x.c1 || x.c0 || y.c1 || y.c0
//   G1/G2 compressed 48/96 B = x (G2: c1 || c0) with bit7 = 1, bit6 = infinity, bit5 = y is
//                              the lexicographically larger root
//   Fr 32 B little-endian canonical (4 x u64 LE limbs, /root/reference/src/serde_impl.rs:296)
// These are the formats behind to_bytes/from_bytes (/root/reference/src/lib.rs:140-153,
// 246-259) and into_uncompressed (:89,163,224,238,276).
#pragma once
#include "tc_curve.h"

namespace tc {

enum : uint8_t {
  TC_JOB_OK = 0,
  TC_JOB_NOT_ENOUGH_SHARES = 1,  // Error::NotEnoughShares  (/root/reference/src/error.rs:9)
  TC_JOB_DUPLICATE_ENTRY = 2,    // Error::DuplicateEntry   (/root/reference/src/error.rs:12)
  TC_JOB_INVALID_ENCODING = 3,   // FromBytesError::Invalid (/root/reference/src/error.rs:39)
};

TC_HD uint32_t load_be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
TC_HD void store_be32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}

// 48 big-endian bytes -> Fq; returns false if the integer is >= q.  mask_top strips the
// three flag bits of byte 0.
TC_HD bool fq_from_be48(const uint8_t* b, bool mask_top, Fq& out) {
  uint32_t l[12];
  TC_UNROLL for (int i = 0; i < 12; i++) l[i] = load_be32(b + 44 - 4 * i);
  if (mask_top) l[11] &= 0x1fffffffu;
  bool ok = limbs_lt_p<FqParams>(l);
  out = Fq::from_canonical(l);
  return ok;
}

TC_HD void fq_to_be48(const Fq& a, uint8_t* b) {
  uint32_t l[12];
  a.to_canonical(l);
  TC_UNROLL for (int i = 0; i < 12; i++) store_be32(b + 44 - 4 * i, l[i]);
}

// 96 big-endian bytes c1 || c0 <-> Fq2.  In the lane-pair build each lane moves its own 48 bytes
// (odd lane: c1, the first half) and the range check is reduced over the pair.
TC_HD bool fq2_from_be96(const uint8_t* b, bool mask_top, Fq2& out) {
#if TC_PAIR
  const bool odd = Fq2::odd();
  Fq m;
  const bool ok = fq_from_be48(b + (odd ? 0 : 48), mask_top && odd, m);
  out = Fq2{m};
  return pair_all(ok);
#else
  Fq re, im;
  bool ok = fq_from_be48(b, mask_top, im);
  ok &= fq_from_be48(b + 48, false, re);
  out = Fq2::make(re, im);
  return ok;
#endif
}
TC_HD void fq2_to_be96(const Fq2& a, uint8_t* b) {
#if TC_PAIR
  fq_to_be48(a.m, b + (Fq2::odd() ? 0 : 48));
#else
  fq_to_be48(a.c1, b);
  fq_to_be48(a.c0, b + 48);
#endif
}
// the lane that wrote byte 0 of a G2 encoding (and therefore ORs the flag bits into it)
TC_HD bool g2_owns_byte0() {
#if TC_PAIR
  return Fq2::odd();
#else
  return true;
#endif
}
// zero nchunks x 96 bytes, each lane of a pair its own halves
TC_HD void g2_zero_bytes(uint8_t* b, int nchunks) {
#if TC_PAIR
  const int o = Fq2::odd() ? 0 : 48;
  for (int c = 0; c < nchunks; c++)
    for (int i = 0; i < 48; i++) b[96 * c + o + i] = 0;
#else
  for (int i = 0; i < 96 * nchunks; i++) b[i] = 0;
#endif
}

// Fr <- 32 LE bytes; returns false if >= r
TC_HD bool fr_from_le32(const uint8_t* b, uint32_t* limbs) {
  TC_UNROLL for (int i = 0; i < 8; i++)
    limbs[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) |
               ((uint32_t)b[4 * i + 3] << 24);
  return limbs_lt_p<FrParams>(limbs);
}

// --- uncompressed affine decode (parse + on-curve; subgroup membership is NOT checked here,
// like EncodedPoint::into_affine_unchecked + is_on_curve) -----------------------------------
TC_HD bool g1_decode_uncompressed(const uint8_t* b, G1Affine& p) {
  const uint8_t f = b[0];
  if (f & 0x80) return false;
  if (f & 0x40) {
    uint32_t o = f & 0x3f;
    for (int i = 1; i < 96; i++) o |= b[i];
    p = G1Affine::infinity();
    return o == 0;
  }
  if (f & 0x20) return false;
  bool ok = fq_from_be48(b, true, p.x);
  ok &= fq_from_be48(b + 48, false, p.y);
  p.inf = false;
  return ok && affine_on_curve(p, g1_b());
}

TC_HD bool g2_decode_uncompressed(const uint8_t* b, G2Affine& p) {
  const uint8_t f = b[0];
  if (f & 0x80) return false;
  if (f & 0x40) {
    uint32_t o = f & 0x3f;
    for (int i = 1; i < 192; i++) o |= b[i];
    p = G2Affine::infinity();
    return o == 0;
  }
  if (f & 0x20) return false;
  bool ok = fq2_from_be96(b, true, p.x);
  ok &= fq2_from_be96(b + 96, false, p.y);
  p.inf = false;
  return ok && affine_on_curve(p, g2_b());
}

TC_HD void g1_encode_uncompressed(const G1Affine& p, uint8_t* b) {
  if (p.inf) {
    for (int i = 0; i < 96; i++) b[i] = 0;
    b[0] = 0x40;
    return;
  }
  fq_to_be48(p.x, b);
  fq_to_be48(p.y, b + 48);
}

TC_HD void g2_encode_uncompressed(const G2Affine& p, uint8_t* b) {
  if (p.inf) {
    g2_zero_bytes(b, 2);
    if (g2_owns_byte0()) b[0] = 0x40;
    return;
  }
  fq2_to_be96(p.x, b);
  fq2_to_be96(p.y, b + 96);
}

// y lexicographically larger than -y ?   (Fq: y > (q-1)/2 ; Fq2: compare c1 first, then c0)
TC_HD bool fq_lex_largest(const Fq& y) {
  uint32_t l[12];
  y.to_canonical(l);
  return fq_canonical_gt_half(l);
}
TC_HD bool fq2_lex_largest(const Fq2& y) {
  const Fq im = y.im();
  if (!im.is_zero()) return fq_lex_largest(im);
  return fq_lex_largest(y.re());
}

TC_HD void g1_encode_compressed(const G1Affine& p, uint8_t* b) {
  if (p.inf) {
    for (int i = 0; i < 48; i++) b[i] = 0;
    b[0] = 0xc0;
    return;
  }
  fq_to_be48(p.x, b);
  b[0] |= 0x80;
  if (fq_lex_largest(p.y)) b[0] |= 0x20;
}

TC_HD void g2_encode_compressed(const G2Affine& p, uint8_t* b) {
  if (p.inf) {
    g2_zero_bytes(b, 1);
    if (g2_owns_byte0()) b[0] = 0xc0;
    return;
  }
  fq2_to_be96(p.x, b);
  const bool largest = fq2_lex_largest(p.y);
  if (g2_owns_byte0()) b[0] |= largest ? 0xa0 : 0x80;
}

}  // namespace tc
