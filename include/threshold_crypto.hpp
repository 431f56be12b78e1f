// threshold_crypto.hpp -- C++17 host-side mirror of the threshold_crypto 0.4.0 public API for the
// accelerated path, header-only, on top of the C ABI of libtc_amd.so (include/tc_amd.h).
//
// The reference is a Rust crate; its toolchain is absent from the build image, so the host side
// above the C ABI is written in C++ (and, for the test-suite, Python: threshold_crypto_amd/api.py).
// Type and method names, argument meaning and error behaviour follow the reference (file:line
// cited per item, relative to the reference repository).  Every method exists in the reference's
// single-item form and in a *_batch form that hands a whole batch to one kernel launch.
// All group / pairing / hash arithmetic happens on the GPU; there is no CPU fallback (constructing
// an Engine without an MI355X throws).
//
// Values are held in the reference's canonical encodings: G1 96 B, G2 192 B uncompressed
// (into_affine().into_uncompressed()), Fr 32 B little-endian.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "tc_amd.h"

namespace threshold_crypto {

constexpr std::size_t PK_SIZE = 48;   // src/lib.rs:71
constexpr std::size_t SIG_SIZE = 96;  // src/lib.rs:75

using Bytes = std::vector<std::uint8_t>;
using G1Bytes = std::array<std::uint8_t, 96>;
using G2Bytes = std::array<std::uint8_t, 192>;
using FrBytes = std::array<std::uint8_t, 32>;

// An index type beyond u64 -- `T: IntoFr` of combine_signatures / decrypt (src/lib.rs:608-622) for T = Fr, i32, i64
// (src/into_fr.rs:10-14, 28-56): the field element as 32 little-endian bytes, ordered by its canonical value like the derived
// Ord of pairing's Fr, so a std::map keyed by it iterates as the reference's BTreeMap<Fr, _> does.
// CAVEAT: that is the order of BTreeMap<Fr, _> ONLY.  A caller whose reference code keys the map by i32 / i64
// iterates NEGATIVE keys first (BTreeMap<i64, _> orders by the signed value), while their field images r - |x| sort after every
// positive key here; with more than t + 1 shares `take(t + 1)` (src/lib.rs:727-730) would then pick a different subset.  For
// signed keys use the std::map<std::int64_t, _> overloads of combine_signatures AND decrypt below: they iterate by the signed value
// and convert afterwards.
struct FrIndex {
  FrBytes le{};
  static FrIndex from_u64(std::uint64_t x) {
    FrIndex f;
    for (int i = 0; i < 8; i++) f.le[i] = (std::uint8_t)(x >> (8 * i));
    return f;
  }
  // IntoFr for i64: a negative value is -(|x|) mod r
  static FrIndex from_i64(std::int64_t x) {
    if (x >= 0) return from_u64((std::uint64_t)x);
    static const std::uint8_t R[32] = {0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
                                       0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};
    const std::uint64_t m = (std::uint64_t)0 - (std::uint64_t)x;  // |x|
    FrIndex f;
    int borrow = 0;
    for (int i = 0; i < 32; i++) {
      const int sub = (i < 8 ? (int)((m >> (8 * i)) & 0xff) : 0) + borrow;
      const int d = (int)R[i] - sub;
      f.le[i] = (std::uint8_t)(d & 0xff);
      borrow = d < 0 ? 1 : 0;
    }
    return f;
  }
  bool operator<(const FrIndex& o) const {
    for (int i = 31; i >= 0; i--)
      if (le[i] != o.le[i]) return le[i] < o.le[i];
    return false;
  }
  bool operator==(const FrIndex& o) const { return le == o.le; }
};

// threshold_crypto::error::Error (src/error.rs:7-17)
enum class Error { NotEnoughShares = 1, DuplicateEntry = 2 };
struct ErrorException : std::runtime_error {
  Error code;
  explicit ErrorException(Error e)
      : std::runtime_error(e == Error::NotEnoughShares ? "Not enough shares for interpolation"
                                                       : "Samples for interpolation contain a duplicate entry"),
        code(e) {}
};
// FromBytesError::Invalid (src/error.rs:37-41)
struct FromBytesError : std::runtime_error {
  FromBytesError() : std::runtime_error("Invalid representation.") {}
};
struct GpuError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// One libtc_amd context = one MI355X + one HIP stream.
class Engine {
 public:
  explicit Engine(int device = 0) {
    if (tc_ctx_create(&ctx_, device) != TC_OK) throw GpuError("tc_ctx_create failed: no gfx950 HIP device (no CPU fallback)");
  }
  ~Engine() { tc_ctx_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  tc_ctx* ctx() const { return ctx_; }
  void check(int rc) const {
    if (rc != TC_OK) throw GpuError(std::string("libtc_amd: ") + tc_last_error(ctx_));
  }
  static Engine& instance() {
    static Engine e(0);
    return e;
  }

 private:
  tc_ctx* ctx_ = nullptr;
};

inline void raise_status(std::uint8_t st) {
  if (st == TC_JOB_NOT_ENOUGH_SHARES) throw ErrorException(Error::NotEnoughShares);
  if (st == TC_JOB_DUPLICATE_ENTRY) throw ErrorException(Error::DuplicateEntry);
  if (st == TC_JOB_INVALID_ENCODING) throw FromBytesError();
}

struct Messages {  // packed variable-length byte strings
  Bytes flat;
  std::vector<std::uint64_t> off{0};
  void push(const void* p, std::size_t n) {
    const auto* b = static_cast<const std::uint8_t*>(p);
    flat.insert(flat.end(), b, b + n);
    off.push_back(flat.size());
  }
  void push(const std::string& s) { push(s.data(), s.size()); }
  void push(const Bytes& s) { push(s.data(), s.size()); }
  std::size_t size() const { return off.size() - 1; }
  const std::uint8_t* data() const { return flat.empty() ? reinterpret_cast<const std::uint8_t*>("") : flat.data(); }
};

// pub fn hash_g2 (src/lib.rs:691-694)
inline std::vector<G2Bytes> hash_g2_batch(const Messages& m, Engine& e = Engine::instance()) {
  std::vector<G2Bytes> out(m.size());
  if (!out.empty()) e.check(tc_hash_g2_batch(e.ctx(), m.data(), m.off.data(), m.size(), out[0].data()));
  return out;
}
inline G2Bytes hash_g2(const std::string& msg) {
  Messages m;
  m.push(msg);
  return hash_g2_batch(m)[0];
}

// struct Signature(G2) (src/lib.rs:202)
struct Signature {
  G2Bytes g2{};
  bool operator==(const Signature& o) const { return g2 == o.g2; }
  bool operator!=(const Signature& o) const { return !(*this == o); }
  // Signature::parity (src/lib.rs:237-243)
  bool parity() const {
    std::uint8_t x = 0;
    for (auto b : g2) x ^= b;
    return __builtin_popcount(x) % 2 != 0;
  }
  // Signature::to_bytes (src/lib.rs:255-259)
  std::array<std::uint8_t, SIG_SIZE> to_bytes(Engine& e = Engine::instance()) const {
    std::array<std::uint8_t, SIG_SIZE> out{};
    std::uint8_t st = 0;
    e.check(tc_g2_compress_batch(e.ctx(), g2.data(), 1, out.data(), &st));
    raise_status(st);
    return out;
  }
  // Signature::from_bytes (src/lib.rs:246-252): checked decode
  static Signature from_bytes(const std::array<std::uint8_t, SIG_SIZE>& b, Engine& e = Engine::instance()) {
    Signature s;
    std::uint8_t st = 0;
    e.check(tc_g2_decompress_batch(e.ctx(), b.data(), 1, s.g2.data(), &st));
    if (st) throw FromBytesError();
    return s;
  }
};
// struct SignatureShare(pub Signature) (src/lib.rs:266)
struct SignatureShare {
  Signature sig;
  bool operator==(const SignatureShare& o) const { return sig == o.sig; }
};
// struct DecryptionShare(G1) (src/lib.rs:517)
struct DecryptionShare {
  G1Bytes g1{};
};
// struct Ciphertext(G1, Vec<u8>, G2) (src/lib.rs:473-478)
struct Ciphertext {
  G1Bytes u{};
  Bytes v;
  G2Bytes w{};
  // Ciphertext::verify (src/lib.rs:508-512)
  bool verify(Engine& e = Engine::instance()) const {
    std::uint64_t off[2] = {0, v.size()};
    std::uint8_t ok = 0;
    const std::uint8_t dummy = 0;
    e.check(tc_ciphertext_verify_batch(e.ctx(), u.data(), v.empty() ? &dummy : v.data(), off, w.data(), 1, &ok));
    return ok != 0;
  }
};

// struct PublicKey(G1) (src/lib.rs:79)
struct PublicKey {
  G1Bytes g1{};
  bool operator==(const PublicKey& o) const { return g1 == o.g1; }
  // PublicKey::verify_g2 (src/lib.rs:108-110)
  bool verify_g2(const Signature& sig, const G2Bytes& hash, Engine& e = Engine::instance()) const {
    std::uint8_t ok = 0;
    e.check(tc_verify_g2_batch(e.ctx(), g1.data(), 0, sig.g2.data(), hash.data(), 1, &ok));
    return ok != 0;
  }
  // PublicKey::verify (src/lib.rs:115-117)
  bool verify(const Signature& sig, const std::string& msg, Engine& e = Engine::instance()) const {
    Messages m;
    m.push(msg);
    return verify_batch({sig}, m, e)[0];
  }
  // one key, many (signature, message) pairs
  std::vector<bool> verify_batch(const std::vector<Signature>& sigs, const Messages& msgs, Engine& e = Engine::instance()) const {
    std::vector<std::uint8_t> ok(sigs.size()), flat(sigs.size() * 192);
    for (std::size_t i = 0; i < sigs.size(); i++) std::memcpy(&flat[i * 192], sigs[i].g2.data(), 192);
    if (!sigs.empty())
      e.check(tc_verify_sig_batch(e.ctx(), g1.data(), 0, flat.data(), msgs.data(), msgs.off.data(), sigs.size(), ok.data()));
    return std::vector<bool>(ok.begin(), ok.end());
  }
  // PublicKey::to_bytes (src/lib.rs:149-153) / from_bytes (:140-146)
  std::array<std::uint8_t, PK_SIZE> to_bytes(Engine& e = Engine::instance()) const {
    std::array<std::uint8_t, PK_SIZE> out{};
    std::uint8_t st = 0;
    e.check(tc_g1_compress_batch(e.ctx(), g1.data(), 1, out.data(), &st));
    raise_status(st);
    return out;
  }
  static PublicKey from_bytes(const std::array<std::uint8_t, PK_SIZE>& b, Engine& e = Engine::instance()) {
    PublicKey p;
    std::uint8_t st = 0;
    e.check(tc_g1_decompress_batch(e.ctx(), b.data(), 1, p.g1.data(), &st));
    if (st) throw FromBytesError();
    return p;
  }
};

// struct PublicKeyShare(PublicKey) (src/lib.rs:159)
struct PublicKeyShare {
  PublicKey pk;
  // PublicKeyShare::verify (src/lib.rs:177-179)
  bool verify(const SignatureShare& s, const std::string& msg, Engine& e = Engine::instance()) const {
    return pk.verify(s.sig, msg, e);
  }
  // PublicKeyShare::verify_decryption_share (src/lib.rs:182-186)
  bool verify_decryption_share(const DecryptionShare& share, const Ciphertext& ct, Engine& e = Engine::instance()) const {
    std::uint64_t off[2] = {0, ct.v.size()};
    std::uint8_t ok = 0;
    const std::uint8_t dummy = 0;
    e.check(tc_verify_decryption_share_batch(e.ctx(), pk.g1.data(), 0, share.g1.data(), ct.u.data(),
                                             ct.v.empty() ? &dummy : ct.v.data(), off, ct.w.data(), 1, &ok));
    return ok != 0;
  }
};

inline const G1Bytes& g1_generator() {
  static const G1Bytes g = [] {
    const char* hex =
        "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
    G1Bytes o{};
    auto nib = [](char c) { return (std::uint8_t)(c <= '9' ? c - '0' : c - 'a' + 10); };
    for (int i = 0; i < 96; i++) o[i] = (std::uint8_t)((nib(hex[2 * i]) << 4) | nib(hex[2 * i + 1]));
    return o;
  }();
  return g;
}

// struct SecretKey(Box<Fr>) (src/lib.rs:302); the scalar is kept as 32 LE bytes and zeroed on drop
class SecretKey {
 public:
  explicit SecretKey(const FrBytes& fr) : fr_(fr) {}
  ~SecretKey() { volatile std::uint8_t* p = fr_.data(); for (int i = 0; i < 32; i++) p[i] = 0; }  // src/lib.rs:304-314
  const FrBytes& fr() const { return fr_; }
  // SecretKey::public_key (src/lib.rs:367-369)
  PublicKey public_key(Engine& e = Engine::instance()) const {
    PublicKey pk;
    std::uint8_t st = 0;
    e.check(tc_g1_mul_batch(e.ctx(), fr_.data(), g1_generator().data(), 1, 1, pk.g1.data(), &st));
    raise_status(st);
    return pk;
  }
  // SecretKey::sign_g2 (src/lib.rs:372-374)
  Signature sign_g2(const G2Bytes& hash, Engine& e = Engine::instance()) const {
    Signature s;
    std::uint8_t st = 0;
    e.check(tc_g2_mul_batch(e.ctx(), fr_.data(), hash.data(), 1, 1, s.g2.data(), &st));
    raise_status(st);
    return s;
  }
  // SecretKey::sign (src/lib.rs:379-381); batch: one key, many messages
  Signature sign(const std::string& msg, Engine& e = Engine::instance()) const {
    Messages m;
    m.push(msg);
    return sign_batch(m, e)[0];
  }
  std::vector<Signature> sign_batch(const Messages& m, Engine& e = Engine::instance()) const {
    std::vector<Signature> out(m.size());
    std::vector<std::uint8_t> flat(m.size() * 192), st(m.size());
    if (!out.empty()) e.check(tc_sign_batch(e.ctx(), fr_.data(), m.data(), m.off.data(), 1, m.size(), flat.data(), st.data()));
    for (std::size_t i = 0; i < out.size(); i++) std::memcpy(out[i].g2.data(), &flat[i * 192], 192);
    return out;
  }
  // SecretKey::decrypt (src/lib.rs:384-391)
  std::optional<Bytes> decrypt(const Ciphertext& ct, Engine& e = Engine::instance()) const {
    // one call: Ciphertext::verify, [sk] u and xor_with_hash on the device
    Bytes out(ct.v.size());
    std::uint64_t off[2] = {0, ct.v.size()};
    std::uint8_t ok = 0, dummy = 0;
    e.check(tc_secret_key_decrypt_batch(e.ctx(), fr_.data(), ct.u.data(), ct.v.empty() ? &dummy : ct.v.data(), off, ct.w.data(), 1,
                                        out.empty() ? &dummy : out.data(), &ok));
    if (!ok) return std::nullopt;
    return out;
  }

 private:
  FrBytes fr_;
};

// struct SecretKeyShare(SecretKey) (src/lib.rs:408)
class SecretKeyShare {
 public:
  explicit SecretKeyShare(const FrBytes& fr) : sk_(fr) {}
  const SecretKey& key() const { return sk_; }
  PublicKeyShare public_key_share(Engine& e = Engine::instance()) const { return PublicKeyShare{sk_.public_key(e)}; }  // :437-439
  SignatureShare sign_g2(const G2Bytes& h, Engine& e = Engine::instance()) const { return SignatureShare{sk_.sign_g2(h, e)}; }  // :442-444
  SignatureShare sign(const std::string& msg, Engine& e = Engine::instance()) const { return SignatureShare{sk_.sign(msg, e)}; }  // :447-449
  // SecretKeyShare::decrypt_share_no_verify (src/lib.rs:460-462)
  DecryptionShare decrypt_share_no_verify(const Ciphertext& ct, Engine& e = Engine::instance()) const {
    DecryptionShare d;
    std::uint8_t st = 0;
    e.check(tc_g1_mul_batch(e.ctx(), sk_.fr().data(), ct.u.data(), 1, 1, d.g1.data(), &st));
    raise_status(st);
    return d;
  }
  // SecretKeyShare::decrypt_share (src/lib.rs:452-457)
  std::optional<DecryptionShare> decrypt_share(const Ciphertext& ct, Engine& e = Engine::instance()) const {
    // one call: Ciphertext::verify and the multiplication on the device; an invalid ciphertext never yields [sk] u
    DecryptionShare d;
    std::uint64_t off[2] = {0, ct.v.size()};
    std::uint8_t ok = 0, dummy = 0;
    e.check(tc_decrypt_share_batch(e.ctx(), sk_.fr().data(), ct.u.data(), ct.v.empty() ? &dummy : ct.v.data(), off, ct.w.data(), 1, d.g1.data(), &ok));
    if (!ok) return std::nullopt;
    return d;
  }

 private:
  SecretKey sk_;
};

// S signers x B messages in one launch: out[j][s] = shares[s].sign(msgs[j])
inline std::vector<std::vector<SignatureShare>> sign_shares_batch(const std::vector<const SecretKeyShare*>& shares,
                                                                  const Messages& m, Engine& e = Engine::instance()) {
  const std::size_t S = shares.size(), B = m.size();
  std::vector<std::uint8_t> fr(S * 32), flat(S * B * 192), st(S * B);
  for (std::size_t s = 0; s < S; s++) std::memcpy(&fr[s * 32], shares[s]->key().fr().data(), 32);
  if (S && B) e.check(tc_sign_batch(e.ctx(), fr.data(), m.data(), m.off.data(), S, B, flat.data(), st.data()));
  volatile std::uint8_t* z = fr.data();
  for (std::size_t i = 0; i < fr.size(); i++) z[i] = 0;
  std::vector<std::vector<SignatureShare>> out(B, std::vector<SignatureShare>(S));
  for (std::size_t j = 0; j < B; j++)
    for (std::size_t s = 0; s < S; s++) std::memcpy(out[j][s].sig.g2.data(), &flat[(j * S + s) * 192], 192);
  return out;
}

// struct PublicKeySet { commit: Commitment } (src/lib.rs:539-543): commit = G1 coefficients
class PublicKeySet {
 public:
  explicit PublicKeySet(std::vector<G1Bytes> commit) : commit_(std::move(commit)) {}
  // PublicKeySet::threshold (src/lib.rs:560-562)
  std::size_t threshold() const { return commit_.size() - 1; }
  // PublicKeySet::public_key (src/lib.rs:565-567)
  PublicKey public_key() const { return PublicKey{commit_[0]}; }
  // PublicKeySet::public_key_share (src/lib.rs:570-573) = Commitment::evaluate(i + 1) (src/poly.rs:497-508):
  // sum_k (i+1)^k commit[k]; the powers are computed by the caller-side helper below for small i
  PublicKeyShare public_key_share(std::uint64_t i, Engine& e = Engine::instance()) const {
    const std::size_t n = commit_.size();
    std::vector<std::uint8_t> scal(n * 32, 0), pts(n * 96);
    // powers of x = i + 1 mod r as little-endian 256-bit integers (schoolbook, x < 2^64 + 1)
    unsigned __int128 carry;
    std::array<std::uint64_t, 5> p = {1, 0, 0, 0, 0};
    const std::uint64_t xlo = i + 1;
    const bool xhi = (i == UINT64_MAX);  // i + 1 == 2^64
    for (std::size_t k = 0; k < n; k++) {
      reduce_mod_r(p);
      std::memcpy(&scal[k * 32], p.data(), 32);
      std::memcpy(&pts[k * 96], commit_[k].data(), 96);
      std::array<std::uint64_t, 5> q = {0, 0, 0, 0, 0};
      carry = 0;
      for (int w = 0; w < 4; w++) {
        unsigned __int128 t = (unsigned __int128)p[w] * xlo + (std::uint64_t)carry;
        q[w] = (std::uint64_t)t;
        carry = t >> 64;
      }
      q[4] = (std::uint64_t)carry;
      if (xhi) {
        q = {0, p[0], p[1], p[2], p[3]};
      }
      p = q;
    }
    PublicKeyShare out;
    std::uint8_t st = 0;
    e.check(tc_g1_lincomb_batch(e.ctx(), n, scal.data(), pts.data(), 1, out.pk.g1.data(), &st));
    raise_status(st);
    return out;
  }

  // PublicKeySet::combine_signatures (src/lib.rs:608-615).  A std::map iterates in ascending index
  // order exactly like the BTreeMap the reference's callers pass.
  Signature combine_signatures(const std::map<std::uint64_t, SignatureShare>& shares, Engine& e = Engine::instance()) const {
    std::vector<std::uint8_t> st;
    auto out = combine_signatures_batch({shares}, st, e);
    raise_status(st[0]);
    return out[0];
  }
  std::vector<Signature> combine_signatures_batch(const std::vector<std::map<std::uint64_t, SignatureShare>>& jobs,
                                                  std::vector<std::uint8_t>& status, Engine& e = Engine::instance()) const {
    const std::size_t B = jobs.size();
    const std::size_t n = B ? jobs[0].size() : 0;
    std::vector<std::uint64_t> idx(B * n + 1);
    std::vector<std::uint8_t> sh(B * n * 192 + 1), flat(B * 192);
    for (std::size_t j = 0; j < B; j++) {
      if (jobs[j].size() != n) throw std::invalid_argument("all jobs of one batch must supply the same number of shares");
      std::size_t k = 0;
      for (const auto& kv : jobs[j]) {
        idx[j * n + k] = kv.first;
        std::memcpy(&sh[(j * n + k) * 192], kv.second.sig.g2.data(), 192);
        k++;
      }
    }
    status.assign(B, 0);
    if (B) e.check(tc_combine_g2_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), B, flat.data(), status.data()));
    std::vector<Signature> out(B);
    for (std::size_t j = 0; j < B; j++) std::memcpy(out[j].g2.data(), &flat[j * 192], 192);
    return out;
  }

  // the same with `T: IntoFr` keys beyond u64 (Fr values, negative i64: FrIndex above) -- tc_combine_g2_fr_batch
  Signature combine_signatures(const std::map<FrIndex, SignatureShare>& shares, Engine& e = Engine::instance()) const {
    const std::size_t n = shares.size();
    std::vector<std::uint8_t> idx(n * 32 + 1), sh(n * 192 + 1);
    std::size_t k = 0;
    for (const auto& kv : shares) {
      std::memcpy(&idx[k * 32], kv.first.le.data(), 32);
      std::memcpy(&sh[k * 192], kv.second.sig.g2.data(), 192);
      k++;
    }
    Signature out;
    std::uint8_t st = 0;
    e.check(tc_combine_g2_fr_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), 1, out.g2.data(), &st));
    raise_status(st);
    return out;
  }
  // `T = i64` (and i32): a BTreeMap<i64, _> iterates by the SIGNED value -- negative keys first -- and interpolate() takes the
  // first t + 1 samples in THAT order (src/lib.rs:727-730, src/into_fr.rs:42-56); the keys become field elements afterwards
  Signature combine_signatures(const std::map<std::int64_t, SignatureShare>& shares, Engine& e = Engine::instance()) const {
    const std::size_t n = shares.size();
    std::vector<std::uint8_t> idx(n * 32 + 1), sh(n * 192 + 1);
    std::size_t k = 0;
    for (const auto& kv : shares) {
      std::memcpy(&idx[k * 32], FrIndex::from_i64(kv.first).le.data(), 32);
      std::memcpy(&sh[k * 192], kv.second.sig.g2.data(), 192);
      k++;
    }
    Signature out;
    std::uint8_t st = 0;
    e.check(tc_combine_g2_fr_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), 1, out.g2.data(), &st));
    raise_status(st);
    return out;
  }
  // Wire-level combine_signatures: (index, SignatureShare::to_bytes) pairs in the order the reference's map would iterate; the
  // shares pass the checked decode of from_bytes (src/lib.rs:246-252) on the device, the result is Signature::to_bytes
  // (src/lib.rs:255-259).  status: 0, 1 = NotEnoughShares, 3 = FromBytesError::Invalid for a job that owns a bad share.
  std::vector<std::array<std::uint8_t, SIG_SIZE>> combine_signatures_wire_batch(
      const std::vector<std::vector<std::pair<std::uint64_t, std::array<std::uint8_t, SIG_SIZE>>>>& jobs, std::vector<std::uint8_t>& status,
      Engine& e = Engine::instance()) const {
    const std::size_t B = jobs.size();
    const std::size_t n = B ? jobs[0].size() : 0;
    std::vector<std::uint64_t> idx(B * n + 1);
    std::vector<std::uint8_t> sh(B * n * SIG_SIZE + 1), flat(B * SIG_SIZE + 1);
    for (std::size_t j = 0; j < B; j++) {
      if (jobs[j].size() != n) throw std::invalid_argument("all jobs of one batch must supply the same number of shares");
      for (std::size_t k = 0; k < n; k++) {
        idx[j * n + k] = jobs[j][k].first;
        std::memcpy(&sh[(j * n + k) * SIG_SIZE], jobs[j][k].second.data(), SIG_SIZE);
      }
    }
    status.assign(B, 0);
    if (B) e.check(tc_combine_signatures_wire_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), B, flat.data(), status.data()));
    std::vector<std::array<std::uint8_t, SIG_SIZE>> out(B);
    for (std::size_t j = 0; j < B; j++) std::memcpy(out[j].data(), &flat[j * SIG_SIZE], SIG_SIZE);
    return out;
  }

  // PublicKeySet::decrypt (src/lib.rs:618-626)
  Bytes decrypt(const std::map<std::uint64_t, DecryptionShare>& shares, const Ciphertext& ct, Engine& e = Engine::instance()) const {
    const std::size_t n = shares.size();
    std::vector<std::uint64_t> idx(n + 1);
    std::vector<std::uint8_t> sh(n * 96 + 1);
    std::size_t k = 0;
    for (const auto& kv : shares) {
      idx[k] = kv.first;
      std::memcpy(&sh[k * 96], kv.second.g1.data(), 96);
      k++;
    }
    Bytes out(ct.v.size() + 1);
    std::uint64_t off[2] = {0, ct.v.size()};
    std::uint8_t st = 0;
    const std::uint8_t dummy = 0;
    e.check(tc_decrypt_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), ct.v.empty() ? &dummy : ct.v.data(), off, 1,
                             out.data(), &st));
    raise_status(st);
    out.resize(ct.v.size());
    return out;
  }
  // the same with `T: IntoFr` keys beyond u64 -- tc_decrypt_fr_batch.  FrIndex keys iterate as BTreeMap<Fr, _> does; std::int64_t
  // keys by the SIGNED value (negative keys first, as BTreeMap<i64, _>: src/lib.rs:618-626 takes the first t + 1 in THAT order,
  // src/into_fr.rs:42-56) and become field elements afterwards (combine_signatures had the signed overload, decrypt not).
  Bytes decrypt(const std::map<FrIndex, DecryptionShare>& shares, const Ciphertext& ct, Engine& e = Engine::instance()) const {
    std::vector<std::pair<FrIndex, const DecryptionShare*>> ordered;
    for (const auto& kv : shares) ordered.emplace_back(kv.first, &kv.second);
    return decrypt_fr(ordered, ct, e);
  }
  Bytes decrypt(const std::map<std::int64_t, DecryptionShare>& shares, const Ciphertext& ct, Engine& e = Engine::instance()) const {
    std::vector<std::pair<FrIndex, const DecryptionShare*>> ordered;
    for (const auto& kv : shares) ordered.emplace_back(FrIndex::from_i64(kv.first), &kv.second);
    return decrypt_fr(ordered, ct, e);
  }

 private:
  Bytes decrypt_fr(const std::vector<std::pair<FrIndex, const DecryptionShare*>>& ordered, const Ciphertext& ct, Engine& e) const {
    const std::size_t n = ordered.size();
    std::vector<std::uint8_t> idx(n * 32 + 1), sh(n * 96 + 1);
    for (std::size_t k = 0; k < n; k++) {
      std::memcpy(&idx[k * 32], ordered[k].first.le.data(), 32);
      std::memcpy(&sh[k * 96], ordered[k].second->g1.data(), 96);
    }
    Bytes out(ct.v.size() + 1);
    std::uint64_t off[2] = {0, ct.v.size()};
    std::uint8_t st = 0;
    const std::uint8_t dummy = 0;
    e.check(tc_decrypt_fr_batch(e.ctx(), threshold(), n, idx.data(), sh.data(), ct.v.empty() ? &dummy : ct.v.data(), off, 1,
                                out.data(), &st));
    raise_status(st);
    out.resize(ct.v.size());
    return out;
  }
  static void reduce_mod_r(std::array<std::uint64_t, 5>& p) {
    // p < 2^320; subtract r * 2^k while possible (binary long division: r is 255 bits)
    static const std::uint64_t R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    for (int shift = 65; shift >= 0; shift--) {
      std::array<std::uint64_t, 6> m = {0, 0, 0, 0, 0, 0};
      for (int w = 0; w < 4; w++) {
        const int bit = shift + 64 * w;
        m[bit / 64] |= R[w] << (bit % 64);
        if (bit % 64) m[bit / 64 + 1] |= R[w] >> (64 - bit % 64);
      }
      if (m[5]) continue;
      bool ge = true;
      for (int w = 4; w >= 0; w--) {
        if (p[w] != m[w]) {
          ge = p[w] > m[w];
          break;
        }
      }
      if (ge) {
        unsigned __int128 borrow = 0;
        for (int w = 0; w < 5; w++) {
          unsigned __int128 t = (unsigned __int128)p[w] - m[w] - (std::uint64_t)borrow;
          p[w] = (std::uint64_t)t;
          borrow = (t >> 64) ? 1 : 0;
        }
      }
    }
  }
  std::vector<G1Bytes> commit_;
};

// ---- membership, the opt-in share validation, DKG algebra, several GPUs (round 2 of the C ABI) ---------------------
// Uncompressed bytes from an untrusted source are NOT yet values of the reference's types: from_bytes
// (src/lib.rs:140-146, 246-252) only admits points of the order-r subgroup, and the kernels rely on it.  Either
// decode through from_bytes, or test with these, or run the Engine in checked-input mode.
inline bool is_member(const G1Bytes& p, Engine& e = Engine::instance()) {
  std::uint8_t ok = 0;
  e.check(tc_g1_subgroup_check_batch(e.ctx(), p.data(), 1, &ok));
  return ok != 0;
}
inline bool is_member(const G2Bytes& p, Engine& e = Engine::instance()) {
  std::uint8_t ok = 0;
  e.check(tc_g2_subgroup_check_batch(e.ctx(), p.data(), 1, &ok));
  return ok != 0;
}
inline void set_input_checks(bool on, Engine& e = Engine::instance()) { e.check(tc_ctx_set_input_checks(e.ctx(), on ? 1 : 0)); }

// The share-validation loop of examples/threshold_sig.rs:115-131 -- ok[j][i] = pk_shares[i].verify(shares[j][i], msgs[j])
// (src/lib.rs:177-179) -- through one random linear combination per message (tc_verify_shares_rlc_batch; per-share checks
// only for messages whose combined check fails).  seed: 32 bytes of fresh secret randomness.
inline std::vector<std::vector<bool>> verify_shares_rlc_batch(const std::vector<PublicKeyShare>& pk_shares,
                                                              const std::vector<std::vector<SignatureShare>>& shares, const Messages& msgs,
                                                              const std::array<std::uint8_t, 32>& seed, std::uint64_t* n_fallback = nullptr,
                                                              Engine& e = Engine::instance()) {
  const std::size_t N = pk_shares.size(), B = shares.size();
  if (B != msgs.size()) throw std::invalid_argument("one share row per message");
  std::vector<std::uint8_t> pk(N * 96 + 1), sg(B * N * 192 + 1), ok(B * N + 1);
  for (std::size_t i = 0; i < N; i++) std::memcpy(&pk[i * 96], pk_shares[i].pk.g1.data(), 96);
  for (std::size_t j = 0; j < B; j++) {
    if (shares[j].size() != N) throw std::invalid_argument("every message needs one share per key");
    for (std::size_t i = 0; i < N; i++) std::memcpy(&sg[(j * N + i) * 192], shares[j][i].sig.g2.data(), 192);
  }
  std::uint64_t nfb = 0;
  if (B && N) e.check(tc_verify_shares_rlc_batch(e.ctx(), pk.data(), N, sg.data(), msgs.data(), msgs.off.data(), B, seed.data(), ok.data(), &nfb));
  if (n_fallback) *n_fallback = nfb;
  std::vector<std::vector<bool>> out(B, std::vector<bool>(N));
  for (std::size_t j = 0; j < B; j++)
    for (std::size_t i = 0; i < N; i++) out[j][i] = ok[j * N + i] != 0;
  return out;
}

// PublicKey::verify (src/lib.rs:115-117) for many (signature, message) pairs under ONE key through one random linear
// combination per group of 64 jobs (tc_verify_sig_rlc_batch, opt-in; groups that fail are re-checked job by job)
inline std::vector<bool> verify_rlc_batch(const PublicKey& pk, const std::vector<Signature>& sigs, const Messages& msgs,
                                          const std::array<std::uint8_t, 32>& seed, std::uint64_t* n_fallback = nullptr,
                                          Engine& e = Engine::instance()) {
  const std::size_t B = sigs.size();
  if (B != msgs.size()) throw std::invalid_argument("one message per signature");
  std::vector<std::uint8_t> sg(B * 192 + 1), ok(B + 1);
  for (std::size_t j = 0; j < B; j++) std::memcpy(&sg[j * 192], sigs[j].g2.data(), 192);
  std::uint64_t nfb = 0;
  if (B) e.check(tc_verify_sig_rlc_batch(e.ctx(), pk.g1.data(), sg.data(), msgs.data(), msgs.off.data(), B, 0, seed.data(), ok.data(), &nfb));
  if (n_fallback) *n_fallback = nfb;
  std::vector<bool> out(B);
  for (std::size_t j = 0; j < B; j++) out[j] = ok[j] != 0;
  return out;
}
// The loop of examples/threshold_enc.rs over PublicKeyShare::verify_decryption_share (src/lib.rs:182-186) for B ciphertexts
// x N nodes through one random linear combination per ciphertext (tc_verify_decryption_shares_rlc_batch, opt-in)
inline std::vector<std::vector<bool>> verify_decryption_shares_rlc_batch(const std::vector<PublicKeyShare>& pk_shares,
                                                                         const std::vector<std::vector<DecryptionShare>>& shares,
                                                                         const std::vector<Ciphertext>& cts,
                                                                         const std::array<std::uint8_t, 32>& seed,
                                                                         std::uint64_t* n_fallback = nullptr, Engine& e = Engine::instance()) {
  const std::size_t N = pk_shares.size(), B = cts.size();
  if (shares.size() != B) throw std::invalid_argument("one share row per ciphertext");
  std::vector<std::uint8_t> pk(N * 96 + 1), sh(B * N * 96 + 1), u(B * 96 + 1), w(B * 192 + 1), ok(B * N + 1);
  Messages v;
  for (std::size_t i = 0; i < N; i++) std::memcpy(&pk[i * 96], pk_shares[i].pk.g1.data(), 96);
  for (std::size_t j = 0; j < B; j++) {
    if (shares[j].size() != N) throw std::invalid_argument("every ciphertext needs one share per key");
    for (std::size_t i = 0; i < N; i++) std::memcpy(&sh[(j * N + i) * 96], shares[j][i].g1.data(), 96);
    std::memcpy(&u[j * 96], cts[j].u.data(), 96);
    std::memcpy(&w[j * 192], cts[j].w.data(), 192);
    v.push(cts[j].v.data(), cts[j].v.size());
  }
  std::uint64_t nfb = 0;
  if (B && N)
    e.check(tc_verify_decryption_shares_rlc_batch(e.ctx(), pk.data(), N, sh.data(), u.data(), v.data(), v.off.data(), w.data(), B, seed.data(), ok.data(),
                                                  &nfb));
  if (n_fallback) *n_fallback = nfb;
  std::vector<std::vector<bool>> out(B, std::vector<bool>(N));
  for (std::size_t j = 0; j < B; j++)
    for (std::size_t i = 0; i < N; i++) out[j][i] = ok[j * N + i] != 0;
  return out;
}

// Poly::commitment (src/poly.rs:372-377) / BivarPoly::commitment (:625-632): coefficient * g1 for every Fr
// coefficient, fixed-base on the device (LDS window table of the generator)
inline std::vector<G1Bytes> commitment(const std::vector<FrBytes>& coeff, Engine& e = Engine::instance()) {
  std::vector<std::uint8_t> fr(coeff.size() * 32 + 1), out(coeff.size() * 96 + 1), st(coeff.size() + 1);
  for (std::size_t i = 0; i < coeff.size(); i++) std::memcpy(&fr[i * 32], coeff[i].data(), 32);
  if (!coeff.empty()) e.check(tc_g1_commitment_batch(e.ctx(), fr.data(), coeff.size(), out.data(), st.data()));
  std::vector<G1Bytes> res(coeff.size());
  for (std::size_t i = 0; i < coeff.size(); i++) {
    raise_status(st[i]);
    std::memcpy(res[i].data(), &out[i * 96], 96);
  }
  return res;
}
// BivarCommitment::row (src/poly.rs:713-727): coeff holds the (degree+1)(degree+2)/2 commitments in coeff_pos order
inline std::vector<G1Bytes> bivar_commitment_row(const std::vector<G1Bytes>& coeff, std::size_t degree, std::uint64_t x,
                                                 Engine& e = Engine::instance()) {
  if (coeff.size() != (degree + 1) * (degree + 2) / 2) throw std::invalid_argument("bivariate commitment size");
  std::vector<std::uint8_t> c(coeff.size() * 96), out((degree + 1) * 96), st(degree + 1);
  for (std::size_t i = 0; i < coeff.size(); i++) std::memcpy(&c[i * 96], coeff[i].data(), 96);
  e.check(tc_bivar_commitment_row_batch(e.ctx(), c.data(), degree, &x, 1, out.data(), st.data()));
  std::vector<G1Bytes> res(degree + 1);
  for (std::size_t i = 0; i <= degree; i++) {
    raise_status(st[i]);
    std::memcpy(res[i].data(), &out[i * 96], 96);
  }
  return res;
}

// Several GPUs of one node from this process (tc_group_*): contiguous job sharding, RCCL broadcast of the key set
class Group {
 public:
  explicit Group(const std::vector<int>& devices) {
    if (tc_group_create(&g_, devices.data(), (int)devices.size()) != TC_OK) throw GpuError("tc_group_create failed");
  }
  ~Group() { tc_group_destroy(g_); }
  Group(const Group&) = delete;
  Group& operator=(const Group&) = delete;
  int size() const { return tc_group_size(g_); }
  void set_keyset(const std::vector<G1Bytes>& commit) {
    std::vector<std::uint8_t> c(commit.size() * 96);
    for (std::size_t i = 0; i < commit.size(); i++) std::memcpy(&c[i * 96], commit[i].data(), 96);
    check(tc_group_set_keyset(g_, commit.size() - 1, c.data()));
  }
  // PublicKeySet::combine_signatures for B jobs of n shares each, sharded over the GPUs
  std::vector<Signature> combine_signatures(std::size_t n, const std::vector<std::uint64_t>& idx, const std::vector<std::uint8_t>& shares,
                                            std::vector<std::uint8_t>& status) {
    const std::size_t B = n ? idx.size() / n : 0;
    // the C ABI trusts its sizes: the vectors must hold exactly B x n indices and B x n x 192 share bytes
    if (n == 0 || idx.size() != B * n || shares.size() != B * n * 192) throw GpuError("Group::combine_signatures: idx / shares sizes do not match n");
    std::vector<std::uint8_t> out(B * 192 + 1);
    status.assign(B, 0);
    if (B) check(tc_group_combine_signatures(g_, n, idx.data(), shares.data(), B, out.data(), status.data()));
    std::vector<Signature> res(B);
    for (std::size_t j = 0; j < B; j++) std::memcpy(res[j].g2.data(), &out[j * 192], 192);
    return res;
  }
  // (host-to-device, device-to-host) bytes the group and its contexts have moved over PCIe
  std::pair<std::uint64_t, std::uint64_t> transfer_bytes() const {
    std::uint64_t up = 0, down = 0;
    check(tc_group_transfer_bytes(g_, &up, &down));
    return {up, down};
  }
  tc_group* raw() const { return g_; }

 private:
  void check(int rc) const {
    if (rc != TC_OK) throw GpuError(std::string("libtc_amd group: ") + tc_group_last_error(g_));
  }
  tc_group* g_ = nullptr;
};

}  // namespace threshold_crypto
