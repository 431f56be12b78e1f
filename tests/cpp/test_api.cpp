// Replays test_threshold_sig / test_threshold_enc / test_from_to_bytes of the reference
// (src/lib.rs:822-873, 907-939, 984-993) through the C++ host mirror include/threshold_crypto.hpp.
// Built and run by tests/test_gpu_cpp_api.py on the GPU box; key material and expected values come
// from a fixture file the Python test writes with the oracle.
//   fixture: u32 t, u32 n | n x 32 B share scalars | (t+1) x 96 B commitment | u32 msg_len, msg |
//            192 B expected combined signature | ciphertext: 96 B u, u32 vlen, v, 192 B w | u32 plen, plaintext |
//            (r04) 4 x (i64 index, 32 B share scalar of that index): negative / large `T: IntoFr` keys
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include "threshold_crypto.hpp"

using namespace threshold_crypto;

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                     \
    }                                                                   \
  } while (0)

template <class T>
static void rd(std::ifstream& f, T* p, std::size_t n) {
  f.read(reinterpret_cast<char*>(p), (std::streamsize)n);
  if (!f) { std::fprintf(stderr, "short fixture\n"); std::exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  std::uint32_t t, n, len;
  rd(f, &t, 4); rd(f, &n, 4);
  std::vector<SecretKeyShare> shares;
  for (std::uint32_t i = 0; i < n; i++) { FrBytes fr; rd(f, fr.data(), 32); shares.emplace_back(fr); }
  std::vector<G1Bytes> commit(t + 1);
  for (auto& c : commit) rd(f, c.data(), 96);
  rd(f, &len, 4); std::string msg(len, '\0'); rd(f, &msg[0], len);
  Signature expected; rd(f, expected.g2.data(), 192);
  Ciphertext ct; rd(f, ct.u.data(), 96); rd(f, &len, 4); ct.v.resize(len); rd(f, ct.v.data(), len); rd(f, ct.w.data(), 192);
  rd(f, &len, 4); Bytes plain(len); rd(f, plain.data(), len);

  std::vector<std::pair<std::int64_t, SecretKeyShare>> odd;   // shares at indices u64 cannot carry (IntoFr for i64)
  for (int i = 0; i < 5; i++) {
    std::int64_t ix; FrBytes fr;
    rd(f, &ix, 8); rd(f, fr.data(), 32);
    odd.emplace_back(ix, SecretKeyShare(fr));
  }

  PublicKeySet pk_set(commit);
  CHECK(pk_set.threshold() == t);

  // -- test_threshold_sig ----------------------------------------------------------------------
  std::map<std::uint64_t, SignatureShare> sigs;
  for (std::uint64_t i : {5, 8, 7, 10}) {
    sigs[i] = shares[i].sign(msg);
    CHECK(pk_set.public_key_share(i).verify(sigs[i], msg));
    CHECK(pk_set.public_key_share(i).pk == shares[i].public_key_share().pk);
  }
  Signature sig = pk_set.combine_signatures(sigs);
  CHECK(sig == expected);
  CHECK(pk_set.public_key().verify(sig, msg));
  CHECK(!pk_set.public_key().verify(sig, msg + "!"));
  std::map<std::uint64_t, SignatureShare> sigs2;
  for (std::uint64_t i : {1, 2, 3, 4}) sigs2[i] = shares[i].sign(msg);
  CHECK(pk_set.combine_signatures(sigs2) == sig);
  sigs.erase(10);
  bool threw = false;
  try { pk_set.combine_signatures(sigs); } catch (const ErrorException& e) { threw = e.code == Error::NotEnoughShares; }
  CHECK(threw);
  // batch form: S signers x B messages, then B combines in one launch
  Messages m;
  for (int j = 0; j < 70; j++) m.push(msg + std::to_string(j));
  std::vector<const SecretKeyShare*> signers;
  for (std::uint32_t i = 0; i < n; i++) signers.push_back(&shares[i]);
  auto all = sign_shares_batch(signers, m);
  std::vector<std::map<std::uint64_t, SignatureShare>> jobs(m.size());
  for (std::size_t j = 0; j < m.size(); j++)
    for (std::uint64_t i : {0 + j % 3, 4 + j % 2, 6ul, 9ul}) jobs[j][i] = all[j][i];
  std::vector<std::uint8_t> st;
  auto combined = pk_set.combine_signatures_batch(jobs, st);
  for (auto s : st) CHECK(s == 0);
  auto ok = pk_set.public_key().verify_batch(combined, m);
  for (bool b : ok) CHECK(b);

  // -- `T: IntoFr` keys beyond u64 (src/into_fr.rs: i64 negatives) and the wire-level combine (round 4) -------------------
  {
    std::map<FrIndex, SignatureShare> fsigs;
    for (auto& kv : odd) fsigs[FrIndex::from_i64(kv.first)] = kv.second.sign(msg);
    CHECK(pk_set.combine_signatures(fsigs) == expected);
    CHECK(FrIndex::from_i64(-1) < FrIndex::from_i64(-1) == false && FrIndex::from_u64(5) < FrIndex::from_i64(-7));
    // (round 5) keys of type i64: a BTreeMap<i64, _> iterates by the SIGNED value, so with t + 2 shares interpolate()
    // takes {-2^63, -2^40, -1, 7} and never looks at key 9 -- a bad share THERE must not matter.  Keyed by FrIndex (the order of
    // BTreeMap<Fr, _>) the first t + 1 are {7, 9, ...}: the same bad share spoils the result.  Both behaviours are the reference's,
    // each for its own key type.
    std::map<std::int64_t, SignatureShare> isigs;
    std::map<FrIndex, SignatureShare> fbad;
    for (auto& kv : odd) {
      SignatureShare sh = kv.second.sign(msg);
      if (kv.first == 9) sh = odd[0].second.sign(msg);   // the share of another node under key 9
      isigs[kv.first] = sh;
      fbad[FrIndex::from_i64(kv.first)] = sh;
    }
    CHECK(isigs.size() == 5 && isigs.begin()->first < 0);
    CHECK(pk_set.combine_signatures(isigs) == expected);
    CHECK(!(pk_set.combine_signatures(fbad) == expected));
    std::vector<std::vector<std::pair<std::uint64_t, std::array<std::uint8_t, SIG_SIZE>>>> wjobs(3);
    for (std::size_t j = 0; j < 3; j++)
      for (const auto& kv : jobs[j]) wjobs[j].emplace_back(kv.first, kv.second.sig.to_bytes());
    wjobs[2][1].second[5] ^= 0x01;   // a share that no longer decodes to a group member
    std::vector<std::uint8_t> wst;
    auto wire = pk_set.combine_signatures_wire_batch(wjobs, wst);
    CHECK(wst[0] == 0 && wst[1] == 0 && wst[2] == 3);
    CHECK(wire[0] == combined[0].to_bytes() && wire[1] == combined[1].to_bytes() && wire[2][0] == 0xc0);
  }

  // -- test_from_to_bytes -----------------------------------------------------------------------
  CHECK(Signature::from_bytes(sig.to_bytes()) == sig);
  CHECK(PublicKey::from_bytes(pk_set.public_key().to_bytes()) == pk_set.public_key());
  auto bad = sig.to_bytes();
  bad[0] &= 0x7f;
  threw = false;
  try { Signature::from_bytes(bad); } catch (const FromBytesError&) { threw = true; }
  CHECK(threw);
  CHECK(hash_g2(msg) == hash_g2(msg) && hash_g2(msg) != hash_g2(msg + "x"));

  // -- test_threshold_enc -------------------------------------------------------------------------
  CHECK(ct.verify());
  std::map<std::uint64_t, DecryptionShare> dsh;
  for (std::uint64_t i : {8, 4, 7, 9}) {
    auto d = shares[i].decrypt_share(ct);
    CHECK(d.has_value());
    CHECK(pk_set.public_key_share(i).verify_decryption_share(*d, ct));
    dsh[i] = *d;
  }
  CHECK(pk_set.decrypt(dsh, ct) == plain);
  {
    // (round 6) decrypt with `T: IntoFr` keys beyond u64, as combine_signatures above: with t + 2 shares keyed by i64 the
    // first t + 1 in SIGNED order are {-2^63, -2^40, -1, 7}, so another node's share under key 9 is never looked at; keyed by FrIndex
    // (the order of BTreeMap<Fr, _>) the first t + 1 start {7, 9, ...} and the same bad share spoils the plaintext.
    std::map<std::int64_t, DecryptionShare> idsh;
    std::map<FrIndex, DecryptionShare> fdsh, fbad;
    for (auto& kv : odd) {
      auto d = kv.second.decrypt_share(ct);
      CHECK(d.has_value());
      fdsh[FrIndex::from_i64(kv.first)] = *d;
      if (kv.first == 9) d = odd[0].second.decrypt_share(ct);
      idsh[kv.first] = *d;
      fbad[FrIndex::from_i64(kv.first)] = *d;
    }
    CHECK(idsh.size() == 5 && idsh.begin()->first < 0);
    CHECK(pk_set.decrypt(fdsh, ct) == plain);
    CHECK(pk_set.decrypt(idsh, ct) == plain);
    CHECK(!(pk_set.decrypt(fbad, ct) == plain));
  }
  Ciphertext fake = ct;
  fake.v[0] ^= 1;
  CHECK(!fake.verify());
  CHECK(!shares[2].decrypt_share(fake).has_value());
  // -- membership, RLC share validation, DKG commitments, a two-worker group (round 2 of the C ABI) -----------------
  CHECK(is_member(sig.g2) && is_member(pk_set.public_key().g1));
  {
    std::vector<PublicKeyShare> pks;
    for (std::uint32_t i = 0; i < n; i++) pks.push_back(pk_set.public_key_share(i));
    std::vector<std::vector<SignatureShare>> rows(all.begin(), all.end());
    rows[3][2] = rows[3][5];  // node 2 sends node 5's share for message 3
    std::array<std::uint8_t, 32> seed{};
    for (int i = 0; i < 32; i++) seed[i] = (std::uint8_t)(i * 7 + 1);
    std::uint64_t nfb = 0;
    auto okm = verify_shares_rlc_batch(pks, rows, m, seed, &nfb);
    CHECK(nfb == 1);
    for (std::size_t j = 0; j < rows.size(); j++)
      for (std::uint32_t i = 0; i < n; i++) CHECK(okm[j][i] == !(j == 3 && i == 2));
    // Poly::commitment of the share scalars = the public key shares (sk_i * g1)
    std::vector<FrBytes> frs;
    for (std::uint32_t i = 0; i < n; i++) frs.push_back(shares[i].key().fr());
    auto cm = commitment(frs);
    for (std::uint32_t i = 0; i < n; i++) CHECK(cm[i] == pks[i].pk.g1);
    // BivarCommitment::row of the degree-0 "matrix" {c} is {c}; of degree 1 with x = 0 it is (c00, c01)
    std::vector<G1Bytes> bc = {cm[0], cm[1], cm[2]};
    auto row0 = bivar_commitment_row(bc, 1, 0);
    CHECK(row0[0] == cm[0] && row0[1] == cm[1]);
    // two workers on GPU 0 (no RCCL with duplicate devices: test configuration), same bytes as one context
    Group grp({0, 0});
    grp.set_keyset(commit);
    std::vector<std::uint64_t> gidx;
    std::vector<std::uint8_t> gsh, gst;
    for (std::size_t j = 0; j < jobs.size(); j++)
      for (const auto& kv : jobs[j]) {
        gidx.push_back(kv.first);
        gsh.insert(gsh.end(), kv.second.sig.g2.begin(), kv.second.sig.g2.end());
      }
    auto gsig = grp.combine_signatures(4, gidx, gsh, gst);
    for (std::size_t j = 0; j < jobs.size(); j++) CHECK(gst[j] == 0 && gsig[j] == combined[j]);
    auto moved = grp.transfer_bytes();
    CHECK(moved.first >= gsh.size() && moved.second >= combined.size() * 192);
    threw = false;
    try { gidx.pop_back(); grp.combine_signatures(4, gidx, gsh, gst); } catch (const GpuError&) { threw = true; }
    CHECK(threw);  // sizes that do not match n are refused before the C ABI sees them
    // -- round 3: same-key signature batches and decryption shares by random linear combination ------------------------
    std::vector<Signature> swapped(combined.begin(), combined.end());
    swapped[11] = combined[12];
    nfb = 0;
    auto okr = verify_rlc_batch(pk_set.public_key(), swapped, m, seed, &nfb);
    for (std::size_t j = 0; j < swapped.size(); j++) CHECK(okr[j] == (j != 11));
    CHECK(nfb == 64);  // the group of 64 jobs that holds job 11 was re-checked job by job
    std::vector<Ciphertext> cts = {ct, ct};
    std::vector<std::vector<DecryptionShare>> drows(2);
    for (std::uint32_t i = 0; i < n; i++) {
      auto d = shares[i].decrypt_share(ct);
      CHECK(d.has_value());
      drows[0].push_back(*d);
      drows[1].push_back(*d);
    }
    drows[1][4] = drows[1][6];  // node 4 hands in node 6's share for the second ciphertext
    nfb = 0;
    auto okd = verify_decryption_shares_rlc_batch(pks, drows, cts, seed, &nfb);
    CHECK(nfb == 1);
    for (std::size_t j = 0; j < 2; j++)
      for (std::uint32_t i = 0; i < n; i++) CHECK(okd[j][i] == !(j == 1 && i == 4));
  }
  std::puts("CPP-API-OK");
  return 0;
}
