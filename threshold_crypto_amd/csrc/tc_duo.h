// Two jobs per lane pair for the Fq-only phases of the G2 kernels (r05).
//
// A lane pair holds every Fq2 value as one coefficient per lane (tc_common.h), so whatever a G2 job does in Fq ALONE -- the
// square-root exponentiations of the checked decode (/root/reference/src/lib.rs:246-252) and of hash_g2 (:691-694), the Jacobi
// symbol of the rejection loop, the inversion of a norm -- both lanes used to execute identically: a third of the multiply-adds
// of a G2 decode and a seventh of a hash were the second copy (tests/count_ops.py).  The kernels of the wire path
// and the hashes therefore give a lane pair TWO jobs, A and B: Fq2 work runs for A, then for B, with both lanes; Fq-only work
// runs ONCE, lane 0 on A's value and lane 1 on B's.
//
//   Duo<T>            one T per job: the device build keeps the lane's own (lane 0: A, lane 1: B), the host build of the same
//                     source (tests/hostsim, one thread per pair) keeps both
//   duo_each(fn)      runs fn(slot) for the lane's slot (device) / for slots 0 and 1 (host)
//   duo_from / duo_to move values both lanes hold identically into the slots and back
#pragma once
#include "tc_tower.h"

namespace tc {

#if defined(TC_COUNT_OPS)
// an Fq operation inside duo_each is executed once per job (each lane works for a different one): counted like the split
// halves of an Fq2 method (tc_field.h)
#define TC_DUO_SCOPE TcSplitScope tc_duo_scope_
#else
#define TC_DUO_SCOPE
#endif

template <class T>
struct Duo {
#if TC_PAIR
  T v;
  TC_HD T& at(int) { return v; }
  TC_HD const T& at(int) const { return v; }
#else
  T v[2];
  TC_HD T& at(int s) { return v[s]; }
  TC_HD const T& at(int s) const { return v[s]; }
#endif
};

template <class FN>
TC_HD void duo_each(FN fn) {
  TC_DUO_SCOPE;
#if TC_PAIR
  fn(pair_odd());
#else
  fn(0);
  fn(1);
#endif
}

// pick only (pointers, lengths: values that never travel back)
template <class T>
TC_HD Duo<T> duo_pick(const T& a, const T& b) {
#if TC_PAIR
  return Duo<T>{pair_odd() ? b : a};
#else
  return Duo<T>{{a, b}};
#endif
}
TC_HD Duo<Fq> duo_from(const Fq& a, const Fq& b) {
#if TC_PAIR
  return Duo<Fq>{Fq::select(pair_odd() != 0, b, a)};
#else
  return Duo<Fq>{{a, b}};
#endif
}
TC_HD void duo_to(const Duo<Fq>& d, Fq& a, Fq& b) {
#if TC_PAIR
  const bool odd = pair_odd() != 0;
  const Fq o = Fq2{d.v}.other();
  a = Fq::select(odd, o, d.v);
  b = Fq::select(odd, d.v, o);
#else
  a = d.v[0];
  b = d.v[1];
#endif
}
TC_HD Duo<uint32_t> duo_from(uint32_t a, uint32_t b) {
#if TC_PAIR
  return Duo<uint32_t>{pair_odd() ? b : a};
#else
  return Duo<uint32_t>{{a, b}};
#endif
}
TC_HD void duo_to(const Duo<uint32_t>& d, uint32_t& a, uint32_t& b) {
#if TC_PAIR
  const bool odd = pair_odd() != 0;
  const uint32_t o = (uint32_t)pair_swap((int32_t)d.v);
  a = odd ? o : d.v;
  b = odd ? d.v : o;
#else
  a = d.v[0];
  b = d.v[1];
#endif
}
TC_HD Duo<bool> duo_from(bool a, bool b) {
#if TC_PAIR
  return Duo<bool>{pair_odd() ? b : a};
#else
  return Duo<bool>{{a, b}};
#endif
}
TC_HD void duo_to(const Duo<bool>& d, bool& a, bool& b) {
#if TC_PAIR
  const bool odd = pair_odd() != 0;
  const bool o = pair_swap(d.v ? 1 : 0) != 0;
  a = odd ? o : d.v;
  b = odd ? d.v : o;
#else
  a = d.v[0];
  b = d.v[1];
#endif
}
// an Fq2 value of job A / job B from per-slot coefficients (re, im both drawn by the slot's own lane)
TC_HD void duo_to_fq2(const Duo<Fq>& re, const Duo<Fq>& im, Fq2& a, Fq2& b) {
#if TC_PAIR
  // A = (lane 0's re, lane 0's im): lane 0 keeps its re, lane 1 takes the partner's im;  B the other way round
  const bool odd = pair_odd() != 0;
  const Fq ore = Fq2{re.v}.other(), oim = Fq2{im.v}.other();
  a = Fq2{Fq::select(odd, oim, re.v)};
  b = Fq2{Fq::select(odd, im.v, ore)};
#else
  a = Fq2::make(re.v[0], im.v[0]);
  b = Fq2::make(re.v[1], im.v[1]);
#endif
}

}  // namespace tc
