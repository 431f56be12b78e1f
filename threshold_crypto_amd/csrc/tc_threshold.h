// Lagrange-at-zero coefficients and the Straus multi-scalar combiner (device side).
// Replaces, for batches, the private `interpolate` of /root/reference/src/lib.rs:719-767.
#pragma once
#include "tc_curve.h"

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

// sum_{k < K} s_k * P_k for K <= 4 points with per-lane 255-bit scalars: joint (Straus)
// double-and-add over a (2^K - 1)-entry subset-sum table held in the lane's scratch.
// One shared doubling chain for the K points; control flow is lane-uniform except the
// "all K bits are zero" skip.  sc[k] points at 8 little-endian u32 words (canonical, < r).
template <class F, int K>
TC_HD Jac<F> straus_chunk(const Affine<F>* pts, const uint32_t (*sc)[8]) {
  Jac<F> tbl[1 << K];
  tbl[0] = Jac<F>::infinity();
  TC_NOUNROLL for (int m = 1; m < (1 << K); m++) {
    int low = 0;
    while (!((m >> low) & 1)) low++;
    const int rest = m & (m - 1);
    tbl[m] = jac_add_mixed(tbl[rest], pts[low]);
  }
  Jac<F> acc = Jac<F>::infinity();
  TC_NOUNROLL for (int bit = 254; bit >= 0; bit--) {
    acc = jac_dbl(acc);
    uint32_t m = 0;
    TC_UNROLL for (int k = 0; k < K; k++) m |= ((sc[k][bit >> 5] >> (bit & 31)) & 1u) << k;
    if (m) acc = jac_add(acc, tbl[m]);
  }
  return acc;
}

}  // namespace tc
