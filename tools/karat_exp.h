#pragma once
// experimental: one-level Karatsuba for the two products of fq_mul2_body (7 + 7 limbs)
namespace tc {
TC_HD void fq_mul2k_body(const int32_t* x, const int32_t* y, const int32_t* z, const int32_t* w, int32_t* out) {
  constexpr int N = FQ_LIMBS;
  constexpr int H = N / 2;
  int32_t xs[H], ys[H], zs[H], ws[H];
  TC_UNROLL for (int i = 0; i < H; i++) {
    xs[i] = x[i] + x[i + H];
    ys[i] = y[i] + y[i + H];
    zs[i] = z[i] + z[i + H];
    ws[i] = w[i] + w[i + H];
  }
  int64_t L[2 * H - 1], U[2 * H - 1], M[2 * H - 1];
  TC_UNROLL for (int k = 0; k < 2 * H - 1; k++) {
    const int lo = (k < H) ? 0 : (k - H + 1);
    const int hi = (k < H) ? k : (H - 1);
    int64_t l = 0, u = 0;
    TC_UNROLL for (int i = lo; i <= hi; i++) {
      l += (int64_t)x[i] * y[k - i];
      l += (int64_t)z[i] * w[k - i];
      u += (int64_t)x[i + H] * y[k - i + H];
      u += (int64_t)z[i + H] * w[k - i + H];
    }
    int64_t m = -l - u;
    TC_UNROLL for (int i = lo; i <= hi; i++) {
      m += (int64_t)xs[i] * ys[k - i];
      m += (int64_t)zs[i] * ws[k - i];
    }
    L[k] = l;
    U[k] = u;
    M[k] = m;
  }
  int32_t mq[N];
  int64_t carry = 0;
  TC_UNROLL for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = (k < N) ? 0 : (k - N + 1);
    const int hi = (k < N) ? k : (N - 1);
    int64_t s1 = carry;
    if (k < 2 * H - 1) s1 += L[k];
    if (k >= H && k - H < 2 * H - 1) s1 += M[k - H];
    if (k >= 2 * H && k - 2 * H < 2 * H - 1) s1 += U[k - 2 * H];
    int64_t s2 = 0;
    if (k < N) {
      TC_UNROLL for (int i = 0; i < k; i++) s2 += (int64_t)mq[i] * FQL_P[k - i];
      int64_t s = s1 + s2;
      mq[k] = (int32_t)(((uint32_t)s * FQL_INV) & (uint32_t)FQ_MASK);
      s += (int64_t)mq[k] * FQL_P[0];
      carry = s >> FQ_RADIX;
    } else {
      TC_UNROLL for (int i = lo; i <= hi; i++) s2 += (int64_t)mq[i] * FQL_P[k - i];
      int64_t s = s1 + s2;
      out[k - N] = (int32_t)((uint32_t)s & (uint32_t)FQ_MASK);
      carry = s >> FQ_RADIX;
    }
  }
  out[N - 1] = (int32_t)carry;
}
}  // namespace tc
