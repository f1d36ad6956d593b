// Score-weighted sampling without replacement on the HOST: Predator_APR/lib/tester.py:83-92 draws
// 5000 interest points per cloud with `np.random.choice(idx, size=n_points, replace=False, p=probs)`.  The draw stays
// on the host's Mersenne Twister (the caller passes the uniforms of `RandomState.random_sample`), and this file
// restates the rest of NumPy's legacy algorithm (numpy/random/mtrand.pyx, RandomState.choice, replace=False with p):
//
//     while n_uniq < size:
//         x = rand(size - n_uniq)                       <- caller
//         p[found[:n_uniq]] = 0
//         cdf = cumsum(p); cdf /= cdf[-1]               sequential float64 adds, IEEE division
//         new = cdf.searchsorted(x, side='right')
//         new = new[sorted first occurrences]           duplicates inside the round dropped, draw order kept
//         found[n_uniq : n_uniq + len(new)] = new
//
// one round per call.  NumPy spends ~2 ms per 5000-of-14000 draw in interpreter round trips, np.unique's sort and
// temporaries; the same arithmetic in one pass is ~0.25 ms, which is what the per-pair tail of the pipeline is made of
// once the encoder runs several pairs per forward.  No GPU work: plain C++, callable without a device.
#include "common.h"

// p: the caller's float64 working copy of the probabilities (zeroed in place for the indices found so far);
// found: int64[size]; n_uniq: entries of `found` already filled; x: the round's uniforms, k = size - n_uniq of them;
// cdf: double[n] work; stamp: int32[n] work, zero-initialised before the first round; round_id: 1, 2, ... (stamp
// value).  Returns the new n_uniq, or a negative apr error code.
APR_API int64_t apr_weighted_choice_round(double* p, int64_t n, int64_t* found, int64_t n_uniq, int64_t n_new_zeroed,
                                          const double* x, int64_t k, double* cdf, int32_t* stamp, int32_t round_id) {
  if (!p || !found || !x || !cdf || !stamp || n <= 0 || k < 0 || n_uniq < 0 || n_new_zeroed < 0 ||
      n_new_zeroed > n_uniq || round_id <= 0) {
    apr_set_error("apr_weighted_choice_round: bad arguments");
    return APR_EINVAL;
  }
  // entries found in the previous round (the earlier ones are already zero)
  for (int64_t i = n_uniq - n_new_zeroed; i < n_uniq; ++i) {
    if (found[i] < 0 || found[i] >= n) {
      apr_set_error("apr_weighted_choice_round: found[%lld] out of range", (long long)i);
      return APR_EINVAL;
    }
    p[found[i]] = 0.0;
  }
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    acc += p[i];          // np.cumsum: strictly sequential
    cdf[i] = acc;
  }
  const double last = cdf[n - 1];
  for (int64_t i = 0; i < n; ++i) cdf[i] /= last;
  // searchsorted(side='right'): first index with cdf[idx] > x.  A round with many draws (the first: all `size` of them)
  // goes through a bucket table over [0, 1) -- bucket b starts at the first index with cdf > b / kBuckets, found for
  // all buckets by one merge pass, and a draw walks on from its bucket's start (1-2 steps) -- instead of `size`
  // binary searches full of mispredicted branches; the answer is the same index either way.
  constexpr int kBuckets = 1 << 14;
  static thread_local int32_t bucket_start[kBuckets + 1];
  const bool tabled = k >= 1024 && n < (1ll << 31);
  if (tabled) {
    int64_t idx = 0;
    for (int b = 0; b <= kBuckets; ++b) {
      const double edge = (double)b / (double)kBuckets;   // exact: power of two
      while (idx < n && cdf[idx] <= edge) ++idx;
      bucket_start[b] = (int32_t)idx;
    }
  }
  for (int64_t j = 0; j < k; ++j) {
    const double v = x[j];
    int64_t lo;
    if (tabled && v >= 0.0 && v < 1.0) {
      lo = bucket_start[(int)(v * (double)kBuckets)];
      while (lo < n && cdf[lo] <= v) ++lo;
    } else {
      lo = 0;
      int64_t hi = n;
      while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] <= v) lo = mid + 1; else hi = mid;
      }
    }
    if (lo >= n) {
      // NumPy would store n here and fail on the next `p[found] = 0` (only with NaN weights)
      apr_set_error("apr_weighted_choice_round: a draw fell past the end of the distribution (NaN weights?)");
      return APR_EINVAL;
    }
    if (stamp[lo] != round_id) {     // first occurrence inside this round
      stamp[lo] = round_id;
      found[n_uniq++] = lo;
    }
  }
  return n_uniq;
}
