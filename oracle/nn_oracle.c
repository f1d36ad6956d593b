/* ORACLE (test infrastructure only -- never linked into the product path).
 *
 * CPU restatement of the squared-L2 feature nearest neighbour of
 * /root/reference/FCGF_APR/lib/metrics.py:22-29 (pdist, 'SquareL2') +
 * lib/eval.py:18-48 (row-wise min / argmin).  The reference sums (a-b)^2 over
 * the channel axis with torch.sum (order unspecified); this restatement fixes
 * the order -- four fma chains over c mod 4, combined as (s0+s1)+(s2+s3) -- so
 * the HIP kernel (apr_amd/csrc/match.hip) can be compared BIT-EXACTLY.
 * Ties resolve to the smallest index, as torch.min(dim=1) does on CPU.
 * Pinned against the imported reference `pdist` by tests/test_oracle_pinning.py
 * (fixture tests/golden/pdist_ref.npz).
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>

static inline float sqdist(const float* a, const float* b, int c) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < c; ++k) {
    float d = a[k] - b[k];
    s[k & 3] = fmaf(d, d, s[k & 3]);
  }
  return (s[0] + s[1]) + (s[2] + s[3]);
}

void nn_sqdist_argmin(const float* f0, int64_t n0, const float* f1, int64_t n1, int c,
                      int64_t* idx, float* d2, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t i = 0; i < n0; ++i) {
    float best = INFINITY;
    int64_t bj = -1;
    for (int64_t j = 0; j < n1; ++j) {
      float d = sqdist(f0 + i * c, f1 + j * c, c);
      if (d < best) {
        best = d;
        bj = j;
      }
    }
    idx[i] = bj;
    if (d2) d2[i] = best;
  }
}

/* full matrix, for pinning against the reference's pdist on small inputs */
void sqdist_matrix(const float* f0, int64_t n0, const float* f1, int64_t n1, int c, float* out) {
  for (int64_t i = 0; i < n0; ++i)
    for (int64_t j = 0; j < n1; ++j) out[i * n1 + j] = sqdist(f0 + i * c, f1 + j * c, c);
}
