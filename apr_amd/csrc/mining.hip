// Hardest-contrastive mining reduction.
//
// Reference: FCGF_APR/lib/trainer.py:400-452 == lib/complement_trainer.py:296-348.  It builds two
// [num_pos, num_hn] pdist matrices on the GPU, pulls the arg-mins back to the host, filters the
// mined negatives that are true positives with np.isin on int64 pair keys (`_hash`,
// util/misc.py:6-18) and pushes masks back: two device<->host round trips per step.  Here the
// nearest negatives come from the fused arg-min kernel (match.hip, nothing materialised) and this
// kernel does the rest in one pass: key = i + j * hash_seed, binary search in the sorted positive
// keys, relu(|f0-f1|^2 - pos_thresh), relu(neg_thresh - sqrt(d2 + 1e-7))^2, masked sums.
#include "common.h"

namespace {

__device__ inline bool key_in_sorted(const long long* __restrict__ keys, int n, long long k) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo < n && keys[lo] == k;
}

__global__ void k_contrastive(const float* __restrict__ pf0, const float* __restrict__ pf1, int p, int c,
                              const unsigned long long* __restrict__ nn01, const unsigned long long* __restrict__ nn10,
                              const long long* __restrict__ sel0, const long long* __restrict__ sel1,
                              const long long* __restrict__ pos0, const long long* __restrict__ pos1,
                              const long long* __restrict__ keys, int nkeys, long long hash_seed, float pos_thresh,
                              float neg_thresh, double* __restrict__ out /*[6]*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v[6] = {0, 0, 0, 0, 0, 0};
  if (i < p) {
    float d = 0.f;
    for (int k = 0; k < c; ++k) {
      const float e = pf0[(int64_t)i * c + k] - pf1[(int64_t)i * c + k];
      d = fmaf(e, e, d);
    }
    v[0] = fmaxf(d - pos_thresh, 0.f);
    v[1] = 1.0;
    {
      const unsigned long long b = nn01[i];
      const float dmin = sqrtf(__uint_as_float((unsigned)(b >> 32)) + 1e-7f);
      const long long j = sel1[(int)(b & 0xffffffffull)];
      if (!key_in_sorted(keys, nkeys, pos0[i] + j * hash_seed)) {
        const float r = fmaxf(neg_thresh - dmin, 0.f);
        v[2] = (double)(r * r);
        v[3] = 1.0;
      }
    }
    {
      const unsigned long long b = nn10[i];
      const float dmin = sqrtf(__uint_as_float((unsigned)(b >> 32)) + 1e-7f);
      const long long j = sel0[(int)(b & 0xffffffffull)];
      if (!key_in_sorted(keys, nkeys, j + pos1[i] * hash_seed)) {
        const float r = fmaxf(neg_thresh - dmin, 0.f);
        v[4] = (double)(r * r);
        v[5] = 1.0;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double s = v[k];
    for (int dd = 32; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd);
    if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(&out[k], s);
  }
}

}  // namespace

APR_API int apr_contrastive_reduce(const float* pos_f0, const float* pos_f1, int32_t p, int32_t c,
                                   const uint64_t* nn01, const uint64_t* nn10, const int64_t* sel0,
                                   const int64_t* sel1, const int64_t* pos_ind0, const int64_t* pos_ind1,
                                   const int64_t* sorted_pos_keys, int32_t n_keys, int64_t hash_seed, float pos_thresh,
                                   float neg_thresh, double* out6, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(p > 0 && c > 0 && n_keys >= 0, "apr_contrastive_reduce: bad arguments");
  APR_HIP(hipMemsetAsync(out6, 0, 6 * sizeof(double), st));
  hipLaunchKernelGGL(k_contrastive, dim3((unsigned)cdiv64(p, 256)), dim3(256), 0, st, pos_f0, pos_f1, p, c,
                     (const unsigned long long*)nn01, (const unsigned long long*)nn10, (const long long*)sel0,
                     (const long long*)sel1, (const long long*)pos_ind0, (const long long*)pos_ind1,
                     (const long long*)sorted_pos_keys, n_keys, (long long)hash_seed, pos_thresh, neg_thresh, out6);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
