// Reverse neighbour table + ordered gather-sum: the deterministic backward of an index gather (the
// input gradient of KPConv, Predator_APR/models/blocks.py:326-374 under lib/trainer.py:142-280; torch's own index
// backward, and round 2's apr_kpconv_dfeat, add through float atomics in arrival order).
//   apr_reverse_table_build: the flat positions t = q * H + h of a neighbour table nbr [nq, H], sorted by the support row
//       nbr[t] they point at (stable radix sort, rocPRIM: equal rows keep ascending t) + the start of every row's run;
//   apr_reverse_gather: out[s, :] = sum of src[t, :] over the run of s, in that order.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

__global__ void k_rev_keys(const int* __restrict__ nbr, int64_t total, int ns, int* __restrict__ keys, int* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int s = nbr[t];
  keys[t] = (s >= 0 && s < ns) ? s : ns;      // padding entries sort behind every real row
  vals[t] = (int)t;
}

// start[s] = first position of the sorted keys that is >= s (s = 0 .. ns): binary search
__global__ void k_rev_starts(const int* __restrict__ keys_sorted, int64_t total, int ns, int* __restrict__ start) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > ns) return;
  int64_t lo = 0, hi = total;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < s) lo = mid + 1;
    else hi = mid;
  }
  start[s] = (int)lo;
}

// thread = (support row, 4 channels): the row's contributions in table order, 4 loads in flight
__global__ void k_rev_gather(const float* __restrict__ src, int c, const int* __restrict__ rev_t, const int* __restrict__ start,
                             int64_t ns, float* __restrict__ out, int64_t ldo) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int c4 = c >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ns * c4) return;
  const int64_t s = t / c4;
  const int col = (int)(t - s * c4) * 4;
  const int e0 = start[s], e1 = start[s + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int e = e0; e < e1; e += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = (e + u < e1) ? *reinterpret_cast<const f32x4*>(src + (int64_t)rev_t[e + u] * c + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  *reinterpret_cast<f32x4*>(out + s * ldo + col) = acc;
}

// The same for the contributions of ONE CHUNK of the table: src holds the rows of flat positions [t_lo, t_hi) only (row
// t - t_lo); a row's run is ascending in t, so the chunk's part of it is one contiguous piece, and with `accumulate` the
// sum continues from the value the previous chunk left in out -- chunk after chunk in ascending t this performs exactly the
// additions of k_rev_gather in exactly their order (same bits), with a contribution buffer 1 / nchunks the size.
__global__ void k_rev_gather_range(const float* __restrict__ src, int c, const int* __restrict__ rev_t,
                                   const int* __restrict__ start, int64_t ns, int t_lo, int t_hi, int accumulate,
                                   float* __restrict__ out, int64_t ldo) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int c4 = c >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ns * c4) return;
  const int64_t s = t / c4;
  const int col = (int)(t - s * c4) * 4;
  int e0 = start[s], e1 = start[s + 1];
  // first entry with rev_t >= t_lo / >= t_hi (the run is sorted): two binary searches
  auto lower = [&](int bound) {
    int lo = e0, hi = e1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rev_t[mid] < bound) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int a = lower(t_lo), b = lower(t_hi);
  f32x4 acc = accumulate ? *reinterpret_cast<const f32x4*>(out + s * ldo + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int e = a; e < b; e += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = (e + u < b) ? *reinterpret_cast<const f32x4*>(src + (int64_t)(rev_t[e + u] - t_lo) * c + col)
                         : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e + u < b) acc += v[u];
  }
  *reinterpret_cast<f32x4*>(out + s * ldo + col) = acc;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int64_t total, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs((void*)nullptr, tmp, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                  (size_t)total, 0u, (unsigned)bits, (hipStream_t)0);
  return tmp;
}

int key_bits(int64_t ns) {
  int b = 1;
  while ((1ll << b) <= ns) ++b;
  return b;
}

}  // namespace

APR_API size_t apr_reverse_table_scratch_bytes(int64_t nq, int32_t H, int64_t ns) {
  if (nq <= 0 || H <= 0 || ns <= 0) return 0;
  const int64_t total = nq * H;
  return 3 * al256((size_t)total * 4) + al256(sort_temp_bytes(total, key_bits(ns))) + 512;
}

// nbr i32 [nq, H] (entries outside [0, ns) = padding) -> rev_t i32 [nq * H] (flat positions sorted by the row they point at,
// ties in ascending position; the padding entries last) and start i32 [ns + 1] (run of row s: rev_t[start[s] .. start[s + 1]))
APR_API int apr_reverse_table_build(const int32_t* nbr, int64_t nq, int32_t H, int64_t ns, int32_t* rev_t, int32_t* start,
                                    void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(nbr && rev_t && start && scratch && nq > 0 && H > 0 && ns > 0 && nq * (int64_t)H < (1ll << 31) && ns < (1ll << 30),
                "apr_reverse_table_build: bad arguments");
  APR_CHECK_ARG(scratch_bytes >= apr_reverse_table_scratch_bytes(nq, H, ns), "apr_reverse_table_build: scratch too small");
  const int64_t total = nq * H;
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  int* keys = (int*)p;          p += al256((size_t)total * 4);
  int* keys_sorted = (int*)p;   p += al256((size_t)total * 4);
  int* vals = (int*)p;          p += al256((size_t)total * 4);
  void* tmp = p;
  const int bits = key_bits(ns);
  size_t tmp_bytes = sort_temp_bytes(total, bits);
  hipLaunchKernelGGL(k_rev_keys, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, nbr, total, (int)ns, keys, vals);
  APR_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const int*)keys, keys_sorted, (const int*)vals, rev_t, (size_t)total, 0u,
                                    (unsigned)bits, st));
  hipLaunchKernelGGL(k_rev_starts, dim3((unsigned)cdiv64(ns + 1, 256)), dim3(256), 0, st, keys_sorted, total, (int)ns, start);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// out f32 [ns, c] (ld ldo) = per row s the sum of src[rev_t[e], :] (src f32 [*, c], contiguous rows), e over the run of s, in
// run order.  c % 4 == 0, 16-byte aligned rows.
APR_API int apr_reverse_gather(const float* src, int32_t c, const int32_t* rev_t, const int32_t* start, int64_t ns, float* out,
                               int64_t ldo, void* stream) {
  APR_CHECK_ARG(src && rev_t && start && out && ns > 0 && c > 0 && c % 4 == 0 && ldo >= c && ldo % 4 == 0 &&
                    ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0,
                "apr_reverse_gather: needs c %% 4 == 0 and 16-byte aligned rows");
  hipLaunchKernelGGL(k_rev_gather, dim3((unsigned)cdiv64(ns * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, c, rev_t,
                     start, ns, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

// apr_reverse_gather over the flat positions [t_lo, t_hi) of the table only: src f32 [t_hi - t_lo, c] holds their rows;
// accumulate != 0 continues the sums already in out.  Calling it chunk by chunk in ascending t (first chunk with
// accumulate = 0) gives the bits of one apr_reverse_gather over the whole table.
APR_API int apr_reverse_gather_range(const float* src, int32_t c, const int32_t* rev_t, const int32_t* start, int64_t ns,
                                     int64_t t_lo, int64_t t_hi, int32_t accumulate, float* out, int64_t ldo, void* stream) {
  APR_CHECK_ARG(src && rev_t && start && out && ns > 0 && c > 0 && c % 4 == 0 && ldo >= c && ldo % 4 == 0 &&
                    ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0,
                "apr_reverse_gather_range: needs c %% 4 == 0 and 16-byte aligned rows");
  APR_CHECK_ARG(t_lo >= 0 && t_hi >= t_lo && t_hi < (1ll << 31), "apr_reverse_gather_range: bad position range");
  hipLaunchKernelGGL(k_rev_gather_range, dim3((unsigned)cdiv64(ns * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, c,
                     rev_t, start, ns, (int)t_lo, (int)t_hi, accumulate, out, ldo);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
