// Adjacent components of the hot path: APG aggregation
// (rigid transform of the complement frames, crop to the key frame's radius, compaction) and the
// 1-NN squared-distance sums of the Chamfer loss.  Pure HBM-bound streaming + a tiled brute-force NN.
#include "common.h"

namespace {

constexpr int kBlock = 256;

// out = pts @ R^T + t  (FCGF_APR/lib/complement_data_loader.py:65-70, float32 like the reference)
__global__ void k_transform(const float* __restrict__ pts, int64_t n, const float* __restrict__ T /*[16] row-major*/,
                            float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  out[3 * i] = x * T[0] + y * T[1] + z * T[2] + T[3];
  out[3 * i + 1] = x * T[4] + y * T[5] + z * T[6] + T[7];
  out[3 * i + 2] = x * T[8] + y * T[9] + z * T[10] + T[11];
}

// max over points of |p|^2 (bit pattern of a non-negative float orders like an unsigned int)
__global__ void k_max_sqnorm(const float* __restrict__ pts, int64_t n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    m = fmaxf(m, x * x + y * y + z * z);
  }
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

// keep[i] = |p_i|^2 < limit ; per-block counts (compaction pass 1)
__global__ void k_crop_flags(const float* __restrict__ pts, int64_t n, const unsigned* __restrict__ limit_bits,
                             uint8_t* __restrict__ flags, int* __restrict__ block_counts) {
  __shared__ int wave_cnt[kBlock / 64];
  const float limit = __uint_as_float(*limit_bits);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = false;
  if (i < n) {
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    f = (x * x + y * y + z * z) < limit;
    flags[i] = f;
  }
  unsigned long long b = __ballot(f);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void k_scan_counts(const int* __restrict__ counts, int nblk, int* __restrict__ offsets, int* __restrict__ total) {
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nblk; base += blockDim.x) {
    int idx = base + threadIdx.x;
    int v = idx < nblk ? counts[idx] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_sum[w];
    int carry = carry_s;
    if (idx < nblk) offsets[idx] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ void k_compact_points(const float* __restrict__ pts, int64_t n, const uint8_t* __restrict__ flags,
                                 const int* __restrict__ block_offsets, float* __restrict__ out) {
  __shared__ int wave_cnt[kBlock / 64];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = (i < n) && flags[i];
  unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(b);
  __syncthreads();
  if (!f) return;
  int pos = block_offsets[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
  out[3 * (int64_t)pos] = pts[3 * i];
  out[3 * (int64_t)pos + 1] = pts[3 * i + 1];
  out[3 * (int64_t)pos + 2] = pts[3 * i + 2];
}

// sum_i min_j |a_i - b_j|^2 : one query per thread, targets tiled through LDS, fp64 sum of the minima
constexpr int kTile = 512;
__global__ __launch_bounds__(256) void k_nn3_min(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                 int64_t m, int chunk, unsigned* __restrict__ best_bits) {
  __shared__ float s_b[kTile * 3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const float x = live ? a[3 * i] : 0.f, y = live ? a[3 * i + 1] : 0.f, z = live ? a[3 * i + 2] : 0.f;
  float best = __builtin_inff();
  const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = min((long long)(t0 + chunk), (long long)m);
  for (int64_t tb = t0; tb < t1; tb += kTile) {
    const int rows = (int)min((long long)kTile, (long long)(t1 - tb));
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 3; e += 256) s_b[e] = b[tb * 3 + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float dx = x - s_b[3 * r], dy = y - s_b[3 * r + 1], dz = z - s_b[3 * r + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  if (live) atomicMin(&best_bits[i], __float_as_uint(best));
}

// the same search keeping the arg-min: packed (bits(d^2) << 32 | j), 64-bit atomicMin (ties: the smallest j)
__global__ __launch_bounds__(256) void k_nn3_arg(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                 int64_t m, int chunk, unsigned long long* __restrict__ best) {
  __shared__ float s_b[kTile * 3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const float x = live ? a[3 * i] : 0.f, y = live ? a[3 * i + 1] : 0.f, z = live ? a[3 * i + 2] : 0.f;
  float bd = __builtin_inff();
  unsigned bj = 0xFFFFFFFFu;
  const int64_t t0 = (int64_t)blockIdx.y * chunk, t1 = min((long long)(t0 + chunk), (long long)m);
  for (int64_t tb = t0; tb < t1; tb += kTile) {
    const int rows = (int)min((long long)kTile, (long long)(t1 - tb));
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 3; e += 256) s_b[e] = b[tb * 3 + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float dx = x - s_b[3 * r], dy = y - s_b[3 * r + 1], dz = z - s_b[3 * r + 2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < bd) {
        bd = d;
        bj = (unsigned)(tb + r);
      }
    }
  }
  if (live && bj != 0xFFFFFFFFu) atomicMin(&best[i], ((unsigned long long)__float_as_uint(bd) << 32) | bj);
}

// sum of the distances of a packed arg-min array in a FIXED order (one workgroup: lane-strided partials, LDS tree)
__global__ __launch_bounds__(1024) void k_sum_packed(const unsigned long long* __restrict__ best, int64_t n,
                                                     double* __restrict__ out) {
  __shared__ double part[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)__uint_as_float((unsigned)(best[i] >> 32));
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}

__global__ void k_sum_bits(const unsigned* __restrict__ bits, int64_t n, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (double)__uint_as_float(bits[i]);
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

APR_API int apr_transform_points(const float* pts, int64_t n, const float* T16_dev, float* out, void* stream) {
  APR_CHECK_ARG(n >= 0, "apr_transform_points: n < 0");
  if (n == 0) return APR_OK;
  hipLaunchKernelGGL(k_transform, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pts, n,
                     T16_dev, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API size_t apr_crop_scratch_bytes(int64_t n) {
  const int64_t nblk = cdiv64(n > 0 ? n : 1, kBlock);
  return align256(n) + 2 * align256(nblk * 4) + 512;
}

APR_API int apr_crop_to_radius(const float* key_pts, int64_t n_key, const float* pts, int64_t n, float* out,
                               int32_t* n_out_dev, void* scratch, size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_key > 0 && n >= 0 && n < (1ll << 31), "apr_crop_to_radius: bad sizes");
  APR_CHECK_ARG(scratch_bytes >= apr_crop_scratch_bytes(n), "apr_crop_to_radius: scratch too small");
  char* p = (char*)scratch;
  unsigned* limit = (unsigned*)p;
  p += 256;
  uint8_t* flags = (uint8_t*)p;
  p += align256(n);
  const int nblk = (int)cdiv64(n > 0 ? n : 1, kBlock);
  int* cnt = (int*)p;
  p += align256((size_t)nblk * 4);
  int* off = (int*)p;
  APR_HIP(hipMemsetAsync(limit, 0, 4, st));
  hipLaunchKernelGGL(k_max_sqnorm, dim3(256), dim3(kBlock), 0, st, key_pts, n_key, limit);
  if (n == 0) {
    APR_HIP(hipMemsetAsync(n_out_dev, 0, 4, st));
    return APR_OK;
  }
  hipLaunchKernelGGL(k_crop_flags, dim3(nblk), dim3(kBlock), 0, st, pts, n, limit, flags, cnt);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, cnt, nblk, off, n_out_dev);
  hipLaunchKernelGGL(k_compact_points, dim3(nblk), dim3(kBlock), 0, st, pts, n, flags, off, out);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_chamfer_sum(const float* a, int64_t n, const float* b, int64_t m, double* out_dev, void* scratch,
                            size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n > 0 && m > 0, "apr_chamfer_sum: empty cloud");
  APR_CHECK_ARG(scratch_bytes >= (size_t)n * 4, "apr_chamfer_sum: scratch too small (need 4*n bytes)");
  unsigned* best = (unsigned*)scratch;
  APR_HIP(hipMemsetAsync(best, 0x7F, (size_t)n * 4, st));   // 0x7F7F7F7F = 3.4e38 as float
  APR_HIP(hipMemsetAsync(out_dev, 0, 8, st));
  const int64_t qb = cdiv64(n, 256);
  int64_t want = cdiv64(2048, qb);
  int64_t chunk = cdiv64(cdiv64(m, want), kTile) * kTile;
  if (chunk < kTile) chunk = kTile;
  hipLaunchKernelGGL(k_nn3_min, dim3((unsigned)qb, (unsigned)cdiv64(m, chunk)), dim3(256), 0, st, a, n, b, m, (int)chunk,
                     best);
  hipLaunchKernelGGL(k_sum_bits, dim3(256), dim3(256), 0, st, best, n, out_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}

APR_API int apr_nn3(const float* a, int64_t n, const float* b, int64_t m, uint64_t* out_packed, double* sum_dev,
                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n > 0 && m > 0 && m < (1ll << 32) - 1, "apr_nn3: empty or oversized cloud");
  APR_HIP(hipMemsetAsync(out_packed, 0xFF, (size_t)n * 8, st));
  const int64_t qb = cdiv64(n, 256);
  int64_t want = cdiv64(2048, qb);
  int64_t chunk = cdiv64(cdiv64(m, want), kTile) * kTile;
  if (chunk < kTile) chunk = kTile;
  hipLaunchKernelGGL(k_nn3_arg, dim3((unsigned)qb, (unsigned)cdiv64(m, chunk)), dim3(256), 0, st, a, n, b, m, (int)chunk,
                     (unsigned long long*)out_packed);
  if (sum_dev) hipLaunchKernelGGL(k_sum_packed, dim3(1), dim3(1024), 0, st, (const unsigned long long*)out_packed, n, sum_dev);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
