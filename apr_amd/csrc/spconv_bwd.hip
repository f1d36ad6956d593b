// Weight gradient of the sparse convolution (FCGF_APR/lib/trainer.py:454-527 calls
// loss.backward() through MinkowskiConvolution / MinkowskiConvolutionTranspose):
//     dW[k][ci][co] = sum_j [nbr[j,k] >= 0]  in[nbr[j,k]][ci] * dout[j][co]
// (the input gradient needs no new kernel: it IS a sparse conv of dout with transposed weights over the reverse
// map — the same table with mirrored offsets for same-level convs, the transposed-conv table for strided ones).
//
// Two deterministic phases, no float atomics:
//   k_wgrad_partial: workgroup = (row chunk of 512 output rows, offset k, 64x64 block of dW[k]); wave w owns input
//       channels 16w..16w+15 and the 4 column blocks: per step of 4 rows one v_mfma_f32_16x16x4_f32 per column block
//       with A[m = ci][kk = row] = gathered input value (0 for an empty table entry) and B[kk = row][n = co] = dout;
//       the 64x64 partial sum goes to scratch[chunk][k][ci][co];
//   k_wgrad_reduce: dW = sum over chunks in ascending order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kChunkRows = 512;

__global__ __launch_bounds__(256) void k_wgrad_partial(const float* __restrict__ in, int64_t ldi,
                                                       const float* __restrict__ dout, int64_t ldo,
                                                       const int* __restrict__ nbr, int64_t n_out, int K, int cin,
                                                       int cout, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, r = lane >> 4;
  const int chunk = blockIdx.x, k = blockIdx.y;
  const int nco = (cout + 63) / 64;
  const int ci0 = (blockIdx.z / nco) * 64 + wave * 16, co0 = (blockIdx.z % nco) * 64;
  const int64_t j0 = (int64_t)chunk * kChunkRows;
  const int64_t j1 = min((long long)(j0 + kChunkRows), (long long)n_out);
  const bool ci_ok = ci0 + m < cin;
  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t jb = j0; jb < j1; jb += 16) {       // 4 MFMA steps per iteration: 4 x 6 loads in flight
    float a[4], b[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t j = jb + s * 4 + r;
      const bool row_ok = j < j1;
      const int idx = (nbr && row_ok) ? nbr[j * K + k] : (row_ok ? (int)j : -1);
      a[s] = (idx >= 0 && ci_ok) ? in[(int64_t)idx * ldi + ci0 + m] : 0.f;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int co = co0 + cb * 16 + m;
        b[s][cb] = (row_ok && co < cout) ? dout[j * ldo + co] : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s][cb], acc[cb], 0, 0, 0);
  }
  // D[row = 4 r + reg][col = m]: row = input channel within the wave's 16, col = output channel within the block
  float* dst = part + (((int64_t)chunk * K + k) * cin) * cout;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ci = ci0 + 4 * r + g, co = co0 + cb * 16 + m;
      if (ci < cin && co < cout) dst[(int64_t)ci * cout + co] = acc[cb][g];
    }
}

__global__ void k_wgrad_reduce(const float* __restrict__ part, int nchunk, int64_t per_chunk, float* __restrict__ dw) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(int64_t)c * per_chunk + t];
  dw[t] = s;
}

}  // namespace

APR_API size_t apr_spconv_wgrad_scratch_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  const int64_t nchunk = cdiv64(n_out > 0 ? n_out : 1, kChunkRows);
  return (size_t)nchunk * K * cin * cout * 4 + 256;
}

APR_API int apr_spconv_wgrad(const float* in, int64_t ldi, const float* dout, int64_t ldo, const int32_t* nbr,
                             int64_t n_out, int32_t K, int32_t cin, int32_t cout, float* dw, void* scratch,
                             size_t scratch_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APR_CHECK_ARG(n_out > 0 && K >= 1 && cin >= 1 && cout >= 1, "apr_spconv_wgrad: bad shape");
  APR_CHECK_ARG(nbr != nullptr || K == 1, "apr_spconv_wgrad: nbr may only be NULL for the identity map (K = 1)");
  APR_CHECK_ARG(scratch_bytes >= apr_spconv_wgrad_scratch_bytes(n_out, K, cin, cout), "apr_spconv_wgrad: scratch too small");
  const int64_t nchunk = cdiv64(n_out, kChunkRows);
  APR_CHECK_ARG(nchunk < 65536 * 32 && K <= 65535, "apr_spconv_wgrad: too many rows / offsets for one launch");
  float* part = (float*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  const unsigned nz = (unsigned)(((cin + 63) / 64) * ((cout + 63) / 64));
  hipLaunchKernelGGL(k_wgrad_partial, dim3((unsigned)nchunk, (unsigned)K, nz), dim3(256), 0, st, in, ldi, dout, ldo,
                     nbr, n_out, K, cin, cout, part);
  const int64_t per_chunk = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cdiv64(per_chunk, 256)), dim3(256), 0, st, part, (int)nchunk,
                     per_chunk, dw);
  APR_LAUNCH_CHECK();
  return APR_OK;
}
