// Microbenchmark: 64-MFMA bursts on 4 accumulators separated by an LDS read-modify-write, as in
// k_spconv_pairs' step, at 1/2/4 waves per SIMD.  Reports cycles per step per wave and pipe utilisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: bursts only, 1: + LDS RMW, 2: + RMW + varying operands from registers
__global__ void k(float* out, int steps, unsigned long long* cyc, const float* src) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* my = lds + wave * 32 * 68;
  for (int i = lane; i < 32 * 68; i += 64) my[i] = 0.f;
  f32x4 a[4], b[4][4];
  for (int j = 0; j < 4; ++j) {
    a[j] = *reinterpret_cast<const f32x4*>(src + (lane + j * 64) * 4);
    for (int c = 0; c < 4; ++c) b[c][j] = *reinterpret_cast<const f32x4*>(src + (lane + (j * 4 + c) * 64) * 4 + 1024);
  }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 2 ? b[c][j][t] : b[0][0][0], MODE == 2 ? a[j][t] : a[0][0],
                                                        acc[c], 0, 0, 0);
    if (MODE >= 1) {
      float* dst = my + ((lane * 7 + s) & 31) * 68 + (lane >> 4) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4* d4 = reinterpret_cast<f32x4*>(dst + c * 16);
        *d4 = *d4 + acc[c];
      }
    } else {
      asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (MODE == 0) out[threadIdx.x] = 0; else out[blockIdx.x * blockDim.x + threadIdx.x] = my[lane];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out, *src; unsigned long long *cyc, h;
  hipMalloc(&out, 8 << 20); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&cyc, 8);
  const int steps = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) {
    size_t lds = (threads / 64) * 32 * 68 * 4;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), lds, 0, out, steps, cyc, src);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double mfma_per_simd = (double)steps * 64 * (threads / 256);
      if (rep) printf("%4d thr (%d waves/SIMD) bursts + RMW, varying operands: %.1f us, %.1f ns per MFMA per SIMD (32 cycles @2.4GHz = 13.3 ns)\n",
             threads, threads / 256, ms * 1e3, ms * 1e6 / mfma_per_simd);
    }
  }
  return 0;
}
