// Microbenchmark: issue rate of fp32 MFMA shapes on gfx950 (one or more waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k32(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(k16<4>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4 f32, 4 acc, %4d thr/WG: %.1f cycles per MFMA per wave\n", threads, (double)h / (iters * 64.0));
    hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4 f32, 1 acc, %4d thr/WG: %.1f cycles per MFMA per wave\n", threads, (double)h / (iters * 16.0));
    hipLaunchKernelGGL(k32, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("32x32x2 f32, 2 acc, %4d thr/WG: %.1f cycles per MFMA per wave\n", threads, (double)h / (iters * 32.0));
  }
  return 0;
}
