// Microbenchmark: sustained chip-wide fp32 MFMA rate in WALL-CLOCK terms (hipEvents), i.e. what fraction of the
// 157.3 TFLOP/s datasheet figure (256 FLOP/clk/CU x 256 CUs x 2.4 GHz) the part holds under load.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_wall.hip -o /tmp/mfma_wall && /tmp/mfma_wall
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 16 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3, 4}) {
    for (int iters : {2000, 20000}) {
      const int grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 /*waves*/ * iters * 64.0 * 2048.0;
      printf("%d WG/CU, %6d iters: %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", wgs_per_cu, iters, ms,
             flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    }
  }
  return 0;
}
