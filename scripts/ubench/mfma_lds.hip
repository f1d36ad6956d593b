// Microbenchmark: the inner step of k_ws_gemm without global memory — per step 4 x (4 ds_read_b128 of B fragments,
// 16 v_mfma_f32_16x16x4_f32 on 4 accumulators) with the A fragments in registers — at 1/2/3 waves per SIMD.
// Reports ns per MFMA per SIMD (13.3 ns = 32 cycles at 2.4 GHz = the pipe's peak).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: B in registers; 1: B from LDS, next j prefetched (as k_ws_gemm); 2: B from LDS, loaded at use
__global__ __launch_bounds__(256) void k(float* out, int steps, const float* src) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];   // [16][64][4] floats = 16 KB
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < 16 * 64 * 4; i += blockDim.x) s_w[i] = src[i];
  f32x4 a[4];
  for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(src + 8192 + (lane + j * 64) * 4);
  __syncthreads();
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0, 0, 0, 0};
  f32x4 breg[4][4];
  if (MODE == 0)
    for (int j = 0; j < 4; ++j)
      for (int cb = 0; cb < 4; ++cb) breg[j][cb] = *reinterpret_cast<const f32x4*>(s_w + ((j * 4 + q) * 64 + cb * 16 + r16) * 4);
  for (int s = 0; s < steps; ++s) {
    f32x4 bq[2][4];
    if (MODE == 1) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bq[0][cb] = *reinterpret_cast<const f32x4*>(s_w + (q * 64 + cb * 16 + r16) * 4);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 1 && j < 3) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          bq[(j + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(s_w + (((j + 1) * 4 + q) * 64 + cb * 16 + r16) * 4);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 2) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          bq[j & 1][cb] = *reinterpret_cast<const f32x4*>(s_w + ((j * 4 + q) * 64 + cb * 16 + r16) * 4);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 0 ? breg[j][cb][t] : bq[j & 1][cb][t], a[j][t], acc[cb], 0, 0, 0);
    }
    if (MODE != 0) asm volatile("" ::: "memory");
  }
  float r = 0;
  for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* what, float* out, const float* src) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 4000;
  for (int wg_per_cu : {1, 2, 3}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 16 * 1024, 0, out, steps, src);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double mfma_per_simd = (double)steps * 64 * wg_per_cu;
    printf("%-34s %d wave(s)/SIMD: %7.1f us  %5.1f ns per MFMA per SIMD  (%.0f %% of peak)\n", what, wg_per_cu, best * 1e3,
           best * 1e6 / mfma_per_simd, 13.33 / (best * 1e6 / mfma_per_simd) * 100);
  }
}

int main() {
  float *out, *src;
  hipMalloc(&out, 8 << 20);
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  run<0>("B in registers", out, src);
  run<1>("B from LDS, prefetched one j ahead", out, src);
  run<2>("B from LDS, loaded at use", out, src);
  return 0;
}
