// Microbenchmark: v_mfma_f32_16x16x32_bf16 rate (wall clock, chip-wide) for the two accumulation orders of the bf16
// 3-way split: CHAIN = 6 dependent MFMAs per accumulator back to back (k_dense_gemm_bf3 / k_ws_gemm_bf3), INTER =
// the same 24 MFMAs term-major over 4 accumulators, and with a split's worth of VALU work (44 ops per 24 MFMAs) mixed in.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_bf16_chain.hip -o /tmp/mfma_bf16_chain && /tmp/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0 chain, 1 interleaved, 2 chain + VALU, 3 interleaved + VALU
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 a[3], b[3];
  for (int i = 0; i < 3; ++i)
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (__bf16)(threadIdx.x * 0.001f + i + e);
      b[i][e] = (__bf16)(threadIdx.x * 0.002f - i + e);
    }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.37f + e;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);   // 40 VALU
    }
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        f32x4 t = acc[cb];
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2], a[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[2], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[0], t, 0, 0, 0);
        acc[cb] = t;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        const int wi = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
        const int ai = (term == 0 || term == 3 || term == 5) ? 0 : (term == 1) ? 2 : 1;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[wi], a[ai], acc[cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3}) {
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 24.0;
    const double cyc_per = ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * 24.0);   // SIMD cycles per MFMA at 2.4 GHz
    printf("%-22s %d wave/SIMD: %8.3f ms  %7.1f bf16 TFLOP/s (%.0f %% of 2517)  %.1f cycles/MFMA/SIMD\n", name, wgs_per_cu, ms,
           mfma * 16384.0 / ms / 1e9, mfma * 16384.0 / ms / 1e9 / 2517 * 100, cyc_per);
  }
}

#include "../../apr_amd/csrc/bf3.h"
// MODE 4: the kernel's inner loop shape -- per (step, 16-column block): 3 W fragments by ds_read_b128 from a 24 KB
// LDS image (prefetched two blocks ahead), 6 chained MFMAs.  MODE 5: + the real split of 8 fp32 per step (apr_split3).
// MODE 6: as 5 with TWO row groups per fragment set (G = 2: 12 MFMAs per 3 LDS reads).
template <int MODE>
__global__ __launch_bounds__(256) void k2(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char s_w[2][3 * 8192];
  for (int i = threadIdx.x; i < 2 * 3 * 8192 / 4; i += 256) reinterpret_cast<float*>(&s_w[0][0])[i] = i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;
  constexpr int G = MODE == 6 ? 2 : 1;
  f32x4 acc[G][4];
  for (int g = 0; g < G; ++g) for (int i = 0; i < 4; ++i) acc[g][i] = (f32x4){0, 0, 0, 0};
  f32x4 x0[G], x1[G];
  for (int g = 0; g < G; ++g) { x0[g] = (f32x4){lane * .1f, 1.f, 2.f, 3.f + g}; x1[g] = (f32x4){lane * .2f, 4.f, 5.f, 6.f + g}; }
  bf16x8 ah[G], am[G], al[G];
  for (int g = 0; g < G; ++g) apr_split3(x0[g], x1[g], ah[g], am[g], al[g]);
  for (int it = 0; it < iters; ++it) {
    const unsigned char* wb = &s_w[it & 1][frag_off];
    bf16x8 wf[3][3];
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wf[i0][pl] = *reinterpret_cast<const bf16x8*>(wb + (i0 * 16) * 64 + pl * 8192);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cb = i & 3;
      if (i < 6) {
        const int s2 = (i + 2) >> 2, cb2 = (i + 2) & 3;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          wf[(i + 2) % 3][pl] = *reinterpret_cast<const bf16x8*>(wb + (s2 * 64 + cb2 * 16) * 64 + pl * 8192);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE >= 5 && (i == 0 || i == 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          x0[g] += acc[g][0] * 1e-30f;      // a fresh input every step, as in the kernel
          apr_split3(x0[g], x1[g], ah[g], am[g], al[g]);
        }
      }
      const bf16x8 wh = wf[i % 3][0], wm = wf[i % 3][1], wl = wf[i % 3][2];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 t = acc[g][cb];
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[g], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[g], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[g], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[g], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[g], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[g], t, 0, 0, 0);
        acc[g][cb] = t;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sacc = 0;
  for (int g = 0; g < G; ++g) for (int i = 0; i < 4; ++i) sacc += acc[g][i][0] + acc[g][i][1] + acc[g][i][2] + acc[g][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

// MODE 7: G = 2, per 32-channel step ONE scheduling region: the 12 fragment reads, the 48 MFMAs of the step and the
// split of the NEXT step's operands, interleaved by sched_group_barrier as (1 MFMA, 2 VALU) x 44 with a fragment read
// every 4 MFMAs -- the vector pipe works in the 12 spare issue cycles behind every 16-cycle MFMA of the same wave.
__global__ __launch_bounds__(256) void k3(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char s_w[2][3 * 8192];
  for (int i = threadIdx.x; i < 2 * 3 * 8192 / 4; i += 256) reinterpret_cast<float*>(&s_w[0][0])[i] = i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int frag_off = (r16 * 4 + ((r16 & 8) ? (q ^ 3) : q)) * 16;
  constexpr int G = 2;
  f32x4 acc[G][4];
  for (int g = 0; g < G; ++g) for (int i = 0; i < 4; ++i) acc[g][i] = (f32x4){0, 0, 0, 0};
  f32x4 x0[G], x1[G];
  for (int g = 0; g < G; ++g) { x0[g] = (f32x4){lane * .1f, 1.f, 2.f, 3.f + g}; x1[g] = (f32x4){lane * .2f, 4.f, 5.f, 6.f + g}; }
  bf16x8 ah[2][G], am[2][G], al[2][G];
  for (int g = 0; g < G; ++g) apr_split3(x0[g], x1[g], ah[0][g], am[0][g], al[0][g]);
  for (int it = 0; it < iters; ++it) {
    const unsigned char* wb = &s_w[it & 1][frag_off];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cu = s & 1, nx = cu ^ 1;
      bf16x8 wf[4][3];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[cb][pl] = *reinterpret_cast<const bf16x8*>(wb + (s * 64 + cb * 16) * 64 + pl * 8192);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        x0[g] += acc[g][0] * 1e-30f;
        apr_split3(x0[g], x1[g], ah[nx][g], am[nx][g], al[nx][g]);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          f32x4 t = acc[g][cb];
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][2], ah[cu][g], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][0], al[cu][g], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][1], am[cu][g], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][1], ah[cu][g], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][0], am[cu][g], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][0], ah[cu][g], t, 0, 0, 0);
          acc[g][cb] = t;
        }
      // 3 fragment reads up front, then per 4 MFMAs: one more read; per MFMA: 2 VALU
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if ((i & 3) == 3 && i < 36) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sacc = 0;
  for (int g = 0; g < G; ++g) for (int i = 0; i < 4; ++i) sacc += acc[g][i][0] + acc[g][i][1] + acc[g][i][2] + acc[g][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

void run3(float* out, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 3}) {
    const int iters = 10000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 96.0;
    printf("%-22s %d WG/CU: %8.3f ms  %7.1f bf16 TFLOP/s (%.0f %% of 2517) = %.0f fp32-equivalent TFLOP/s\n", name, wgs_per_cu, ms,
           mfma * 16384.0 / ms / 1e9, mfma * 16384.0 / ms / 1e9 / 2517 * 100, mfma * 16384.0 / ms / 1e9 / 6);
  }
}

template <int MODE>
void run2(float* out, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  constexpr int G = MODE == 6 ? 2 : 1;
  for (int wgs_per_cu : {1, 2, 3}) {
    const int iters = 10000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k2<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k2<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 48.0 * G;
    printf("%-22s %d WG/CU: %8.3f ms  %7.1f bf16 TFLOP/s (%.0f %% of 2517) = %.0f fp32-equivalent TFLOP/s\n", name, wgs_per_cu, ms,
           mfma * 16384.0 / ms / 1e9, mfma * 16384.0 / ms / 1e9 / 2517 * 100, mfma * 16384.0 / ms / 1e9 / 6);
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 16 << 20);
  run2<4>(out, "LDS frags + chain");
  run2<5>(out, "LDS frags + split");
  run2<6>(out, "same, G = 2");
  run3(out, "G = 2, interleaved");
  run<0>(out, "chain of 6");
  run<1>(out, "term-major x4 acc");
  run<2>(out, "chain + 40 VALU");
  run<3>(out, "term-major + 40 VALU");
  return 0;
}
