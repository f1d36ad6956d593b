// Build: hipcc --offload-arch=gfx950 -O2 -w -o /tmp/lds_occupancy scripts/ubench/lds_occupancy.hip
// How many workgroups of 256 threads does a CU hold as a function of their LDS footprint?  Every workgroup spins for a
// fixed number of clock ticks; 256 CUs x NPER workgroups are launched; elapsed / spin = rounds = NPER / residency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int BYTES>
__global__ __launch_bounds__(256) void k_spin_static(long long ticks, int* sink) {
  __shared__ int s[BYTES / 4];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (s[(threadIdx.x * 7) & 255] == -1) sink[0] = 1;
}

__global__ __launch_bounds__(256) void k_spin_dyn(long long ticks, int* sink) {
  extern __shared__ int sd[];
  sd[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sd[(threadIdx.x * 7) & 255] == -1) sink[0] = 1;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  int* sink;
  hipMalloc(&sink, 4);
  const int NPER = 12, NCU = 256;
  const long long ticks = 100 * 100;          // wall_clock64 runs at 100 MHz: 100 us
  const dim3 grid(NCU * NPER), block(256);
#define RUN_STATIC(B)                                                                                      \
  {                                                                                                        \
    float ms = time_ms([&] { hipLaunchKernelGGL(k_spin_static<B>, grid, block, 0, 0, ticks, sink); });    \
    printf("static %6d B: %7.3f ms -> %.1f rounds -> %.2f workgroups per CU\n", B, ms, ms / 0.1, NPER / (ms / 0.1)); \
  }
  RUN_STATIC(1024) RUN_STATIC(16384) RUN_STATIC(24576) RUN_STATIC(32768) RUN_STATIC(40960) RUN_STATIC(49152) RUN_STATIC(65536)
  const int dyn[] = {16384, 32768, 49152, 65536, 81920, 98304, 131072, 163840};
  for (int b : dyn) {
    if (hipFuncSetAttribute((const void*)k_spin_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, b) != hipSuccess) {
      printf("dynamic %6d B: opt-in refused\n", b);
      continue;
    }
    float ms = time_ms([&] { hipLaunchKernelGGL(k_spin_dyn, grid, block, b, 0, ticks, sink); });
    if (hipGetLastError() != hipSuccess) { printf("dynamic %6d B: launch failed\n", b); continue; }
    printf("dynamic %6d B: %7.3f ms -> %.1f rounds -> %.2f workgroups per CU\n", b, ms, ms / 0.1, NPER / (ms / 0.1));
  }
  return 0;
}
