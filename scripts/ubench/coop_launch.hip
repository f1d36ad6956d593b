// Host cost and GPU gap of hipLaunchCooperativeKernel against a plain launch (is a grid barrier inside one kernel cheaper than
// a second launch?).  hipcc --offload-arch=gfx950 -O3 coop_launch.hip -o coop_launch && ./coop_launch
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k_plain(float* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1.f;
}
__global__ void k_coop(float* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1.f;
  cg::this_grid().sync();
  if (i < n) x[n - 1 - i] += 1.f;
}
int main() {
  float* x;
  int n = 512 * 256;
  hipMalloc(&x, n * 4);
  hipMemset(x, 0, n * 4);
  hipStream_t st;
  hipStreamCreate(&st);
  const int reps = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) {
        hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, st, x, n);
      } else if (mode == 1) {
        hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, st, x, n);
        hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, st, x, n);
      } else {
        void* args[] = {&x, &n};
        hipLaunchCooperativeKernel((const void*)k_coop, dim3(512), dim3(256), args, 0, st);
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    printf("%s: host %.2f us per iteration, stream %.2f us per iteration\n",
           mode == 0 ? "one plain launch" : mode == 1 ? "two plain launches" : "one cooperative launch (grid sync inside)", host_us,
           ms * 1e3 / reps);
  }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
