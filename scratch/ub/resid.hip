// micro-benchmark: how many single-wave workgroups stay resident per SIMD (sleep-bound waves), with and without LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS_FLOATS>
__global__ __launch_bounds__(64) void k_sleep(float* out, int iters) {
  __shared__ float buf[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  if (LDS_FLOATS > 0) buf[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  if (LDS_FLOATS > 0 && buf[threadIdx.x] == 12345.f) out[0] = 1.f;
}

template <int LDS_FLOATS>
int run(float* d, int wps) {
  const int blocks = 256 * 4 * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_sleep<LDS_FLOATS>, dim3(blocks), dim3(64), 0, 0, d, 20);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_sleep<LDS_FLOATS>, dim3(blocks), dim3(64), 0, 0, d, 200);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("lds %5d B  blocks/SIMD %2d: %.3f ms\n", LDS_FLOATS * 4, wps, ms);
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  for (int wps : {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16}) if (run<0>(d, wps)) return 1;
  for (int wps : {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16}) if (run<1536>(d, wps)) return 1;
  return 0;
}
