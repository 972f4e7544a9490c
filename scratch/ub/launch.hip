// micro-benchmark: workgroup dispatch rate of single-wave workgroups (near-empty kernel), by VGPR count and LDS use
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS_FLOATS, int SLEEP>
__global__ __launch_bounds__(64) void k_empty(float* out) {
  __shared__ float buf[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  if (LDS_FLOATS > 0) buf[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < SLEEP; ++i) __builtin_amdgcn_s_sleep(127);
  if (LDS_FLOATS > 0 && buf[threadIdx.x] == 12345.f) out[0] = 1.f;
}

template <int LDS_FLOATS, int SLEEP>
int run(float* d, int blocks, int tpb) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_empty<LDS_FLOATS, SLEEP>), dim3(blocks), dim3(tpb), 0, 0, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_empty<LDS_FLOATS, SLEEP>), dim3(blocks), dim3(tpb), 0, 0, d);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("lds %5d B sleep %2d tpb %3d blocks %8d: %.3f ms = %.0f waves/us\n", LDS_FLOATS * 4, SLEEP, tpb, blocks, ms,
         (double)blocks * (tpb / 64) / (ms * 1e3));
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  if (run<0, 0>(d, 1 << 20, 64)) return 1;
  if (run<1536, 0>(d, 1 << 20, 64)) return 1;
  if (run<0, 0>(d, 1 << 18, 256)) return 1;
  if (run<1536, 0>(d, 1 << 18, 256)) return 1;
  // waves of ~12 us (sleep 127*64 cycles = 8k cycles each x 4)
  if (run<1536, 4>(d, 1 << 18, 64)) return 1;
  if (run<1536, 4>(d, 1 << 16, 256)) return 1;
  return 0;
}
