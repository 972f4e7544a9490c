// micro-benchmark: achieved residency of short (~12 us) single-wave workgroups by VGPR count (LDS 4608 B)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NV>
__global__ __launch_bounds__(64) void k_sleep(float* out, int iters) {
  __shared__ float buf[1152];
  buf[threadIdx.x] = threadIdx.x;
  float r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = threadIdx.x + i;
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(r[j]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += r[i];
  if (buf[threadIdx.x] + s == 12345.f) out[0] = 1.f;
}

template <int NV>
int run(float* d) {
  const int blocks = 1 << 18, iters = 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_sleep<NV>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_sleep<NV>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_sleep<NV>));
  const double life_us = iters * 127.0 * 64 / 2400.0;   // s_sleep 127 = 127 * 64 cycles
  printf("regs %3d: %.3f ms, %.0f waves/us, residency ~%.2f waves/SIMD (limit %d)\n", fa.numRegs, ms, blocks / (ms * 1e3),
         blocks / (ms * 1e3) * life_us / 1024, 512 / ((fa.numRegs + 7) / 8 * 8));
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  if (run<8>(d) || run<60>(d) || run<100>(d) || run<110>(d) || run<150>(d)) return 1;
  return 0;
}
