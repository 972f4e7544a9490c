// micro-benchmark: do scattered stores issued just before a wave ends keep its slot occupied?  Sleep-bound single-wave
// workgroups (~13.5 us) that finish with NS scattered 256-B row stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NV, int NS, bool WAIT>
__global__ __launch_bounds__(64) void k_sleep(float* out, size_t rows, int iters) {
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
  size_t h = blockIdx.x * 2654435761u;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    h = h * 6364136223846793005ull + 1442695040888963407ull;
    out[((h >> 20) % rows) * 64 + threadIdx.x] = s + k;
  }
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NV, int NS, bool WAIT>
int run(float* d, size_t rows) {
  const int blocks = 1 << 18, iters = 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_sleep<NV, NS, WAIT>), dim3(blocks), dim3(64), 0, 0, d, rows, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_sleep<NV, NS, WAIT>), dim3(blocks), dim3(64), 0, 0, d, rows, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_sleep<NV, NS, WAIT>));
  const double life_us = iters * 127.0 * 64 / 2400.0;
  printf("regs %3d stores/wave %2d wait %d: %.3f ms, %.0f waves/us, sleep-only residency %.2f of limit %d, store BW %.2f TB/s\n", fa.numRegs, NS,
         (int)WAIT, ms, blocks / (ms * 1e3), blocks / (ms * 1e3) * life_us / 1024, 512 / ((fa.numRegs + 7) / 8 * 8),
         (double)blocks * NS * 256 / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const size_t rows = (size_t)1 << 24;   // 4 GiB of 256-B rows
  float* d; CK(hipMalloc(&d, rows * 256));
  if (run<100, 0, false>(d, rows) || run<100, 20, false>(d, rows) || run<100, 60, false>(d, rows) || run<100, 60, true>(d, rows) ||
      run<60, 20, false>(d, rows) || run<60, 60, false>(d, rows))
    return 1;
  return 0;
}
