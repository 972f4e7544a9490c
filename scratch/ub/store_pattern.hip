// scratch: write-only kernels over Y[R, 320] floats: (A) the thin-GEMM epilogue pattern -- per instruction 8 rows x 128 B,
// ten column tiles per 32-row block; (B) the same bytes as whole rows (1 KB contiguous per instruction); (C) pattern A with
// 4 rows x 256 B per instruction (two tiles staged).  Prints ms and TB/s for R rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int C = 320;
__global__ __launch_bounds__(512) void kA(float* __restrict__ Y, long R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, er = lane >> 3, ec = 4 * (lane & 7);
  const long tiles = (R + 255) / 256;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long row0 = t * 256 + wave * 32;
    for (int n0 = 0; n0 < C; n0 += 32)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long r = row0 + 8 * i + er;
        if (r < R) *reinterpret_cast<float4*>(Y + r * C + n0 + ec) = make_float4(1.f, 2.f, 3.f, (float)n0);
      }
  }
}
__global__ __launch_bounds__(512) void kB(float* __restrict__ Y, long R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long tiles = (R + 255) / 256;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long row0 = t * 256 + wave * 32;
    if (row0 + 32 > R) continue;
    float* p = Y + row0 * C;
    for (int i = 0; i < 32 * C / 256; ++i) *reinterpret_cast<float4*>(p + i * 256 + 4 * lane) = make_float4(1.f, 2.f, 3.f, (float)i);
  }
}
__global__ __launch_bounds__(512) void kC(float* __restrict__ Y, long R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, er = lane >> 4, ec = 4 * (lane & 15);
  const long tiles = (R + 255) / 256;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long row0 = t * 256 + wave * 32;
    for (int n0 = 0; n0 < C; n0 += 64)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long r = row0 + 4 * i + er;
        if (r < R) *reinterpret_cast<float4*>(Y + r * C + n0 + ec) = make_float4(1.f, 2.f, 3.f, (float)n0);
      }
  }
}
template <class K> void run(const char* name, K k, float* Y, long R, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, Y, R);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, Y, R);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%s grid %d: %.3f ms  %.2f TB/s\n", name, grid, ms, (double)R * C * 4 / ms / 1e9);
}
int main(int argc, char** argv) {
  const long R = argc > 1 ? atol(argv[1]) : 10000384;
  float* Y; if (hipMalloc(&Y, (size_t)R * C * 4) != hipSuccess) return 1;
  for (int grid : {512, 1024, 4096}) { run("A tiles 8x128B ", kA, Y, R, grid); run("B whole rows    ", kB, Y, R, grid); run("C tiles 4x256B ", kC, Y, R, grid); }
  hipFree(Y); return 0;
}
