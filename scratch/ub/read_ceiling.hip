// scratch: read-only streaming ceiling: every thread sums float4s with U loads in flight; grid-stride over 12.8 GB
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ __launch_bounds__(256) void kread(const float4* __restrict__ X, long n4, float* __restrict__ out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = X[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < n4; i += stride) { float4 v = X[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int U> void run(const float4* X, long n4, float* out, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kread<U>, dim3(grid), dim3(256), 0, 0, X, n4, out); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kread<U>, dim3(grid), dim3(256), 0, 0, X, n4, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("U %d grid %d: %.3f ms  %.2f TB/s\n", U, grid, ms, (double)n4 * 16 / ms / 1e9);
}
int main() {
  const long n4 = 800000000L;  // 12.8 GB
  float4* X; float* out; hipMalloc(&X, n4 * 16); hipMalloc(&out, 4); hipMemset(X, 0, n4 * 16);
  for (int grid : {2048, 8192, 32768}) { run<1>(X, n4, out, grid); run<4>(X, n4, out, grid); run<8>(X, n4, out, grid); }
  return 0;
}
