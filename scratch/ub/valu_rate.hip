// micro-benchmark: VALU issue rate per SIMD for v_fma_f32 / v_pk_fma_f32 / v_fmac_f32_dpp at 1..8 waves per SIMD,
// plus a semantic check of DPP row_newbcast.  scratch tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k_rate(float* out, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  float b = 1.0001f, c = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      } else if (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                     "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
      } else {
        asm volatile("v_fmac_f32_dpp %0, %8, %9 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %8, %9 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %2, %8, %9 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %3, %8, %9 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %4, %8, %9 row_ror:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %5, %8, %9 row_ror:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %6, %8, %9 row_ror:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %7, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      }
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (s == 123.456f) out[0] = s;
}

__global__ void k_bcast(const float* in, float* out) {
  float v = in[threadIdx.x], w = 1.f, acc = 0.f;
  asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w));
  out[threadIdx.x] = acc;
  float acc2 = 0.f;
  asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc2) : "v"(v), "v"(w));
  out[64 + threadIdx.x] = acc2;
}

template <int MODE>
int run(const char* name, float* d, int wps) {
  // wps waves per SIMD: single-wave blocks, 256 CUs x 4 SIMDs x wps blocks (one round, all resident)
  const int iters = 2000, blocks = 256 * 4 * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)blocks * iters * 16 * 8;
  // cycles per wave-instruction per SIMD at 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 * 1024 / instr;
  printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (2.4 GHz assumed)\n", name, wps, ms, cyc);
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  for (int wps : {1, 2, 4, 8}) {
    if (run<0>("v_fma_f32", d, wps)) return 1;
    if (run<1>("v_pk_fma_f32", d, wps)) return 1;
    if (run<2>("v_fmac_f32_dpp", d, wps)) return 1;
  }
  std::vector<float> h(64), o(128);
  for (int i = 0; i < 64; ++i) h[i] = i;
  float *di, *dout; CK(hipMalloc(&di, 256)); CK(hipMalloc(&dout, 512));
  CK(hipMemcpy(di, h.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_bcast, dim3(1), dim3(64), 0, 0, di, dout);
  CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
  printf("row_newbcast:5 ->"); for (int i = 0; i < 64; ++i) printf(" %g", o[i]); printf("\n");
  printf("row_ror:3      ->"); for (int i = 0; i < 64; ++i) printf(" %g", o[64 + i]); printf("\n");
  return 0;
}
