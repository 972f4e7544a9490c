// Probe of the f32 MFMA operand / result lane maps on gfx950 with exact integer data:
//   v_mfma_f32_16x16x4_f32      (one block)   and   v_mfma_f32_16x16x1_4b_f32 (four blocks, K = 1)
// plus v_permlane32_swap / v_permlane16_swap semantics.  Prints OK / the first mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_probe(const float* A, const float* B, float* D1, float* D4, unsigned* perm) {
  const int l = threadIdx.x, g = l >> 4, x = l & 15;
  // 16x16x4: A[i][k] (16x4) at lane (g = k, x = i); B[k][j] at lane (g = k, x = j)
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[x * 4 + g], B[g * 16 + x], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D1[(l * 4 + r)] = c[r];
  // 16x16x1 4 blocks: A[blk][i] at lane (g = blk, x = i), B[blk][j] at lane (g = blk, x = j)
  f16v c4;
  for (int r = 0; r < 16; ++r) c4[r] = 0.f;
  c4 = __builtin_amdgcn_mfma_f32_16x16x1f32(A[64 + l], B[64 + l], c4, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D4[l * 16 + r] = c4[r];
  // permlane swaps: both operands = lane id + tag
  unsigned a = 1000 + l, b = 2000 + l;
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  perm[l * 4 + 0] = r32[0]; perm[l * 4 + 1] = r32[1];
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  perm[l * 4 + 2] = r16[0]; perm[l * 4 + 3] = r16[1];
}

int main() {
  float hA[128], hB[128];
  for (int i = 0; i < 128; ++i) { hA[i] = (float)(1 + (i * 7) % 13); hB[i] = (float)(2 + (i * 5) % 11); }
  float *dA, *dB, *d1, *d4; unsigned* dp;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&d1, 256 * 4); hipMalloc(&d4, 1024 * 4); hipMalloc(&dp, 256 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, d1, d4, dp);
  float h1[256], h4[1024]; unsigned hp[256];
  hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost); hipMemcpy(h4, d4, sizeof(h4), hipMemcpyDeviceToHost);
  hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  int bad = 0;
  // expected 16x16x4: D[i][j] = sum_k A[i][k] B[k][j], at lane (g, x = j), reg r: i = 4g + r
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int i = 4 * (l >> 4) + r, j = l & 15; float e = 0;
    for (int k = 0; k < 4; ++k) e += hA[i * 4 + k] * hB[k * 16 + j];
    if (e != h1[l * 4 + r]) { if (bad++ < 5) printf("16x16x4 mismatch lane %d reg %d: %g vs %g\n", l, r, h1[l * 4 + r], e); }
  }
  printf("16x16x4 map: %s\n", bad ? "MISMATCH" : "OK");
  int bad4 = 0;
  // expected 4B: block b: D_b[i][j] = A[b][i] * B[b][j], at reg 4b + r, lane (g, x = j): i = 4g + r
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) {
    int b = q >> 2, r = q & 3, i = 4 * (l >> 4) + r, j = l & 15;
    float e = hA[64 + b * 16 + i] * hB[64 + b * 16 + j];
    if (e != h4[l * 16 + q]) { if (bad4++ < 5) printf("4B mismatch lane %d reg %d: %g vs %g\n", l, q, h4[l * 16 + q], e); }
  }
  printf("16x16x1_4B map: %s\n", bad4 ? "MISMATCH" : "OK");
  // permlane32_swap(a, b): expect r[0] = (a.lo, b.lo), r[1] = (a.hi, b.hi)?  print what it is
  printf("permlane32_swap lane0: %u %u  lane40: %u %u\n", hp[0], hp[1], hp[160], hp[161]);
  printf("permlane16_swap lane0: %u %u  lane20: %u %u  lane40: %u %u lane 60: %u %u\n", hp[2], hp[3], hp[82], hp[83], hp[162], hp[163], hp[242], hp[243]);
  return (bad || bad4) ? 2 : 0;
}
