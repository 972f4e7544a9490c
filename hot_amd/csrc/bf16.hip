// bf16-STORAGE variants of the row primitives around the fused attention core (hotk_attn22_args.storage =
// HOTK_STORAGE_BF16): the wide entry-level tensors h2 [R, 8 dqk + D] and its gradient are bf16 in memory, every sum is fp32.
// Reference: the same code as the fp32 forms (models/sparse/linear.py:143-161,293-306,345-346 for the assembly,
// :187-231 and the gather backward for the segment sums); the reference itself trains in fp32 only.
// Four consecutive channels per thread (8-byte loads / stores): C, ld multiples of 4, 8-byte aligned rows.
#include "common.h"

namespace {

typedef __bf16 bf16;

__device__ __forceinline__ float4 ld4b(const bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4b(bf16* p, const float4& v) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  const bf16x2 a = __builtin_convertvector((f2v){v.x, v.y}, bf16x2), b = __builtin_convertvector((f2v){v.z, v.w}, bf16x2);
  uint2 u;
  u.x = __builtin_bit_cast(uint32_t, a);
  u.y = __builtin_bit_cast(uint32_t, b);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// k_assemble22 (prims.hip) on a bf16 z: the node-level products yn and the biases stay fp32, the sum is formed in fp32
// and rounded once
__global__ void k_assemble22_b(bf16* __restrict__ z, int64_t ld, const float* __restrict__ yn, int64_t ld_yn,
                               const uint8_t* __restrict__ kind, const int32_t* __restrict__ rown,
                               const int32_t* __restrict__ coln, const float* __restrict__ b_off,
                               const float* __restrict__ b_diag, int64_t R, int64_t C) {
  const int64_t CV = C / 4, total = R * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / CV, c = (t - r * CV) * 4;
    const uint8_t kd = kind[r];
    bf16* p = z + r * ld + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kd == 2) {
      v = ld4b(p);
      v = add4(v, *reinterpret_cast<const float4*>(yn + (int64_t)rown[r] * ld_yn + C + c));
      v = add4(v, *reinterpret_cast<const float4*>(yn + (int64_t)coln[r] * ld_yn + 2 * C + c));
      if (b_off) v = add4(v, *reinterpret_cast<const float4*>(b_off + c));
    } else if (kd == 1) {
      v = *reinterpret_cast<const float4*>(yn + (int64_t)rown[r] * ld_yn + c);
      if (b_diag) v = add4(v, *reinterpret_cast<const float4*>(b_diag + c));
    }
    st4b(p, v);
  }
}

// k_seg_sum_rows (prims.hip) with a bf16 source: rows are added in segment order in fp32, the result rows are fp32
constexpr int SU = 8;
// PERM / KIND compile-time and opaque uses of the fetched values, as in prims.hip: the index, kind and row loads of a
// step are issued together instead of one dependent round trip each
template <bool PERM, bool KIND>
__global__ void k_seg_sum_rows_b(const bf16* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ seg_begin,
                                 const int32_t* __restrict__ seg_end, const int32_t* __restrict__ perm,
                                 const uint8_t* __restrict__ kind, unsigned keep, float* __restrict__ out, int64_t ld_out,
                                 int64_t n_seg, int64_t CV) {
  const int64_t total = n_seg * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / CV, cv = t - s * CV;
    const int64_t lo = seg_begin[s], hi = seg_end[s];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p = lo; p < hi; p += SU) {
      int pr[SU], kk[SU];
      float4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int64_t pp = p + u < hi ? p + u : hi - 1;
        pr[u] = PERM ? perm[pp] : (int)pp;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if (PERM) asm volatile("" : "+v"(pr[u]));
#pragma unroll
      for (int u = 0; u < SU; ++u) kk[u] = KIND ? (int)kind[pr[u]] : 0;
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = ld4b(src + (int64_t)pr[u] * ld_src + cv * 4);
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (KIND) asm volatile("" : "+v"(kk[u]));
        asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if ((p + u < hi) && (!KIND || hotk_keep(keep, (uint8_t)kk[u]))) acc = add4(acc, v[u]);
    }
    *reinterpret_cast<float4*>(out + s * ld_out + cv * 4) = acc;
  }
}

bool aligned8(const void* p) { return (((uintptr_t)p) & 7u) == 0; }

}  // namespace

extern "C" int hotk_assemble22_bf16(void* z, int64_t ld, const float* yn, int64_t ld_yn, const uint8_t* kind,
                                    const int32_t* rown, const int32_t* coln, const float* b_off, const float* b_diag,
                                    int64_t R, int64_t C, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && C >= 0 && ld >= C && ld_yn >= 3 * C);
  if (R == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(z && yn && kind && rown && coln);
  if ((C & 3) || (ld & 3) || (ld_yn & 3) || !aligned8(z) || !hotk_aligned16(yn) || (b_off && !hotk_aligned16(b_off)) ||
      (b_diag && !hotk_aligned16(b_diag)))
    return HOTK_EUNSUPPORTED;
  hipLaunchKernelGGL(k_assemble22_b, dim3(hotk_grid(R * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, (bf16*)z, ld, yn,
                     ld_yn, kind, rown, coln, b_off, b_diag, R, C);
  return hotk_launch_status();
}

extern "C" int hotk_seg_sum_rows_bf16(const void* src, int64_t ld_src, const int32_t* seg_begin, const int32_t* seg_end,
                                      const int32_t* perm, const uint8_t* kind, unsigned keep, float* out, int64_t ld_out,
                                      int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream) {
  HOTK_CHECK_ARG(n_seg >= 0 && C >= 0 && ld_src >= C && ld_out >= C && max_seg_len >= 0);
  if (n_seg == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(src && out && seg_begin && seg_end);
  if (max_seg_len > HOTK_SEG_CHUNK) return HOTK_EUNSUPPORTED;   // as hotk_seg_sum_rows: the caller splits longer segments
  if ((C & 3) || (ld_src & 3) || (ld_out & 3) || !aligned8(src) || !hotk_aligned16(out)) return HOTK_EUNSUPPORTED;
  const dim3 grid(hotk_grid(n_seg * (C / 4), 256)), block(256);
#define HOTK_SSB(PP, KK)                                                                                              \
  hipLaunchKernelGGL((k_seg_sum_rows_b<PP, KK>), grid, block, 0, (hipStream_t)stream, (const bf16*)src, ld_src, seg_begin, \
                     seg_end, perm, kind, keep, out, ld_out, n_seg, C / 4)
  if (perm) {
    if (kind) HOTK_SSB(true, true); else HOTK_SSB(true, false);
  } else {
    if (kind) HOTK_SSB(false, true); else HOTK_SSB(false, false);
  }
#undef HOTK_SSB
  return hotk_launch_status();
}
