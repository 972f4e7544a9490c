// Segmented softmax pooling: the linear-cost order-0 readout of the softmax attention
// (hot_pytorch/models/sparse/attncoef.py:25-59 `_vector_query` + apply_attn with a zeroth-order query,
// attncoef.py:165-173; used by SelfAttn._1_to_0 / _2_to_0, sparse/attn.py:82-131).
//
//   s[r][h]     = q[b][h*d : (h+1)*d] . k[r][h*d : (h+1)*d] / sqrt(d)          for the kept rows r of segment b
//   alpha[r][h] = softmax over the kept rows of segment b (0 on every other row; all 0 for an empty set)
//   out[b][h*c + ch] = sum_r alpha[r][h] * v[r][h*c + ch]
//
// One wavefront per (segment, head); rows are lane-strided, so a lane only ever re-reads alpha entries it wrote
// itself.  alpha [n_rows, H] is an output (saved for the backward pass).  Segments are graphs here: tens of rows in
// the PCQM-like configurations, so the kernel favours simplicity over bandwidth.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int CT = 8;  // channel tile

__global__ __launch_bounds__(64) void k_softmax_pool_fwd(const float* __restrict__ q, int64_t ld_q,
                                                         const float* __restrict__ k, int64_t ld_k,
                                                         const float* __restrict__ v, int64_t ld_v,
                                                         const int32_t* __restrict__ seg_begin,
                                                         const int32_t* __restrict__ seg_end,
                                                         const uint8_t* __restrict__ kind, unsigned keep,
                                                         float* __restrict__ out, int64_t ld_out,
                                                         float* __restrict__ alpha, float* __restrict__ stats,
                                                         const int32_t* __restrict__ q_row, int64_t n_seg, int H, int d,
                                                         int c) {
  const int lane = threadIdx.x;
  const float scale = rsqrtf((float)d);
  for (int64_t w = blockIdx.x; w < n_seg * H; w += gridDim.x) {
    const int64_t b = w / H;
    const int h = (int)(w - b * H);
    const int32_t r0 = seg_begin[b], r1 = seg_end[b];
    const float* qh = q + (q_row ? (int64_t)q_row[b] : b) * ld_q + h * d;
    // pass 1: raw scores (-inf on rows that are not kept) and their maximum
    float mx = -INFINITY;
    for (int32_t r = r0 + lane; r < r1; r += 64) {
      float s = -INFINITY;
      if (hotk_keep(keep, kind[r])) {
        const float* kr = k + (int64_t)r * ld_k + h * d;
        s = 0.f;
        for (int j = 0; j < d; ++j) s = fmaf(qh[j], kr[j], s);
        s *= scale;
      }
      alpha[(int64_t)r * H + h] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    // pass 2: exponentials and their sum
    float sum = 0.f;
    for (int32_t r = r0 + lane; r < r1; r += 64) {
      const float s = alpha[(int64_t)r * H + h];
      const float e = (s == -INFINITY) ? 0.f : __expf(s - mx);
      alpha[(int64_t)r * H + h] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (stats && lane == 0) {      // (maximum, sum of exponentials) of this segment: lets a caller merge pieces of one set
      stats[(b * H + h) * 2 + 0] = mx;
      stats[(b * H + h) * 2 + 1] = sum;
    }
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    // pass 3: normalise and pool the values, CT channels at a time
    for (int c0 = 0; c0 < c; c0 += CT) {
      float acc[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = 0.f;
      for (int32_t r = r0 + lane; r < r1; r += 64) {
        float a = alpha[(int64_t)r * H + h];
        if (c0 == 0) {
          a *= inv;
          alpha[(int64_t)r * H + h] = a;
        }
        if (a != 0.f) {
          const float* vr = v + (int64_t)r * ld_v + h * c + c0;
#pragma unroll
          for (int t = 0; t < CT; ++t)
            if (c0 + t < c) acc[t] = fmaf(a, vr[t], acc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float tot = wave_sum(acc[t]);
        if (lane == 0 && c0 + t < c) out[b * ld_out + h * c + c0 + t] = tot;
      }
    }
  }
}

// backward: dalpha[r] = dout_h . v[r];  ds = alpha (dalpha - dout_h . out_h);  dv[r] = alpha dout_h;
// dk[r] = ds q_h / sqrt(d);  dq_h = sum_r ds k[r] / sqrt(d).  dk / dv rows that are not kept are left untouched
// (the caller zero-fills them).
__global__ __launch_bounds__(64) void k_softmax_pool_bwd(const float* __restrict__ q, int64_t ld_q,
                                                         const float* __restrict__ k, int64_t ld_k,
                                                         const float* __restrict__ v, int64_t ld_v,
                                                         const int32_t* __restrict__ seg_begin,
                                                         const int32_t* __restrict__ seg_end,
                                                         const float* __restrict__ alpha, const float* __restrict__ out,
                                                         int64_t ld_out, const float* __restrict__ dout, int64_t ld_dout,
                                                         float* __restrict__ dq, int64_t ld_dq, float* __restrict__ dk,
                                                         int64_t ld_dk, float* __restrict__ dv, int64_t ld_dv,
                                                         const int32_t* __restrict__ q_row, int64_t n_seg, int H, int d,
                                                         int c) {
  const int lane = threadIdx.x;
  const float scale = rsqrtf((float)d);
  for (int64_t w = blockIdx.x; w < n_seg * H; w += gridDim.x) {
    const int64_t b = w / H;
    const int h = (int)(w - b * H);
    const int32_t r0 = seg_begin[b], r1 = seg_end[b];
    const int64_t bq = q_row ? (int64_t)q_row[b] : b;      // row of q / out / dout this segment belongs to
    const float* qh = q + bq * ld_q + h * d;
    const float* gh = dout + bq * ld_dout + h * c;
    const float* oh = out + bq * ld_out + h * c;
    float dot = 0.f;
    for (int t = 0; t < c; ++t) dot = fmaf(gh[t], oh[t], dot);
    // d q_h is accumulated CT components at a time (the per-row ds is recomputed for every tile)
    for (int j0 = 0; j0 < d; j0 += CT) {
      float acc[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = 0.f;
      for (int32_t r = r0 + lane; r < r1; r += 64) {
        const float a = alpha[(int64_t)r * H + h];
        if (a == 0.f) continue;
        const float* vr = v + (int64_t)r * ld_v + h * c;
        float da = 0.f;
        for (int t = 0; t < c; ++t) da = fmaf(gh[t], vr[t], da);
        const float ds = a * (da - dot) * scale;
        const float* kr = k + (int64_t)r * ld_k + h * d;
#pragma unroll
        for (int t = 0; t < CT; ++t)
          if (j0 + t < d) acc[t] = fmaf(ds, kr[j0 + t], acc[t]);
        if (j0 == 0) {
          float* dkr = dk + (int64_t)r * ld_dk + h * d;
          for (int j = 0; j < d; ++j) dkr[j] = ds * qh[j];
          float* dvr = dv + (int64_t)r * ld_dv + h * c;
          for (int t = 0; t < c; ++t) dvr[t] = a * gh[t];
        }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float tot = wave_sum(acc[t]);
        if (lane == 0 && j0 + t < d) dq[b * ld_dq + h * d + j0 + t] = tot;
      }
    }
  }
}

}  // namespace

extern "C" int hotk_seg_softmax_pool_fwd(const float* q, int64_t ld_q, const float* k, int64_t ld_k, const float* v,
                                         int64_t ld_v, const int32_t* seg_begin, const int32_t* seg_end,
                                         const uint8_t* kind, unsigned keep, float* out, int64_t ld_out, float* alpha,
                                         float* stats, const int32_t* q_row, int64_t n_seg, int H, int d, int c,
                                         void* stream) {
  HOTK_CHECK_ARG(n_seg >= 0 && H > 0 && d > 0 && c > 0 && ld_q >= (int64_t)H * d && ld_k >= (int64_t)H * d &&
                 ld_v >= (int64_t)H * c && ld_out >= (int64_t)H * c);
  if (n_seg == 0) return HOTK_OK;
  HOTK_CHECK_ARG(q && k && v && seg_begin && seg_end && kind && out && alpha);
  hipLaunchKernelGGL(k_softmax_pool_fwd, dim3(hotk_grid(n_seg * H, 1, 256 * 64)), dim3(64), 0, (hipStream_t)stream, q,
                     ld_q, k, ld_k, v, ld_v, seg_begin, seg_end, kind, keep, out, ld_out, alpha, stats, q_row, n_seg, H, d, c);
  return hotk_launch_status();
}

extern "C" int hotk_seg_softmax_pool_bwd(const float* q, int64_t ld_q, const float* k, int64_t ld_k, const float* v,
                                         int64_t ld_v, const int32_t* seg_begin, const int32_t* seg_end,
                                         const float* alpha, const float* out, int64_t ld_out, const float* dout,
                                         int64_t ld_dout, float* dq, int64_t ld_dq, float* dk, int64_t ld_dk, float* dv,
                                         int64_t ld_dv, const int32_t* q_row, int64_t n_seg, int H, int d, int c,
                                         void* stream) {
  HOTK_CHECK_ARG(n_seg >= 0 && H > 0 && d > 0 && c > 0 && ld_q >= (int64_t)H * d && ld_k >= (int64_t)H * d &&
                 ld_v >= (int64_t)H * c && ld_out >= (int64_t)H * c && ld_dout >= (int64_t)H * c &&
                 ld_dq >= (int64_t)H * d && ld_dk >= (int64_t)H * d && ld_dv >= (int64_t)H * c);
  if (n_seg == 0) return HOTK_OK;
  HOTK_CHECK_ARG(q && k && v && seg_begin && seg_end && alpha && out && dout && dq && dk && dv);
  hipLaunchKernelGGL(k_softmax_pool_bwd, dim3(hotk_grid(n_seg * H, 1, 256 * 64)), dim3(64), 0, (hipStream_t)stream, q,
                     ld_q, k, ld_k, v, ld_v, seg_begin, seg_end, alpha, out, ld_out, dout, ld_dout, dq, ld_dq, dk, ld_dk,
                     dv, ld_dv, q_row, n_seg, H, d, c);
  return hotk_launch_status();
}
