// K3 (general sizes): segmented key-value reductions and their application, any HH / m / c.
// Reference: models/sparse/kernelattncoef.py:74-91 (per-graph sums), :103-122 (per-column sums through
// torch.sparse coalesce), :124-175 (query application and 1e-5 normaliser guard).
//
// KV layout: [n_seg][F = HH*m][c+1] floats; channel c is the normaliser (sum of keys) in the forward
// pass and the normaliser gradient in the backward pass.
#include "common.h"

namespace {

constexpr int AT_THREADS = 256;
constexpr int AT_CT = 9;  // channels accumulated per thread and pass: c = 8 values + the normaliser in ONE walk of the segment

// work item = (segment, chunk, channel tile, feature); consecutive lanes -> consecutive features, so the
// key row is read as contiguous 256-B pieces and the value channels are wave-broadcast loads.
// C8: c == 8 with 16-byte aligned value rows -- the head's 8 values arrive as two 16-byte loads instead of 8 scalar ones
template <bool C8>
__global__ __launch_bounds__(AT_THREADS) void k_seg_outer_sum(
    const float* __restrict__ K, int64_t ld_k, const int32_t* __restrict__ kidx, const float* __restrict__ V,
    int64_t ld_v, const int32_t* __restrict__ vidx, const float* __restrict__ aug, int64_t ld_aug,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, const int32_t* __restrict__ perm,
    float* __restrict__ KV, int64_t n_seg, int HH, int m, int c, int64_t n_chunk, int n_tile) {
  const int F = HH * m, Fp = (F + 63) & ~63, c1 = c + 1;
  const int64_t total = n_seg * n_chunk * n_tile * Fp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int f = (int)(t % Fp);
    int64_t rest = t / Fp;
    int tile = (int)(rest % n_tile);
    rest /= n_tile;
    int64_t ch = rest % n_chunk, s = rest / n_chunk;
    if (f >= F) continue;
    const int hh = f / m;
    const int64_t b = seg_begin[s], e = seg_end[s];
    int64_t lo = b + ch * HOTK_SEG_CHUNK, hi = lo + HOTK_SEG_CHUNK;
    if (hi > e) hi = e;
    if (lo >= hi && ch > 0) continue;
    const int c0 = tile * AT_CT;
    float acc[AT_CT];
#pragma unroll
    for (int k = 0; k < AT_CT; ++k) acc[k] = 0.f;
    for (int64_t p = lo; p < hi; ++p) {
      int64_t r = perm ? (int64_t)perm[p] : p;
      int64_t kr = kidx ? (int64_t)kidx[r] : r;
      if (kr < 0) continue;
      const float kv = K[kr * ld_k + f];
      int64_t vr = vidx ? (int64_t)vidx[r] : r;
      const float* vp = V + vr * ld_v + hh * c;
      if (C8) {
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (vr >= 0) {
          v0 = *reinterpret_cast<const float4*>(vp);
          v1 = *reinterpret_cast<const float4*>(vp + 4);
        }
        const float va = aug ? aug[r * ld_aug + hh] : 1.f;
        acc[0] = fmaf(kv, v0.x, acc[0]); acc[1] = fmaf(kv, v0.y, acc[1]); acc[2] = fmaf(kv, v0.z, acc[2]);
        acc[3] = fmaf(kv, v0.w, acc[3]); acc[4] = fmaf(kv, v1.x, acc[4]); acc[5] = fmaf(kv, v1.y, acc[5]);
        acc[6] = fmaf(kv, v1.z, acc[6]); acc[7] = fmaf(kv, v1.w, acc[7]); acc[8] = fmaf(kv, va, acc[8]);
        continue;
      }
#pragma unroll
      for (int k = 0; k < AT_CT; ++k) {
        int chn = c0 + k;
        float val = 0.f;
        if (chn < c) val = (vr >= 0) ? vp[chn] : 0.f;
        else if (chn == c) val = aug ? aug[r * ld_aug + hh] : 1.f;
        acc[k] = fmaf(kv, val, acc[k]);
      }
    }
    float* o = KV + ((int64_t)s * F + f) * c1;
#pragma unroll
    for (int k = 0; k < AT_CT; ++k) {
      int chn = c0 + k;
      if (chn <= c) o[chn] = acc[k];
    }
  }
}

__device__ __forceinline__ int64_t seg_of(const int32_t* __restrict__ sidx, int64_t seg_div, int64_t r) {
  return sidx ? (int64_t)sidx[r] : r / seg_div;
}

// thread per (row, head, channel)
__global__ __launch_bounds__(AT_THREADS) void k_row_contract(
    const float* __restrict__ Q, int64_t ld_q, const int32_t* __restrict__ qidx, const float* __restrict__ KV,
    const int32_t* __restrict__ sidx, int64_t seg_div, const uint8_t* __restrict__ kind, unsigned keep,
    const int32_t* __restrict__ oidx, float* __restrict__ out, int64_t ld_out, int64_t n, int HH, int m, int c,
    int normalize) {
  const int F = HH * m, c1 = c + 1, W = HH * c;
  const int64_t total = n * W;
  const bool q4 = (m & 3) == 0 && (ld_q & 3) == 0 && (((uintptr_t)Q) & 15u) == 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / W;
    int w = (int)(t - r * W);
    int hh = w / c, chn = w - hh * c;
    int64_t orow = oidx ? (int64_t)oidx[r] : r;
    if (orow < 0) continue;
    float o = 0.f;
    int64_t s = seg_of(sidx, seg_div, r);
    int64_t qr = qidx ? (int64_t)qidx[r] : r;
    if (s >= 0 && qr >= 0 && (!kind || hotk_keep(keep, kind[r]))) {
      const float* q = Q + qr * ld_q + hh * m;
      const float* kv = KV + ((int64_t)s * F + hh * m) * c1;
      float num = 0.f, den = 0.f;
      if (q4) {   // m % 4 == 0, 16-byte aligned rows: the query slice as vector loads (same fmaf order)
        for (int mm = 0; mm < m; mm += 4) {
          const float4 qv = *reinterpret_cast<const float4*>(q + mm);
          const float* k0 = kv + mm * c1;
          num = fmaf(qv.x, k0[chn], num);          den = fmaf(qv.x, k0[c], den);
          num = fmaf(qv.y, k0[c1 + chn], num);     den = fmaf(qv.y, k0[c1 + c], den);
          num = fmaf(qv.z, k0[2 * c1 + chn], num); den = fmaf(qv.z, k0[2 * c1 + c], den);
          num = fmaf(qv.w, k0[3 * c1 + chn], num); den = fmaf(qv.w, k0[3 * c1 + c], den);
        }
      } else {
        for (int mm = 0; mm < m; ++mm) {
          float qv = q[mm];
          num = fmaf(qv, kv[mm * c1 + chn], num);
          den = fmaf(qv, kv[mm * c1 + c], den);
        }
      }
      if (normalize) {
        if (den == 0.f) den = 1e-5f;
        o = num / den;
      } else {
        o = num;
      }
    }
    out[orow * ld_out + w] = o;
  }
}

// thread per (row, head): X = dO / den, XA = -(dO . O) / den
__global__ __launch_bounds__(AT_THREADS) void k_apply_bwd_stats(
    const float* __restrict__ Q, int64_t ld_q, const float* __restrict__ KV, const int32_t* __restrict__ sidx,
    int64_t seg_div, const uint8_t* __restrict__ kind, unsigned keep, const float* __restrict__ dO, int64_t ld_do,
    float* __restrict__ X, int64_t ld_x, float* __restrict__ XA, int64_t ld_xa, int64_t n, int HH, int m, int c) {
  const int F = HH * m, c1 = c + 1;
  const int64_t total = n * HH;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / HH;
    int hh = (int)(t - r * HH);
    float* xo = X + r * ld_x + hh * c;
    int64_t s = seg_of(sidx, seg_div, r);
    if (s < 0 || (kind && !hotk_keep(keep, kind[r]))) {
      for (int chn = 0; chn < c; ++chn) xo[chn] = 0.f;
      XA[r * ld_xa + hh] = 0.f;
      continue;
    }
    const float* q = Q + r * ld_q + hh * m;
    const float* kv = KV + ((int64_t)s * F + hh * m) * c1;
    const float* g = dO + r * ld_do + hh * c;
    float den = 0.f;
    for (int mm = 0; mm < m; ++mm) den = fmaf(q[mm], kv[mm * c1 + c], den);
    const bool guarded = (den == 0.f);
    if (guarded) den = 1e-5f;
    const float inv = 1.f / den;
    float dot = 0.f;
    for (int chn = 0; chn < c; ++chn) {
      float num = 0.f;
      for (int mm = 0; mm < m; ++mm) num = fmaf(q[mm], kv[mm * c1 + chn], num);
      float gv = g[chn];
      xo[chn] = gv * inv;
      dot = fmaf(gv, num * inv, dot);
    }
    XA[r * ld_xa + hh] = guarded ? 0.f : -dot * inv;
  }
}

// The same with a thread per (row, head, channel) for c in {1, 2, 4, 8, 16, 32, 64}: the c threads of a head are c
// consecutive lanes (groups aligned in the wave), each contracts ITS channel and the normaliser with the query -- 8
// lanes read 8 consecutive floats of a key-value row instead of one lane walking all 144 -- and the dot product
// dO . O meets through xor shuffles.  Same fmaf chains per channel as above; the channel sum is a tree instead of a loop.
template <int C>
__global__ __launch_bounds__(AT_THREADS) void k_apply_bwd_stats_c(
    const float* __restrict__ Q, int64_t ld_q, const float* __restrict__ KV, const int32_t* __restrict__ sidx,
    int64_t seg_div, const uint8_t* __restrict__ kind, unsigned keep, const float* __restrict__ dO, int64_t ld_do,
    float* __restrict__ X, int64_t ld_x, float* __restrict__ XA, int64_t ld_xa, int64_t n, int HH, int m) {
  constexpr int c1 = C + 1;
  const int F = HH * m, W = HH * C;
  const int64_t total = n * W;
  // every lane of a wave runs the same number of iterations (total is padded), so the shuffles see whole groups
  const int64_t padded = (total + 63) & ~(int64_t)63;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < padded; t += (int64_t)gridDim.x * blockDim.x) {
    const bool live = t < total;
    const int64_t r = live ? t / W : 0;
    const int w = (int)(t - r * W);
    const int hh = live ? w / C : 0, chn = w - hh * C;
    const int64_t s = live ? seg_of(sidx, seg_div, r) : -1;
    const bool on = live && s >= 0 && (!kind || hotk_keep(keep, kind[r]));
    float x = 0.f, part = 0.f, inv = 0.f;
    bool guarded = false;
    if (on) {
      const float* q = Q + r * ld_q + hh * m;
      const float* kv = KV + ((int64_t)s * F + hh * m) * c1;
      float num = 0.f, den = 0.f;
      for (int mm = 0; mm < m; ++mm) {
        const float qv = q[mm];
        num = fmaf(qv, kv[mm * c1 + chn], num);
        den = fmaf(qv, kv[mm * c1 + C], den);
      }
      guarded = (den == 0.f);
      if (guarded) den = 1e-5f;
      inv = 1.f / den;
      const float gv = dO[r * ld_do + w];
      x = gv * inv;
      part = gv * (num * inv);
    }
#pragma unroll
    for (int o = 1; o < C; o <<= 1) part += __shfl_xor(part, o, 64);
    if (live) {
      X[r * ld_x + w] = x;
      if (chn == 0) XA[r * ld_xa + hh] = (on && !guarded) ? -part * inv : 0.f;
    }
  }
}

// thread per (row, feature)
__global__ __launch_bounds__(AT_THREADS) void k_row_expand(
    const float* __restrict__ KV, const int32_t* __restrict__ sidx, int64_t seg_div, const float* __restrict__ X,
    int64_t ld_x, const int32_t* __restrict__ xidx, const float* __restrict__ XA, int64_t ld_xa,
    const uint8_t* __restrict__ kind, unsigned keep, float* __restrict__ out, int64_t ld_out, int64_t n, int HH, int m,
    int c) {
  const int F = HH * m, c1 = c + 1;
  const int64_t total = n * F;
  const bool x8 = c == 8 && (ld_x & 3) == 0 && (((uintptr_t)X) & 15u) == 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / F;
    int f = (int)(t - r * F);
    int hh = f / m;
    float o = 0.f;
    int64_t s = seg_of(sidx, seg_div, r);
    if (s >= 0 && (!kind || hotk_keep(keep, kind[r]))) {
      const float* kv = KV + ((int64_t)s * F + f) * c1;
      int64_t xr = xidx ? (int64_t)xidx[r] : r;
      if (xr >= 0) {
        const float* xp = X + xr * ld_x + hh * c;
        if (x8) {   // c == 8, 16-byte aligned rows: two vector loads (same fmaf order)
          const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
          o = fmaf(kv[0], x0.x, o); o = fmaf(kv[1], x0.y, o); o = fmaf(kv[2], x0.z, o); o = fmaf(kv[3], x0.w, o);
          o = fmaf(kv[4], x1.x, o); o = fmaf(kv[5], x1.y, o); o = fmaf(kv[6], x1.z, o); o = fmaf(kv[7], x1.w, o);
        } else {
          for (int chn = 0; chn < c; ++chn) o = fmaf(kv[chn], xp[chn], o);
        }
      }
      o = fmaf(kv[c], XA ? XA[r * ld_xa + hh] : 1.f, o);
    }
    out[r * ld_out + f] = o;
  }
}

int dims_ok(int HH, int m, int c) { return HH > 0 && m > 0 && c > 0; }

}  // namespace

extern "C" int hotk_seg_outer_sum(const float* K, int64_t ld_k, const int32_t* kidx, const float* V, int64_t ld_v,
                                  const int32_t* vidx, const float* aug, int64_t ld_aug, const int32_t* seg_begin,
                                  const int32_t* seg_end, const int32_t* perm, float* KV, int64_t n_seg, int HH, int m,
                                  int c, int64_t max_seg_len, void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n_seg >= 0 && max_seg_len >= 0);
  if (n_seg == 0) return HOTK_OK;
  HOTK_CHECK_ARG(K && V && KV && seg_begin && seg_end && ld_k >= (int64_t)HH * m && ld_v >= (int64_t)HH * c);
  HOTK_CHECK_ARG(!aug || ld_aug >= HH);
  // one thread walks a whole segment in order; longer segments are split by the caller (hot_amd.ops.refine_seg) and the
  // partial sums added by hotk_seg_sum_rows: no float atomics, bitwise reproducible
  if (max_seg_len > HOTK_SEG_CHUNK) return HOTK_EUNSUPPORTED;
  const int64_t n_chunk = 1;
  int n_tile = (c + 1 + AT_CT - 1) / AT_CT;
  int Fp = (HH * m + 63) & ~63;
  int64_t work = n_seg * n_chunk * n_tile * Fp;
  static_assert(AT_CT == 9, "the C8 form keeps 8 values + the normaliser in one tile");
  const bool c8 = c == 8 && (ld_v & 3) == 0 && hotk_aligned16(V);
  if (c8)
    hipLaunchKernelGGL(k_seg_outer_sum<true>, dim3(hotk_grid(work, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                       (hipStream_t)stream, K, ld_k, kidx, V, ld_v, vidx, aug, ld_aug, seg_begin, seg_end, perm, KV,
                       n_seg, HH, m, c, n_chunk, n_tile);
  else
    hipLaunchKernelGGL(k_seg_outer_sum<false>, dim3(hotk_grid(work, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                       (hipStream_t)stream, K, ld_k, kidx, V, ld_v, vidx, aug, ld_aug, seg_begin, seg_end, perm, KV,
                       n_seg, HH, m, c, n_chunk, n_tile);
  return hotk_launch_status();
}

extern "C" int hotk_row_contract(const float* Q, int64_t ld_q, const int32_t* qidx, const float* KV,
                                 const int32_t* sidx, int64_t seg_div, const uint8_t* kind, unsigned keep,
                                 const int32_t* oidx, float* out, int64_t ld_out, int64_t n, int HH, int m, int c,
                                 int normalize, void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(Q && KV && out && ld_q >= (int64_t)HH * m && ld_out >= (int64_t)HH * c);
  hipLaunchKernelGGL(k_row_contract, dim3(hotk_grid(n * HH * c, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                     (hipStream_t)stream, Q, ld_q, qidx, KV, sidx, seg_div, kind, keep, oidx, out, ld_out, n, HH, m, c,
                     normalize);
  return hotk_launch_status();
}

extern "C" int hotk_apply_bwd_stats(const float* Q, int64_t ld_q, const float* KV, const int32_t* sidx,
                                    int64_t seg_div, const uint8_t* kind, unsigned keep, const float* dO, int64_t ld_do,
                                    float* X, int64_t ld_x, float* XA, int64_t ld_xa, int64_t n, int HH, int m, int c,
                                    void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(Q && KV && dO && X && XA && ld_q >= (int64_t)HH * m && ld_do >= (int64_t)HH * c &&
                 ld_x >= (int64_t)HH * c && ld_xa >= HH);
  hipStream_t st = (hipStream_t)stream;
#define HOTK_ABS_CASE(CC)                                                                                              \
  case CC:                                                                                                             \
    hipLaunchKernelGGL(k_apply_bwd_stats_c<CC>, dim3(hotk_grid(n * HH * CC, AT_THREADS, 256 * 32)), dim3(AT_THREADS),  \
                       0, st, Q, ld_q, KV, sidx, seg_div, kind, keep, dO, ld_do, X, ld_x, XA, ld_xa, n, HH, m);        \
    break;
  switch (c) {
    HOTK_ABS_CASE(1) HOTK_ABS_CASE(2) HOTK_ABS_CASE(4) HOTK_ABS_CASE(8) HOTK_ABS_CASE(16) HOTK_ABS_CASE(32)
    HOTK_ABS_CASE(64)
    default:
      hipLaunchKernelGGL(k_apply_bwd_stats, dim3(hotk_grid(n * HH, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0, st, Q,
                         ld_q, KV, sidx, seg_div, kind, keep, dO, ld_do, X, ld_x, XA, ld_xa, n, HH, m, c);
  }
#undef HOTK_ABS_CASE
  return hotk_launch_status();
}

extern "C" int hotk_row_expand(const float* KV, const int32_t* sidx, int64_t seg_div, const float* X, int64_t ld_x,
                               const int32_t* xidx, const float* XA, int64_t ld_xa, const uint8_t* kind, unsigned keep,
                               float* out, int64_t ld_out, int64_t n, int HH, int m, int c, void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(KV && X && out && ld_x >= (int64_t)HH * c && ld_out >= (int64_t)HH * m && (!XA || ld_xa >= HH));
  hipLaunchKernelGGL(k_row_expand, dim3(hotk_grid(n * HH * m, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                     (hipStream_t)stream, KV, sidx, seg_div, X, ld_x, xidx, XA, ld_xa, kind, keep, out, ld_out, n, HH,
                     m, c);
  return hotk_launch_status();
}
