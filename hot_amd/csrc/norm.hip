// LayerNorm over the rows of a batch with the padding mask folded in (epilogue fusion of SURVEY 8f-1).
// Reference: EncLayer's Apply(nn.LayerNorm) followed by batch_like(..., skip_masking=False), i.e. aten's layer norm over
// every padded row plus a masked_fill clone (hot_pytorch/models/encoder.py:36,40-41, sparse/f.py:20-27,
// batch/sparse.py:115-118, utils/set.py:47-48); under autograd three more aten kernels and the mask's backward.
//
//   forward : y[r] = kind[r] kept ? (x[r] - mean) * rstd * gamma + beta : 0        stats[r] = {mean, rstd}
//   backward: dx[r] = kept ? rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)) : 0
//             dgamma = sum_r g * xhat,  dbeta = sum_r g   over the kept rows: per-workgroup partials [n][2][D], summed by
//             the caller in block order (fixed: reproducible)
// A wave owns a row: lane l holds the float4s l, l + 64, ... of it (D <= 1024, D % 4 == 0), mean and variance by a
// two-pass sum over registers (the row is read once), cross-lane sums by DPP / swizzle adds.  HBM-bound: x in, y out.
#include "common.h"

namespace {

constexpr int LN_WAVES = 4;      // waves (rows in flight) per workgroup
constexpr int LN_MAXV = 4;       // float4s per lane: D <= 64 * 4 * 4

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NV>
__global__ __launch_bounds__(64 * LN_WAVES) void k_layernorm_fwd(const float* __restrict__ x, int64_t ld_x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const uint8_t* __restrict__ kind, unsigned keep, float eps,
                                                                 float* __restrict__ y, int64_t ld_y, float* __restrict__ stats,
                                                                 int64_t R, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv4 = D >> 2;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    const bool in = c < nv4;
    gm[i] = in ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[i] = in ? reinterpret_cast<const float4*>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t r = (int64_t)blockIdx.x * LN_WAVES + wave; r < R; r += (int64_t)gridDim.x * LN_WAVES) {
    const bool kept = !kind || hotk_keep(keep, kind[r]);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      v[i] = (kept && c < nv4) ? reinterpret_cast<const float4*>(x + r * ld_x)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kept) {
          o.x = (v[i].x - mean) * rstd * gm[i].x + bt[i].x;
          o.y = (v[i].y - mean) * rstd * gm[i].y + bt[i].y;
          o.z = (v[i].z - mean) * rstd * gm[i].z + bt[i].z;
          o.w = (v[i].w - mean) * rstd * gm[i].w + bt[i].w;
        }
        reinterpret_cast<float4*>(y + r * ld_y)[c] = o;
      }
    }
    if (lane == 0) {
      stats[2 * r] = mean;
      stats[2 * r + 1] = rstd;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(64 * LN_WAVES) void k_layernorm_bwd(const float* __restrict__ g, int64_t ld_g,
                                                                 const float* __restrict__ x, int64_t ld_x,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const uint8_t* __restrict__ kind, unsigned keep,
                                                                 float* __restrict__ dx, int64_t ld_dx, float* __restrict__ partial,
                                                                 int64_t R, int D) {
  __shared__ float4 fold[LN_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv4 = D >> 2;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    gm[i] = c < nv4 ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t r = (int64_t)blockIdx.x * LN_WAVES + wave; r < R; r += (int64_t)gridDim.x * LN_WAVES) {
    const bool kept = !kind || hotk_keep(keep, kind[r]);
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float4 gv[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool in = kept && c < nv4;
      gv[i] = in ? reinterpret_cast<const float4*>(g + r * ld_g)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 xv = in ? reinterpret_cast<const float4*>(x + r * ld_x)[c] : make_float4(mean, mean, mean, mean);
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      dg[i].x += gv[i].x * xh[i].x; dg[i].y += gv[i].y * xh[i].y; dg[i].z += gv[i].z * xh[i].z; dg[i].w += gv[i].w * xh[i].w;
      db[i].x += gv[i].x; db[i].y += gv[i].y; db[i].z += gv[i].z; db[i].w += gv[i].w;
      // g * gamma from here on
      gv[i].x *= gm[i].x; gv[i].y *= gm[i].y; gv[i].z *= gm[i].z; gv[i].w *= gm[i].w;
      s1 += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
      s2 += (gv[i].x * xh[i].x + gv[i].y * xh[i].y) + (gv[i].z * xh[i].z + gv[i].w * xh[i].w);
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        float4 o;   // zeros on rows that are not kept: their gv is zero and xh is zero
        o.x = rstd * (gv[i].x - m1 - xh[i].x * m2);
        o.y = rstd * (gv[i].y - m1 - xh[i].y * m2);
        o.z = rstd * (gv[i].z - m1 - xh[i].z * m2);
        o.w = rstd * (gv[i].w - m1 - xh[i].w * m2);
        if (!kept) o = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(dx + r * ld_dx)[c] = o;
      }
    }
  }
  // the workgroup's waves add their dgamma / dbeta registers in wave order; one partial [2][D] per workgroup
  float* pg = partial + (int64_t)blockIdx.x * 2 * D;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      __syncthreads();
      fold[wave][lane] = half ? db[i] : dg[i];
      __syncthreads();
      const int c = lane + 64 * i;
      if (wave == 0 && c < nv4) {
        float4 t = fold[0][lane];
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) {
          const float4 u = fold[w][lane];
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        reinterpret_cast<float4*>(pg + half * D)[c] = t;
      }
    }
  }
}

unsigned ln_grid(int64_t R) {
  const int64_t g = (R + LN_WAVES - 1) / LN_WAVES;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int hotk_layernorm_supported(int64_t D) { return (D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV) ? 1 : 0; }

extern "C" int hotk_layernorm_partials(int64_t R, int64_t* n_partials) {
  HOTK_CHECK_ARG(n_partials && R >= 0);
  *n_partials = (int64_t)ln_grid(R);
  return HOTK_OK;
}

extern "C" int hotk_layernorm_fwd(const float* x, int64_t ld_x, const float* gamma, const float* beta, const uint8_t* kind,
                                  unsigned keep, float eps, float* y, int64_t ld_y, float* stats, int64_t R, int64_t D,
                                  void* stream) {
  HOTK_CHECK_ARG(R >= 0 && ld_x >= D && ld_y >= D);
  if (!hotk_layernorm_supported(D)) return HOTK_EUNSUPPORTED;
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && gamma && beta && y && stats);
  HOTK_CHECK_ARG(hotk_aligned16(x) && hotk_aligned16(y) && hotk_aligned16(gamma) && hotk_aligned16(beta) && ((ld_x | ld_y) & 3) == 0);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ln_grid(R)), block(64 * LN_WAVES);
  const int nv = (int)((D / 4 + 63) / 64);
#define HOTK_LN_F(NV) hipLaunchKernelGGL(k_layernorm_fwd<NV>, grid, block, 0, st, x, ld_x, gamma, beta, kind, keep, eps, y, ld_y, stats, R, (int)D)
  if (nv == 1) HOTK_LN_F(1); else if (nv == 2) HOTK_LN_F(2); else if (nv == 3) HOTK_LN_F(3); else HOTK_LN_F(4);
#undef HOTK_LN_F
  return hotk_launch_status();
}

extern "C" int hotk_layernorm_bwd(const float* g, int64_t ld_g, const float* x, int64_t ld_x, const float* stats,
                                  const float* gamma, const uint8_t* kind, unsigned keep, float* dx, int64_t ld_dx,
                                  float* partial, int64_t R, int64_t D, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && ld_g >= D && ld_x >= D && ld_dx >= D && partial);
  if (!hotk_layernorm_supported(D)) return HOTK_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (R == 0) return hipMemsetAsync(partial, 0, 2 * D * sizeof(float), st) == hipSuccess ? HOTK_OK : HOTK_ELAUNCH;
  HOTK_CHECK_ARG(g && x && stats && gamma && dx);
  HOTK_CHECK_ARG(hotk_aligned16(g) && hotk_aligned16(x) && hotk_aligned16(dx) && hotk_aligned16(gamma) && hotk_aligned16(partial) &&
                 ((ld_g | ld_x | ld_dx) & 3) == 0);
  const dim3 grid(ln_grid(R)), block(64 * LN_WAVES);
  const int nv = (int)((D / 4 + 63) / 64);
#define HOTK_LN_B(NV) hipLaunchKernelGGL(k_layernorm_bwd<NV>, grid, block, 0, st, g, ld_g, x, ld_x, stats, gamma, kind, keep, dx, ld_dx, partial, R, (int)D)
  if (nv == 1) HOTK_LN_B(1); else if (nv == 2) HOTK_LN_B(2); else if (nv == 3) HOTK_LN_B(3); else HOTK_LN_B(4);
#undef HOTK_LN_B
  return hotk_launch_status();
}
