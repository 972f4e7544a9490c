// K2: FAVOR+ random-feature maps, fused with the head split and the loop/padding masks.
// Reference: models/common/kernel.py:81-138 (softmax_map / generalized_map) and
// models/sparse/kernelattncoef.py:19-28 (split into heads, loop mask, concat, padding mask).
//
// x [n, HH*d] -> phi [n, HH*m].  One thread per output feature (generalized map) or per (row, head)
// (softmax maps, which need a max over the head's m features).  The projection matrix (m x d floats) is
// staged in LDS once per block; x rows are read as whole contiguous pieces by consecutive lanes.
#include "common.h"

namespace {

constexpr int PHI_THREADS = 256;
constexpr int PHI_MAX_PROJ = 8192;  // floats of LDS for the projection matrix (m*d <= 8192)

// one shared definition so that every kernel reproduces the projected value bit for bit
__device__ __forceinline__ float proj_dot(const float* __restrict__ xr, const float* __restrict__ pr, int d,
                                          float scale) {
  float s = 0.f;
  for (int k = 0; k < d; ++k) s = fmaf(xr[k] * scale, pr[k], s);
  return s;
}

__device__ __forceinline__ void stage_proj(float* lds, const float* __restrict__ proj, int md) {
  for (int i = threadIdx.x; i < md; i += blockDim.x) lds[i] = proj[i];
  __syncthreads();
}

// ------------------------------------------------------------------ generalized (ReLU) map
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_fwd(const float* __restrict__ x, int64_t ld_x,
                                                              const float* __restrict__ proj,
                                                              const uint8_t* __restrict__ kind, unsigned keep,
                                                              float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                              int HH, int d, int m, float scale, float eps) {
  __shared__ float P[PHI_MAX_PROJ];
  stage_proj(P, proj, m * d);
  const int64_t F = (int64_t)HH * m, total = n * F;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / F;
    int f = (int)(t - r * F);
    int hh = f / m, mm = f - hh * m;
    float o = 0.f;
    if (!kind || hotk_keep(keep, kind[r])) {
      float pre = proj_dot(x + r * ld_x + hh * d, P + mm * d, d, scale);
      o = fmaxf(pre, 0.f) + eps;
    }
    phi[r * ld_phi + f] = o;
  }
}

// d == 8, m == 16 (the C1 / C2 head shape), 16-byte aligned rows: a thread keeps its feature's projection row in
// registers (256 threads = 16 whole heads, so the feature index of a thread never changes) and reads the 8-float head
// slice as two 16-byte loads shared by the 16 threads of the head.  Same fmaf chain as proj_dot: identical bits.
// PHI_U (row, feature) items per thread and step, flattened over rows, with the kinds and head slices of all of them
// requested together and unconditionally (a masked row's slice is fetched and dropped): the block-per-row form paid two
// dependent round trips per row -- kind, then the slice under a branch on it -- with one row in flight per block
// (1.27 ms for the 978 k-node key table of the bench: 2.8 TB/s).  Same fmaf chain per element.
constexpr int PHI_U = 8;
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_fwd_8x16(const float* __restrict__ x, int64_t ld_x,
                                                                   const float* __restrict__ proj,
                                                                   const uint8_t* __restrict__ kind, unsigned keep,
                                                                   float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                   int HH, float scale, float eps) {
  const int mm = threadIdx.x & 15;
  float pr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) pr[k] = proj[mm * 8 + k];
  const int F = HH * 16;                  // (a multiple of 16 and PHI_THREADS % 16 == 0: item t keeps mm = t & 15)
  const int64_t total = n * F;
  const bool small = total < (int64_t)0xffffffffll;
  const int64_t stride = (int64_t)PHI_THREADS * PHI_U;
  for (int64_t base = (int64_t)blockIdx.x * stride; base < total; base += (int64_t)gridDim.x * stride) {
    int64_t r[PHI_U];
    int f[PHI_U], kd[PHI_U];
    float4 a[PHI_U], b[PHI_U];
#pragma unroll
    for (int u = 0; u < PHI_U; ++u) {
      const int64_t t = base + threadIdx.x + (int64_t)PHI_THREADS * u;
      const int64_t tc = t < total ? t : total - 1;
      r[u] = small ? (int64_t)((uint32_t)tc / (uint32_t)F) : tc / F;
      f[u] = (int)(tc - r[u] * F);
    }
#pragma unroll
    for (int u = 0; u < PHI_U; ++u) kd[u] = kind ? (int)kind[r[u]] : 0;
#pragma unroll
    for (int u = 0; u < PHI_U; ++u) {
      const float* xr = x + r[u] * ld_x + (f[u] >> 4) * 8;
      a[u] = *reinterpret_cast<const float4*>(xr);
      b[u] = *reinterpret_cast<const float4*>(xr + 4);
    }
#pragma unroll
    for (int u = 0; u < PHI_U; ++u)     // (opaque uses: the loads stay together, in front of the arithmetic)
      asm volatile("" : "+v"(kd[u]), "+v"(a[u].x), "+v"(a[u].y), "+v"(a[u].z), "+v"(a[u].w), "+v"(b[u].x), "+v"(b[u].y),
                   "+v"(b[u].z), "+v"(b[u].w));
#pragma unroll
    for (int u = 0; u < PHI_U; ++u) {
      float s = 0.f;
      s = fmaf(a[u].x * scale, pr[0], s); s = fmaf(a[u].y * scale, pr[1], s);
      s = fmaf(a[u].z * scale, pr[2], s); s = fmaf(a[u].w * scale, pr[3], s);
      s = fmaf(b[u].x * scale, pr[4], s); s = fmaf(b[u].y * scale, pr[5], s);
      s = fmaf(b[u].z * scale, pr[6], s); s = fmaf(b[u].w * scale, pr[7], s);
      const bool on = !kind || hotk_keep(keep, (uint8_t)kd[u]);
      const float o = on ? fmaxf(s, 0.f) + eps : 0.f;
      if (base + threadIdx.x + (int64_t)PHI_THREADS * u < total) phi[r[u] * ld_phi + f[u]] = o;
    }
  }
}

// A WAVE per row: the row's HH x 8 inputs arrive as ONE coalesced 16-byte load per lane (HH <= 32), pass through a
// wave-private LDS buffer and are read back as the lane's head slice (two broadcast ds_read_b128 per 64 features).  The
// block-per-row form had every lane fetch its head's 32 bytes itself -- 16 lanes per address, two 1 KB requests through
// the texture path per 64 outputs -- and paid two dependent round trips per row (kind, then the slice under a branch on
// it) with one row in flight per block: 1.27 ms for the 978 k-node key table of the bench (2.3 TB/s), and batching the
// loads alone changed nothing.  Here the next two rows of a wave are in flight behind the current one (two named
// register sets, unconditional loads: rows past the end repeat the last row and store the same values again).
// Needs HH % 4 == 0 and HH <= 32; same fmaf chain per element.
template <int NGT>      // groups of 64 features per row when known at compile time (the stores of a step are then counted), else 0
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_fwd_8x16_w(const float* __restrict__ x, int64_t ld_x,
                                                                     const float* __restrict__ proj,
                                                                     const uint8_t* __restrict__ kind, unsigned keep,
                                                                     float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                     int HH, float scale, float eps) {
  constexpr int WAVES = PHI_THREADS / 64;
  __shared__ __align__(16) float xs_all[WAVES][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mm = lane & 15;
  float pr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) pr[k] = proj[mm * 8 + k];
  const int nq = HH * 2, NG = NGT ? NGT : HH >> 2;          // 16-byte pieces of a row; groups of 64 features
  const int lq = lane < nq ? lane : 0;          // (idle lanes re-read piece 0)
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const uint8_t* kbase = kind ? kind : reinterpret_cast<const uint8_t*>(proj);
  auto fetch = [&](int64_t r, float4& v, int& kd) __attribute__((always_inline)) {
    const int64_t rc = r < n ? r : n - 1;
    v = *reinterpret_cast<const float4*>(x + rc * ld_x + 4 * lq);
    kd = (int)kbase[kind ? rc : 0];       // (unconditional: a load under `kind ? ... : 0` cannot be counted)
  };
  auto step = [&](float4& v, int& kd, const int b, const int64_t r) __attribute__((always_inline)) {
    const int64_t rc = r < n ? r : n - 1;
    float* xs = xs_all[wave][b];
    *reinterpret_cast<float4*>(xs + 4 * lane) = v;
    const bool on = !kind || hotk_keep(keep, (uint8_t)kd);
    fetch(r + 2 * stride, v, kd);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* out = phi + rc * ld_phi + lane;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* xh = xs + (4 * g + (lane >> 4)) * 8;
      const float4 a = *reinterpret_cast<const float4*>(xh), c = *reinterpret_cast<const float4*>(xh + 4);
      float t = 0.f;
      t = fmaf(a.x * scale, pr[0], t); t = fmaf(a.y * scale, pr[1], t);
      t = fmaf(a.z * scale, pr[2], t); t = fmaf(a.w * scale, pr[3], t);
      t = fmaf(c.x * scale, pr[4], t); t = fmaf(c.y * scale, pr[5], t);
      t = fmaf(c.z * scale, pr[6], t); t = fmaf(c.w * scale, pr[7], t);
      out[64 * g] = on ? fmaxf(t, 0.f) + eps : 0.f;
    }
  };
  int64_t r = (int64_t)blockIdx.x * WAVES + wave;
  float4 v0, v1;
  int k0, k1;
  fetch(r, v0, k0);
  fetch(r + stride, v1, k1);
  for (; r < n; r += 2 * stride) {
    step(v0, k0, 0, r);
    step(v1, k1, 1, r + stride);
  }
}

__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_bwd(const float* __restrict__ x, int64_t ld_x,
                                                              const float* __restrict__ proj,
                                                              const uint8_t* __restrict__ kind, unsigned keep,
                                                              const float* __restrict__ dphi, int64_t ld_dphi,
                                                              float* __restrict__ dx, int64_t ld_dx, int64_t n, int HH,
                                                              int d, int m, float scale) {
  __shared__ float P[PHI_MAX_PROJ];
  stage_proj(P, proj, m * d);
  const int64_t C = (int64_t)HH * d, total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / C;
    int c = (int)(t - r * C);
    int hh = c / d, k = c - hh * d;
    float acc = 0.f;
    if (!kind || hotk_keep(keep, kind[r])) {
      const float* xr = x + r * ld_x + hh * d;
      const float* gr = dphi + r * ld_dphi + hh * m;
      for (int mm = 0; mm < m; ++mm) {
        float pre = proj_dot(xr, P + mm * d, d, scale);
        if (pre > 0.f) acc = fmaf(gr[mm], P[mm * d + k], acc);
      }
      acc *= scale;
    }
    dx[r * ld_dx + c] = acc;
  }
}

// Head widths d in {4, 8, 16, 32} with 16-byte aligned rows (the other shipped shape, enc.sh: d = 16, m = 44).  A block walks
// whole rows (no 64-bit division per element); the projection sits in LDS with rows padded to d + 1 floats so that 64
// consecutive features read 64 distinct banks; the head slice of x arrives as d / 4 16-byte loads shared by the lanes of
// the head.  Same fmaf chains as proj_dot: identical bits (and identical gates in the backward).
template <int D>
__device__ __forceinline__ float proj_dot_v(const float* __restrict__ xs, const float* __restrict__ pr, float scale) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < D / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(xs + 4 * q);
    s = fmaf(v.x * scale, pr[4 * q + 0], s);
    s = fmaf(v.y * scale, pr[4 * q + 1], s);
    s = fmaf(v.z * scale, pr[4 * q + 2], s);
    s = fmaf(v.w * scale, pr[4 * q + 3], s);
  }
  return s;
}
template <int D>
__device__ __forceinline__ void stage_proj_padded(float* lds, const float* __restrict__ proj, int m) {
  for (int i = threadIdx.x; i < m * D; i += blockDim.x) lds[(i / D) * (D + 1) + (i % D)] = proj[i];
  __syncthreads();
}
constexpr int PHI_MAX_M = 128;                 // features per head these kernels hold in LDS (padded rows of D + 1 floats)

template <int D>
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_fwd_d(const float* __restrict__ x, int64_t ld_x,
                                                                const float* __restrict__ proj,
                                                                const uint8_t* __restrict__ kind, unsigned keep,
                                                                float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                int HH, int m, float scale, float eps) {
  __shared__ float P[PHI_MAX_M * (D + 1)];
  stage_proj_padded<D>(P, proj, m);
  const int F = HH * m;
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const bool on = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x;
    float* pr = phi + r * ld_phi;
    for (int f = threadIdx.x; f < F; f += PHI_THREADS) {
      const int hh = f / m, mm = f - hh * m;
      float o = 0.f;
      if (on) o = fmaxf(proj_dot_v<D>(xr + hh * D, P + mm * (D + 1), scale), 0.f) + eps;
      pr[f] = o;
    }
  }
}

// The same map with a WAVE per row: the row's HH x D inputs arrive as coalesced 16-byte loads (one or a few per lane)
// into a wave-private LDS row, and every lane reads its head's slice from there.  In the block-per-row form all m lanes
// of a head fetch the same D floats from global memory themselves: D / 4 requests of 1 KB through the texture path per
// 64 outputs for D x 4 bytes of new data (the 8 x 16 twin went 1.29 -> 0.58 ms by this alone).  Same fmaf chains.
constexpr int PHI_ROW_MAX = 1024;      // floats of an input row held per wave
template <int D>
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_fwd_dw(const float* __restrict__ x, int64_t ld_x,
                                                                 const float* __restrict__ proj,
                                                                 const uint8_t* __restrict__ kind, unsigned keep,
                                                                 float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                 int HH, int m, float scale, float eps) {
  constexpr int WAVES = PHI_THREADS / 64;
  __shared__ float P[PHI_MAX_M * (D + 1)];
  __shared__ __align__(16) float xs_all[WAVES][PHI_ROW_MAX];
  stage_proj_padded<D>(P, proj, m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = xs_all[wave];
  const int F = HH * m, nq = (HH * D) >> 2;
  for (int64_t r = (int64_t)blockIdx.x * WAVES + wave; r < n; r += (int64_t)gridDim.x * WAVES) {
    const bool on = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (on)
      for (int i = lane; i < nq; i += 64) *reinterpret_cast<float4*>(xs + 4 * i) = *reinterpret_cast<const float4*>(xr + 4 * i);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* pr = phi + r * ld_phi;
    for (int f = lane; f < F; f += 64) {
      const int hh = f / m, mm = f - hh * m;
      float o = 0.f;
      if (on) o = fmaxf(proj_dot_v<D>(xs + hh * D, P + mm * (D + 1), scale), 0.f) + eps;
      pr[f] = o;
    }
  }
}

// backward: thread (head-slot hh, k) of a row; the D threads of a head-slot are consecutive lanes.  Phase A: thread k
// recomputes the pre-activations of features k, k + D, ... (the forward's chain: identical gate) and leaves the gated
// gradients in LDS; phase B: it contracts them with column k of the projection in feature order, as the one-thread-per-
// element kernel does (same bits), at (m + D ceil(m / D)) instead of m (D + 1) multiply-adds per output element.
template <int D>
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_bwd_d(const float* __restrict__ x, int64_t ld_x,
                                                                const float* __restrict__ proj,
                                                                const uint8_t* __restrict__ kind, unsigned keep,
                                                                const float* __restrict__ dphi, int64_t ld_dphi,
                                                                float* __restrict__ dx, int64_t ld_dx, int64_t n, int HH,
                                                                int m, float scale) {
  __shared__ float P[PHI_MAX_M * (D + 1)];
  __shared__ float gs[(PHI_THREADS / D) * PHI_MAX_M];
  stage_proj_padded<D>(P, proj, m);
  const int C = HH * D;
  const int k = threadIdx.x % D, slot = threadIdx.x / D;
  float* g = gs + slot * m;
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const bool on = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x;
    const float* gr = dphi + r * ld_dphi;
    for (int c0 = 0; c0 < C; c0 += PHI_THREADS) {   // block-uniform trip count: the barriers below are reached by all
      const int c = c0 + threadIdx.x, hh = c / D;
      const bool act = on && c < C;
      if (act) {
        for (int mm = k; mm < m; mm += D) {
          const float pre = proj_dot_v<D>(xr + hh * D, P + mm * (D + 1), scale);
          g[mm] = pre > 0.f ? gr[hh * m + mm] : 0.f;
        }
      }
      __syncthreads();
      if (c < C) {
        float acc = 0.f;
        if (on) {
          for (int mm = 0; mm < m; ++mm) acc = fmaf(g[mm], P[mm * (D + 1) + k], acc);
          acc *= scale;
        }
        dx[r * ld_dx + c] = acc;
      }
      __syncthreads();
    }
  }
}

// d == 8, m == 16: the backward twin of k_phi_relu_fwd_8x16.  Thread (head, feature mm) recomputes ITS pre-activation
// (the forward's fmaf chain: identical gate), the 16 gated gradients of a head meet in LDS, and the head's threads
// mm < 8 contract them with column k = mm of the projection (held in registers), in feature order like the generic
// kernel.  8 + 16 multiply-adds per thread instead of 16 x 8 + 16 per output element and no 64-bit division.
__global__ __launch_bounds__(PHI_THREADS) void k_phi_relu_bwd_8x16(const float* __restrict__ x, int64_t ld_x,
                                                                   const float* __restrict__ proj,
                                                                   const uint8_t* __restrict__ kind, unsigned keep,
                                                                   const float* __restrict__ dphi, int64_t ld_dphi,
                                                                   float* __restrict__ dx, int64_t ld_dx, int64_t n, int HH,
                                                                   float scale) {
  __shared__ float gs[PHI_THREADS];
  const int mm = threadIdx.x & 15, k = mm & 7;
  float pr[8], pk[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) pr[j] = proj[mm * 8 + j];
#pragma unroll
  for (int j = 0; j < 16; ++j) pk[j] = proj[j * 8 + k];
  const int F = HH * 16;
  const float* g16 = gs + (threadIdx.x & ~15);
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const bool on = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x;
    for (int f0 = 0; f0 < F; f0 += PHI_THREADS) {   // block-uniform trip count: the barriers below are reached by all
      const int f = f0 + threadIdx.x, hh = f >> 4;
      const bool act = f < F;
      float gm = 0.f;
      if (on && act) {
        const float4 a = *reinterpret_cast<const float4*>(xr + hh * 8);
        const float4 b = *reinterpret_cast<const float4*>(xr + hh * 8 + 4);
        float s = 0.f;
        s = fmaf(a.x * scale, pr[0], s); s = fmaf(a.y * scale, pr[1], s);
        s = fmaf(a.z * scale, pr[2], s); s = fmaf(a.w * scale, pr[3], s);
        s = fmaf(b.x * scale, pr[4], s); s = fmaf(b.y * scale, pr[5], s);
        s = fmaf(b.z * scale, pr[6], s); s = fmaf(b.w * scale, pr[7], s);
        if (s > 0.f) gm = dphi[r * ld_dphi + f];
      }
      gs[threadIdx.x] = gm;
      __syncthreads();
      if (act && mm < 8) {
        float acc = 0.f;
        if (on) {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc = fmaf(g16[j], pk[j], acc);
          acc *= scale;
        }
        dx[r * ld_dx + hh * 8 + k] = acc;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ softmax maps
// mode 1 (query): stabiliser = max over the head's m features; mode 2 (key): stabiliser = gstat[0]
__global__ __launch_bounds__(PHI_THREADS) void k_phi_softmax_fwd(const float* __restrict__ x, int64_t ld_x,
                                                                 const float* __restrict__ proj,
                                                                 const uint8_t* __restrict__ kind, unsigned keep,
                                                                 float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                 int HH, int d, int m, float scale, float eps,
                                                                 float ratio, int mode,
                                                                 const float* __restrict__ gstat) {
  __shared__ float P[PHI_MAX_PROJ];
  stage_proj(P, proj, m * d);
  const int64_t total = n * HH;
  const float gmax = (mode == HOTK_PHI_SOFTMAX_K) ? gstat[0] : 0.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / HH;
    int hh = (int)(t - r * HH);
    float* o = phi + r * ld_phi + (int64_t)hh * m;
    if (kind && !hotk_keep(keep, kind[r])) {
      for (int mm = 0; mm < m; ++mm) o[mm] = 0.f;
      continue;
    }
    const float* xr = x + r * ld_x + hh * d;
    float nrm = 0.f;
    for (int k = 0; k < d; ++k) nrm = fmaf(xr[k], xr[k], nrm);
    const float diag = nrm * 0.5f * scale * scale;
    float stab = gmax;
    if (mode == HOTK_PHI_SOFTMAX_Q) {
      stab = -INFINITY;
      for (int mm = 0; mm < m; ++mm) stab = fmaxf(stab, proj_dot(xr, P + mm * d, d, scale));
    }
    for (int mm = 0; mm < m; ++mm) {
      float pre = proj_dot(xr, P + mm * d, d, scale);
      o[mm] = ratio * (expf(pre - diag - stab) + eps);
    }
  }
}

// thread per (row, head, k)
__global__ __launch_bounds__(PHI_THREADS) void k_phi_softmax_bwd(const float* __restrict__ x, int64_t ld_x,
                                                                 const float* __restrict__ proj,
                                                                 const uint8_t* __restrict__ kind, unsigned keep,
                                                                 const float* __restrict__ dphi, int64_t ld_dphi,
                                                                 float* __restrict__ dx, int64_t ld_dx, int64_t n,
                                                                 int HH, int d, int m, float scale, float ratio,
                                                                 int mode, const float* __restrict__ gstat) {
  __shared__ float P[PHI_MAX_PROJ];
  stage_proj(P, proj, m * d);
  const int64_t C = (int64_t)HH * d, total = n * C;
  float gmax = 0.f, gshare = 0.f;
  if (mode == HOTK_PHI_SOFTMAX_K) {
    gmax = gstat[0];
    gshare = gstat[1] / fmaxf(gstat[2], 1e-30f);  // d loss / d(-max), evenly split over the maxima
  }
  // rows that every rank of an edge-sharded graph holds (everything but off-diagonal entries) take their share of the
  // maximum's gradient with weight gstat[3] = 1/world: the caller adds the gradients of replicated rows over the ranks
  const float wrep = (mode == HOTK_PHI_SOFTMAX_K && gstat[3] > 0.f) ? gstat[3] : 1.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / C;
    int c = (int)(t - r * C);
    int hh = c / d, k = c - hh * d;
    const bool live = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x + hh * d;
    const float* gr = dphi + r * ld_dphi + (int64_t)hh * m;
    float nrm = 0.f;
    for (int kk = 0; kk < d; ++kk) nrm = fmaf(xr[kk], xr[kk], nrm);
    const float diag = nrm * 0.5f * scale * scale;
    float acc = 0.f, asum = 0.f;
    if (mode == HOTK_PHI_SOFTMAX_Q) {
      if (live) {
        float stab = -INFINITY;
        int arg = 0;
        for (int mm = 0; mm < m; ++mm) {
          float pre = proj_dot(xr, P + mm * d, d, scale);
          if (pre > stab) { stab = pre; arg = mm; }
        }
        for (int mm = 0; mm < m; ++mm) {
          float pre = proj_dot(xr, P + mm * d, d, scale);
          float a = gr[mm] * ratio * expf(pre - diag - stab);
          asum += a;
          acc = fmaf(a, P[mm * d + k], acc);
        }
        acc -= asum * P[arg * d + k];
      }
    } else {
      for (int mm = 0; mm < m; ++mm) {
        float pre = proj_dot(xr, P + mm * d, d, scale);
        float a = live ? gr[mm] * ratio * expf(pre - diag - gmax) : 0.f;
        asum += a;
        float dpre = a - ((pre == gmax) ? gshare * ((kind && kind[r] == 2) ? 1.f : wrep) : 0.f);
        acc = fmaf(dpre, P[mm * d + k], acc);
      }
    }
    dx[r * ld_dx + c] = scale * acc - asum * scale * scale * xr[k];
  }
}

// The softmax maps for head widths 8 / 16 / 32 with 16-byte aligned rows, as row-walking kernels like the ReLU ones
// above (the thread-per-(row, head) kernels write a head's m features as m stores 4 m bytes apart and recompute all m
// pre-activations in every one of a head's d backward threads).  Same fmaf chains, same maximum / summation order:
// identical bits.
constexpr int PHI_SM_MAXF = 4096;      // HH m floats of pre-activations per row held in LDS by the forward kernel
template <int D>
__device__ __forceinline__ float head_half_sqnorm(const float* __restrict__ xs, float scale) {
  float nrm = 0.f;
#pragma unroll
  for (int q = 0; q < D / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(xs + 4 * q);
    nrm = fmaf(v.x, v.x, nrm); nrm = fmaf(v.y, v.y, nrm); nrm = fmaf(v.z, v.z, nrm); nrm = fmaf(v.w, v.w, nrm);
  }
  return nrm * 0.5f * scale * scale;
}

// forward: a block walks rows, a thread takes features f, f + 256, ...; the pre-activations of the row meet in LDS so
// that a query feature finds the maximum of its head (scanned in feature order, as the reference loop does)
template <int D>
__global__ __launch_bounds__(PHI_THREADS) void k_phi_softmax_fwd_d(const float* __restrict__ x, int64_t ld_x,
                                                                   const float* __restrict__ proj,
                                                                   const uint8_t* __restrict__ kind, unsigned keep,
                                                                   float* __restrict__ phi, int64_t ld_phi, int64_t n,
                                                                   int HH, int m, float scale, float eps, float ratio,
                                                                   int mode, const float* __restrict__ gstat) {
  __shared__ float P[PHI_MAX_M * (D + 1)];
  __shared__ float pre_s[PHI_SM_MAXF];
  stage_proj_padded<D>(P, proj, m);
  const int F = HH * m;
  const bool qmode = mode == HOTK_PHI_SOFTMAX_Q;
  const float gmax = qmode ? 0.f : gstat[0];
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const bool on = !kind || hotk_keep(keep, kind[r]);      // block-uniform
    const float* xr = x + r * ld_x;
    float* pr = phi + r * ld_phi;
    if (!on) {
      for (int f = threadIdx.x; f < F; f += PHI_THREADS) pr[f] = 0.f;
      continue;
    }
    for (int f = threadIdx.x; f < F; f += PHI_THREADS) {
      const int hh = f / m, mm = f - hh * m;
      pre_s[f] = proj_dot_v<D>(xr + hh * D, P + mm * (D + 1), scale);
    }
    if (qmode) __syncthreads();
    for (int f = threadIdx.x; f < F; f += PHI_THREADS) {
      const int hh = f / m;
      float stab = gmax;
      if (qmode) {
        stab = -INFINITY;
        const float* ph = pre_s + hh * m;
        for (int j = 0; j < m; ++j) stab = fmaxf(stab, ph[j]);
      }
      const float diag = head_half_sqnorm<D>(xr + hh * D, scale);
      pr[f] = ratio * (expf(pre_s[f] - diag - stab) + eps);
    }
    if (qmode) __syncthreads();      // the next row overwrites pre_s
  }
}

// backward: thread (head-slot, k) like k_phi_relu_bwd_d.  Phase A: the head's pre-activations (thread k: features k,
// k + D, ...) -> LDS; phase B: a[mm] = g ratio exp(pre - diag - stab) and, for the key map, the share of the global
// maximum's gradient -> LDS; phase C: thread k contracts with column k of the projection in feature order.
template <int D>
__global__ __launch_bounds__(PHI_THREADS) void k_phi_softmax_bwd_d(const float* __restrict__ x, int64_t ld_x,
                                                                   const float* __restrict__ proj,
                                                                   const uint8_t* __restrict__ kind, unsigned keep,
                                                                   const float* __restrict__ dphi, int64_t ld_dphi,
                                                                   float* __restrict__ dx, int64_t ld_dx, int64_t n,
                                                                   int HH, int m, float scale, float ratio, int mode,
                                                                   const float* __restrict__ gstat) {
  constexpr int SLOTS = PHI_THREADS / D;
  __shared__ float P[PHI_MAX_M * (D + 1)];
  __shared__ float pre_s[SLOTS * PHI_MAX_M];
  __shared__ float a_s[SLOTS * PHI_MAX_M];
  stage_proj_padded<D>(P, proj, m);
  const int C = HH * D;
  const int k = threadIdx.x % D, slot = threadIdx.x / D;
  float* pre = pre_s + slot * m;
  float* av = a_s + slot * m;
  const bool qmode = mode == HOTK_PHI_SOFTMAX_Q;
  float gmax = 0.f, gshare = 0.f, wrep = 1.f;
  if (!qmode) {
    gmax = gstat[0];
    gshare = gstat[1] / fmaxf(gstat[2], 1e-30f);
    if (gstat[3] > 0.f) wrep = gstat[3];
  }
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const bool live = !kind || hotk_keep(keep, kind[r]);
    const float wr = (kind && kind[r] == 2) ? 1.f : wrep;
    const float* xr = x + r * ld_x;
    const float* gr = dphi + r * ld_dphi;
    for (int c0 = 0; c0 < C; c0 += PHI_THREADS) {   // block-uniform trip count: the barriers below are reached by all
      const int c = c0 + threadIdx.x, hh = c / D;
      const bool act = c < C;
      if (act)
        for (int mm = k; mm < m; mm += D) pre[mm] = proj_dot_v<D>(xr + hh * D, P + mm * (D + 1), scale);
      __syncthreads();
      float stab = gmax, diag = 0.f;
      int arg = 0;
      if (act) {
        diag = head_half_sqnorm<D>(xr + hh * D, scale);
        if (qmode) {
          stab = -INFINITY;
          for (int mm = 0; mm < m; ++mm)
            if (pre[mm] > stab) { stab = pre[mm]; arg = mm; }
        }
        for (int mm = k; mm < m; mm += D) av[mm] = live ? gr[hh * m + mm] * ratio * expf(pre[mm] - diag - stab) : 0.f;
      }
      __syncthreads();
      if (act) {
        float acc = 0.f, asum = 0.f;
        if (qmode) {
          if (live) {
            for (int mm = 0; mm < m; ++mm) {
              const float a = av[mm];
              asum += a;
              acc = fmaf(a, P[mm * (D + 1) + k], acc);
            }
            acc -= asum * P[arg * (D + 1) + k];
          }
        } else {
          for (int mm = 0; mm < m; ++mm) {
            const float a = av[mm];
            asum += a;
            const float dpre = a - ((pre[mm] == gmax) ? gshare * wr : 0.f);
            acc = fmaf(dpre, P[mm * (D + 1) + k], acc);
          }
        }
        dx[r * ld_dx + c] = scale * acc - asum * scale * scale * xr[c];
      }
      __syncthreads();
    }
  }
}

bool phi_softmax_vec_ok(const float* x, int64_t ld_x, int HH, int d, int m) {
  return (d == 8 || d == 16 || d == 32) && (ld_x & 3) == 0 && ((uintptr_t)x & 15) == 0 && m <= PHI_MAX_M &&
         (int64_t)HH * m <= PHI_SM_MAXF;
}

// ------------------------------------------------------------------ global statistics for the key branch
__device__ __forceinline__ float block_reduce_max(float v, float* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = lds[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = fmaxf(r, lds[k]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += lds[k];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(PHI_THREADS) void k_phi_key_max_partial(const float* __restrict__ x, int64_t ld_x,
                                                                     const float* __restrict__ proj, int64_t n, int HH,
                                                                     int d, int m, float scale,
                                                                     float* __restrict__ part) {
  __shared__ float P[PHI_MAX_PROJ];
  __shared__ float red[PHI_THREADS / 64];
  stage_proj(P, proj, m * d);
  const int64_t F = (int64_t)HH * m, total = n * F;
  float mx = -INFINITY;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / F;
    int f = (int)(t - r * F);
    int hh = f / m, mm = f - hh * m;
    mx = fmaxf(mx, proj_dot(x + r * ld_x + hh * d, P + mm * d, d, scale));
  }
  mx = block_reduce_max(mx, red);
  if (threadIdx.x == 0) part[blockIdx.x] = mx;
}

__global__ __launch_bounds__(PHI_THREADS) void k_reduce_max_final(const float* __restrict__ part, int nb,
                                                                  float* __restrict__ out) {
  __shared__ float red[PHI_THREADS / 64];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) mx = fmaxf(mx, part[i]);
  mx = block_reduce_max(mx, red);
  if (threadIdx.x == 0) out[0] = mx;
}

__global__ __launch_bounds__(PHI_THREADS) void k_phi_key_bwd_partial(const float* __restrict__ x, int64_t ld_x,
                                                                     const float* __restrict__ proj,
                                                                     const uint8_t* __restrict__ kind, unsigned keep,
                                                                     const float* __restrict__ dphi, int64_t ld_dphi,
                                                                     int64_t n, int HH, int d, int m, float scale,
                                                                     float ratio, const float* __restrict__ gstat,
                                                                     float* __restrict__ part, int nb) {
  __shared__ float P[PHI_MAX_PROJ];
  __shared__ float red[PHI_THREADS / 64];
  stage_proj(P, proj, m * d);
  const float gmax = gstat[0];
  const float wrep = gstat[3] > 0.f ? gstat[3] : 1.f;   // weight of a maximum on a row every rank holds (see k_phi_softmax_bwd)
  const int64_t total = n * HH;
  float s = 0.f, cnt = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / HH;
    int hh = (int)(t - r * HH);
    const bool live = !kind || hotk_keep(keep, kind[r]);
    const float* xr = x + r * ld_x + hh * d;
    const float* gr = dphi + r * ld_dphi + (int64_t)hh * m;
    float nrm = 0.f;
    for (int k = 0; k < d; ++k) nrm = fmaf(xr[k], xr[k], nrm);
    const float diag = nrm * 0.5f * scale * scale;
    for (int mm = 0; mm < m; ++mm) {
      float pre = proj_dot(xr, P + mm * d, d, scale);
      if (pre == gmax) cnt += (kind && kind[r] == 2) ? 1.f : wrep;
      if (live) s += gr[mm] * ratio * expf(pre - diag - gmax);
    }
  }
  s = block_reduce_sum(s, red);
  cnt = block_reduce_sum(cnt, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    part[nb + blockIdx.x] = cnt;
  }
}

__global__ __launch_bounds__(PHI_THREADS) void k_reduce_sum2_final(const float* __restrict__ part, int nb,
                                                                   float* __restrict__ out) {
  __shared__ float red[PHI_THREADS / 64];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    s += part[i];
    c += part[nb + i];
  }
  s = block_reduce_sum(s, red);
  c = block_reduce_sum(c, red);
  if (threadIdx.x == 0) {
    out[1] = s;
    out[2] = c;
  }
}

constexpr int STATS_BLOCKS = 1024;

// the row-walking ReLU kernels: head width 4 / 8 / 16 / 32, 16-byte aligned head slices, padded projection fits LDS
bool phi_vec_ok(const float* x, int64_t ld_x, int d, int m) {
  return (d == 4 || d == 8 || d == 16 || d == 32) && (ld_x & 3) == 0 && ((uintptr_t)x & 15) == 0 && m <= PHI_MAX_M;
}

int phi_check(int64_t n, int HH, int d, int m, int mode) {
  if (n < 0 || HH <= 0 || d <= 0 || m <= 0) return HOTK_EINVAL;
  if (mode < 0 || mode > 2) return HOTK_EINVAL;
  if ((int64_t)m * d > PHI_MAX_PROJ) return HOTK_EUNSUPPORTED;
  return HOTK_OK;
}

}  // namespace

extern "C" int hotk_phi_fwd(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind, unsigned keep,
                            float* phi, int64_t ld_phi, int64_t n, int HH, int d, int m, int mode, const float* gstat,
                            void* stream) {
  int rc = phi_check(n, HH, d, m, mode);
  if (rc) return rc;
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && proj && phi && ld_x >= (int64_t)HH * d && ld_phi >= (int64_t)HH * m);
  HOTK_CHECK_ARG(mode != HOTK_PHI_SOFTMAX_K || gstat);
  hipStream_t st = (hipStream_t)stream;
  const float scale = powf((float)d, -0.25f);
  if (mode == HOTK_PHI_RELU && d == 8 && m == 16 && (ld_x & 3) == 0 && ((uintptr_t)x & 15) == 0) {
    if ((HH & 3) == 0 && HH <= 32 && (ld_x & 3) == 0 && hotk_aligned16(x)) {
      if (HH == 32)
        hipLaunchKernelGGL(k_phi_relu_fwd_8x16_w<8>, dim3(hotk_grid(n, PHI_THREADS / 64, 256 * 8)), dim3(PHI_THREADS), 0, st, x,
                           ld_x, proj, kind, keep, phi, ld_phi, n, HH, scale, 1e-3f);
      else
        hipLaunchKernelGGL(k_phi_relu_fwd_8x16_w<0>, dim3(hotk_grid(n, PHI_THREADS / 64, 256 * 8)), dim3(PHI_THREADS), 0, st, x,
                           ld_x, proj, kind, keep, phi, ld_phi, n, HH, scale, 1e-3f);
    } else
      hipLaunchKernelGGL(k_phi_relu_fwd_8x16, dim3(hotk_grid(n * HH * 16, PHI_THREADS * PHI_U, 256 * 32)), dim3(PHI_THREADS), 0, st, x,
                         ld_x, proj, kind, keep, phi, ld_phi, n, HH, scale, 1e-3f);
  } else if (mode == HOTK_PHI_RELU && phi_vec_ok(x, ld_x, d, m) && HH * d <= PHI_ROW_MAX && (d == 8 || d == 16 || d == 32)) {
    const dim3 grid(hotk_grid(n, PHI_THREADS / 64, 256 * 8)), block(PHI_THREADS);
    switch (d) {
      case 8: hipLaunchKernelGGL(k_phi_relu_fwd_dw<8>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f); break;
      case 16: hipLaunchKernelGGL(k_phi_relu_fwd_dw<16>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f); break;
      default: hipLaunchKernelGGL(k_phi_relu_fwd_dw<32>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f);
    }
  } else if (mode == HOTK_PHI_RELU && phi_vec_ok(x, ld_x, d, m)) {
    const dim3 grid(hotk_grid(n, 1, 256 * 32)), block(PHI_THREADS);
    switch (d) {
      case 4: hipLaunchKernelGGL(k_phi_relu_fwd_d<4>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f); break;
      case 8: hipLaunchKernelGGL(k_phi_relu_fwd_d<8>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f); break;
      case 16: hipLaunchKernelGGL(k_phi_relu_fwd_d<16>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f); break;
      default: hipLaunchKernelGGL(k_phi_relu_fwd_d<32>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-3f);
    }
  } else if (mode == HOTK_PHI_RELU) {
    hipLaunchKernelGGL(k_phi_relu_fwd, dim3(hotk_grid(n * HH * m, PHI_THREADS)), dim3(PHI_THREADS), 0, st, x, ld_x, proj,
                       kind, keep, phi, ld_phi, n, HH, d, m, scale, 1e-3f);
  } else if (phi_softmax_vec_ok(x, ld_x, HH, d, m)) {
    const dim3 grid(hotk_grid(n, 1, 256 * 32)), block(PHI_THREADS);
    const float ratio = 1.0f / sqrtf((float)m);
    switch (d) {
      case 8: hipLaunchKernelGGL(k_phi_softmax_fwd_d<8>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-4f, ratio, mode, gstat); break;
      case 16: hipLaunchKernelGGL(k_phi_softmax_fwd_d<16>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-4f, ratio, mode, gstat); break;
      default: hipLaunchKernelGGL(k_phi_softmax_fwd_d<32>, grid, block, 0, st, x, ld_x, proj, kind, keep, phi, ld_phi, n, HH, m, scale, 1e-4f, ratio, mode, gstat);
    }
  } else {
    hipLaunchKernelGGL(k_phi_softmax_fwd, dim3(hotk_grid(n * HH, PHI_THREADS)), dim3(PHI_THREADS), 0, st, x, ld_x, proj,
                       kind, keep, phi, ld_phi, n, HH, d, m, scale, 1e-4f, 1.0f / sqrtf((float)m), mode, gstat);
  }
  return hotk_launch_status();
}

extern "C" int hotk_phi_bwd(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind, unsigned keep,
                            const float* dphi, int64_t ld_dphi, float* dx, int64_t ld_dx, int64_t n, int HH, int d,
                            int m, int mode, const float* gstat, void* stream) {
  int rc = phi_check(n, HH, d, m, mode);
  if (rc) return rc;
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && proj && dphi && dx && ld_x >= (int64_t)HH * d && ld_dx >= (int64_t)HH * d &&
                 ld_dphi >= (int64_t)HH * m);
  HOTK_CHECK_ARG(mode != HOTK_PHI_SOFTMAX_K || gstat);
  hipStream_t st = (hipStream_t)stream;
  const float scale = powf((float)d, -0.25f);
  if (mode == HOTK_PHI_RELU && d == 8 && m == 16 && (ld_x & 3) == 0 && ((uintptr_t)x & 15) == 0) {
    hipLaunchKernelGGL(k_phi_relu_bwd_8x16, dim3(hotk_grid(n, 1, 256 * 32)), dim3(PHI_THREADS), 0, st, x, ld_x, proj,
                       kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, scale);
  } else if (mode == HOTK_PHI_RELU && phi_vec_ok(x, ld_x, d, m)) {
    const dim3 grid(hotk_grid(n, 1, 256 * 32)), block(PHI_THREADS);
    switch (d) {
      case 4: hipLaunchKernelGGL(k_phi_relu_bwd_d<4>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale); break;
      case 8: hipLaunchKernelGGL(k_phi_relu_bwd_d<8>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale); break;
      case 16: hipLaunchKernelGGL(k_phi_relu_bwd_d<16>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale); break;
      default: hipLaunchKernelGGL(k_phi_relu_bwd_d<32>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale);
    }
  } else if (mode == HOTK_PHI_RELU) {
    hipLaunchKernelGGL(k_phi_relu_bwd, dim3(hotk_grid(n * HH * d, PHI_THREADS)), dim3(PHI_THREADS), 0, st, x, ld_x, proj,
                       kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, d, m, scale);
  } else if (phi_softmax_vec_ok(x, ld_x, HH, d, m)) {
    const dim3 grid(hotk_grid(n, 1, 256 * 32)), block(PHI_THREADS);
    const float ratio = 1.0f / sqrtf((float)m);
    switch (d) {
      case 8: hipLaunchKernelGGL(k_phi_softmax_bwd_d<8>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale, ratio, mode, gstat); break;
      case 16: hipLaunchKernelGGL(k_phi_softmax_bwd_d<16>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale, ratio, mode, gstat); break;
      default: hipLaunchKernelGGL(k_phi_softmax_bwd_d<32>, grid, block, 0, st, x, ld_x, proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, m, scale, ratio, mode, gstat);
    }
  } else {
    hipLaunchKernelGGL(k_phi_softmax_bwd, dim3(hotk_grid(n * HH * d, PHI_THREADS)), dim3(PHI_THREADS), 0, st, x, ld_x,
                       proj, kind, keep, dphi, ld_dphi, dx, ld_dx, n, HH, d, m, scale, 1.0f / sqrtf((float)m), mode,
                       gstat);
  }
  return hotk_launch_status();
}

extern "C" int hotk_phi_stats_workspace_bytes(int64_t n, int HH, int m, size_t* bytes) {
  HOTK_CHECK_ARG(bytes);
  (void)n; (void)HH; (void)m;
  *bytes = 2 * STATS_BLOCKS * sizeof(float);
  return HOTK_OK;
}

extern "C" int hotk_phi_key_stats(const float* x, int64_t ld_x, const float* proj, int64_t n, int HH, int d, int m,
                                  float* gstat, void* workspace, size_t ws_bytes, void* stream) {
  int rc = phi_check(n, HH, d, m, HOTK_PHI_SOFTMAX_K);
  if (rc) return rc;
  HOTK_CHECK_ARG(n > 0 && x && proj && gstat && workspace && ld_x >= (int64_t)HH * d);
  if (ws_bytes < 2 * STATS_BLOCKS * sizeof(float)) return HOTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float scale = powf((float)d, -0.25f);
  int nb = (int)hotk_grid(n * HH * m, PHI_THREADS, STATS_BLOCKS);
  hipLaunchKernelGGL(k_phi_key_max_partial, dim3(nb), dim3(PHI_THREADS), 0, st, x, ld_x, proj, n, HH, d, m, scale,
                     (float*)workspace);
  hipLaunchKernelGGL(k_reduce_max_final, dim3(1), dim3(PHI_THREADS), 0, st, (const float*)workspace, nb, gstat);
  return hotk_launch_status();
}

extern "C" int hotk_phi_key_bwd_stats(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind,
                                      unsigned keep, const float* dphi, int64_t ld_dphi, int64_t n, int HH, int d,
                                      int m, float* gstat, void* workspace, size_t ws_bytes, void* stream) {
  int rc = phi_check(n, HH, d, m, HOTK_PHI_SOFTMAX_K);
  if (rc) return rc;
  HOTK_CHECK_ARG(n > 0 && x && proj && dphi && gstat && workspace && ld_x >= (int64_t)HH * d &&
                 ld_dphi >= (int64_t)HH * m);
  if (ws_bytes < 2 * STATS_BLOCKS * sizeof(float)) return HOTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float scale = powf((float)d, -0.25f);
  int nb = (int)hotk_grid(n * HH, PHI_THREADS, STATS_BLOCKS);
  hipLaunchKernelGGL(k_phi_key_bwd_partial, dim3(nb), dim3(PHI_THREADS), 0, st, x, ld_x, proj, kind, keep, dphi,
                     ld_dphi, n, HH, d, m, scale, 1.0f / sqrtf((float)m), gstat, (float*)workspace, nb);
  hipLaunchKernelGGL(k_reduce_sum2_final, dim3(1), dim3(PHI_THREADS), 0, st, (const float*)workspace, nb, gstat);
  return hotk_launch_status();
}
