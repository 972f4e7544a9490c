// Backward of a narrow-input linear layer  y[R, C] = x[R, 32] . W[32, C] + b  (fc_v of models/sparse/kernelattn.py:51,
// 65-68: C = 10 dim_v = 320) in ONE pass over the wide gradient g = dy [R, C]:
//     dx[R, 32] = g . W^T            dW[32, C] = x^T . g            db[C] = column sums of g
// The three results were a library GEMM, a split-K batched GEMM over [x | 1] and a concatenation: g (12.8 GB at 10 M
// rows) crossed HBM twice and [x | 1] was written and read once more.  Here a wave takes 32 rows at a time, walks g in
// ten 32-column chunks staged through LDS (full 128-byte row pieces in, next chunk in flight during the products) and
// feeds both products from the same staged chunk with v_mfma_f32_32x32x2_f32 (exact fp32):
//     dW chunk  D[k][c]   += x[row][k] * g[row][c]      A = x (lane = k,  half = row parity), B = g (lane = c,  half = row parity)
//     dx        D[k][row] += W[k][c]   * g[row][c]      A = W (lane = k,  half = c parity),   B = g (lane = row, half = c parity)
// dW stays in 16 registers per chunk for the life of the wave (persistent workgroups) and leaves as one partial
// [34, C] block per wave (rows 32 / 33: the column sums of even / odd rows), summed by the caller in a fixed order.
// A launch covers at most five chunks (80 accumulator registers: two waves per SIMD without spilling); C = 320 is two
// launches over the two column halves, the second adding its part of dx to the first's (every byte of g is still read
// once; dx, 1/10 of g, is re-read once).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// fp32 products as six bf16 MFMAs on exact three-way bf16 splits of both operands (see csrc/gemm.hip)
__device__ __forceinline__ uint32_t lb_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float lb_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void lb_split3_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t h0 = lb_f2u(x0) & 0xffff0000u, h1 = lb_f2u(x1) & 0xffff0000u;
  const float r0 = x0 - lb_u2f(h0), r1 = x1 - lb_u2f(h1);
  const uint32_t m0 = lb_f2u(r0) & 0xffff0000u, m1 = lb_f2u(r1) & 0xffff0000u;
  const float s0 = r0 - lb_u2f(m0), s1 = r1 - lb_u2f(m1);
  hi = (h0 >> 16) | h1;
  mid = (m0 >> 16) | m1;
  lo = (lb_f2u(s0) >> 16) | (lb_f2u(s1) & 0xffff0000u);
}
struct Planes { u32x4 h, m, l; };
// eight fp32 values that are exact bf16 numbers (widened bf16 rows) -> their single plane
__device__ __forceinline__ u32x4 lb_pack8(const float (&v)[8]) {
  u32x4 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) h[q] = (lb_f2u(v[2 * q]) >> 16) | (lb_f2u(v[2 * q + 1]) & 0xffff0000u);
  return h;
}
// four bf16 values (as loaded: two packed pairs) -> fp32
__device__ __forceinline__ float4 lb_widen(const uint2& u) {
  return make_float4(lb_u2f(u.x << 16), lb_u2f(u.x & 0xffff0000u), lb_u2f(u.y << 16), lb_u2f(u.y & 0xffff0000u));
}
__device__ __forceinline__ Planes lb_split8(const float (&v)[8]) {
  Planes p;
  uint32_t h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    lb_split3_pair(v[2 * q], v[2 * q + 1], h, m, l);
    p.h[q] = h; p.m[q] = m; p.l[q] = l;
  }
  return p;
}
__device__ __forceinline__ f32x16 lb_mfma(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A . B with A = ah + am + al, B = bh + bm + bl (the three products below 2^-24 of the result dropped)
__device__ __forceinline__ f32x16 lb_mfma6(const Planes& a, const Planes& b, f32x16 c) {
  c = lb_mfma(a.l, b.h, c);
  c = lb_mfma(a.h, b.l, c);
  c = lb_mfma(a.m, b.m, c);
  c = lb_mfma(a.m, b.h, c);
  c = lb_mfma(a.h, b.m, c);
  c = lb_mfma(a.h, b.h, c);
  return c;
}

constexpr int LB_K = 32;         // input width
constexpr int LB_WAVES = 8;      // waves per workgroup (one LDS copy of W)
constexpr int LB_GS = 34;        // floats per staged row: conflict-free for the row-wise AND the column-wise operand reads
constexpr int LB_OS = 36;        // floats per staged dx row (16-byte aligned)

__device__ __forceinline__ void lb_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// columns [c_lo, c_lo + 32 NCH) of g / W; Ctot = full width (row stride of W and of the partial blocks)
// B6: both products as bf16 MFMAs (32x32x16) on split operands; W then sits in LDS as three bf16 planes
// [plane][chunk][k step][k (32)][16 c] (a lane's A fragment = one ds_read_b128), behind the fp32 copy's place
// GB: g and x are bf16 tensors (the bf16 storage mode; leading dimensions in bf16 elements): half the bytes in, widened to
// fp32 when a piece is staged, the same products and fp32 results
template <int NCH, bool WANT_DX, bool B6 = false, bool GB = false>
__global__ __launch_bounds__(64 * LB_WAVES) void k_linear_bwd(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X,
                                                             int64_t ldx, const float* __restrict__ W, float* __restrict__ DX,
                                                             int64_t lddx, float* __restrict__ partial, int64_t R, int c_lo,
                                                             int Ctot, int add_dx) {
  constexpr int C = 32 * NCH;
  constexpr int CS = C + 2;      // W row stride in LDS: 2 or 34 mod 64 -> bank (2 or 34) k + c parity, all 64 distinct
  extern __shared__ __align__(16) float lds[];
  float* w_lds = lds;                                  // [32][CS]   (B6: 3 x NCH x 2 x 32 x 16 bf16 = 1.5 x 32 x C floats)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, half = lane >> 5;
  constexpr int WF = B6 ? (3 * LB_K * C) / 2 : LB_K * CS;   // floats of LDS taken by W
  float* gt = lds + WF + wave * (32 * LB_GS + 32 * LB_OS);   // staged g chunk [32][GS]
  float* xt = gt + 32 * LB_GS;                                        // staged x tile / dx image [32][OS]
  if (WANT_DX) {
    if constexpr (B6) {
      uint16_t* wb = reinterpret_cast<uint16_t*>(lds);
      constexpr int PL = NCH * 2 * 32 * 16;            // bf16 elements per plane
      for (int i = tid; i < LB_K * (C / 2); i += 64 * LB_WAVES) {   // two consecutive columns of one k
        const int k = i / (C / 2), c = 2 * (i - k * (C / 2));
        uint32_t hi, mid, lo;
        lb_split3_pair(W[k * Ctot + c_lo + c], W[k * Ctot + c_lo + c + 1], hi, mid, lo);
        const int o = (((c >> 5) * 2 + ((c >> 4) & 1)) * 32 + k) * 16 + (c & 15);
        wb[o] = (uint16_t)hi; wb[o + 1] = (uint16_t)(hi >> 16);
        wb[PL + o] = (uint16_t)mid; wb[PL + o + 1] = (uint16_t)(mid >> 16);
        wb[2 * PL + o] = (uint16_t)lo; wb[2 * PL + o + 1] = (uint16_t)(lo >> 16);
      }
    } else {
      for (int i = tid; i < LB_K * C; i += 64 * LB_WAVES) w_lds[(i / C) * CS + (i % C)] = W[(i / C) * Ctot + c_lo + (i % C)];
    }
    __syncthreads();
  }
  f32x16 dw[NCH];
  float dbp[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    dbp[ch] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[ch][r] = 0.f;
  }
  const int lr = lane >> 3, lc = 4 * (lane & 7);       // staging: rows 8 i + lr, float4 column lc
  const int64_t n_tiles = (R + 32 * LB_WAVES - 1) / (32 * LB_WAVES);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * (32 * LB_WAVES) + wave * 32;
    if (row0 >= R) continue;
    // Every load below is UNCONDITIONAL (rows past the end re-read row R - 1 and are zeroed by a select) and addressed
    // as G / X / DX + offset: behind `live ? *p : 0` each load sat in a block of its own with a full wait after it, and
    // an array of row POINTERS went through scratch memory and came back as generic pointers -- flat loads, which
    // count against the LDS counter as well (found in the ISA at the end of round 2: 13 dependent round trips per
    // 32-row tile at two waves per SIMD).
    int64_t goff[4], xoff[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t rr = row0 + 8 * i + lr;
      live[i] = rr < R;                                // rows past the end enter as zeros
      const int64_t rc = live[i] ? rr : R - 1;
      goff[i] = rc * ldg + c_lo + lc;
      xoff[i] = rc * ldx + lc;
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the zeros of a partial tile -- the last one only -- are written over the staged rows on a wave-uniform branch: a
    // select on the loaded value is scheduled right behind the load and drags the wait for it in front of the products)
    const bool partial_tile = __builtin_amdgcn_readfirstlane((int)(row0 + 32 > R)) != 0;
    // x tile -> LDS -> A operands of dW: xa[s] = x[row 2s + half][k = j]
    lb_fence();
    {
      float4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (GB) t[i] = lb_widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(X) + xoff[i]));
        else t[i] = *reinterpret_cast<const float4*>(X + xoff[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(xt + (8 * i + lr) * LB_OS + lc) = t[i];
      if (partial_tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!live[i]) *reinterpret_cast<float4*>(xt + (8 * i + lr) * LB_OS + lc) = z4;
      }
    }
    lb_fence();
    float xa[B6 ? 1 : 16];
    Planes xp[B6 ? 2 : 1];   // B6: A fragments of dW, x[row 16 ks + 8 half + e][k = j]
    if constexpr (B6) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xt[(16 * ks + 8 * half + e) * LB_OS + j];
        if constexpr (GB) xp[ks].h = lb_pack8(v);      // (bf16 rows: a value is its own plane, the other two are zero)
        else xp[ks] = lb_split8(v);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) xa[s] = xt[(2 * s + half) * LB_OS + j];
    }
    f32x16 dxa;
#pragma unroll
    for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
    typename std::conditional<GB, uint2, float4>::type nx[4];      // (GB: kept as loaded, widened at the staging write)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (GB) nx[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(G) + goff[i]);
      else nx[i] = *reinterpret_cast<const float4*>(G + goff[i]);
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      lb_fence();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* d = gt + (8 * i + lr) * LB_GS + lc;
        float4 w4;
        if constexpr (GB) w4 = lb_widen(nx[i]);
        else w4 = nx[i];
        *reinterpret_cast<float2*>(d) = make_float2(w4.x, w4.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(w4.z, w4.w);
      }
      if (partial_tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!live[i]) {
            float* d = gt + (8 * i + lr) * LB_GS + lc;
            *reinterpret_cast<float2*>(d) = make_float2(0.f, 0.f);
            *reinterpret_cast<float2*>(d + 2) = make_float2(0.f, 0.f);
          }
      }
      lb_fence();
      if (ch + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (GB) nx[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(G) + goff[i] + 32 * (ch + 1));
          else nx[i] = *reinterpret_cast<const float4*>(G + goff[i] + 32 * (ch + 1));
        }
      }
      // (the next chunk's loads stay HERE, in front of this chunk's products: at 250 registers the scheduler otherwise
      // sinks them to just before their use and the wave waits out the whole round trip)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B6) {
        const u32x4* wq = reinterpret_cast<const u32x4*>(lds);
        constexpr int PLQ = NCH * 2 * 32 * 2;            // u32x4 (8 bf16) per plane
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          // dW: B fragment g[row 16 ks + 8 half + e][c = j]; the same values feed the column sums
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = gt[(16 * ks + 8 * half + e) * LB_GS + j];
            dbp[ch] += v[e];
          }
          Planes gb;
          if constexpr (GB) {
            gb.h = lb_pack8(v);
            dw[ch] = lb_mfma(xp[ks].h, gb.h, dw[ch]);      // bf16 x bf16: one exact product
          } else {
            gb = lb_split8(v);
            dw[ch] = lb_mfma6(xp[ks], gb, dw[ch]);
          }
          if (WANT_DX) {
            // dx: A fragment W[k = j][32 ch + 16 ks + 8 half + e], B fragment g[row j][16 ks + 8 half + e]
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const float2 t = *reinterpret_cast<const float2*>(gt + j * LB_GS + 16 * ks + 8 * half + e);
              u[e] = t.x; u[e + 1] = t.y;
            }
            const int wo = ((ch * 2 + ks) * 32 + j) * 2 + half;
            Planes wp;
            wp.h = wq[wo]; wp.m = wq[PLQ + wo]; wp.l = wq[2 * PLQ + wo];
            if constexpr (GB) {
              const u32x4 gh = lb_pack8(u);               // fp32 weight (three planes) x bf16 gradient (one): three products
              dxa = lb_mfma(wp.l, gh, dxa);
              dxa = lb_mfma(wp.m, gh, dxa);
              dxa = lb_mfma(wp.h, gh, dxa);
            } else {
              const Planes gr = lb_split8(u);
              dxa = lb_mfma6(wp, gr, dxa);
            }
          }
        }
      } else {
        const float* wl = w_lds + j * CS + 32 * ch + half;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const float gb = gt[(2 * s + half) * LB_GS + j];     // g[row 2s + half][c = j]
          dw[ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], gb, dw[ch], 0, 0, 0);
          dbp[ch] += gb;
          if (WANT_DX) {
            const float gr = gt[j * LB_GS + 2 * s + half];      // g[row j][c = 2s + half]
            dxa = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[2 * s], gr, dxa, 0, 0, 0);
          }
        }
      }
    }
    if (!WANT_DX) continue;
    // dx: lane (row j, half) owns k = 8 q + 4 half + (0..3); through the LDS image to full 128-byte row stores
    lb_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(xt + j * LB_OS + 8 * q + 4 * half) =
          make_float4(dxa[4 * q + 0], dxa[4 * q + 1], dxa[4 * q + 2], dxa[4 * q + 3]);
    lb_fence();
    int64_t doff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) doff[i] = (live[i] ? row0 + 8 * i + lr : R - 1) * lddx + lc;
    // Rows past the end of a partial tile (the dead lanes) load DX row R - 1 while the live lane of that row in the same
    // tile stores to it: a benign race on an output buffer -- the loaded value is discarded (the store below is under
    // live[i]); the clamp only keeps the load stream unconditional and on a valid address.
    float4 prev[4];
    if (add_dx) {                                      // (uniform: the second column half adds to the first's result)
#pragma unroll
      for (int i = 0; i < 4; ++i) prev[i] = *reinterpret_cast<const float4*>(DX + doff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) prev[i] = z4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = *reinterpret_cast<const float4*>(xt + (8 * i + lr) * LB_OS + lc);
      v.x += prev[i].x; v.y += prev[i].y; v.z += prev[i].z; v.w += prev[i].w;
      if (live[i]) *reinterpret_cast<float4*>(DX + doff[i]) = v;
    }
  }
  // The eight waves of the workgroup add their dW / column-sum registers in LDS, in wave order (fixed: reproducible), and
  // the workgroup leaves ONE partial [34][C] block: dW rows k = (reg & 3) + 8 (reg >> 2) + 4 half, column 32 ch + j; rows
  // 32 + half: column sums of the even / odd rows.  (One block per wave was 2 048 blocks = 89 MB for the caller to sum.)
  __syncthreads();                                     // every wave is done with the staging rows: LDS is reused from 0
  float* acc = lds;                                    // [34][C]
#pragma unroll 1
  for (int w = 0; w < LB_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* a = acc + ((r & 3) + 8 * (r >> 2) + 4 * half) * C + 32 * ch + j;
          *a = (w ? *a : 0.f) + dw[ch][r];
        }
        float* b = acc + (LB_K + half) * C + 32 * ch + j;
        *b = (w ? *b : 0.f) + dbp[ch];
      }
    }
    __syncthreads();
  }
  float* pw = partial + (int64_t)blockIdx.x * (LB_K + 2) * Ctot + c_lo;
  for (int i = tid; i < (LB_K + 2) * C; i += 64 * LB_WAVES) pw[(i / C) * Ctot + (i % C)] = acc[i];
}

unsigned lb_grid(int64_t R) {
  const int64_t n_tiles = (R + 32 * LB_WAVES - 1) / (32 * LB_WAVES);
  return (unsigned)(n_tiles < 256 ? (n_tiles < 1 ? 1 : n_tiles) : 256);
}

// HOTK_LINBWD_B6=1 selects the split-bf16 form.  Measured at 10 M rows x 320 and NOT adopted: 5.5 ms against 5.0 ms for
// the fp32 MFMAs -- here every staged gradient value is split twice (once per product orientation: 32 values and ~10
// VALU instructions each per lane and chunk), which costs more than the 1 280 matrix cycles per chunk it saves; the
// thin GEMMs split a value once per 288 output columns.
bool lb_f32() {
  static const int v = [] { const char* e = getenv("HOTK_LINBWD_B6"); return (e && e[0] == '1') ? 0 : 1; }();
  return v != 0;
}

template <int NCH, bool WANT_DX, bool B6, bool GB = false>
int lb_launch_t(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W, float* DX, int64_t lddx,
              float* partial, int64_t R, int c_lo, int Ctot, int add_dx, hipStream_t st) {
  constexpr int C = 32 * NCH;
  const size_t lds = ((size_t)(B6 ? (3 * LB_K * C) / 2 : LB_K * (C + 2)) + (size_t)LB_WAVES * (32 * LB_GS + 32 * LB_OS)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_linear_bwd<NCH, WANT_DX, B6, GB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HOTK_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_linear_bwd<NCH, WANT_DX, B6, GB>), dim3(lb_grid(R)), dim3(64 * LB_WAVES), lds, st, G, ldg, X, ldx, W, DX, lddx, partial, R,
                     c_lo, Ctot, add_dx);
  return hotk_launch_status();
}

template <int NCH, bool WANT_DX, bool GB = false>
int lb_launch(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W, float* DX, int64_t lddx,
              float* partial, int64_t R, int c_lo, int Ctot, int add_dx, hipStream_t st) {
  // bf16 rows: everything on the bf16 matrix pipe (8 products of 32 cycles per chunk instead of 32 fp32 ones of 64)
  if constexpr (GB) return lb_launch_t<NCH, WANT_DX, true, true>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
  return lb_f32() ? lb_launch_t<NCH, WANT_DX, false>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st)
                  : lb_launch_t<NCH, WANT_DX, true>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
}

}  // namespace

namespace {
// one launch over the chunks [c_lo, c_lo + 32 nch) of g / W
template <bool WANT_DX, bool GB>
int lb_launch_n(int nch, const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W, float* DX, int64_t lddx,
                float* partial, int64_t R, int c_lo, int Ctot, int add_dx, hipStream_t st) {
  switch (nch) {
    case 1: return lb_launch<1, WANT_DX, GB>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
    case 2: return lb_launch<2, WANT_DX, GB>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
    case 3: return lb_launch<3, WANT_DX, GB>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
    case 4: return lb_launch<4, WANT_DX, GB>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
    default: return lb_launch<5, WANT_DX, GB>(G, ldg, X, ldx, W, DX, lddx, partial, R, c_lo, Ctot, add_dx, st);
  }
}
constexpr int LB_MAX_C = 1024;
}  // namespace

// K = 32 and any width that is a multiple of 32 up to 1 024: launches of at most five 32-column chunks side by side (every
// byte of g is read once; dx, 32 floats per row, is re-read and added to by each launch after the first)
extern "C" int hotk_linear_bwd_supported(int64_t K, int64_t C) {
  return (K == LB_K && C >= 32 && C <= LB_MAX_C && C % 32 == 0) ? 1 : 0;
}

extern "C" int hotk_linear_bwd_partials(int64_t R, int64_t* n_partials) {
  HOTK_CHECK_ARG(n_partials && R >= 0);
  *n_partials = (int64_t)lb_grid(R);
  return HOTK_OK;
}

namespace {
template <bool GB>
int lb_run(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t K, int64_t C, float* dx,
           int64_t lddx, float* partial, int64_t R, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && ldg >= C && ldx >= K && (!dx || lddx >= K));
  if (!hotk_linear_bwd_supported(K, C)) return HOTK_EUNSUPPORTED;
  HOTK_CHECK_ARG(g && x && partial && (!dx || W));
  // row pieces of four values: 16 bytes of fp32, 8 bytes of bf16
  HOTK_CHECK_ARG((((uintptr_t)g | (uintptr_t)x) & (GB ? 7u : 15u)) == 0 && (!dx || hotk_aligned16(dx)) && ((ldg | ldx) & 3) == 0 &&
                 (!dx || (lddx & 3) == 0));
  if (R == 0) {
    // the caller sums lb_grid(0) = 1 partial block: give it zeros
    return hipMemsetAsync(partial, 0, (size_t)(LB_K + 2) * C * sizeof(float), (hipStream_t)stream) == hipSuccess
               ? HOTK_OK : HOTK_ELAUNCH;
  }
  hipStream_t st = (hipStream_t)stream;
  // chunk groups of equal size where the width allows it (C = 320: two launches of five; 864 = 27 chunks: 5 5 5 4 4 4)
  const int chunks = (int)(C / 32), launches = (chunks + 4) / 5;
  int c_lo = 0;
  for (int l = 0; l < launches; ++l) {
    const int nch = chunks / launches + (l < chunks % launches ? 1 : 0);
    const int rc = dx ? lb_launch_n<true, GB>(nch, g, ldg, x, ldx, W, dx, lddx, partial, R, c_lo, (int)C, l > 0 ? 1 : 0, st)
                      : lb_launch_n<false, GB>(nch, g, ldg, x, ldx, W, dx, lddx, partial, R, c_lo, (int)C, 0, st);
    if (rc) return rc;
    c_lo += 32 * nch;
  }
  return HOTK_OK;
}
}  // namespace

extern "C" int hotk_linear_bwd_fused(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t K,
                                     int64_t C, float* dx, int64_t lddx, float* partial, int64_t R, void* stream) {
  return lb_run<false>(g, ldg, x, ldx, W, K, C, dx, lddx, partial, R, stream);
}

// g and x as bf16 tensors (leading dimensions in bf16 elements, multiples of 4); W, dx and the partial blocks are fp32
extern "C" int hotk_linear_bwd_fused_bf16(const void* g, int64_t ldg, const void* x, int64_t ldx, const float* W, int64_t K,
                                          int64_t C, float* dx, int64_t lddx, float* partial, int64_t R, void* stream) {
  return lb_run<true>(reinterpret_cast<const float*>(g), ldg, reinterpret_cast<const float*>(x), ldx, W, K, C, dx, lddx, partial,
                      R, stream);
}
