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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LB_K = 32;         // input width
constexpr int LB_WAVES = 8;      // waves per workgroup (one LDS copy of W)
constexpr int LB_GS = 34;        // floats per staged row: conflict-free for the row-wise AND the column-wise operand reads
constexpr int LB_OS = 36;        // floats per staged dx row (16-byte aligned)

__device__ __forceinline__ void lb_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// columns [c_lo, c_lo + 32 NCH) of g / W; Ctot = full width (row stride of W and of the partial blocks)
template <int NCH, bool WANT_DX>
__global__ __launch_bounds__(64 * LB_WAVES) void k_linear_bwd(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X,
                                                             int64_t ldx, const float* __restrict__ W, float* __restrict__ DX,
                                                             int64_t lddx, float* __restrict__ partial, int64_t R, int c_lo,
                                                             int Ctot, int add_dx) {
  constexpr int C = 32 * NCH;
  constexpr int CS = C + 2;      // W row stride in LDS: 2 or 34 mod 64 -> bank (2 or 34) k + c parity, all 64 distinct
  extern __shared__ __align__(16) float lds[];
  float* w_lds = lds;                                  // [32][CS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, half = lane >> 5;
  float* gt = lds + LB_K * CS + wave * (32 * LB_GS + 32 * LB_OS);   // staged g chunk [32][GS]
  float* xt = gt + 32 * LB_GS;                                        // staged x tile / dx image [32][OS]
  if (WANT_DX) {
    for (int i = tid; i < LB_K * C; i += 64 * LB_WAVES) w_lds[(i / C) * CS + (i % C)] = W[(i / C) * Ctot + c_lo + (i % C)];
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
    const float* grow[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t rr = row0 + 8 * i + lr;
      live[i] = rr < R;                                // rows past the end enter as zeros
      grow[i] = G + (live[i] ? rr : R - 1) * ldg + c_lo + lc;
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // x tile -> LDS -> A operands of dW: xa[s] = x[row 2s + half][k = j]
    lb_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = live[i] ? *reinterpret_cast<const float4*>(X + (row0 + 8 * i + lr) * ldx + lc) : z4;
      *reinterpret_cast<float4*>(xt + (8 * i + lr) * LB_OS + lc) = t;
    }
    lb_fence();
    float xa[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) xa[s] = xt[(2 * s + half) * LB_OS + j];
    f32x16 dxa;
#pragma unroll
    for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
    float4 nx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nx[i] = live[i] ? *reinterpret_cast<const float4*>(grow[i]) : z4;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      lb_fence();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* d = gt + (8 * i + lr) * LB_GS + lc;
        *reinterpret_cast<float2*>(d) = make_float2(nx[i].x, nx[i].y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(nx[i].z, nx[i].w);
      }
      lb_fence();
      if (ch + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nx[i] = live[i] ? *reinterpret_cast<const float4*>(grow[i] + 32 * (ch + 1)) : z4;
      }
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
    if (!WANT_DX) continue;
    // dx: lane (row j, half) owns k = 8 q + 4 half + (0..3); through the LDS image to full 128-byte row stores
    lb_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(xt + j * LB_OS + 8 * q + 4 * half) =
          make_float4(dxa[4 * q + 0], dxa[4 * q + 1], dxa[4 * q + 2], dxa[4 * q + 3]);
    lb_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (live[i]) {
        float4* d = reinterpret_cast<float4*>(DX + (row0 + 8 * i + lr) * lddx + lc);
        float4 v = *reinterpret_cast<const float4*>(xt + (8 * i + lr) * LB_OS + lc);
        if (add_dx) {
          const float4 o = *d;
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *d = v;
      }
  }
  // partial [34][C] of this wave: dW rows k = (reg & 3) + 8 (reg >> 2) + 4 half, column 32 ch + j; rows 32 + half: column sums
  float* pw = partial + ((int64_t)blockIdx.x * LB_WAVES + wave) * (LB_K + 2) * Ctot + c_lo;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pw[((r & 3) + 8 * (r >> 2) + 4 * half) * Ctot + 32 * ch + j] = dw[ch][r];
    pw[(LB_K + half) * Ctot + 32 * ch + j] = dbp[ch];
  }
}

unsigned lb_grid(int64_t R) {
  const int64_t n_tiles = (R + 32 * LB_WAVES - 1) / (32 * LB_WAVES);
  return (unsigned)(n_tiles < 256 ? (n_tiles < 1 ? 1 : n_tiles) : 256);
}

template <int NCH, bool WANT_DX>
int lb_launch(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W, float* DX, int64_t lddx,
              float* partial, int64_t R, int c_lo, int Ctot, int add_dx, hipStream_t st) {
  constexpr int C = 32 * NCH;
  const size_t lds = ((size_t)LB_K * (C + 2) + (size_t)LB_WAVES * (32 * LB_GS + 32 * LB_OS)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_linear_bwd<NCH, WANT_DX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HOTK_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_linear_bwd<NCH, WANT_DX>), dim3(lb_grid(R)), dim3(64 * LB_WAVES), lds, st, G, ldg, X, ldx, W, DX, lddx, partial, R,
                     c_lo, Ctot, add_dx);
  return hotk_launch_status();
}

}  // namespace

extern "C" int hotk_linear_bwd_supported(int64_t K, int64_t C) { return (K == LB_K && (C == 320 || C == 64)) ? 1 : 0; }

extern "C" int hotk_linear_bwd_partials(int64_t R, int64_t* n_partials) {
  HOTK_CHECK_ARG(n_partials && R >= 0);
  *n_partials = (int64_t)lb_grid(R) * LB_WAVES;
  return HOTK_OK;
}

extern "C" int hotk_linear_bwd_fused(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t K,
                                     int64_t C, float* dx, int64_t lddx, float* partial, int64_t R, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && ldg >= C && ldx >= K && (!dx || lddx >= K));
  if (!hotk_linear_bwd_supported(K, C)) return HOTK_EUNSUPPORTED;
  HOTK_CHECK_ARG(g && x && partial && (!dx || W));
  HOTK_CHECK_ARG(hotk_aligned16(g) && hotk_aligned16(x) && (!dx || hotk_aligned16(dx)) && ((ldg | ldx) & 3) == 0 &&
                 (!dx || (lddx & 3) == 0));
  if (R == 0) {
    // the caller sums lb_grid(0) * WAVES partial blocks: give it zeros
    return hipMemsetAsync(partial, 0, (size_t)LB_WAVES * (LB_K + 2) * C * sizeof(float), (hipStream_t)stream) == hipSuccess
               ? HOTK_OK : HOTK_ELAUNCH;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!dx) {   // weight / bias gradients only (W is not read)
    if (C == 320) {
      const int rc = lb_launch<5, false>(g, ldg, x, ldx, W, dx, lddx, partial, R, 0, 320, 0, st);
      return rc ? rc : lb_launch<5, false>(g, ldg, x, ldx, W, dx, lddx, partial, R, 160, 320, 0, st);
    }
    return lb_launch<2, false>(g, ldg, x, ldx, W, dx, lddx, partial, R, 0, 64, 0, st);
  }
  if (C == 320) {
    const int rc = lb_launch<5, true>(g, ldg, x, ldx, W, dx, lddx, partial, R, 0, 320, 0, st);
    return rc ? rc : lb_launch<5, true>(g, ldg, x, ldx, W, dx, lddx, partial, R, 160, 320, 1, st);
  }
  return lb_launch<2, true>(g, ldg, x, ldx, W, dx, lddx, partial, R, 0, 64, 0, st);
}
