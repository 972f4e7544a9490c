// Thin GEMMs of the layer with the weight resident in LDS and the row epilogue fused:  Y[R, C] = A[R, K] . W[K, C]
// for R ~ 1e7 rows and K, C <= 320 (the dense projections fc_2 / fc_v / fc_o of models/sparse/kernelattn.py:46-51 and
// their data gradients).  These products are streaming problems: 14 GB of rows move per call while the weight is
// 40-74 KB, so the kernel keeps W in LDS for the life of a persistent workgroup, reads every A row once, feeds
// v_mfma_f32_32x32x2_f32 (exact fp32, MI355X_MICROARCH.md: 256 FLOP/clk/CU) and applies the epilogue on the
// accumulators before the only store:
//   MODE 0: + bias[col] + res[row][col]                                       (nn.Linear forward / data gradient, skip path)
//   MODE 1: the order-2 'light' assembly of models/sparse/linear.py:143-161,293-306,345-346 in factored form:
//           off-diagonal (i,j): acc + yn[i][C..2C) + yn[j][2C..3C) + b_off;  diagonal (v,v): yn[v][0..C) + b_diag;
//           padding: 0   (replaces a library GEMM followed by hotk_assemble22: one pass over Y instead of three)
//
// Work split: a workgroup of 4 waves takes 128 consecutive rows, a wave 32 of them, ONE ROW PER LANE PAIR: the MFMA
// computes the transposed tile  Y^T[32 cols, 32 rows] = W^T . A^T, so lane (j = lane & 31, half = lane >> 5) supplies
// A[row0 + j][half*K/2 + s] as the B operand (its K/2 values sit in registers, loaded as contiguous 16-byte pieces)
// and one conflict-free ds_read_b32 of W as the A operand; MFMA step s contracts the K-pair {s, K/2 + s}.  In the C/D
// map (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); cdna_hip_programming.md, fragment layout)
// a lane then owns, of ITS row, the four consecutive output columns 8 q + 4 half + (0..3) of every quad q = reg >> 2:
// the epilogue is float4 loads / stores, the row's kind and node ids are read once per lane, and the gathered
// addends of the next 32-column tile are requested before the current tile is finished (and before the next MFMA
// loop), so their latency overlaps the matrix work.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- fp32 products on the bf16 matrix pipe (B6 kernels) -------------------------------------------------------------
// An fp32 number is EXACTLY the sum of three bf16 numbers: hi = its top 8 significant bits (mantissa truncated), mid =
// the top 8 bits of the exact remainder, lo = what is left (<= 8 bits).  x . w then is nine bf16 products; the six of
// relative weight >= 2^-16 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) are summed in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16 and the other three (<= 2^-24 of the product each) dropped: 1.9e-7 relative against fp64 on
// [1 M, 64] x [64, 288] (scratch/bf16_split_probe.py; the fp32 library GEMM: 4.8e-7), at 6 x 32 cycles per 16 k instead
// of 8 x 64 for v_mfma_f32_32x32x2_f32.
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// planes of two consecutive k values packed as bf16 pairs (low half = first k)
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t h0 = f2u(x0) & 0xffff0000u, h1 = f2u(x1) & 0xffff0000u;
  const float r0 = x0 - u2f(h0), r1 = x1 - u2f(h1);
  const uint32_t m0 = f2u(r0) & 0xffff0000u, m1 = f2u(r1) & 0xffff0000u;
  const float s0 = r0 - u2f(m0), s1 = r1 - u2f(m1);
  hi = (h0 >> 16) | h1;
  mid = (m0 >> 16) | m1;
  lo = (f2u(s0) >> 16) | (f2u(s1) & 0xffff0000u);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#ifndef HOTK_WIDE_WAVES
#define HOTK_WIDE_WAVES 8
#endif
#ifndef HOTK_ENTRY_WAVES
#define HOTK_ENTRY_WAVES 8   /* measured at 10 M rows: 4 waves 7.9 ms, 8 waves 6.1 ms, 16 waves (128 VGPRs, spills) 6.7 ms */
#endif

struct Epi {
  const float* bias;      // MODE 0: [C] or null
  const float* res;       // MODE 0: residual rows [R, >= C] added to the result, or null
  int64_t ld_res;
  const float* yn;        // MODE 1: node table [Nn, 3C]
  int64_t ld_yn;
  const uint8_t* kind;    // MODE 1: row kinds
  const int32_t* rown;
  const int32_t* coln;
  const float* b_off;     // MODE 1: [C] or null
  const float* b_diag;
  int out_bf16;           // Y is a bf16 tensor (ldy in bf16 elements): the fp32 result is rounded once at the store
};

// The epilogue runs on a row-major image of the 32 x 32 result tile: the accumulators (lane pair = one row, 16 bytes
// per quad) go through a wave-private LDS image and come back as lane = (row r = 8 i + (lane >> 3), float4 column
// c4 = lane & 7), so every global access of the epilogue -- the result store, the gathered node rows, the residual --
// is a full 128-byte row piece shared by 8 consecutive lanes.  (Stores straight from the accumulator layout are 32 rows
// x 32 B per instruction; measured on the 10 M-row entry GEMM: 6.7 ms with them, 3.5 ms with stores and gathers
// compiled out = the MFMA time.)
constexpr int MAX_WAVES = 16;      // waves per workgroup (template parameter NW), all sharing one LDS copy of the weight
constexpr int OS = 36;             // floats per staged result row (32 + 4: conflict-free 16-byte writes)

struct RowMeta {                   // epilogue operands of one result row, kept small (the kernel runs at 128 VGPRs):
  uint32_t o1, o2;                 // MODE 1: element offsets into the node table yn (host checks Nn * ld_yn < 2^32)
  int kd;                          // MODE 1: row kind (0 padding, 1 diagonal, 2 off-diagonal); MODE 0: 1 = row exists
  const float* pres;               // MODE 0: residual row or null
};

// addends of one 32-column tile for this lane's four rows
struct Addend {
  float4 u[4], w[4];
};

template <int MODE>
__device__ __forceinline__ void fetch_addend(Addend& ad, const RowMeta (&rm)[4], const float* __restrict__ yn, bool has_res,
                                             int col) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (MODE == 1) {
      ad.u[i] = *reinterpret_cast<const float4*>(yn + rm[i].o1 + col);
      ad.w[i] = *reinterpret_cast<const float4*>(yn + rm[i].o2 + col);
    } else if (has_res) {
      ad.u[i] = *reinterpret_cast<const float4*>(rm[i].pres + col);
    }
  }
}

__device__ __forceinline__ void gemm_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// B6: the products run as six bf16 MFMAs per 16 k (see above).  The weight sits in LDS as three bf16 planes
// [plane][k / 16][col][16 k] (a lane's A fragment = one ds_read_b128), the lane's 32 row values are split once per tile.
// RES: MODE 0 with a residual; OB: bf16 result.  Compile-time, like FULL below (a row tile with all 32 rows inside the
// matrix): the epilogue then has no branch around its stores, the compiler can COUNT them, and the wait for the
// prefetched addends of the next tile becomes vmcnt(4) instead of vmcnt(0) -- a wait for every store of the tile to
// complete (loads and stores share the in-order counter).
// IB (wide-input stream form only): A is a bf16 tensor (lda in bf16 elements) -- a value is its own plane: the lane's eight k
// values of a step are ONE 16-byte load and the B fragment as loaded, three products per step and column tile, no split
template <int KH, int TG, int MODE, int NW, bool B6 = false, bool RES = false, bool OB = false, bool IB = false>
__global__ __launch_bounds__(64 * NW) void k_thin_gemm(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                          float* __restrict__ Y, int64_t ldy, int64_t R, int C, Epi ep) {
  constexpr int WAVES = NW;
  extern __shared__ __align__(16) float w_lds[];   // [2 KH][C], then NW x [32][OS] result images
  constexpr int K = 2 * KH;
  const int tid = threadIdx.x;
  constexpr int KS = K / 16;                         // B6: k steps
  // B6 with a wide input (K > 64): the row values are loaded and split per k step inside the MFMA loop (one step ahead)
  // instead of being held for the whole tile, and the TG column tiles of a group share each split
  constexpr bool STREAM = B6 && KS > 4;
  static_assert(!B6 || (K % 16 == 0 && (STREAM || TG == 1)), "B6: whole 16-k steps; narrow inputs: one column tile per loop");
  static_assert(!IB || STREAM, "bf16 rows: the wide-input stream form only");
  const int w_floats = B6 ? (3 * K * C) / 2 : K * C; // LDS floats taken by the weight
  if (B6) {
    uint16_t* wb = reinterpret_cast<uint16_t*>(w_lds);
    for (int i = tid * 2; i < K * C; i += 64 * WAVES * 2) {   // two consecutive columns of one k
      const int k = i / C, c = i - k * C;
      const float2 t = *reinterpret_cast<const float2*>(W + i);
      uint32_t hi, mid, lo;
      split3_pair(t.x, t.y, hi, mid, lo);
      const int o = ((k >> 4) * C + c) * 16 + (k & 15);
      wb[o] = (uint16_t)hi; wb[o + 16] = (uint16_t)(hi >> 16);
      wb[KS * C * 16 + o] = (uint16_t)mid; wb[KS * C * 16 + o + 16] = (uint16_t)(mid >> 16);
      wb[2 * KS * C * 16 + o] = (uint16_t)lo; wb[2 * KS * C * 16 + o + 16] = (uint16_t)(lo >> 16);
    }
  } else {
    for (int i = tid * 4; i < K * C; i += 64 * WAVES * 4) *reinterpret_cast<float4*>(w_lds + i) = *reinterpret_cast<const float4*>(W + i);
  }
  // both bias vectors behind the result images (zeros when absent): an epilogue that fetched them from global memory
  // between its stores waited for every store to complete first -- loads and stores share one in-order counter
  float* b_lds = w_lds + w_floats + WAVES * 32 * OS;     // [2][C]: MODE 0: bias | unused; MODE 1: b_diag | b_off
  for (int i = tid; i < C; i += 64 * WAVES) {
    const float* b0 = MODE == 1 ? ep.b_diag : ep.bias;
    b_lds[i] = b0 ? b0[i] : 0.f;
    if (MODE == 1) b_lds[C + i] = ep.b_off ? ep.b_off[i] : 0.f;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, half = lane >> 5;
  float* img = w_lds + w_floats + wave * 32 * OS;
  const int er = lane >> 3, ec = 4 * (lane & 7);       // epilogue: rows 8 i + er, columns ec .. ec + 3 of a tile
  constexpr bool has_res = MODE == 0 && RES;
  const int64_t n_tiles = (R + 32 * WAVES - 1) / (32 * WAVES);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * (32 * WAVES) + wave * 32;
    if (row0 >= R) continue;
    auto row_tile = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int64_t row = (FULL || row0 + j < R) ? row0 + j : R - 1;      // rows past the end re-read the last row and store nothing
    // per-row epilogue operands: MODE 1 folds the three row kinds into two gathered rows, two multipliers and a bias
    // selector (x2 is zero on diagonal and padding rows, so the accumulator contributes nothing there):
    //   off-diagonal: acc + yn[i][C..) + yn[j][2C..) + b_off    diagonal: 0 + yn[v][0..) + 0 + b_diag    padding: 0
    float a[B6 ? 1 : KH];
    constexpr int NF = (B6 && !STREAM) ? KS : 1;
    u32x4 ah[NF], am[NF], al[NF];   // B6: the lane's B fragments, k = 16 ks + 8 half + (0..7)
    // the row values are REQUESTED first, then the epilogue's row kinds / node ids (they do not depend on each other:
    // one round trip for both), then the values are split
    float4 t0[NF], t1[NF];
    if constexpr (STREAM) {
    } else if constexpr (B6) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* ap = A + row * lda + 16 * ks + 8 * half;
        t0[ks] = *reinterpret_cast<const float4*>(ap);
        t1[ks] = *reinterpret_cast<const float4*>(ap + 4);
      }
    } else {
      const float* ap = A + row * lda + half * KH;
#pragma unroll
      for (int s = 0; s < KH; s += 4) {
        const float4 t = *reinterpret_cast<const float4*>(ap + s);
        a[s] = t.x; a[s + 1] = t.y; a[s + 2] = t.z; a[s + 3] = t.w;
      }
    }
    RowMeta rm[4];
    // the twelve kind / node-id loads of a tile are issued together (opaque uses): written as `kd == 2 ? coln[r] : 0` the
    // column id was fetched under a branch on the kind just loaded -- eight dependent round trips per tile in the ISA
    int kd4[4] = {0, 0, 0, 0}, rn4[4] = {0, 0, 0, 0}, cn4[4] = {0, 0, 0, 0};
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t rr = row0 + 8 * i + er;
        const int64_t rc = (FULL || rr < R) ? rr : R - 1;
        kd4[i] = ep.kind[rc];
        rn4[i] = ep.rown[rc];
        cn4[i] = ep.coln[rc];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(kd4[i]), "+v"(rn4[i]), "+v"(cn4[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t rr = row0 + 8 * i + er;
      const bool live = FULL || rr < R;
      const int64_t rc = live ? rr : R - 1;
      rm[i].o1 = rm[i].o2 = 0;
      rm[i].kd = live ? 1 : 0;
      rm[i].pres = nullptr;
      if (MODE == 1) {
        const int kd = kd4[i];
        // padding rows carry node id -1 (plan): they read node row 0 and multiply it by zero
        const int32_t rn = rn4[i], cn = cn4[i];
        const uint32_t yi = (uint32_t)(kd != 0 ? rn : 0) * (uint32_t)ep.ld_yn;
        const uint32_t yj = (uint32_t)(kd == 2 ? cn : 0) * (uint32_t)ep.ld_yn;
        rm[i].o1 = (kd == 2 ? yi + C : yi) + ec;
        rm[i].o2 = (kd == 2 ? yj + 2 * C : yi) + ec;
        rm[i].kd = live ? kd : -1;            // -1: past the end (nothing stored)
      } else if (has_res) {
        rm[i].pres = ep.res + rc * ep.ld_res + ec;
      }
    }
    if constexpr (B6 && !STREAM) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint32_t h, m, l;
        split3_pair(t0[ks].x, t0[ks].y, h, m, l); ah[ks][0] = h; am[ks][0] = m; al[ks][0] = l;
        split3_pair(t0[ks].z, t0[ks].w, h, m, l); ah[ks][1] = h; am[ks][1] = m; al[ks][1] = l;
        split3_pair(t1[ks].x, t1[ks].y, h, m, l); ah[ks][2] = h; am[ks][2] = m; al[ks][2] = l;
        split3_pair(t1[ks].z, t1[ks].w, h, m, l); ah[ks][3] = h; am[ks][3] = m; al[ks][3] = l;
      }
    }
    const float* wl = w_lds + half * KH * C + j;
    const u32x4* wq = reinterpret_cast<const u32x4*>(w_lds) + (j * 2 + half);   // B6: + ((plane * KS + ks) * C + n0) * 2
    // addends are requested one tile ahead (two buffers), BEFORE the current tile's stores: a load issued behind a store
    // completes behind it
    Addend cur, nxt;
    fetch_addend<MODE>(cur, rm, ep.yn, has_res, 0);
    for (int n0 = 0; n0 < C; n0 += 32 * TG) {
      f32x16 acc[TG];
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      if constexpr (STREAM && IB) {
        const uint16_t* ap = reinterpret_cast<const uint16_t*>(A) + row * lda + 8 * half;
        auto fetch = [&](int ks, u32x4& x) __attribute__((always_inline)) {
          const int kc = ks < KS ? ks : KS - 1;
          x = *reinterpret_cast<const u32x4*>(ap + 16 * kc);
        };
        // the weight fragments of a step are read from LDS one step ahead as well (two named sets): read right before their
        // products, their latency stood in front of every step (3.87 ms for the two bf16 products of a layer step against 3.2)
        struct WF { u32x4 h[TG], m[TG], l[TG]; };
        auto wload = [&](int ks, WF& w) __attribute__((always_inline)) {
          const int kc = ks < KS ? ks : KS - 1;
#pragma unroll
          for (int t = 0; t < TG; ++t) {
            w.h[t] = wq[((0 * KS + kc) * C + n0 + 32 * t) * 2];
            w.m[t] = wq[((1 * KS + kc) * C + n0 + 32 * t) * 2];
            w.l[t] = wq[((2 * KS + kc) * C + n0 + 32 * t) * 2];
          }
        };
        auto products = [&](const WF& w, const u32x4& bh) __attribute__((always_inline)) {
#pragma unroll
          for (int t = 0; t < TG; ++t) {
            acc[t] = mfma_bf16(w.l[t], bh, acc[t]);
            acc[t] = mfma_bf16(w.m[t], bh, acc[t]);
            acc[t] = mfma_bf16(w.h[t], bh, acc[t]);
          }
        };
        // rows three steps ahead in four named register sets (16 bytes per lane and step)
        static_assert(KS % 2 == 0 && KS >= 6, "pairs of k steps, at least one round of four");
        if constexpr (!RES) {
          // without a residual the whole k loop is unrolled (every address an immediate, the fragment reads scheduled across
          // steps: 3.2 ms for the two bf16 products of a layer step against 3.7 for the loop below); with one, the epilogue's
          // addends push that form over 256 registers
          u32x4 ring[4];
#pragma unroll
          for (int d = 0; d < 3; ++d) fetch(d, ring[d]);
#pragma unroll
          for (int k2 = 0; k2 < KS; ++k2) {
            fetch(k2 + 3, ring[(k2 + 3) % 4]);
            WF w;
            wload(k2, w);
            __builtin_amdgcn_sched_barrier(0);
            products(w, ring[k2 % 4]);
          }
        } else {
        u32x4 r0, r1, r2, r3;
        WF wa, wb;
        fetch(0, r0); fetch(1, r1); fetch(2, r2);
        wload(0, wa);
        int ks = 0;
#pragma unroll 1
        for (; ks + 4 <= KS; ks += 4) {
          fetch(ks + 3, r3); wload(ks + 1, wb);
          __builtin_amdgcn_sched_barrier(0);
          products(wa, r0);
          fetch(ks + 4, r0); wload(ks + 2, wa);              // (past the end: the last piece again)
          __builtin_amdgcn_sched_barrier(0);
          products(wb, r1);
          fetch(ks + 5, r1); wload(ks + 3, wb);
          __builtin_amdgcn_sched_barrier(0);
          products(wa, r2);
          fetch(ks + 6, r2); wload(ks + 4, wa);
          __builtin_amdgcn_sched_barrier(0);
          products(wb, r3);
        }
        if constexpr (KS % 4 == 2) {      // r0, r1 hold the last two steps, wa the first one's weight fragments
          wload(KS - 1, wb);
          products(wa, r0);
          products(wb, r1);
        }
        }
      } else if constexpr (STREAM) {
        // One step ahead.  At K = 320 -> 32 with a residual the kernel moves 15.4 GB in 3.1-3.4 ms (4.5-5 TB/s): it is near the
        // streaming ceiling, and every deeper or wider form of the row stream measured slower -- 8-k pieces 2 / 4 / 6
        // steps ahead 5.5 / 5.1 / 3.3 ms, a lane's 64 contiguous bytes of each line two lines ahead (k order permuted
        // in the weight planes) 3.65 ms, whole lines staged through the result image 9.7 ms.
        // (two named register sets in turn: with `r = q` at the end of a step the compiler's moves wait for the load
        // just issued -- vmcnt(0) right behind it in the ISA -- and nothing is ahead at all)
        const float* ap = A + row * lda + 8 * half;
        auto fetch = [&](int ks, float4& x0, float4& x1) __attribute__((always_inline)) {
          const int kc = ks < KS ? ks : KS - 1;
          x0 = *reinterpret_cast<const float4*>(ap + 16 * kc);
          x1 = *reinterpret_cast<const float4*>(ap + 16 * kc + 4);
        };
        auto products = [&](int ks, const float4& x0, const float4& x1) __attribute__((always_inline)) {
          u32x4 bh, bm, bl;
          uint32_t h, m, l;
          split3_pair(x0.x, x0.y, h, m, l); bh[0] = h; bm[0] = m; bl[0] = l;
          split3_pair(x0.z, x0.w, h, m, l); bh[1] = h; bm[1] = m; bl[1] = l;
          split3_pair(x1.x, x1.y, h, m, l); bh[2] = h; bm[2] = m; bl[2] = l;
          split3_pair(x1.z, x1.w, h, m, l); bh[3] = h; bm[3] = m; bl[3] = l;
#pragma unroll
          for (int t = 0; t < TG; ++t) {
            const u32x4 wh = wq[((0 * KS + ks) * C + n0 + 32 * t) * 2];
            const u32x4 wm = wq[((1 * KS + ks) * C + n0 + 32 * t) * 2];
            const u32x4 wo = wq[((2 * KS + ks) * C + n0 + 32 * t) * 2];
            acc[t] = mfma_bf16(wo, bh, acc[t]);
            acc[t] = mfma_bf16(wh, bl, acc[t]);
            acc[t] = mfma_bf16(wm, bm, acc[t]);
            acc[t] = mfma_bf16(wm, bh, acc[t]);
            acc[t] = mfma_bf16(wh, bm, acc[t]);
            acc[t] = mfma_bf16(wh, bh, acc[t]);
          }
        };
        static_assert(KS % 2 == 0, "the row stream runs in pairs of k steps");
        float4 r0, r1, q0, q1;
        fetch(0, r0, r1);
#pragma unroll 1
        for (int ks = 0; ks < KS; ks += 2) {
          // (the barriers keep each request IN FRONT of the other set's products: the scheduler otherwise sinks it to
          // just before its own use)
          fetch(ks + 1, q0, q1);
          __builtin_amdgcn_sched_barrier(0);
          products(ks, r0, r1);
          fetch(ks + 2, r0, r1);          // (the last pair re-reads the row's last piece)
          __builtin_amdgcn_sched_barrier(0);
          products(ks + 1, q0, q1);
        }
      } else if constexpr (B6) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4 wh = wq[((0 * KS + ks) * C + n0) * 2];
          const u32x4 wm = wq[((1 * KS + ks) * C + n0) * 2];
          const u32x4 wo = wq[((2 * KS + ks) * C + n0) * 2];
          // small terms first
          acc[0] = mfma_bf16(wo, ah[ks], acc[0]);
          acc[0] = mfma_bf16(wh, al[ks], acc[0]);
          acc[0] = mfma_bf16(wm, am[ks], acc[0]);
          acc[0] = mfma_bf16(wm, ah[ks], acc[0]);
          acc[0] = mfma_bf16(wh, am[ks], acc[0]);
          acc[0] = mfma_bf16(wh, ah[ks], acc[0]);
        }
      } else {
#pragma unroll
        for (int s = 0; s < KH; ++s) {
#pragma unroll
          for (int t = 0; t < TG; ++t) {
            const float b = wl[s * C + n0 + 32 * t];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a[s], acc[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        // request the next tile's addends (this group or the first tile of the next group) before finishing this one
        const int ncol = (t + 1 < TG) ? n0 + 32 * (t + 1) : n0 + 32 * TG;
        if (ncol < C) fetch_addend<MODE>(nxt, rm, ep.yn, has_res, ncol);
        const int col0 = n0 + 32 * t;
        gemm_wave_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v;
          v.x = acc[t][4 * q + 0]; v.y = acc[t][4 * q + 1]; v.z = acc[t][4 * q + 2]; v.w = acc[t][4 * q + 3];
          *reinterpret_cast<float4*>(img + j * OS + 8 * q + 4 * half) = v;
        }
        gemm_wave_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 v = *reinterpret_cast<const float4*>(img + (8 * i + er) * OS + ec);
          const int kd = rm[i].kd;
          if (MODE == 1) {
            const float mb = kd > 0 ? 1.f : 0.f;
            const float4 bb = *reinterpret_cast<const float4*>(b_lds + (kd == 2 ? C : 0) + col0 + ec);
            const float m1 = kd > 0 ? 1.f : 0.f, m2 = kd == 2 ? 1.f : 0.f;
            v.x = fmaf(mb, bb.x, fmaf(m2, cur.w[i].x, fmaf(m1, cur.u[i].x, v.x)));
            v.y = fmaf(mb, bb.y, fmaf(m2, cur.w[i].y, fmaf(m1, cur.u[i].y, v.y)));
            v.z = fmaf(mb, bb.z, fmaf(m2, cur.w[i].z, fmaf(m1, cur.u[i].z, v.z)));
            v.w = fmaf(mb, bb.w, fmaf(m2, cur.w[i].w, fmaf(m1, cur.u[i].w, v.w)));
          } else {
            if (ep.bias) {
              const float4 bb = *reinterpret_cast<const float4*>(b_lds + col0 + ec);
              v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (has_res) { v.x += cur.u[i].x; v.y += cur.u[i].y; v.z += cur.u[i].z; v.w += cur.u[i].w; }
          }
          if (OB && (FULL || (MODE == 1 ? kd >= 0 : kd != 0))) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f2v __attribute__((ext_vector_type(2)));
            const bf16x2 lo2 = __builtin_convertvector((f2v){v.x, v.y}, bf16x2), hi2 = __builtin_convertvector((f2v){v.z, v.w}, bf16x2);
            uint2 u;
            u.x = __builtin_bit_cast(uint32_t, lo2);
            u.y = __builtin_bit_cast(uint32_t, hi2);
            *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(Y) + (row0 + 8 * i + er) * ldy + col0 + ec) = u;
          } else if (!OB && (FULL || (MODE == 1 ? kd >= 0 : kd != 0))) {
            float4* yp = reinterpret_cast<float4*>(Y + (row0 + 8 * i + er) * ldy + col0 + ec);
#ifdef HOTK_GEMM_NT   /* nontemporal result stores: measured within noise at 10 M rows (entry GEMM 5.07 vs 4.88 ms, the others +-1 %) */
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(yp));
#else
            *yp = v;
#endif
          }
        }
        cur = nxt;
      }
    }
    };
    // (the wide-input kernels have one store group per row tile and nothing to gain from a second copy)
    if (!STREAM && row0 + 32 <= R) row_tile(std::true_type{});
    else row_tile(std::false_type{});
  }
}

template <int KH, int TG, int MODE, int NW, bool B6, bool RES, bool OB, bool IB = false>
int launch_v(const float* A, int64_t lda, const float* W, float* Y, int64_t ldy, int64_t R, int C, const Epi& ep,
             hipStream_t st) {
  constexpr int WAVES = NW;
  const size_t lds = ((size_t)(B6 ? 3 : 2) * KH * C + (size_t)WAVES * 32 * OS + 2 * (size_t)C) * sizeof(float);
  static bool attr_set = false;   // raising the dynamic-LDS limit of a kernel is a per-process property of the function
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_thin_gemm<KH, TG, MODE, NW, B6, RES, OB, IB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return HOTK_ELAUNCH;
    attr_set = true;
  }
  const int64_t n_tiles = (R + 32 * WAVES - 1) / (32 * WAVES);
  // persistent workgroups: two per CU where the weight and the result images allow it
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const unsigned grid = (unsigned)(n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu);
  hipLaunchKernelGGL((k_thin_gemm<KH, TG, MODE, NW, B6, RES, OB, IB>), dim3(grid), dim3(64 * WAVES), lds, st, A, lda, W, Y, ldy, R, C, ep);
  return hotk_launch_status();
}

// the variants in use: a residual only with MODE 0, bf16 results from the entry GEMM (MODE 1) and the K = 32 products
template <int KH, int TG, int MODE, int NW = 8, bool B6 = false>
int launch(const float* A, int64_t lda, const float* W, float* Y, int64_t ldy, int64_t R, int C, const Epi& ep,
           hipStream_t st) {
  if constexpr (MODE == 0) {
    if (ep.out_bf16) {   // bf16 results without a residual: the narrow-to-wide products of the bf16 storage mode (fc_v, d att)
      if (ep.res || KH != 16) return HOTK_EUNSUPPORTED;
      if constexpr (KH == 16) return launch_v<KH, TG, MODE, NW, B6, false, true>(A, lda, W, Y, ldy, R, C, ep, st);
    }
    return ep.res ? launch_v<KH, TG, MODE, NW, B6, true, false>(A, lda, W, Y, ldy, R, C, ep, st)
                  : launch_v<KH, TG, MODE, NW, B6, false, false>(A, lda, W, Y, ldy, R, C, ep, st);
  } else {
    return ep.out_bf16 ? launch_v<KH, TG, MODE, NW, B6, false, true>(A, lda, W, Y, ldy, R, C, ep, st)
                       : launch_v<KH, TG, MODE, NW, B6, false, false>(A, lda, W, Y, ldy, R, C, ep, st);
  }
}

bool gemm_f32() {
  static const int v = [] { const char* e = getenv("HOTK_GEMM_F32"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}

bool b6_narrow() {   // HOTK_GEMM_B6_NARROW=1: the K = 32 products in the split-bf16 form as well (A/B runs)
  static const int v = [] { const char* e = getenv("HOTK_GEMM_B6_NARROW"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}

template <int MODE>
int dispatch(const float* A, int64_t lda, const float* W, int64_t K, int64_t C, float* Y, int64_t ldy, int64_t R,
             const Epi& ep, hipStream_t st) {
  // gathered addends (MODE 1) are requested one 32-column tile ahead: with one tile per MFMA loop their latency hides
  // behind the whole loop
  if (K == 64 && C % 96 == 0) {
    if (MODE == 1) {
      // the split-bf16 form when its three weight planes fit in LDS beside the result images (HOTK_GEMM_F32=1: A/B runs)
      const bool b6 = !gemm_f32() && ((size_t)3 * 32 * C + (size_t)HOTK_ENTRY_WAVES * 32 * OS + 2 * (size_t)C) * sizeof(float) <= 160 * 1024 && C % 2 == 0;
      return b6 ? launch<32, 1, MODE, HOTK_ENTRY_WAVES, true>(A, lda, W, Y, ldy, R, (int)C, ep, st)
                : launch<32, 1, MODE, HOTK_ENTRY_WAVES>(A, lda, W, Y, ldy, R, (int)C, ep, st);
    }
    const bool b6m0 = !gemm_f32() && ((size_t)3 * 32 * C + (size_t)8 * 32 * OS + 2 * (size_t)C) * sizeof(float) <= 160 * 1024 && C % 2 == 0;
    return b6m0 ? launch<32, 1, MODE, 8, true>(A, lda, W, Y, ldy, R, (int)C, ep, st)
                : launch<32, 3, MODE>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  }
  if (K == 32 && C % 160 == 0) {
    // bf16 results (half the bytes to store): the fp32 matrix time shows, the split-bf16 form takes it out
    // (10 M rows x 320: 2.49 -> 2.04 ms); fp32 results: within noise either way, the fp32 MFMAs stay
    if (MODE == 0 && (b6_narrow() || ep.out_bf16)) return launch<16, 1, MODE, 8, true>(A, lda, W, Y, ldy, R, (int)C, ep, st);
    return launch<16, 5, MODE>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  }
  // (the node-level products of the light basis at the headline widths: 32 -> 384, and 32 -> 864 as 480 + 384 columns)
  if (K == 32 && C % 96 == 0 && MODE == 0 && !ep.out_bf16) return launch<16, 3, MODE>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  // wide input, narrow output: split-bf16 form while the three weight planes fit in LDS beside the result images
  const bool b6_fits = !gemm_f32() && ((size_t)3 * (K / 2) * C + (size_t)8 * 32 * OS + 2 * (size_t)C) * sizeof(float) <= 160 * 1024 && C % 2 == 0;
  if (K == 288 && C % 64 == 0) return b6_fits ? launch<144, 2, MODE, 8, true>(A, lda, W, Y, ldy, R, (int)C, ep, st)
                                              : launch<144, 2, MODE>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  const bool b6_fits16 = !gemm_f32() && ((size_t)3 * (K / 2) * C + (size_t)16 * 32 * OS + 2 * (size_t)C) * sizeof(float) <= 160 * 1024 && C % 2 == 0;
  if (K == 320 && C % 32 == 0 && b6_fits16 && HOTK_WIDE_WAVES == 16) return launch<160, 1, MODE, 16, true>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  if (K == 320 && C % 32 == 0) return b6_fits ? launch<160, 1, MODE, 8, true>(A, lda, W, Y, ldy, R, (int)C, ep, st)
                                              : launch<160, 1, MODE>(A, lda, W, Y, ldy, R, (int)C, ep, st);
  return HOTK_EUNSUPPORTED;
}

bool shape_ok(int64_t K, int64_t C) {
  const bool known = (K == 64 && C % 96 == 0) || (K == 32 && (C % 160 == 0 || C % 96 == 0)) || (K == 288 && C % 64 == 0) ||
                     (K == 320 && C % 32 == 0);
  return known && (K * C + MAX_WAVES * 32 * OS + 2 * C) * (int64_t)sizeof(float) <= 150 * 1024;
}

}  // namespace

extern "C" int hotk_thin_gemm_supported(int64_t K, int64_t C) { return shape_ok(K, C) ? 1 : 0; }

extern "C" int hotk_thin_gemm_bias(const float* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                                   const float* res, int64_t ld_res, float* Y, int64_t ldy, int64_t R, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && K > 0 && C > 0 && lda >= K && ldy >= C && (!res || ld_res >= C));
  if (!shape_ok(K, C)) return HOTK_EUNSUPPORTED;
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(A && W && Y && hotk_aligned16(A) && hotk_aligned16(W) && (lda & 3) == 0);
  HOTK_CHECK_ARG(hotk_aligned16(Y) && (ldy & 3) == 0 && (!bias || hotk_aligned16(bias)) &&
                 (!res || (hotk_aligned16(res) && (ld_res & 3) == 0)));
  Epi ep{};
  ep.bias = bias; ep.res = res; ep.ld_res = ld_res;
  return dispatch<0>(A, lda, W, K, C, Y, ldy, R, ep, (hipStream_t)stream);
}

// the same product with the result rounded to bf16 at the store (K = 32 only: fc_v and d att of the bf16 storage mode)
extern "C" int hotk_thin_gemm_bias_bf16(const float* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                                        void* Y, int64_t ldy, int64_t R, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && K > 0 && C > 0 && lda >= K && ldy >= C);
  if (K != 32 || !shape_ok(K, C)) return HOTK_EUNSUPPORTED;
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(A && W && Y && hotk_aligned16(A) && hotk_aligned16(W) && (lda & 3) == 0);
  HOTK_CHECK_ARG((((uintptr_t)Y) & 7u) == 0 && (ldy & 3) == 0 && (!bias || hotk_aligned16(bias)));
  Epi ep{};
  ep.bias = bias;
  ep.out_bf16 = 1;
  return dispatch<0>(A, lda, W, K, C, (float*)Y, ldy, R, ep, (hipStream_t)stream);
}

// Y[R, C] (fp32) = A[R, K] (bf16 rows) . W[K, C] (fp32) (+ bias) (+ res, fp32): the wide-input products of the bf16 storage mode
// (fc_o: K = 320 -> 32; the entry blocks' data gradient: K = 288 -> 64); every product of a bf16 value with a three-way split
// weight is exact, fp32 accumulation
extern "C" int hotk_thin_gemm_bias_bf16in(const void* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                                          const float* res, int64_t ld_res, float* Y, int64_t ldy, int64_t R, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && K > 0 && C > 0 && lda >= K && ldy >= C && (!res || ld_res >= C));
  const bool fits = ((size_t)3 * (K / 2) * C + (size_t)8 * 32 * OS + 2 * (size_t)C) * sizeof(float) <= 160 * 1024 && C % 2 == 0;
  if (!((K == 288 && C % 64 == 0) || (K == 320 && C % 32 == 0)) || !fits) return HOTK_EUNSUPPORTED;
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(A && W && Y && hotk_aligned16(A) && hotk_aligned16(W) && (lda & 7) == 0);
  HOTK_CHECK_ARG(hotk_aligned16(Y) && (ldy & 3) == 0 && (!bias || hotk_aligned16(bias)) &&
                 (!res || (hotk_aligned16(res) && (ld_res & 3) == 0)));
  Epi ep{};
  ep.bias = bias; ep.res = res; ep.ld_res = ld_res;
  const float* a = reinterpret_cast<const float*>(A);
  hipStream_t st = (hipStream_t)stream;
  if (K == 288)
    return res ? launch_v<144, 2, 0, 8, true, true, false, true>(a, lda, W, Y, ldy, R, (int)C, ep, st)
               : launch_v<144, 2, 0, 8, true, false, false, true>(a, lda, W, Y, ldy, R, (int)C, ep, st);
  return res ? launch_v<160, 1, 0, 8, true, true, false, true>(a, lda, W, Y, ldy, R, (int)C, ep, st)
             : launch_v<160, 1, 0, 8, true, false, false, true>(a, lda, W, Y, ldy, R, (int)C, ep, st);
}

namespace {
int entry_fwd(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C, const float* yn, int64_t ld_yn,
              const uint8_t* kind, const int32_t* rown, const int32_t* coln, const float* b_off, const float* b_diag,
              float* Y, int64_t ldy, int64_t R, int64_t Nn, int out_bf16, void* stream);
}

extern "C" int hotk_linear22_entry_fwd(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C, const float* yn,
                                       int64_t ld_yn, const uint8_t* kind, const int32_t* rown, const int32_t* coln,
                                       const float* b_off, const float* b_diag, float* Y, int64_t ldy, int64_t R,
                                       int64_t Nn, void* stream) {
  return entry_fwd(x2, ld_x2, W, K, C, yn, ld_yn, kind, rown, coln, b_off, b_diag, Y, ldy, R, Nn, 0, stream);
}

extern "C" int hotk_linear22_entry_fwd_bf16(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C,
                                            const float* yn, int64_t ld_yn, const uint8_t* kind, const int32_t* rown,
                                            const int32_t* coln, const float* b_off, const float* b_diag, void* Y,
                                            int64_t ldy, int64_t R, int64_t Nn, void* stream) {
  return entry_fwd(x2, ld_x2, W, K, C, yn, ld_yn, kind, rown, coln, b_off, b_diag, (float*)Y, ldy, R, Nn, 1, stream);
}

namespace {
int entry_fwd(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C, const float* yn, int64_t ld_yn,
              const uint8_t* kind, const int32_t* rown, const int32_t* coln, const float* b_off, const float* b_diag,
              float* Y, int64_t ldy, int64_t R, int64_t Nn, int out_bf16, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && K > 0 && C > 0 && ld_x2 >= K && ldy >= C && ld_yn >= 3 * C);
  if (!shape_ok(K, C)) return HOTK_EUNSUPPORTED;
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x2 && W && Y && yn && kind && rown && coln && hotk_aligned16(x2) && hotk_aligned16(W) && (ld_x2 & 3) == 0);
  HOTK_CHECK_ARG(hotk_aligned16(Y) && (ldy & 3) == 0 && hotk_aligned16(yn) && (ld_yn & 3) == 0 && (C & 3) == 0 &&
                 (!b_off || hotk_aligned16(b_off)) && (!b_diag || hotk_aligned16(b_diag)));
  // node-table rows (Nn of them) are addressed with 32-bit element offsets
  if (Nn < 0 || Nn * ld_yn >= ((int64_t)1 << 32)) return HOTK_EUNSUPPORTED;
  Epi ep{};
  ep.yn = yn; ep.ld_yn = ld_yn; ep.kind = kind; ep.rown = rown; ep.coln = coln; ep.b_off = b_off; ep.b_diag = b_diag;
  ep.out_bf16 = out_bf16;
  return dispatch<1>(x2, ld_x2, W, K, C, Y, ldy, R, ep, (hipStream_t)stream);
}
}  // namespace
