// fp32-exact tiled GEMMs on the bf16 matrix pipe, for the dense projections that are too wide for the LDS-resident
// thin GEMM (csrc/gemm.hip): C2's hidden width 256 / 512 and the dqk = 64 head shape, where the projections ARE
// GEMM-shaped and MFMA-bound (5.2 MFLOP per entry at C2) and the library's fp32 kernels run at the fp32 matrix
// rate (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU).  An fp32 number is exactly the sum of three bf16 numbers (hi / mid /
// lo, see csrc/gemm.hip), so a product is nine bf16 products, of which the six of relative weight >= 2^-16 are summed in
// the fp32 accumulator of v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k against 8 x 64 for the fp32 MFMA, 1.8e-7
// relative against fp64 (the library's fp32 GEMM: 2.1e-7).
//
//   hotk_gemm6_nn :  C[M, N]      = A[M, K] . B (+ bias)       rows M ~ 1e5..1e7; B = weight, pre-split once per call into
//                                                              bf16 planes [3][N][K] by hotk_gemm6_split_weight
//   hotk_gemm6_tn :  P[S][K][N]   = A[M_s, K]^T . G[M_s, N]    weight gradients: the row dimension is cut into S slabs, one
//                                                              partial product per slab, summed by the caller in order
// Reference: the aten addmm / mm calls under nn.Linear and _Weight.get_output (models/sparse/linear.py:46-48,
// kernelattn.py:65-72) and their autograd.
//
// Tile: 256 x 256 per workgroup of 8 waves (2 x 4, a wave owns 128 x 64 = eight 32 x 32 MFMA tiles, 128 accumulator
// registers), k step 32.  Operands of a step sit in LDS as three bf16 planes [plane][row or column][32 k], 80-byte rows
// (16 lanes x 16 B hit 64 distinct banks), a lane's MFMA fragment = one ds_read_b128.  The next step's global loads are
// issued into registers before the current step's products and written to LDS after them (one LDS buffer of 120 KB, one
// workgroup per CU).  Why this tile: with 128 x 128 tiles (two workgroups per CU) the kernel was bound by L2 -> CU traffic,
// not by the matrix pipe -- every workgroup pulls 16 KB of A and 24 KB of B planes per k step, 5.1 GB per [254 k, 512] x
// [512, 512] product, and a build with the products compiled out still took 0.94 of its 1.14 ms; a 256 x 256 tile moves half
// the bytes per product.
#include "common.h"
#include <cstdlib>

// (kernels of the two builds of this file -- see csrc/gemm6_n128.hip -- must not share their symbol names: template
// instantiations are merged across translation units even inside unnamed namespaces)
#ifdef HOTK_G6_VARIANT
#define HOTK_G6_NS g6_tail
#else
#define HOTK_G6_NS g6_wide
#endif
namespace {
namespace HOTK_G6_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t g6_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float g6_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// the three bf16 planes of two values, packed (low half = x0)
__device__ __forceinline__ void g6_split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t h0 = g6_f2u(x0) & 0xffff0000u, h1 = g6_f2u(x1) & 0xffff0000u;
  const float r0 = x0 - g6_u2f(h0), r1 = x1 - g6_u2f(h1);
  const uint32_t m0 = g6_f2u(r0) & 0xffff0000u, m1 = g6_f2u(r1) & 0xffff0000u;
  const float s0 = r0 - g6_u2f(m0), s1 = r1 - g6_u2f(m1);
  hi = (h0 >> 16) | h1;
  mid = (m0 >> 16) | m1;
  lo = (g6_f2u(s0) >> 16) | (g6_f2u(s1) & 0xffff0000u);
}
__device__ __forceinline__ f32x16 g6_mfma(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
struct Frag { u32x4 h, m, l; };
// acc += A . B with both operands as three planes; the three products below 2^-24 of the result are dropped, the small
// terms are added first
__device__ __forceinline__ f32x16 g6_mfma6(const Frag& a, const Frag& b, f32x16 c) {
  c = g6_mfma(a.l, b.h, c);
  c = g6_mfma(a.h, b.l, c);
  c = g6_mfma(a.m, b.m, c);
  c = g6_mfma(a.m, b.h, c);
  c = g6_mfma(a.h, b.m, c);
  c = g6_mfma(a.h, b.h, c);
  return c;
}

// ---- loads the compiler does not count ------------------------------------------------------------------------------
// With more than one k step in flight in registers (HOTK_G6_DEPTH > 1), hipcc's wait-count pass cannot follow the register
// ring across the loop's back edge and puts s_waitcnt vmcnt(0) at the loop header.  So the ring's loads are issued by
// inline asm -- which the compiler treats as complete -- and waited for by hand: vmcnt counts in order, the oldest set is
// complete once at most the loads of the younger sets are outstanding.  g6_arrived() makes the registers "defined" only
// after that wait, so no use can move above it; g6_drain() runs after the loop, before any register of the ring is reused.
__device__ __forceinline__ void g6_ld16(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void g6_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
__device__ __forceinline__ void g6_arrived(u32x4& v) { asm volatile("" : "+v"(v) : : "memory"); }
// After the loop: loads of steps past the end are still landing.  The wait alone does not keep the compiler from giving
// the ring's registers to something else BEFORE it (register arithmetic may move above a volatile asm): follow it with
// g6_arrived() on every register of the ring, which keeps them allocated up to this point.  scratch/check_async_regs.py
// scans the assembly for violations of this discipline.
__device__ __forceinline__ void g6_drain() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }

// timing probes (A/B builds only; results are wrong with either): NOLOAD re-reads the first k step (cache hits: no memory
// time), NOPROD skips the products of all but the first k step (data movement, splits and barriers only)
#ifdef HOTK_G6_NOLOAD
#define HOTK_G6_PROBE_KSEL(k) 0
#else
#define HOTK_G6_PROBE_KSEL(k) (k)
#endif
#ifdef HOTK_G6_NOPROD
#define HOTK_G6_PROBE_PROD(c, first) ((c) && (first))
#else
#define HOTK_G6_PROBE_PROD(c, first) (c)
#endif

#ifndef HOTK_G6_BM
#define HOTK_G6_BM 256
#endif
#ifndef HOTK_G6_BN
#define HOTK_G6_BN 256
#endif
#ifndef HOTK_G6_WM
#define HOTK_G6_WM 2
#endif
#ifndef HOTK_G6_WN
#define HOTK_G6_WN 4
#endif
#ifndef HOTK_G6_DEPTH
#define HOTK_G6_DEPTH 1
#endif
#ifndef HOTK_G6_WGS
#define HOTK_G6_WGS 1      /* workgroups per CU the register budget is set for */
#endif
constexpr int BM = HOTK_G6_BM, BN = HOTK_G6_BN, BK = 32, DEPTH = HOTK_G6_DEPTH;
constexpr int WM = HOTK_G6_WM, WN = HOTK_G6_WN, TM = BM / (32 * WM), TN = BN / (32 * WN);
static_assert(WM * TM * 32 == BM && WN * TN * 32 == BN, "the waves tile the block");
constexpr int G6_THREADS = 64 * WM * WN;
constexpr int ROWB = 80;                       // bytes per LDS row of a plane: 32 bf16 + 16 B padding
constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
constexpr int OPER_A = 3 * PLANE_A, OPER_B = 3 * PLANE_B;
constexpr int AP = BM * 4 / G6_THREADS, BP = BN * 4 / G6_THREADS;   // 8-element pieces (row, quarter of the 32 k) per thread
static_assert(AP * G6_THREADS == BM * 4 && BP * G6_THREADS == BN * 4, "pieces divide over the threads");

// fragments of one k16 step s for NB 32-row (or 32-column) blocks of the wave starting at `base`
template <int NB>
__device__ __forceinline__ void g6_frags(const unsigned char* oper, int plane, int base, int s, int lane, Frag (&f)[NB]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const unsigned char* p = oper + (base + 32 * i + r) * ROWB + (16 * s + 8 * h) * 2;
    f[i].h = *reinterpret_cast<const u32x4*>(p);
    f[i].m = *reinterpret_cast<const u32x4*>(p + plane);
    f[i].l = *reinterpret_cast<const u32x4*>(p + 2 * plane);
  }
}

// the products of a k step: each of the six plane pairs over the wave's tiles, so that independent accumulators separate
// two MFMAs of one chain (a dependent 32x32x16 MFMA issues only after the previous one's 8 passes)
__device__ __forceinline__ void g6_products(const unsigned char* lds_a, const unsigned char* lds_b, int wm, int wn, int lane,
                                            f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Frag fa[TM], fb[TN];
    g6_frags<TM>(lds_a, PLANE_A, 32 * TM * wm, s, lane, fa);
    g6_frags<TN>(lds_b, PLANE_B, 32 * TN * wn, s, lane, fb);
#define HOTK_G6_STEP(PA, PB)                                                     \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                 \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = g6_mfma(fa[i].PA, fb[j].PB, acc[i][j]);
    HOTK_G6_STEP(l, h)
    HOTK_G6_STEP(h, l)
    HOTK_G6_STEP(m, m)
    HOTK_G6_STEP(m, h)
    HOTK_G6_STEP(h, m)
    HOTK_G6_STEP(h, h)
#undef HOTK_G6_STEP
  }
}

// Result tile of a wave (TM x TN accumulators; C/D map: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
// -> global rows.  FULL tiles go through a wave-private LDS image, 32 rows at a time, and leave as 16-byte pieces of
// 256-byte row runs in straight-line code; tiles that cross the matrix edge store element by element under their guards
// (one row tile and one column tile of a product at most).
constexpr int G6_ST = 32 * TN + 4;   // floats per staged row
// Row map (RM): the tile's rows are entries row0 .. of the ascending list `rows` [Mc] of the rows of C [Mall, .] that carry
// data (the others are padding).  The lane that stores listed row i also writes zeros to the padding rows between it and
// the next listed row (the first one: and before it; the last one: up to Mall), so that C leaves fully written.
// Row MODE of the NN kernel: 0 = all rows, 1 = row list, 2 = row RECORDS with the assembly of Linear(2, 2, 'light') as the
// epilogue (hotk_gemm6_entry22): rec[i] = {row, next listed row (or Mall), row node, column node or -1 on the diagonal},
// y = z + yn[row node][C..2C) + yn[column node][2C..3C) + b_off off the diagonal, yn[node][0..C) + b_diag on it.
struct RowMap {
  const int32_t* rows; int64_t Mc, Mall;
  const int4* rec; const float* yn; int64_t ld_yn; const float* b_off; const float* b_diag; int C;
};
__device__ __forceinline__ int64_t g6_next_row(const RowMap& rm, int64_t i) { return i + 1 < rm.Mc ? rm.rows[i + 1] : rm.Mall; }
// (nxt = g6_next_row(rm, i), fetched by the caller together with its other list reads: one round trip, not one per row)
template <class V>
__device__ __forceinline__ void g6_zero_gap(float* __restrict__ C, int64_t ldc, int col, int64_t i, int64_t r, int64_t nxt) {
  V z;
  __builtin_memset(&z, 0, sizeof(V));
  for (int64_t g = r + 1; g < nxt; ++g) *reinterpret_cast<V*>(C + g * ldc + col) = z;
  if (i == 0)
    for (int64_t g = 0; g < r; ++g) *reinterpret_cast<V*>(C + g * ldc + col) = z;
}
template <int MODE = 0>
__device__ __forceinline__ void g6_store_tile(float* __restrict__ C, int64_t ldc, int64_t row0, int64_t n_rows, int col0,
                                              int n_cols, const float* __restrict__ bias, unsigned char* lds, int wave, int lane,
                                              const f32x16 (&acc)[TM][TN], bool full, const RowMap rm = RowMap{}) {
  constexpr bool RM = MODE != 0;
  const int cl = lane & 31, rh = 4 * (lane >> 5);
  // `full` here: 16-byte pieces are possible (alignment of C, ldc, bias: the caller's check).  Rows past n_rows and
  // 4-column pieces past n_cols are skipped lane by lane, so edge tiles take this path too when n_cols % 4 == 0.
  if (full && (n_cols >= 32 * TN || (n_cols & 3) == 0) && n_rows > 0 && n_cols > 0) {
    static_assert(TN == 2, "16 lanes x 16 bytes span the staged row");
    float* st = reinterpret_cast<float*>(lds) + wave * (32 * G6_ST);
    const int row = lane >> 4, c4 = lane & 15;
    const bool col_in = 4 * c4 < n_cols;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), bd4 = b4;
    const int cc = col0 + 4 * c4;                     // this lane's four columns
    if (MODE == 2) {
      if (rm.b_off && col_in) b4 = *reinterpret_cast<const float4*>(rm.b_off + cc);
      if (rm.b_diag && col_in) bd4 = *reinterpret_cast<const float4*>(rm.b_diag + cc);
    } else if (bias && col_in) {
      b4 = *reinterpret_cast<const float4*>(bias + cc);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int64_t dst[8], nxt[8];
      float4 ga[8], gb[8];                            // MODE 2: the two gathered node rows of each stored row
      bool diag[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t ti = 32 * i + 4 * q + row;      // row of the wave's tile
        dst[q] = row0 + (ti < n_rows ? ti : n_rows - 1);
        if (MODE == 1) {
          nxt[q] = g6_next_row(rm, dst[q]);
          dst[q] = rm.rows[dst[q]];
        }
      }
      if (MODE == 2) {
        int4 rc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rc[q] = rm.rec[dst[q]];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          dst[q] = rc[q].x;
          nxt[q] = rc[q].y;
          diag[q] = rc[q].w < 0;
          const float* pa = rm.yn + (int64_t)rc[q].z * rm.ld_yn + (diag[q] ? 0 : rm.C) + cc;
          const float* pb = rm.yn + (int64_t)(diag[q] ? rc[q].z : rc[q].w) * rm.ld_yn + 2 * rm.C + cc;
          ga[q] = col_in ? *reinterpret_cast<const float4*>(pa) : make_float4(0.f, 0.f, 0.f, 0.f);
          gb[q] = (col_in && !diag[q]) ? *reinterpret_cast<const float4*>(pb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + rh) * G6_ST + 32 * j + cl] = acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!col_in || 32 * i + 4 * q + row >= n_rows) continue;
        float4 v = *reinterpret_cast<const float4*>(st + (4 * q + row) * G6_ST + 4 * c4);
        if (MODE == 2) {
          // (the order of hotk_assemble22's additions: the two forms agree bit for bit)
          if (diag[q]) {
            v = make_float4(ga[q].x + bd4.x, ga[q].y + bd4.y, ga[q].z + bd4.z, ga[q].w + bd4.w);
          } else {
            v.x = ((v.x + ga[q].x) + gb[q].x) + b4.x; v.y = ((v.y + ga[q].y) + gb[q].y) + b4.y;
            v.z = ((v.z + ga[q].z) + gb[q].z) + b4.z; v.w = ((v.w + ga[q].w) + gb[q].w) + b4.w;
          }
        } else {
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        *reinterpret_cast<float4*>(C + dst[q] * ldc + cc) = v;
        if (RM) g6_zero_gap<float4>(C, ldc, cc, row0 + 32 * i + 4 * q + row, dst[q], nxt[q]);
      }
    }
    return;
  }
  if (RM) return;      // (row lists: the entry point insists on the alignment of the staged path; nothing left to store here)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = 32 * j + cl;
    if (n >= n_cols) continue;
    const float b = bias ? bias[col0 + n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t rr = 32 * i + rh + (r & 3) + 8 * (r >> 2);
        if (rr < n_rows) C[(row0 + rr) * ldc + col0 + n] = acc[i][j][r] + b;
      }
  }
}

// ---- weight -> bf16 planes [3][K / 32][N][32 k]: plane[p][kb][n][kk] = split_p(B[32 kb + kk][n]), B[k][n] = trans ? W[k][n]
// : W[n][k].  A k step's slice for consecutive columns is one contiguous run of whole cache lines (with k contiguous over
// all of K a step read 64 bytes out of every 1 KB: half of each line fetched, twice) ------------------------------------
__global__ void k_gemm6_split_weight(const float* __restrict__ W, int64_t ldw, int K, int N, int trans,
                                     uint16_t* __restrict__ planes, int kb_log2) {
  const int64_t total = (int64_t)N * (K / 2);
  uint32_t* p32 = reinterpret_cast<uint32_t*>(planes);
  const int64_t P = (int64_t)N * K / 2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    // consecutive threads walk the contiguous dimension of W: n when W is [K, N], k when it is [N, K]
    int n, k;
    if (trans) {
      k = 2 * (int)(t / N);
      n = (int)(t - (int64_t)(k / 2) * N);
    } else {
      n = (int)(t / (K / 2));
      k = 2 * (int)(t - (int64_t)n * (K / 2));
    }
    const float x0 = trans ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
    const float x1 = trans ? W[(int64_t)(k + 1) * ldw + n] : W[(int64_t)n * ldw + k + 1];
    uint32_t hi, mid, lo;
    g6_split_pair(x0, x1, hi, mid, lo);
    const int64_t o = ((((int64_t)(k >> kb_log2) * N + n) << kb_log2) + (k & ((1 << kb_log2) - 1))) / 2;   // [K / kb][N][kb]
    p32[o] = hi;
    p32[P + o] = mid;
    p32[2 * P + o] = lo;
  }
}

// tile of workgroup `orig` of `nwg`: the blocks that share an XCD (orig % 8) take consecutive tiles
__device__ __forceinline__ int64_t g6_xcd_tile(unsigned orig, unsigned nwg) {
#ifdef HOTK_G6_NOXCD
  return orig;
#else
  const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  return (int64_t)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
#endif
}

// ---- C = A . B (+ bias) ----------------------------------------------------------------------------------------------
// Piece pi of a tile = (row or column pi & 7 + 8 (pi >> 5), quarter (pi >> 3) & 3 of the 32 k): 8 floats of A (two 16-byte
// loads, one 16-byte LDS write per plane after the split) or 8 bf16 per plane of B (one 16-byte load and write per plane).
// Eight consecutive lanes write the same quarter of eight consecutive rows: 80-byte rows put them on distinct banks.
struct NNRaw { u32x4 a[AP][2]; u32x4 b[BP][3]; };
constexpr int NN_LOADS = 2 * AP + 3 * BP;      // loads per set

__device__ __forceinline__ int g6_piece_row(int pi) { return (pi & 7) + 8 * (pi >> 5); }
__device__ __forceinline__ int g6_piece_quarter(int pi) { return (pi >> 3) & 3; }

__device__ __forceinline__ void g6_load_nn(const float* const (&ap)[AP], const uint16_t* const (&bp)[BP], int64_t PL, int Nld,
                                           int k0, NNRaw& raw) {
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    g6_ld16(raw.a[q][0], ap[q] + k0);
    g6_ld16(raw.a[q][1], ap[q] + k0 + 4);
  }
#pragma unroll
  for (int q = 0; q < BP; ++q)
#pragma unroll
    for (int p = 0; p < 3; ++p) g6_ld16(raw.b[q][p], bp[q] + p * PL + (int64_t)k0 * Nld);
}

// wait until this set (the oldest of DEPTH in flight) has arrived
__device__ __forceinline__ void g6_wait_nn(NNRaw& raw) {
  g6_wait_vm<(DEPTH - 1) * NN_LOADS>();
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    g6_arrived(raw.a[q][0]);
    g6_arrived(raw.a[q][1]);
  }
#pragma unroll
  for (int q = 0; q < BP; ++q)
#pragma unroll
    for (int p = 0; p < 3; ++p) g6_arrived(raw.b[q][p]);
}

__device__ __forceinline__ void g6_store_nn(unsigned char* lds_a, unsigned char* lds_b, int t, const NNRaw& raw) {
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    const int pi = t + q * G6_THREADS;
    const int off = g6_piece_row(pi) * ROWB + g6_piece_quarter(pi) * 16;
    const u32x4 x = raw.a[q][0], y = raw.a[q][1];
    uint32_t h[4], m[4], l[4];
    g6_split_pair(g6_u2f(x[0]), g6_u2f(x[1]), h[0], m[0], l[0]);
    g6_split_pair(g6_u2f(x[2]), g6_u2f(x[3]), h[1], m[1], l[1]);
    g6_split_pair(g6_u2f(y[0]), g6_u2f(y[1]), h[2], m[2], l[2]);
    g6_split_pair(g6_u2f(y[2]), g6_u2f(y[3]), h[3], m[3], l[3]);
    *reinterpret_cast<u32x4*>(lds_a + off) = (u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(lds_a + PLANE_A + off) = (u32x4){m[0], m[1], m[2], m[3]};
    *reinterpret_cast<u32x4*>(lds_a + 2 * PLANE_A + off) = (u32x4){l[0], l[1], l[2], l[3]};
  }
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int pi = t + q * G6_THREADS;
    const int off = g6_piece_row(pi) * ROWB + g6_piece_quarter(pi) * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(lds_b + p * PLANE_B + off) = raw.b[q][p];
  }
}

template <int MODE>
__global__ __launch_bounds__(G6_THREADS, HOTK_G6_WGS) void k_gemm6_nn(const float* __restrict__ A, int64_t lda,
                                                                      const uint16_t* __restrict__ planes,
                                                                      const float* __restrict__ bias, float* __restrict__ C,
                                                                      int64_t ldc, int64_t M, int K, int N, int tiles_n, int col0,
                                                                      int ncap, RowMap rm) {
  // columns [col0, ncap) of the product; N = all columns of B (the stride of its planes).  RM: M = number of LISTED rows,
  // tile row i = row rm.rows[i] of A and of C (padding rows of C: zeros, see g6_zero_gap)
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* lds_a = lds;
  unsigned char* lds_b = lds + OPER_A;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
  // Workgroups b, b + 8, b + 16, .. share an XCD (round-robin dispatch; speed only): consecutive tiles are dealt to the
  // blocks of ONE XCD, so the column tiles of a row tile read its A rows through the same L2 (bijective remap).
  const int64_t tile = g6_xcd_tile(blockIdx.x, gridDim.x);
  const int64_t tm = tile / tiles_n;
  const int tn = (int)(tile - tm * tiles_n);
  const int64_t m0 = tm * BM;
  const int n0 = col0 + tn * BN;
  const int64_t PL = (int64_t)N * K;            // bf16 elements per plane
  const float* ap[AP];
  const uint16_t* bp[BP];
#pragma unroll
  for (int q = 0; q < AP; ++q) {
    const int pi = t + q * G6_THREADS;
    int64_t arow = m0 + g6_piece_row(pi);
    if (arow >= M) arow = M - 1;
    if (MODE == 1) arow = rm.rows[arow];
    if (MODE == 2) arow = rm.rec[arow].x;
    ap[q] = A + arow * lda + 8 * g6_piece_quarter(pi);
  }
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int pi = t + q * G6_THREADS;
    int bn = n0 + g6_piece_row(pi);
    if (bn >= ncap) bn = ncap - 1;
    bp[q] = planes + (int64_t)bn * 32 + 8 * g6_piece_quarter(pi);      // + k0 * N per step: planes are [K / 32][N][32]
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Operands of DEPTH k steps are in flight in registers (a ring of sets).  Every load is unconditional (steps past K
  // re-read the last one and skip their products), so exactly (DEPTH - 1) NN_LOADS younger loads are outstanding at a wait.
  const int klast = K - BK;
  NNRaw ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) g6_load_nn(ap, bp, PL, N, d * BK < K ? d * BK : klast, ring[d]);
  for (int k0 = 0; k0 < K; k0 += DEPTH * BK) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int kk = k0 + d * BK;
      g6_wait_nn(ring[d]);
      g6_store_nn(lds_a, lds_b, t, ring[d]);
      __syncthreads();
      g6_load_nn(ap, bp, PL, N, HOTK_G6_PROBE_KSEL(kk + DEPTH * BK < K ? kk + DEPTH * BK : klast), ring[d]);
      if (HOTK_G6_PROBE_PROD(kk < K, kk == 0)) g6_products(lds_a, lds_b, wm, wn, lane, acc);
      __syncthreads();
    }
  }
  g6_drain();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) g6_wait_nn(ring[d]);      // the ring's registers stay allocated until the drain (see g6_drain)
  const bool full = (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0);
  g6_store_tile<MODE>(C, ldc, m0 + 32 * TM * wm, M - (m0 + 32 * TM * wm), n0 + 32 * TN * wn, ncap - (n0 + 32 * TN * wn), bias, lds,
                    wave, lane, acc, full, rm);
}

#if HOTK_G6_BN == 256
// ---- C = A . B (+ bias), pipelined form -------------------------------------------------------------------------------
// The kernel above runs its data phase (loads -> split -> LDS writes -> barrier, ~3 us per k step) and its product phase
// (1.3 us) one after the other: one workgroup per CU and a barrier per step keep all eight waves in phase.  Here the k step
// is 16 and LDS holds TWO stages of A planes and THREE of B planes (147 KB), so that stage i + 1 is filled while stage i is
// multiplied, with ONE barrier per step:
//   * B planes go global -> LDS by DMA (global_load_lds_dwordx4: no registers, no ds_write), two steps ahead.  A DMA
//     instruction writes 1 KB lane-linearly, so the B stage is unpadded [plane][column][2 x 16 B] and the bank swizzle of the
//     fragment reads -- chunk position = k half ^ bit 3 of the column -- is applied to the per-lane SOURCE address;
//   * A goes through a two-deep register ring (inline-asm loads, hand-counted vmcnt, see above), split and written one
//     step ahead;
//   * the barrier is a raw s_barrier behind s_waitcnt lgkmcnt(0): no vmcnt(0), the loads of the next two steps stay in
//     flight across it.
// Weight planes for this kernel are [3][K / 16][N][16].
constexpr int ROWB16 = 48;                               // A stage: 16 bf16 + 16 B padding per row and plane
constexpr int PLANE16_A = BM * ROWB16, STAGE16_A = 3 * PLANE16_A;
constexpr int PLANE16_B = BN * 32, STAGE16_B = 3 * PLANE16_B;
constexpr int P16_A0 = 0, P16_B0 = 2 * STAGE16_A;
constexpr size_t G6_LDS16 = (size_t)(2 * STAGE16_A + 3 * STAGE16_B);
static_assert(BN * 32 * 3 == (G6_THREADS / 64) * 3 * 1024, "three DMA instructions per wave fill a B stage");
static_assert(BM * 2 == G6_THREADS, "one A piece (row, k half) per thread");

// 16 bytes per lane from gsrc to LDS byte address lds_dst + 16 * lane (lds_dst wave-uniform)
__device__ __forceinline__ void g6_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the workgroup barrier of the pipelined loop: this wave's LDS writes and reads are done, nothing is said about vmcnt
__device__ __forceinline__ void g6_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

__device__ __forceinline__ void g6_products16(const unsigned char* la, const unsigned char* lb, int wm, int wn, int lane,
                                              f32x16 (&acc)[TM][TN]) {
  const int r = lane & 31, h = lane >> 5;
  Frag fa[TM], fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned char* p = la + (32 * TM * wm + 32 * i + r) * ROWB16 + 16 * h;
    fa[i].h = *reinterpret_cast<const u32x4*>(p);
    fa[i].m = *reinterpret_cast<const u32x4*>(p + PLANE16_A);
    fa[i].l = *reinterpret_cast<const u32x4*>(p + 2 * PLANE16_A);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = 32 * TN * wn + 32 * j + r;
    const unsigned char* p = lb + n * 32 + 16 * (h ^ ((n >> 3) & 1));
    fb[j].h = *reinterpret_cast<const u32x4*>(p);
    fb[j].m = *reinterpret_cast<const u32x4*>(p + PLANE16_B);
    fb[j].l = *reinterpret_cast<const u32x4*>(p + 2 * PLANE16_B);
  }
#define HOTK_G6_STEP(PA, PB)                                                     \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                 \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = g6_mfma(fa[i].PA, fb[j].PB, acc[i][j]);
  HOTK_G6_STEP(l, h)
  HOTK_G6_STEP(h, l)
  HOTK_G6_STEP(m, m)
  HOTK_G6_STEP(m, h)
  HOTK_G6_STEP(h, m)
  HOTK_G6_STEP(h, h)
#undef HOTK_G6_STEP
}

// PP ("ping-pong"): the two wave groups (rows 0..127 / 128..255 of the tile: one wave of each per SIMD) take turns --
// while one multiplies step i, the other requests step i + 2 and stages its share of step i + 1 -- so that one wave's
// fragment reads, splits and LDS writes run under the other's MFMAs instead of both waves of a SIMD doing the same
// thing at the same time.  Two barriers per step; every wave passes the same number of them.
template <bool PP>
__global__ __launch_bounds__(G6_THREADS, 1) void k_gemm6_nn16(const float* __restrict__ A, int64_t lda,
                                                              const uint16_t* __restrict__ planes, const float* __restrict__ bias,
                                                              float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                              int tiles_n) {
  extern __shared__ __align__(16) unsigned char lds[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
  const int64_t tile = g6_xcd_tile(blockIdx.x, gridDim.x);
  const int64_t tm = tile / tiles_n;
  const int tn = (int)(tile - tm * tiles_n);
  const int64_t m0 = tm * BM;
  const int n0 = tn * BN;
  const int S = K >> 4;                                       // k steps
  // A piece of this thread: row (t & 7) + 8 (t >> 4), k half (t >> 3) & 1
  const int arow_l = (t & 7) + 8 * (t >> 4), ahalf = (t >> 3) & 1;
  int64_t arow = m0 + arow_l;
  if (arow >= M) arow = M - 1;
  const float* ap = A + arow * lda + 8 * ahalf;
  const int aoff = arow_l * ROWB16 + 16 * ahalf;
  // B: this wave's DMA covers columns 32 wave .. 32 wave + 31; lane -> (column, chunk position); source chunk swizzled
  const int bn_l = 32 * wave + (lane >> 1);
  int bn = n0 + bn_l;
  if (bn >= N) bn = N - 1;
  const int bchunk = (lane & 1) ^ ((bn_l >> 3) & 1);
  const int64_t PL = (int64_t)N * K;                          // bf16 elements per plane
  const uint16_t* bp = planes + (int64_t)bn * 16 + 8 * bchunk;   // + (plane * K / 16 + step) * N * 16
  const uint32_t bdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + P16_B0 + wave * 1024));
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[2][2];                                             // A register ring: [set][two 16-byte halves of the piece]
  auto step_of = [&](int i) __attribute__((always_inline)) { return i < S ? i : S - 1; };   // steps past K re-read the last one
#define HOTK_G6_LOAD_A(SET, I)                                  \
  {                                                             \
    const float* q_ = ap + 16 * step_of(I);                     \
    g6_ld16(ra[SET][0], q_);                                    \
    g6_ld16(ra[SET][1], q_ + 4);                                \
  }
#define HOTK_G6_DMA_B(I, STAGE)                                                                 \
  {                                                                                             \
    const uint16_t* q_ = bp + (int64_t)step_of(I) * N * 16;                                     \
    const uint32_t d_ = bdst + (STAGE) * STAGE16_B;                                             \
    g6_glds16(q_, d_);                                                                          \
    g6_glds16(q_ + PL, d_ + PLANE16_B);                                                         \
    g6_glds16(q_ + 2 * PL, d_ + 2 * PLANE16_B);                                                 \
  }
#define HOTK_G6_WRITE_A(SET, STAGE)                                                                               \
  {                                                                                                               \
    g6_arrived(ra[SET][0]);                                                                                       \
    g6_arrived(ra[SET][1]);                                                                                       \
    const u32x4 x_ = ra[SET][0], y_ = ra[SET][1];                                                                 \
    uint32_t h_[4], m_[4], l_[4];                                                                                 \
    g6_split_pair(g6_u2f(x_[0]), g6_u2f(x_[1]), h_[0], m_[0], l_[0]);                                             \
    g6_split_pair(g6_u2f(x_[2]), g6_u2f(x_[3]), h_[1], m_[1], l_[1]);                                             \
    g6_split_pair(g6_u2f(y_[0]), g6_u2f(y_[1]), h_[2], m_[2], l_[2]);                                             \
    g6_split_pair(g6_u2f(y_[2]), g6_u2f(y_[3]), h_[3], m_[3], l_[3]);                                             \
    unsigned char* w_ = lds + P16_A0 + (STAGE) * STAGE16_A + aoff;                                                \
    *reinterpret_cast<u32x4*>(w_) = (u32x4){h_[0], h_[1], h_[2], h_[3]};                                          \
    *reinterpret_cast<u32x4*>(w_ + PLANE16_A) = (u32x4){m_[0], m_[1], m_[2], m_[3]};                              \
    *reinterpret_cast<u32x4*>(w_ + 2 * PLANE16_A) = (u32x4){l_[0], l_[1], l_[2], l_[3]};                          \
  }
  // (wave-uniform) PP: the group that stages first -- it runs the same loop half a step behind: its staging of iteration i
  // is that of step i + 1, and it names its two register sets the other way round, so that the loop body is one code
  const int off = (PP && __builtin_amdgcn_readfirstlane(wm) != 0) ? 1 : 0;
  // prologue: steps 0 and 1 requested; step 0 staged
  if (off == 0) {
    HOTK_G6_LOAD_A(0, 0)
    HOTK_G6_DMA_B(0, 0)
    HOTK_G6_LOAD_A(1, 1)
    HOTK_G6_DMA_B(1, 1)
    g6_wait_vm<5>();                                          // A(0), B(0) landed; A(1), B(1) may be in flight
    HOTK_G6_WRITE_A(0, 0)
    g6_barrier_lds();
  } else {
    HOTK_G6_LOAD_A(1, 0)
    HOTK_G6_DMA_B(0, 0)
    HOTK_G6_LOAD_A(0, 1)
    HOTK_G6_DMA_B(1, 1)
    g6_wait_vm<5>();
    HOTK_G6_WRITE_A(1, 0)
    g6_barrier_lds();
    // its extra half step, while the first group multiplies step 0: request step 2, stage step 1
    HOTK_G6_LOAD_A(1, 2)
    HOTK_G6_DMA_B(2, 2)
    g6_wait_vm<5>();
    HOTK_G6_WRITE_A(0, 1)
    g6_barrier_lds();
  }
  // iteration i: multiply step i; request step j + 2 and stage A of step j + 1, j = i + off (requested an iteration ago)
  for (int i0 = 0; i0 < S; i0 += 6) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {                             // 6 = lcm of the ring depths 2 and 3: static stage indices
      const int i = i0 + u;
      if (!PP) {
        HOTK_G6_LOAD_A(u & 1, i + 2)
        HOTK_G6_DMA_B(i + 2, (u + 2) % 3)
      }
      if (HOTK_G6_PROBE_PROD(i < S, i == 0)) {
        if (PP) __builtin_amdgcn_s_setprio(1);
        g6_products16(lds + P16_A0 + (u & 1) * STAGE16_A, lds + P16_B0 + (u % 3) * STAGE16_B, wm, wn, lane, acc);
        if (PP) __builtin_amdgcn_s_setprio(0);
      }
      if (PP) {
        g6_barrier_lds();
        const int sb = (u + 2) % 3, sb1 = (u + 3) % 3;        // B stage of step j + 2
        HOTK_G6_LOAD_A(u & 1, i + off + 2)
        HOTK_G6_DMA_B(i + off + 2, off ? sb1 : sb)
      }
      g6_wait_vm<5>();                                        // step j + 1 landed (A registers and B stage); step j + 2 in flight
      HOTK_G6_WRITE_A((u + 1) & 1, (u + 1 + off) & 1)
      g6_barrier_lds();
    }
  }
  if (PP && off == 0) g6_barrier_lds();                        // the half step the other group spent in its prologue
#undef HOTK_G6_LOAD_A
#undef HOTK_G6_DMA_B
#undef HOTK_G6_WRITE_A
  g6_drain();
  g6_arrived(ra[0][0]); g6_arrived(ra[0][1]); g6_arrived(ra[1][0]); g6_arrived(ra[1][1]);   // (allocated until here: see g6_drain)
  g6_barrier_lds();                                           // the stages become the epilogue's staging image
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                    (!bias || ((uintptr_t)bias & 15) == 0);
  g6_store_tile(C, ldc, m0 + 32 * TM * wm, M - (m0 + 32 * TM * wm), n0 + 32 * TN * wn, N - (n0 + 32 * TN * wn), bias, lds, wave,
                lane, acc, full);
}

#endif  // HOTK_G6_BN == 256

// ---- P[s] = A[rows of slab s]^T . G[rows of slab s] -------------------------------------------------------------------
// Both operands are read row-major and must enter the MFMA with the ROW index as its k: the LDS planes are [k or n][32
// rows], written as packed row PAIRS: thread (row pair mp = t & 15, column group cg = t >> 4) loads 4 consecutive columns
// of its two rows and writes, per column and plane, one 32-bit word (rows 2 mp, 2 mp + 1) -- 16 consecutive lanes fill 64
// contiguous bytes of one LDS row.  The threads cover G6_THREADS / 4 columns per pass.
constexpr int T_COLS = G6_THREADS / 4;
constexpr int TPA = BM / T_COLS, TPG = BN / T_COLS;   // column passes over the A tile / the G tile
static_assert(TPA * T_COLS == BM && TPG * T_COLS == BN, "column passes cover the tile");
struct TRaw { u32x4 a[TPA][2], g[TPG][2]; };   // [column pass][row of the pair]
constexpr int TN_LOADS = 2 * (TPA + TPG);

// in0 / in1: whether the two rows of the pair lie inside the slab (rows past it enter as zeros)
template <int NP>
__device__ __forceinline__ void g6_store_t(unsigned char* oper, int plane, int t, const u32x4 (&v)[NP][2], bool in0, bool in1) {
  const int mp = t & 15, cg = t >> 4;
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t hi, mid, lo;
      g6_split_pair(in0 ? g6_u2f(v[pass][0][c]) : 0.f, in1 ? g6_u2f(v[pass][1][c]) : 0.f, hi, mid, lo);
      unsigned char* p = oper + (T_COLS * pass + 4 * cg + c) * ROWB + mp * 4;
      *reinterpret_cast<uint32_t*>(p) = hi;
      *reinterpret_cast<uint32_t*>(p + plane) = mid;
      *reinterpret_cast<uint32_t*>(p + 2 * plane) = lo;
    }
  }
}

// rows rw[0], rw[1] of the operands: the thread's row pair of a step
__device__ __forceinline__ void g6_load_tn(const float* __restrict__ A, int64_t lda, const float* __restrict__ G, int64_t ldg,
                                           const int64_t (&rw)[2], const int (&ka)[TPA], const int (&na)[TPG], TRaw& raw) {
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int64_t rc = rw[w];
#pragma unroll
    for (int pass = 0; pass < TPA; ++pass) g6_ld16(raw.a[pass][w], A + rc * lda + ka[pass]);
#pragma unroll
    for (int pass = 0; pass < TPG; ++pass) g6_ld16(raw.g[pass][w], G + rc * ldg + na[pass]);
  }
}

__device__ __forceinline__ void g6_wait_tn(TRaw& raw) {
  g6_wait_vm<(DEPTH - 1) * TN_LOADS>();
#pragma unroll
  for (int w = 0; w < 2; ++w) {
#pragma unroll
    for (int pass = 0; pass < TPA; ++pass) g6_arrived(raw.a[pass][w]);
#pragma unroll
    for (int pass = 0; pass < TPG; ++pass) g6_arrived(raw.g[pass][w]);
  }
}

// rows `row`, `row + 1` of a step, clamped into the slab [., r1); RM: entries of the row list
template <bool RM>
__device__ __forceinline__ void g6_rows_tn(int64_t row, int64_t r1, const int32_t* __restrict__ rows, int64_t (&rw)[2]) {
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    rw[w] = row + w < r1 ? row + w : r1 - 1;
    if (RM) rw[w] = rows[rw[w]];
  }
}

// RM: M = number of LISTED rows, the sum runs over rows rows[0 .. M) of A and G (the list is read with ordinary loads
// next to the hand-counted ones: sound with one step in flight, where every wait is a wait for everything)
static_assert(DEPTH == 1, "row lists: see above");
template <bool RM>
__global__ __launch_bounds__(G6_THREADS, HOTK_G6_WGS) void k_gemm6_tn(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ G, int64_t ldg,
                                                                      float* __restrict__ partial, int64_t M, int K, int N,
                                                                      int tiles_k, int tiles_n, int64_t rows_per_slab,
                                                                      const int32_t* __restrict__ rows, int col0, int ncap) {
  // columns [col0, ncap) of the product (N = all columns of G: the row stride of the partial blocks)
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* lds_a = lds;
  unsigned char* lds_b = lds + OPER_A;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
  const int64_t bid = blockIdx.x;
  const int64_t slab = bid / (tiles_k * tiles_n);
  const int rem = (int)(bid - slab * (tiles_k * tiles_n));
  const int tk = rem / tiles_n, tn = rem - tk * tiles_n;
  const int k0 = tk * BM, n0 = col0 + tn * BN;
  const int64_t r0 = slab * rows_per_slab;
  int64_t r1 = r0 + rows_per_slab;
  if (r1 > M) r1 = M;
  const int mp = t & 15, cg = t >> 4;
  // columns past the end of a partial tile read the last valid column group (their products land in rows / columns that
  // are never stored)
  int ka[TPA], na[TPG];
#pragma unroll
  for (int pass = 0; pass < TPA; ++pass) {
    ka[pass] = k0 + T_COLS * pass + 4 * cg;
    if (ka[pass] + 4 > K) ka[pass] = K - 4;
  }
#pragma unroll
  for (int pass = 0; pass < TPG; ++pass) {
    na[pass] = n0 + T_COLS * pass + 4 * cg;
    if (na[pass] + 4 > ncap) na[pass] = ncap - 4;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // DEPTH 32-row steps in flight in registers, as in k_gemm6_nn; steps past the slab re-read its last row and enter as zeros
  TRaw ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    int64_t rw[2];
    g6_rows_tn<RM>(r0 + d * BK + 2 * mp, r1, rows, rw);
    g6_load_tn(A, lda, G, ldg, rw, ka, na, ring[d]);
  }
  for (int64_t m = r0; m < r1; m += DEPTH * BK) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int64_t mm = m + d * BK;
      g6_wait_tn(ring[d]);
      // (RM: the list entries of the next loads are fetched here, a split + LDS write + barrier ahead of their use)
      int64_t rw[2];
      g6_rows_tn<RM>(mm + DEPTH * BK + 2 * mp, r1, rows, rw);
      const bool in0 = mm + 2 * mp < r1, in1 = mm + 2 * mp + 1 < r1;
      g6_store_t<TPA>(lds_a, PLANE_A, t, ring[d].a, in0, in1);
      g6_store_t<TPG>(lds_b, PLANE_B, t, ring[d].g, in0, in1);
      __syncthreads();
      g6_load_tn(A, lda, G, ldg, rw, ka, na, ring[d]);
      if (mm < r1) g6_products(lds_a, lds_b, wm, wn, lane, acc);
      __syncthreads();
    }
  }
  g6_drain();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) g6_wait_tn(ring[d]);
  float* P = partial + slab * (int64_t)K * N;
  const bool full = (N & 3) == 0 && ((uintptr_t)partial & 15) == 0;
  g6_store_tile(P, N, k0 + 32 * TM * wm, K - (k0 + 32 * TM * wm), n0 + 32 * TN * wn, ncap - (n0 + 32 * TN * wn), nullptr, lds, wave,
                lane, acc, full);
}

// operand planes of a k step, re-used by the epilogue as one 32-row staging image per wave
constexpr size_t G6_STAGE = (size_t)WM * WN * 32 * G6_ST * sizeof(float);
constexpr size_t G6_LDS = (size_t)(OPER_A + OPER_B) > G6_STAGE ? (size_t)(OPER_A + OPER_B) : G6_STAGE;
static_assert(HOTK_G6_WGS * G6_LDS <= 160 * 1024, "LDS of the workgroups of a CU");

#ifndef HOTK_G6_TN_WGS
#define HOTK_G6_TN_WGS 256
#endif
// A column tail of at most 128 goes to the 256 x 128 build -- when the wide part is ONE tile column (N = 320, 384): both
// launches are then a full round of workgroups over the same slabs.  With more wide tiles the tail launch would fill
// 1 / tiles of the chip for a whole round (256 x 864: 0.76 against 0.70 ms with the tail padded to a fourth 256-tile).
inline int64_t g6_tn_tail(int64_t N) {
  const int64_t tail = N % 256;
  return (tail > 0 && tail <= 128 && N < 512) ? tail : 0;
}
int g6_slabs(int64_t M, int64_t K, int64_t N, int64_t* rows_per_slab) {
  // (with a tail, both launches use the slabs of the wide part: each is one full round; a tail alone: its own tiles)
  const int64_t nw = N - g6_tn_tail(N);
  const int64_t tiles = ((K + 255) / 256) * (nw > 0 ? (nw + 255) / 256 : 1);
  // one round of workgroups over the 256 CUs (one is resident per CU; a second round that is not full costs a whole
  // round, and every slab costs a K x N partial block written and read again), slabs of at least 1 024 rows (a multiple
  // of the 32-row step)
  int64_t s = HOTK_G6_TN_WGS / tiles;
  const int64_t max_s = (M + 1023) / 1024;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int64_t rps = (M + s - 1) / s;
  rps = (rps + BK - 1) / BK * BK;
  *rows_per_slab = rps;
  return (int)((M + rps - 1) / rps);
}

template <class F>
int g6_attr(F f) {
  return hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G6_LDS) == hipSuccess ? HOTK_OK
                                                                                                                    : HOTK_ELAUNCH;
}

// columns [col0, ncap) of C = A . B on this build's tiles; rm.rows / rm.rec: see hotk_gemm6_nn_rows / hotk_gemm6_entry22
template <int MODE>
int g6_launch_mode(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc, int64_t Mt, int64_t K,
                   int64_t N, int64_t col0, int64_t ncap, const RowMap& rm, hipStream_t st) {
  const int64_t tiles_m = (Mt + BM - 1) / BM, tiles_n = (ncap - col0 + BN - 1) / BN;
  if (tiles_m * tiles_n > 0x7fffffffll) return HOTK_EUNSUPPORTED;
  static int attr = g6_attr(k_gemm6_nn<MODE>);
  if (attr) return attr;
  hipLaunchKernelGGL(k_gemm6_nn<MODE>, dim3((unsigned)(tiles_m * tiles_n)), dim3(G6_THREADS), G6_LDS, st, A, lda,
                     (const uint16_t*)planes, bias, C, ldc, Mt, (int)K, (int)N, (int)tiles_n, (int)col0, (int)ncap, rm);
  return hotk_launch_status();
}
// columns [col0, ncap) of the per-slab partial products on this build's tiles: `slabs` slabs of rps rows over Ms rows (or list
// entries)
int g6_launch_tn(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t Ms, int64_t K, int64_t N,
                 int64_t col0, int64_t ncap, int slabs, int64_t rps, const int32_t* rows, hipStream_t st) {
  const int tiles_k = (int)((K + BM - 1) / BM), tiles_n = (int)((ncap - col0 + BN - 1) / BN);
  const dim3 grid((unsigned)((int64_t)slabs * tiles_k * tiles_n));
  if (rows) {
    static int attr = g6_attr(k_gemm6_tn<true>);
    if (attr) return attr;
    hipLaunchKernelGGL(k_gemm6_tn<true>, grid, dim3(G6_THREADS), G6_LDS, st, A, lda, G, ldg, partial, Ms, (int)K, (int)N, tiles_k,
                       tiles_n, rps, rows, (int)col0, (int)ncap);
  } else {
    static int attr = g6_attr(k_gemm6_tn<false>);
    if (attr) return attr;
    hipLaunchKernelGGL(k_gemm6_tn<false>, grid, dim3(G6_THREADS), G6_LDS, st, A, lda, G, ldg, partial, Ms, (int)K, (int)N,
                       tiles_k, tiles_n, rps, rows, (int)col0, (int)ncap);
  }
  return hotk_launch_status();
}
int g6_launch_nn(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc, int64_t M, int64_t K,
                 int64_t N, int64_t col0, int64_t ncap, const RowMap& rm, hipStream_t st) {
  if (rm.rec) return g6_launch_mode<2>(A, lda, planes, bias, C, ldc, rm.Mc, K, N, col0, ncap, rm, st);
  if (rm.rows) return g6_launch_mode<1>(A, lda, planes, bias, C, ldc, rm.Mc, K, N, col0, ncap, rm, st);
  return g6_launch_mode<0>(A, lda, planes, bias, C, ldc, M, K, N, col0, ncap, rm, st);
}

}  // namespace HOTK_G6_NS
using namespace HOTK_G6_NS;
}  // namespace

#ifdef HOTK_G6_VARIANT
// A second build of this file (csrc/gemm6_n128.hip: 256 x 128 tiles) exports ONE entry point: columns [col0, ncap) of
// C = A . B for the column tail that would leave a 256-wide tile mostly empty.  Called by hotk_gemm6_nn below.
#define HOTK_G6_CAT2(a, b) a##b
#define HOTK_G6_CAT(a, b) HOTK_G6_CAT2(a, b)
extern "C" int HOTK_G6_CAT(hotk_gemm6_nn_cols, HOTK_G6_VARIANT)(const float* A, int64_t lda, const void* planes, const float* bias,
                                                                 float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                                                                 int64_t col0, int64_t ncap, const void* row_map, void* stream) {
  return g6_launch_nn(A, lda, planes, bias, C, ldc, M, K, N, col0, ncap, *static_cast<const RowMap*>(row_map), (hipStream_t)stream);
}
extern "C" int HOTK_G6_CAT(hotk_gemm6_tn_cols, HOTK_G6_VARIANT)(const float* A, int64_t lda, const float* G, int64_t ldg,
                                                                 float* partial, int64_t Ms, int64_t K, int64_t N, int64_t col0,
                                                                 int64_t ncap, int slabs, int64_t rps, const int32_t* rows,
                                                                 void* stream) {
  return g6_launch_tn(A, lda, G, ldg, partial, Ms, K, N, col0, ncap, slabs, rps, rows, (hipStream_t)stream);
}
#else
extern "C" int hotk_gemm6_tn_cols_n128(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t Ms,
                                       int64_t K, int64_t N, int64_t col0, int64_t ncap, int slabs, int64_t rps,
                                       const int32_t* rows, void* stream);
// (row_map: the RowMap of this file, same layout in both builds)
extern "C" int hotk_gemm6_nn_cols_n128(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc,
                                       int64_t M, int64_t K, int64_t N, int64_t col0, int64_t ncap, const void* row_map,
                                       void* stream);
extern "C" int hotk_gemm6_nn_supported(int64_t M, int64_t K, int64_t N) {
  return (M > 0 && K >= 32 && K % 32 == 0 && N >= 1 && K <= (1 << 20) && N <= (1 << 20) && M < ((int64_t)1 << 40)) ? 1 : 0;
}

extern "C" int hotk_gemm6_tn_supported(int64_t M, int64_t K, int64_t N) {
  return (M > 0 && K >= 4 && K % 4 == 0 && N >= 4 && N % 4 == 0 && K <= (1 << 20) && N <= (1 << 20)) ? 1 : 0;
}

namespace {
// HOTK_G6_PIPE=1 selects the pipelined kernel (k step 16, planes [K/16][N][16]) for A/B runs; default: the two-phase one
// (k step 32, planes [K/32][N][32]) -- they measure the same (0.93 / 0.90 ms on [254 k, 512] x [512, 512]), the simpler one
// is kept.  One switch for hotk_gemm6_split_weight and hotk_gemm6_nn: the planes are theirs.
int g6_pipe_mode() {
  static const int v = [] { const char* e = getenv("HOTK_G6_PIPE"); return (e && e[0] >= '1' && e[0] <= '2') ? e[0] - '0' : 0; }();
  return v;
}
bool g6_pipe() { return g6_pipe_mode() != 0; }

// rows == nullptr: all M rows.  Otherwise the Mc listed rows (ascending, each < M); the other rows of C are zeroed.
int g6_nn(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc, int64_t M, int64_t K,
          int64_t N, const RowMap& rm, hipStream_t st) {
  const bool rows = rm.rows || rm.rec;      // a list of rows (plain, or as records)
  const int64_t Mc = rm.Mc;
  HOTK_CHECK_ARG(M >= 0 && lda >= K && ldc >= N && (!rows || (Mc >= 0 && Mc <= M && M <= 0x7fffffffll)));
  if (M == 0) return HOTK_OK;
  if (!hotk_gemm6_nn_supported(M, K, N)) return HOTK_EUNSUPPORTED;
  HOTK_CHECK_ARG(A && planes && C && hotk_aligned16(A) && hotk_aligned16(planes) && (lda & 3) == 0);
  // (row lists: 16-byte pieces only -- the kernel's element-wise store path does not know the list)
  if (rows && !((N & 3) == 0 && (ldc & 3) == 0 && hotk_aligned16(C) && (!bias || hotk_aligned16(bias)))) return HOTK_EUNSUPPORTED;
  if (rows && Mc == 0)
    return hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st) == hipSuccess
               ? HOTK_OK : HOTK_ELAUNCH;
  // a column tail of at most 128 goes to the 256 x 128 build of the kernel (N = 320: one 256-wide and one 128-wide tile
  // instead of two 256-wide ones, the second three quarters empty)
  const int64_t tail = N % BN;
  const bool pipe = g6_pipe() && !rows && BN == 256;
  const bool split_tail = !pipe && tail > 0 && tail <= 128;
  const int64_t nwide = split_tail ? N - tail : N;
  if (split_tail) {
    const int rc = hotk_gemm6_nn_cols_n128(A, lda, planes, bias, C, ldc, M, K, N, nwide, N, &rm, st);
    if (rc || nwide == 0) return rc;
  }
#if HOTK_G6_BN == 256
  if (pipe) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (nwide + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffll) return HOTK_EUNSUPPORTED;
    static int attr16 = (hipFuncSetAttribute((const void*)k_gemm6_nn16<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)G6_LDS16) == hipSuccess &&
                         hipFuncSetAttribute((const void*)k_gemm6_nn16<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)G6_LDS16) == hipSuccess) ? HOTK_OK : HOTK_ELAUNCH;
    if (attr16) return attr16;
    if (g6_pipe_mode() == 2)
      hipLaunchKernelGGL(k_gemm6_nn16<true>, dim3((unsigned)(tiles_m * tiles_n)), dim3(G6_THREADS), G6_LDS16, st, A, lda,
                         (const uint16_t*)planes, bias, C, ldc, M, (int)K, (int)N, (int)tiles_n);
    else
      hipLaunchKernelGGL(k_gemm6_nn16<false>, dim3((unsigned)(tiles_m * tiles_n)), dim3(G6_THREADS), G6_LDS16, st, A, lda,
                         (const uint16_t*)planes, bias, C, ldc, M, (int)K, (int)N, (int)tiles_n);
    return hotk_launch_status();
  }
#endif
  return g6_launch_nn(A, lda, planes, bias, C, ldc, M, K, N, 0, nwide, rm, st);
}

int g6_tn(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t M, int64_t K, int64_t N,
          const int32_t* rows, int64_t Mc, hipStream_t st) {
  HOTK_CHECK_ARG(M >= 0 && lda >= K && ldg >= N && partial && (!rows || (Mc >= 0 && Mc <= M && M <= 0x7fffffffll)));
  const int64_t Ms = rows ? Mc : M;      // rows summed over
  if (Ms == 0) return hipMemsetAsync(partial, 0, (size_t)K * N * sizeof(float), st) == hipSuccess ? HOTK_OK : HOTK_ELAUNCH;
  if (!hotk_gemm6_tn_supported(M, K, N)) return HOTK_EUNSUPPORTED;
  HOTK_CHECK_ARG(A && G && hotk_aligned16(A) && hotk_aligned16(G) && ((lda | ldg) & 3) == 0);
  int64_t rps;
  const int slabs = g6_slabs(Ms, K, N, &rps);
  const int64_t tail = g6_tn_tail(N), nw = N - tail;
  if (tail) {
    const int rc = hotk_gemm6_tn_cols_n128(A, lda, G, ldg, partial, Ms, K, N, nw, N, slabs, rps, rows, st);
    if (rc || nw == 0) return rc;
  }
  return g6_launch_tn(A, lda, G, ldg, partial, Ms, K, N, 0, nw, slabs, rps, rows, st);
}
}  // namespace

extern "C" int hotk_gemm6_split_weight(const float* W, int64_t ldw, int64_t K, int64_t N, int transposed, void* planes,
                                       void* stream) {
  HOTK_CHECK_ARG(W && planes && K > 0 && N > 0 && K % 32 == 0 && ldw >= (transposed ? N : K));
  hipLaunchKernelGGL(k_gemm6_split_weight, dim3(hotk_grid(N * (K / 2), 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, (int)K,
                     (int)N, transposed, (uint16_t*)planes, g6_pipe() ? 4 : 5);
  return hotk_launch_status();
}

extern "C" int hotk_gemm6_nn(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc,
                             int64_t M, int64_t K, int64_t N, void* stream) {
  return g6_nn(A, lda, planes, bias, C, ldc, M, K, N, RowMap{}, (hipStream_t)stream);
}

extern "C" int hotk_gemm6_nn_rows(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc,
                                  int64_t M, int64_t K, int64_t N, const int32_t* rows, int64_t n_rows, void* stream) {
  HOTK_CHECK_ARG(rows || n_rows == 0);
  if (!rows) return hotk_gemm6_nn(A, lda, planes, bias, C, ldc, M, K, N, stream);
  RowMap rm{};
  rm.rows = rows; rm.Mc = n_rows; rm.Mall = M;
  return g6_nn(A, lda, planes, bias, C, ldc, M, K, N, rm, (hipStream_t)stream);
}

extern "C" int hotk_gemm6_entry22(const float* A, int64_t lda, const void* planes, float* Y, int64_t ldy, int64_t M, int64_t K,
                                  int64_t N, const int32_t* rec, int64_t n_rows, const float* yn, int64_t ld_yn,
                                  const float* b_off, const float* b_diag, void* stream) {
  HOTK_CHECK_ARG(rec && yn && n_rows > 0 && ld_yn >= 3 * N && (ld_yn & 3) == 0 && hotk_aligned16(rec) && hotk_aligned16(yn) &&
                 (!b_off || hotk_aligned16(b_off)) && (!b_diag || hotk_aligned16(b_diag)));
  RowMap rm{};
  rm.rec = reinterpret_cast<const int4*>(rec); rm.Mc = n_rows; rm.Mall = M;
  rm.yn = yn; rm.ld_yn = ld_yn; rm.b_off = b_off; rm.b_diag = b_diag; rm.C = (int)N;
  return g6_nn(A, lda, planes, nullptr, Y, ldy, M, K, N, rm, (hipStream_t)stream);
}

extern "C" int hotk_gemm6_tn_partials(int64_t M, int64_t K, int64_t N, int64_t* n_partials) {
  HOTK_CHECK_ARG(n_partials && M >= 0 && K > 0 && N > 0);
  int64_t rps;
  *n_partials = M == 0 ? 1 : g6_slabs(M, K, N, &rps);
  return HOTK_OK;
}

extern "C" int hotk_gemm6_tn(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t M, int64_t K,
                             int64_t N, void* stream) {
  return g6_tn(A, lda, G, ldg, partial, M, K, N, nullptr, 0, (hipStream_t)stream);
}

extern "C" int hotk_gemm6_tn_rows(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t M, int64_t K,
                                  int64_t N, const int32_t* rows, int64_t n_rows, void* stream) {
  HOTK_CHECK_ARG(rows || n_rows == 0);
  if (!rows) return hotk_gemm6_tn(A, lda, G, ldg, partial, M, K, N, stream);
  return g6_tn(A, lda, G, ldg, partial, M, K, N, rows, n_rows, (hipStream_t)stream);
}
#endif  // HOTK_G6_VARIANT
