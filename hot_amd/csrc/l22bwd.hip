// Entry-level reductions of the backward pass of the factored Linear(2, 2, 'light') (models/sparse/linear.py:143-161 under
// autograd; ops._l22_backward) in ONE pass over the incoming gradient g = dy [R, C]:
//     dyn[v][0 .. C)    = g on the diagonal entry (v, v)                       (was hotk_gather_rows)
//     dyn[v][C .. 2C)   = sum of g over the off-diagonal entries of ROW v      (was hotk_seg_sum_rows_pair: g read from HBM
//     dyn[v][2C .. 3C)  = sum of g over the off-diagonal entries of COLUMN v    and once more from the last-level cache)
//     dW2 [64, C]       = x2^T . g     (x2 = [x_ij | x_ji], zero on diagonal rows)   (was a split-K library GEMM: g again)
// At the headline shape (C = 288, 10 M entries) g is 11.5 GB and the three passes took 4.2 + 0.4 + 3.4 ms.
//
// All four results are products with g as the B operand of v_mfma_f32_32x32x16_bf16 contracted over ROWS: the weight
// gradient with A = x2^T, the two sums with A = a one-hot matrix [node of the graph][row of the tile] (exact in bf16), g
// as its exact three-way bf16 split (csrc/gemm.hip): six products for dW2 (fp32-exact to 2e-7), three for each sum
// (exact products, fp32 accumulation).  No table in LDS, no atomics: every sum is accumulated in matrix-core registers
// in a fixed order -- bitwise reproducible.
//
// Work split: a workgroup of nine waves walks whole graphs (persistent, graph b = block, block + grid, ...) in tiles of
// 32 rows; wave w owns the columns [32 w, 32 w + 32): its 32 x 32 piece of the tile comes into a wave-private LDS image
// by DMA (full 128-byte row pieces in, column fragments out), it keeps dW2[:, its columns] for the life of the workgroup
// and the two [Nmax <= 64, 32] sums for the life of a graph, then writes them to dyn.  The A fragments of x2 (split
// planes) and the rows' node ids are built once per tile by a few waves and shared through LDS, one tile ahead, with
// one workgroup barrier per tile.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rb_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float rb_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void rb_split3_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t h0 = rb_f2u(x0) & 0xffff0000u, h1 = rb_f2u(x1) & 0xffff0000u;
  const float r0 = x0 - rb_u2f(h0), r1 = x1 - rb_u2f(h1);
  const uint32_t m0 = rb_f2u(r0) & 0xffff0000u, m1 = rb_f2u(r1) & 0xffff0000u;
  const float s0 = r0 - rb_u2f(m0), s1 = r1 - rb_u2f(m1);
  hi = (h0 >> 16) | h1;
  mid = (m0 >> 16) | m1;
  lo = (rb_f2u(s0) >> 16) | (rb_f2u(s1) & 0xffff0000u);
}
struct Planes { u32x4 h, m, l; };
__device__ __forceinline__ Planes rb_split8(const float (&v)[8]) {
  Planes p;
  uint32_t h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    rb_split3_pair(v[2 * q], v[2 * q + 1], h, m, l);
    p.h[q] = h; p.m[q] = m; p.l[q] = l;
  }
  return p;
}
__device__ __forceinline__ f32x16 rb_mfma(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 rb_mfma6(const Planes& a, const Planes& b, f32x16 c) {   // small terms first
  c = rb_mfma(a.l, b.h, c);
  c = rb_mfma(a.h, b.l, c);
  c = rb_mfma(a.m, b.m, c);
  c = rb_mfma(a.m, b.h, c);
  c = rb_mfma(a.h, b.m, c);
  c = rb_mfma(a.h, b.h, c);
  return c;
}
constexpr int RB_NW = 9;          // waves = 32-column chunks: C = 288
constexpr int RB_C = 32 * RB_NW;
constexpr int RB_K = 64;          // width of x2
constexpr int RB_MAXN = 64;       // nodes per graph covered by two one-hot tiles

__device__ __forceinline__ void rb_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct Tile {          // wave-uniform
  int b;               // graph (>= B: past the end)
  int r0, e1;          // first row of the tile, end of the graph's rows
};

struct RbArgs {
  const void* G; int64_t ldg;       // float or (BF) bf16 rows; leading dimensions in elements
  const void* X2; int64_t ldx;
  const uint8_t* kind; const int32_t* rown; const int32_t* coln;
  int B, Nmax;
  float* dyn; int64_t ld_dyn;
  float* partial;
};

// (the per-graph tables come as __restrict__ kernel arguments: read-only for the life of the kernel, so these are scalar
// loads -- a vector load here is one the compiler waits for with vmcnt(0), i.e. for every DMA request in flight)
struct Graphs { const int32_t* __restrict__ ebeg; const int32_t* __restrict__ eend; int B; };
__device__ __forceinline__ Tile rb_first(const Graphs& a, int b) {
  Tile t;
  t.b = b;
  t.r0 = t.e1 = 0;
  if (b < a.B) { t.r0 = a.ebeg[b]; t.e1 = a.eend[b]; }
  t.b = __builtin_amdgcn_readfirstlane(t.b);
  t.r0 = __builtin_amdgcn_readfirstlane(t.r0);
  t.e1 = __builtin_amdgcn_readfirstlane(t.e1);
  return t;
}
__device__ __forceinline__ Tile rb_next(const Graphs& a, Tile t, int stride) {
  t.r0 += 32;
  if (t.r0 >= t.e1) t = rb_first(a, t.b + stride);     // (a graph without rows still is one all-dead tile: its dyn rows are written)
  return t;
}


// ---- global -> LDS by DMA (no registers, no ds_write): 16 / 4 bytes per lane to LDS byte address lds_dst + 16 / 4 * lane ----
__device__ __forceinline__ void rb_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same from a wave-uniform base and a 32-bit byte offset per lane (no 64-bit address arithmetic in vector registers)
__device__ __forceinline__ void rb_glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void rb_glds4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The DMA loads are invisible to the compiler's wait counting, so every load of the main loop is a DMA and every wait for
// one is written here: loads return in order, so "at most n outstanding" guarantees everything older than the n youngest
// LOADS has landed -- stores in flight (the diagonal rows, a graph's sums) can only make the wait longer, never too short.
template <int N>
__device__ __forceinline__ void rb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// workgroup barrier that says nothing about the loads in flight (a __syncthreads() would wait for all of them)
__device__ __forceinline__ void rb_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

constexpr int RB_G_STAGE = RB_NW * 4096;                   // bytes of a g stage: per wave [32 rows][32 floats]
constexpr int RB_OFF_XRAW = 3 * RB_G_STAGE;                // [2][fragment 4][16 rows][32 floats]
constexpr int RB_OFF_PLANES = RB_OFF_XRAW + 2 * 4 * 2048;  // [2][plane 3][fragment 4][64] u32x4
constexpr int RB_OFF_MRAW = RB_OFF_PLANES + 2 * 12 * 1024; // [2][row node 32 | column node 32]
constexpr int RB_OFF_META = RB_OFF_MRAW + 2 * 256;         // [2][32]: node of a diagonal row, -1 elsewhere
constexpr int RB_OFF_MASK = RB_OFF_META + 2 * 128;         // [2][row 64 | column 64] per node of the graph: bit r = tile row r is its off-diagonal entry
constexpr int RB_OFF_TAB = RB_OFF_MASK + 2 * 512;          // [256] u32x4: eight mask bits -> one-hot A fragment (bf16 1.0 / 0.0)
constexpr int RB_LDS_BYTES = RB_OFF_TAB + 256 * 16;

// NT: one-hot tiles of 32 nodes (Nmax <= 32 NT).  BF: g and x2 are bf16 tensors (the bf16 storage mode): a value is its own
// single plane -- one product per weight-gradient tile and per sum instead of six and three, no split; a wave's piece of a
// tile row is 64 bytes, the DMA instruction covers 16 rows, and the load counts below are halved (G = 2, X = 1).
//
// Pipeline.  Tile t + 2 is requested (DMA) while tile t is multiplied: three stages of g per wave, two of the raw x2
// fragments and node ids, two of the shared operands built from them (x2 planes, local ids).  Per iteration a wave issues
// X loads for the shared operands it builds (two for an x2 fragment, one for the ids, none otherwise) and then four for
// its g piece, always in that order and always all of them (past the last tile they re-read row 0), so the waits are
// constants: g of tile t is older than X + 4 loads, the raw operands of tile t + 1 older than X + 8.
template <int NT, bool BF>
__global__ __launch_bounds__(64 * RB_NW) void k_l22_bwd_reduce(RbArgs a, const int32_t* __restrict__ g_ebeg,
                                                               const int32_t* __restrict__ g_eend,
                                                               const int32_t* __restrict__ n_nodes) {
  Graphs gr;
  gr.ebeg = g_ebeg; gr.eend = g_eend; gr.B = a.B;
  extern __shared__ __align__(16) unsigned char lds[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col0 = 32 * wave;
  const int lr = lane >> 3, lc = 4 * (lane & 7);                               // DMA pieces: rows 8 i + lr, float4 column lc
  // builders of the shared operands (waves on the SIMDs that hold two of the nine waves)
  const int frag = wave == 1 ? 0 : wave == 2 ? 1 : wave == 3 ? 2 : wave == 5 ? 3 : -1;   // x2 fragment (kg = frag >> 1, mt = frag & 1)
  const bool meta_wave = wave == 6;

  f32x16 dw[2], rs[NT], cs[NT];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dw[0][r] = dw[1][r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rs[nt][r] = cs[nt][r] = 0.f;
  }

  // request tile t: raw shared operands -> slot s2, this wave's g piece -> stage s3
  auto issue = [&](const Tile& t, int s3, int s2) __attribute__((always_inline)) {
    const bool ok = t.b < a.B;
    const int64_t r0 = ok ? t.r0 : 0, e1 = ok ? t.e1 : 1;
    // Addresses = a wave-uniform row base (scalar arithmetic) + a 32-bit lane offset.  Rows past the graph's end re-read the
    // base row (zeroed after landing); a graph without rows reads the row before it (or row 0).
    const int64_t rb = r0 < e1 ? r0 : (e1 > 0 ? e1 - 1 : 0);
    const int nlive = (int)(e1 - r0 < 32 ? (e1 > r0 ? e1 - r0 : 0) : 32);      // live rows of the tile
    if (frag >= 0) {
      const int kg = frag >> 1, mt = frag & 1;
      if constexpr (BF) {
        const uint16_t* xb = reinterpret_cast<const uint16_t*>(a.X2) + rb * a.ldx + 32 * mt;
        const int rl = 16 * kg + (lane >> 2);
        const uint32_t off = (uint32_t)(((rl < nlive ? rl : 0) * (int)a.ldx + 8 * (lane & 3)) * 2);
        rb_glds16s(xb, off, lds0 + RB_OFF_XRAW + (s2 * 4 + frag) * 2048);
      } else {
        const float* xb = reinterpret_cast<const float*>(a.X2) + rb * a.ldx + 32 * mt;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int rl = 16 * kg + 8 * q + lr;
          const uint32_t off = (uint32_t)(((rl < nlive ? rl : 0) * (int)a.ldx + lc) * 4);
          rb_glds16s(xb, off, lds0 + RB_OFF_XRAW + ((s2 * 4 + frag) * 2 + q) * 1024);
        }
      }
    } else if (meta_wave) {
      const int64_t rc = j < nlive ? rb + j : rb;
      rb_glds4((half ? a.coln : a.rown) + rc, lds0 + RB_OFF_MRAW + s2 * 256);
    }
    if constexpr (BF) {
      const uint16_t* gb_ = reinterpret_cast<const uint16_t*>(a.G) + rb * a.ldg + col0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = 16 * i + (lane >> 2);
        const uint32_t off = (uint32_t)(((rl < nlive ? rl : 0) * (int)a.ldg + 8 * (lane & 3)) * 2);
        rb_glds16s(gb_, off, lds0 + s3 * RB_G_STAGE + (wave * 4 + i) * 1024);
      }
    } else {
      const float* gb_ = reinterpret_cast<const float*>(a.G) + rb * a.ldg + col0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 8 * i + lr;
#ifdef RB_PROBE_NOLOAD
        const uint32_t off = (uint32_t)((rl * 0 + lc) * 4);
#else
        const uint32_t off = (uint32_t)(((rl < nlive ? rl : 0) * (int)a.ldg + lc) * 4);
#endif
        rb_glds16s(gb_, off, lds0 + s3 * RB_G_STAGE + (wave * 4 + i) * 1024);
      }
    }
  };
  // "the raw operands of the tile before the youngest two are in LDS" / "the g piece ..." for this wave's load count
  auto wait_raw = [&]() __attribute__((always_inline)) {
    if (frag >= 0) rb_wait_vm<BF ? 5 : 10>();
    else if (meta_wave) rb_wait_vm<BF ? 5 : 9>();
  };
  auto wait_g = [&]() __attribute__((always_inline)) {
    if (frag >= 0) rb_wait_vm<BF ? 3 : 6>();
    else if (meta_wave) rb_wait_vm<BF ? 3 : 5>();
    else rb_wait_vm<BF ? 2 : 4>();
  };
  // shared operands of tile t from its raw slot s2 -> planes / ids buffer s2
  auto build_shared = [&](const Tile& t, int s2) __attribute__((always_inline)) {
    if (frag >= 0) {
      const int kg = frag >> 1;
      u32x4* d = reinterpret_cast<u32x4*>(lds + RB_OFF_PLANES) + s2 * 12 * 64 + frag * 64 + lane;
      if constexpr (BF) {
        const uint16_t* raw = reinterpret_cast<const uint16_t*>(lds + RB_OFF_XRAW + (s2 * 4 + frag) * 2048);
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t x = raw[(8 * half + e) * 32 + j];
          v[e] = ((int64_t)t.r0 + 16 * kg + 8 * half + e < t.e1) ? x : 0u;
        }
        u32x4 h;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = v[2 * q] | (v[2 * q + 1] << 16);
        d[0] = h;
      } else {
        const float* raw = reinterpret_cast<const float*>(lds + RB_OFF_XRAW) + (s2 * 4 + frag) * 512;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = raw[(8 * half + e) * 32 + j];
          v[e] = ((int64_t)t.r0 + 16 * kg + 8 * half + e < t.e1) ? x : 0.f;
        }
        const Planes p = rb_split8(v);
        d[0] = p.h; d[4 * 64] = p.m; d[8 * 64] = p.l;
      }
    }
    if (meta_wave) {
      // per node of the graph, the tile rows that are its off-diagonal row / column entries as a 32-bit mask: the one-hot
      // operands are then one table look-up per fragment instead of eight compares (ds_or: order-free, one wave)
      unsigned* msk = reinterpret_cast<unsigned*>(lds + RB_OFF_MASK) + s2 * 128;
      msk[lane] = 0u;
      msk[64 + lane] = 0u;
      rb_fence();
      if (half == 0) {
        const int* raw = reinterpret_cast<const int*>(lds + RB_OFF_MRAW) + s2 * 64;
        int jj = j;
        asm volatile("" : "+v"(jj));     // (not hoisted out of the tile loop: a spilled address is reloaded behind vmcnt(0))
        const int rn = raw[jj], cn = raw[32 + jj];
        const bool live = (int64_t)t.r0 + j < t.e1;     // (every row of [g_ebeg, g_eend) is an entry: diagonal iff row node == column node)
        const int nb = t.b * a.Nmax;
        if (live && rn != cn) {
          __hip_atomic_fetch_or(msk + ((rn - nb) & 63), 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_or(msk + 64 + ((cn - nb) & 63), 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        reinterpret_cast<int*>(lds + RB_OFF_META)[s2 * 32 + j] = (live && rn == cn) ? rn : -1;
      }
    }
  };

  if (tid < 256) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = ((tid >> (2 * q)) & 1 ? 0x3F80u : 0u) | ((tid >> (2 * q + 1)) & 1 ? 0x3F800000u : 0u);
    reinterpret_cast<u32x4*>(lds + RB_OFF_TAB)[tid] = o;
  }
  const int stride = gridDim.x;
  Tile cur = rb_first(gr, blockIdx.x);
  Tile nxt = rb_next(gr, cur, stride);
  Tile nn2 = rb_next(gr, nxt, stride);
  issue(cur, 0, 0);
  issue(nxt, 1, 1);
  wait_raw();
  build_shared(cur, 0);
  int s3 = 0, buf = 0;
  while (cur.b < a.B) {
    float* gt = reinterpret_cast<float*>(lds + s3 * RB_G_STAGE + wave * 4096);   // [32][32] this wave's piece of the tile
    wait_g();
    const uint16_t* gtb = reinterpret_cast<const uint16_t*>(gt);                    // BF: [32][32] bf16
    if (cur.r0 + 32 > cur.e1) {           // the graph's last tile: rows past its end enter as zeros
      if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rl = 16 * i + (lane >> 2);
          if ((int64_t)cur.r0 + rl >= cur.e1)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(gt) + rl * 64 + 16 * (lane & 3)) = make_uint4(0u, 0u, 0u, 0u);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((int64_t)cur.r0 + 8 * i + lr >= cur.e1) *reinterpret_cast<float4*>(gt + (8 * i + lr) * 32 + lc) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    rb_fence();
    {
      const int s3n = s3 == 0 ? 2 : s3 - 1;            // (s3 + 2) % 3: the stage of the tile before this one
      issue(nn2, s3n, buf);
    }
    rb_barrier_lds();      // the shared operands of this tile (built during the previous one) are in place

    const int* mt_ = reinterpret_cast<const int*>(lds + RB_OFF_META) + buf * 32;
    const unsigned* msk = reinterpret_cast<const unsigned*>(lds + RB_OFF_MASK) + buf * 128;
    const u32x4* tab = reinterpret_cast<const u32x4*>(lds + RB_OFF_TAB);
    unsigned rmask[NT], cmask[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { rmask[nt] = msk[32 * nt + j]; cmask[nt] = msk[64 + 32 * nt + j]; }
    const u32x4* xpl = reinterpret_cast<const u32x4*>(lds + RB_OFF_PLANES) + buf * 12 * 64;
    // diagonal rows of the tile: their g is the first block of the node's dyn row
    {
      const int dn = mt_[j];
      uint32_t mask = (uint32_t)__builtin_amdgcn_ballot_w64(dn >= 0);
      while (mask) {
        const int r = __builtin_amdgcn_readfirstlane(__builtin_ctz(mask));
        mask &= mask - 1;
        const int node = __builtin_amdgcn_readlane(dn, r);
        if (half == 0) a.dyn[(int64_t)node * a.ld_dyn + col0 + j] = BF ? rb_u2f((uint32_t)gtb[r * 32 + j] << 16) : gt[r * 32 + j];
      }
    }
#ifdef RB_PROBE_NOCOMP
    if (a.B < 0)
#endif
#pragma unroll 1
    for (int kg = 0; kg < 2; ++kg) {
      Planes gb;
      if constexpr (BF) {
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gtb[(16 * kg + 8 * half + e) * 32 + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) gb.h[q] = v[2 * q] | (v[2 * q + 1] << 16);
      } else {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gt[(16 * kg + 8 * half + e) * 32 + j];
        gb = rb_split8(v);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const u32x4* s = xpl + (kg * 2 + mt) * 64 + lane;
        if constexpr (BF) {
          dw[mt] = rb_mfma(s[0], gb.h, dw[mt]);
        } else {
          Planes xa;
          xa.h = s[0]; xa.m = s[4 * 64]; xa.l = s[8 * 64];
          dw[mt] = rb_mfma6(xa, gb, dw[mt]);
        }
      }
#ifdef RB_PROBE_NOSUMS
      if (a.B < 0)
#endif
      {
        const int sh = 16 * kg + 8 * half;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // (a one-hot tile without a single row of this half-tile is skipped: with entries sorted by row node the 16 rows
          // of a k group belong to a few nodes, i.e. to one of the two node tiles)
          const unsigned rb_ = (rmask[nt] >> (16 * kg)) & 0xffffu, cb_ = (cmask[nt] >> (16 * kg)) & 0xffffu;
          if (__builtin_amdgcn_ballot_w64(rb_ != 0u) != 0) {
            const u32x4 oh = tab[(rmask[nt] >> sh) & 0xffu];
            if constexpr (!BF) {
              rs[nt] = rb_mfma(oh, gb.l, rs[nt]);
              rs[nt] = rb_mfma(oh, gb.m, rs[nt]);
            }
            rs[nt] = rb_mfma(oh, gb.h, rs[nt]);
          }
          if (__builtin_amdgcn_ballot_w64(cb_ != 0u) != 0) {
            const u32x4 oc = tab[(cmask[nt] >> sh) & 0xffu];
            if constexpr (!BF) {
              cs[nt] = rb_mfma(oc, gb.l, cs[nt]);
              cs[nt] = rb_mfma(oc, gb.m, cs[nt]);
            }
            cs[nt] = rb_mfma(oc, gb.h, cs[nt]);
          }
        }
      }
    }
    if (nxt.b < a.B) {
      wait_raw();
      build_shared(nxt, buf ^ 1);
    }
    if (nxt.b != cur.b) {
      // the graph is complete: its node sums leave (C/D map: node = (reg & 3) + 8 (reg >> 2) + 4 half, column j)
      const int nn = __builtin_amdgcn_readfirstlane(n_nodes[cur.b]);
      const int64_t nb = (int64_t)cur.b * a.Nmax;
      // (one lane-dependent base, opaque to the optimiser: written as dyn + (nb + m) ld + .. it hoists the 32 lane-dependent
      // row offsets out of the tile loop into 64 registers, spills them, and every reload waits for all loads in flight)
      float* d0 = a.dyn + (nb + 4 * half) * a.ld_dyn + col0 + j;
      asm volatile("" : "+v"(d0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mu = 32 * nt + (r & 3) + 8 * (r >> 2);      // node = mu + 4 half
          if (mu + 4 * half < a.Nmax) {
            float* d = d0 + (int64_t)mu * a.ld_dyn;
            d[RB_C] = rs[nt][r];
            d[2 * RB_C] = cs[nt][r];
            if (mu + 4 * half >= nn) d[0] = 0.f;        // (a padding node has no diagonal entry)
          }
          rs[nt][r] = 0.f;
          cs[nt][r] = 0.f;
        }
      }
    }
    cur = nxt;
    nxt = nn2;
    nn2 = rb_next(gr, nn2, stride);
    s3 = s3 == 2 ? 0 : s3 + 1;
    buf ^= 1;
  }
  rb_wait_vm<0>();         // the requests past the last tile land before the workgroup gives its LDS back
  // dW2[:, this wave's columns] of the workgroup's graphs: one partial block per workgroup, summed by the caller in order
  float* pw = a.partial + (int64_t)blockIdx.x * RB_K * RB_C + col0 + j;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) pw[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half) * RB_C] = dw[mt][r];
}

unsigned rb_grid(int64_t B) { return (unsigned)(B < 256 ? (B < 1 ? 1 : B) : 256); }

}  // namespace

extern "C" int hotk_linear22_bwd_reduce_supported(int64_t K2, int64_t C, int64_t Nmax) {
  return (K2 == RB_K && C == RB_C && Nmax >= 1 && Nmax <= RB_MAXN) ? 1 : 0;
}

extern "C" int hotk_linear22_bwd_reduce_partials(int64_t B, int64_t* n_partials) {
  HOTK_CHECK_ARG(n_partials && B >= 0);
  *n_partials = (int64_t)rb_grid(B);
  return HOTK_OK;
}

namespace {
template <bool BF>
int rb_run(const void* g, int64_t ldg, const void* x2, int64_t ld_x2, int64_t K2, int64_t C, const int32_t* rown,
           const int32_t* coln, const int32_t* g_ebeg, const int32_t* g_eend, const int32_t* n_nodes, int64_t B, int64_t Nmax,
           int64_t R, float* dyn, int64_t ld_dyn, float* partial, void* stream) {
  HOTK_CHECK_ARG(B >= 0 && R >= 0 && ldg >= C && ld_x2 >= K2 && ld_dyn >= 3 * C);
  if (!hotk_linear22_bwd_reduce_supported(K2, C, Nmax)) return HOTK_EUNSUPPORTED;
  HOTK_CHECK_ARG(partial && B < ((int64_t)1 << 31) / (Nmax > 0 ? Nmax : 1) && R < ((int64_t)1 << 31));
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || R == 0) {
    // no entry rows: zero sums and a zero weight gradient (one partial block)
    if (B > 0 && Nmax > 0) {
      HOTK_CHECK_ARG(dyn != nullptr);
      if (hipMemset2DAsync(dyn, (size_t)ld_dyn * sizeof(float), 0, (size_t)3 * C * sizeof(float), (size_t)(B * Nmax), st) != hipSuccess)
        return HOTK_ELAUNCH;
    }
    return hipMemsetAsync(partial, 0, (size_t)rb_grid(B) * RB_K * RB_C * sizeof(float), st) == hipSuccess ? HOTK_OK : HOTK_ELAUNCH;
  }
  HOTK_CHECK_ARG(g && x2 && rown && coln && g_ebeg && g_eend && n_nodes && dyn);      // (kind is not read)
  // 16-byte row pieces: fp32 rows of 4 k elements, bf16 rows of 8 k
  HOTK_CHECK_ARG(hotk_aligned16(g) && hotk_aligned16(x2) && ((ldg | ld_x2) & (BF ? 7 : 3)) == 0);
  RbArgs a;
  a.G = g; a.ldg = ldg; a.X2 = x2; a.ldx = ld_x2; a.kind = nullptr; a.rown = rown; a.coln = coln;
  a.B = (int)B; a.Nmax = (int)Nmax;
  a.dyn = dyn; a.ld_dyn = ld_dyn; a.partial = partial;
  const size_t lds = RB_LDS_BYTES;
  const unsigned grid = rb_grid(B);
  static bool attr_set = false;   // (149 KB of dynamic LDS: above the 64 KB a kernel gets without asking)
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_l22_bwd_reduce<1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_l22_bwd_reduce<2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HOTK_ELAUNCH;
    attr_set = true;
  }
  if (Nmax <= 32) hipLaunchKernelGGL((k_l22_bwd_reduce<1, BF>), dim3(grid), dim3(64 * RB_NW), lds, st, a, g_ebeg, g_eend, n_nodes);
  else hipLaunchKernelGGL((k_l22_bwd_reduce<2, BF>), dim3(grid), dim3(64 * RB_NW), lds, st, a, g_ebeg, g_eend, n_nodes);
  return hotk_launch_status();
}
}  // namespace

extern "C" int hotk_linear22_bwd_reduce(const float* g, int64_t ldg, const float* x2, int64_t ld_x2, int64_t K2, int64_t C,
                                        const uint8_t* kind, const int32_t* rown, const int32_t* coln, const int32_t* g_ebeg,
                                        const int32_t* g_eend, const int32_t* n_nodes, int64_t B, int64_t Nmax, int64_t R,
                                        float* dyn, int64_t ld_dyn, float* partial, void* stream) {
  (void)kind;
  return rb_run<false>(g, ldg, x2, ld_x2, K2, C, rown, coln, g_ebeg, g_eend, n_nodes, B, Nmax, R, dyn, ld_dyn, partial, stream);
}

// g and x2 as bf16 tensors (leading dimensions in bf16 elements, multiples of 8); dyn and the partial blocks are fp32
extern "C" int hotk_linear22_bwd_reduce_bf16(const void* g, int64_t ldg, const void* x2, int64_t ld_x2, int64_t K2, int64_t C,
                                             const uint8_t* kind, const int32_t* rown, const int32_t* coln,
                                             const int32_t* g_ebeg, const int32_t* g_eend, const int32_t* n_nodes, int64_t B,
                                             int64_t Nmax, int64_t R, float* dyn, int64_t ld_dyn, float* partial,
                                             void* stream) {
  (void)kind;
  return rb_run<true>(g, ldg, x2, ld_x2, K2, C, rown, coln, g_ebeg, g_eend, n_nodes, B, Nmax, R, dyn, ld_dyn, partial, stream);
}
