// K4: coalesced row gather / mask / segment-sum primitives and the 'light' order-2 basis.
// Every kernel maps consecutive lanes to consecutive channels of one row, so each wave instruction
// touches whole 256-B (scalar) or 1-KiB (float4) row pieces; index arrays are read once per row piece.
#include "common.h"

namespace {

template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

__device__ __forceinline__ float vzero(float) { return 0.f; }
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
// an opaque use (no instruction): the value must be in registers here
__device__ __forceinline__ void vtouch(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void vtouch(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void vtouch(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vfill(float, float f) { return f; }
__device__ __forceinline__ float4 vfill(float4, float f) { return make_float4(f, f, f, f); }

// ------------------------------------------------------------------ gather
template <int V>
__global__ void k_gather_rows(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx,
                              float* __restrict__ out, int64_t ld_out, int64_t n_out, int64_t CV, int accumulate) {
  using T = typename VecT<V>::type;
  const int64_t total = n_out * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV, c = (t - r * CV) * V;
    int64_t s = idx ? (int64_t)idx[r] : r;
    T v = vzero(T());
    if (s >= 0) v = *reinterpret_cast<const T*>(src + s * ld_src + c);
    T* o = reinterpret_cast<T*>(out + r * ld_out + c);
    *o = accumulate ? vadd(*o, v) : v;
  }
}

// ------------------------------------------------------------------ mask
template <int V>
__global__ void k_mask_rows(const float* __restrict__ src, int64_t ld_src, const uint8_t* __restrict__ kind,
                            unsigned keep, float fill, float* __restrict__ out, int64_t ld_out, int64_t n, int64_t CV) {
  using T = typename VecT<V>::type;
  const int64_t total = n * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV, c = (t - r * CV) * V;
    T v = hotk_keep(keep, kind[r]) ? *reinterpret_cast<const T*>(src + r * ld_src + c) : vfill(T(), fill);
    *reinterpret_cast<T*>(out + r * ld_out + c) = v;
  }
}

// ------------------------------------------------------------------ segment sum of rows
// work item = (segment, chunk, channel-vector); a segment longer than HOTK_SEG_CHUNK is split and the
// chunks are combined with float atomics into the (zero-filled) output.
#ifndef HOTK_SEGSUM_UNROLL
#define HOTK_SEGSUM_UNROLL 8   /* measured at 10 M rows x 288: 2 -> 2.79, 4 -> 2.68, 8 -> 2.56 ms per call */
#endif
constexpr int SU = HOTK_SEGSUM_UNROLL;   // entries per step
// PERM / KIND: whether the call has a position -> row map and a row-kind filter.  Compile-time, because a load behind
// a run-time `perm ? perm[p] : p` sits in a block of its own and the compiler waits for it (vmcnt(0)) before the next
// one is issued: the eight index loads and eight kind loads of a step came out as SIXTEEN dependent round trips instead
// of two (found in the ISA at the end of round 2; the rows were always fetched together).
template <int V, bool PERM, bool KIND>
__global__ void k_seg_sum_rows(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ seg_begin,
                               const int32_t* __restrict__ seg_end, const int32_t* __restrict__ perm,
                               const uint8_t* __restrict__ kind, unsigned keep, float* __restrict__ out,
                               int64_t ld_out, int64_t n_seg, int64_t CV, int64_t n_chunk) {
  using T = typename VecT<V>::type;
  const int64_t total = n_seg * n_chunk * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    // (segment, chunk, column vector) of this item; 32-bit division when the item count allows it (a 64-bit division is
    // ~100 instructions, as many as the whole walk of a 10-entry segment)
    int64_t cv, ch, s;
    if (total <= 0xffffffffll) {
      const uint32_t tt = (uint32_t)t, cvu = (uint32_t)CV, ncu = (uint32_t)n_chunk;
      const uint32_t rest = tt / cvu;
      cv = tt - rest * cvu;
      const uint32_t su = ncu == 1 ? rest : rest / ncu;
      ch = rest - su * ncu;
      s = su;
    } else {
      cv = t % CV;
      const int64_t rest = t / CV;
      ch = rest % n_chunk;
      s = rest / n_chunk;
    }
    int64_t b = seg_begin[s], e = seg_end[s];
    int64_t lo = b + ch * HOTK_SEG_CHUNK, hi = lo + HOTK_SEG_CHUNK;
    if (hi > e) hi = e;
    if (lo >= hi && ch > 0) continue;
    T acc = vzero(T());
    // SU entries per step: index, kind and row loads of a step are independent of each other (one latency per
    // step instead of three per entry) and unconditional; rows past the end re-read the last entry and are not added.
    // The additions stay in segment order, so the sum is bitwise the same as the one-at-a-time loop.
    for (int64_t p = lo; p < hi; p += SU) {
      int64_t r[SU];
      uint8_t kd[SU];
      T v[SU];
      // (opaque uses of everything fetched: otherwise the compiler sinks a row's loads into the branch that adds it,
      // or waits for the first index on its own, and the round trips run one after the other)
      int pr[SU], kk[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int64_t pp = p + u < hi ? p + u : hi - 1;
        pr[u] = PERM ? perm[pp] : (int)pp;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (PERM) vtouch(pr[u]);
        r[u] = (int64_t)pr[u];
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) kk[u] = KIND ? (int)kind[r[u]] : 0;
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = *reinterpret_cast<const T*>(src + r[u] * ld_src + cv * V);
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (KIND) vtouch(kk[u]);
        kd[u] = (uint8_t)kk[u];
        vtouch(v[u]);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if ((p + u < hi) && (!KIND || hotk_keep(keep, kd[u]))) acc = vadd(acc, v[u]);
    }
    *reinterpret_cast<T*>(out + s * ld_out + cv * V) = acc;
  }
}

// ------------------------------------------------------------------ two segment sums of the same rows in one pass
// out_a[s] / out_b[s] = sum of the kept rows of segment s of list A / list B (the row and the column CSR of a plan: every
// entry row sits in exactly one segment of each list).  As two launches of k_seg_sum_rows the source crossed HBM twice --
// the second launch starts after the first has streamed the whole tensor through the 256 MB last-level cache, so nothing
// of it is left.  Here the thread of (node s, column vector) walks segment s of A and then segment s of B: the two reads of
// a row are made by the threads of two nodes of the SAME graph, a few workgroups and a few microseconds apart, and the
// second one is served from the last-level cache (the rows in flight over the whole chip are ~100 MB).  Same walk, same
// fixed order per segment: bitwise the results of the two separate calls.
template <int V>
__device__ __forceinline__ typename VecT<V>::type seg_walk(const float* __restrict__ src, int64_t ld_src, int64_t lo, int64_t hi,
                                                          const int32_t* __restrict__ perm, const uint8_t* __restrict__ kind,
                                                          unsigned keep, int64_t col) {
  using T = typename VecT<V>::type;
  T acc = vzero(T());
  for (int64_t p = lo; p < hi; p += SU) {
    int64_t r[SU];
    T v[SU];
    int pr[SU], kk[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int64_t pp = p + u < hi ? p + u : hi - 1;
      pr[u] = perm[pp];
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      vtouch(pr[u]);
      r[u] = (int64_t)pr[u];
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) kk[u] = (int)kind[r[u]];
#pragma unroll
    for (int u = 0; u < SU; ++u) v[u] = *reinterpret_cast<const T*>(src + r[u] * ld_src + col);
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      vtouch(kk[u]);
      vtouch(v[u]);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u)
      if ((p + u < hi) && hotk_keep(keep, (uint8_t)kk[u])) acc = vadd(acc, v[u]);
  }
  return acc;
}

template <int V>
__global__ void k_seg_sum_rows_pair(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ a_begin,
                                    const int32_t* __restrict__ a_end, const int32_t* __restrict__ a_perm,
                                    const int32_t* __restrict__ b_begin, const int32_t* __restrict__ b_end,
                                    const int32_t* __restrict__ b_perm, const uint8_t* __restrict__ kind, unsigned keep,
                                    float* __restrict__ out_a, float* __restrict__ out_b, int64_t ld_out, int64_t n_seg,
                                    uint32_t CV, const int32_t* __restrict__ own_row, float* __restrict__ out_own) {
  using T = typename VecT<V>::type;
  const int64_t total = n_seg * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t s, cv;
    if (total <= 0xffffffffll) {
      const uint32_t su = (uint32_t)t / CV;
      s = su;
      cv = (uint32_t)t - su * CV;
    } else {
      s = t / CV;
      cv = t - s * CV;
    }
    const T va = seg_walk<V>(src, ld_src, a_begin[s], a_end[s], a_perm, kind, keep, cv * V);
    *reinterpret_cast<T*>(out_a + s * ld_out + cv * V) = va;
    const T vb = seg_walk<V>(src, ld_src, b_begin[s], b_end[s], b_perm, kind, keep, cv * V);
    *reinterpret_cast<T*>(out_b + s * ld_out + cv * V) = vb;
    if (own_row) {      // (hotk_seg_sum_rows_pair_own: the segment's own source row too, e.g. a node's diagonal entry)
      const int64_t r = own_row[s];
      T v = vzero(T());
      if (r >= 0) v = *reinterpret_cast<const T*>(src + r * ld_src + cv * V);
      *reinterpret_cast<T*>(out_own + s * ld_out + cv * V) = v;
    }
  }
}

// ------------------------------------------------------------------ bias + mask (in place)
template <int V>
__global__ void k_bias_mask_rows(float* __restrict__ y, int64_t ld, const uint8_t* __restrict__ kind,
                                 const float* __restrict__ b_off, const float* __restrict__ b_diag, int64_t n,
                                 int64_t CV) {
  using T = typename VecT<V>::type;
  const int64_t total = n * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV, c = (t - r * CV) * V;
    uint8_t kd = kind[r];
    T* p = reinterpret_cast<T*>(y + r * ld + c);
    T v = vzero(T());
    if (kd != 0) {
      v = *p;
      const float* b = (kd == 2) ? b_off : b_diag;
      if (b) v = vadd(v, *reinterpret_cast<const T*>(b + c));
    }
    *p = v;
  }
}

// bias gradients: column sums of dy over the rows of each kind, in two passes without atomics (bitwise reproducible):
// pass 1 leaves one partial row pair per block (part[block][0 = off-diagonal | 1 = diagonal][C]), pass 2 adds the
// partials of a channel in block order.
__global__ __launch_bounds__(256) void k_bias_grad_rows(const float* __restrict__ dy, int64_t ld,
                                                        const uint8_t* __restrict__ kind, float* __restrict__ part,
                                                        int64_t n, int64_t C, int64_t rows_per_block) {
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float* o = part + (int64_t)blockIdx.x * 2 * C;
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
    float a_off = 0.f, a_diag = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      uint8_t kd = kind[r];
      if (kd == 0) continue;
      float v = dy[r * ld + c];
      if (kd == 2) a_off += v; else a_diag += v;
    }
    o[c] = a_off;
    o[C + c] = a_diag;
  }
}

// float4 variant: thread = (channel vector cv, row lane ty); TY rows are read in parallel, each row as whole
// contiguous pieces; the TY partial sums meet in LDS (added in lane order) and leave as the block's partial row pair
__global__ __launch_bounds__(256) void k_bias_grad_rows4(const float* __restrict__ dy, int64_t ld,
                                                         const uint8_t* __restrict__ kind, float* __restrict__ part,
                                                         int64_t n, int C4, int TY, int64_t rows_per_block) {
  __shared__ float4 red[2][256];
  const int cv = threadIdx.x % C4, ty = threadIdx.x / C4;
  float4 a_off = make_float4(0.f, 0.f, 0.f, 0.f), a_diag = a_off;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  if (ty < TY) {
    // four rows per step so that four loads are in flight (a short strip is latency bound otherwise)
    for (int64_t r = r0 + ty; r < r1; r += 4 * TY) {
      uint8_t kd[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t rr = r + (int64_t)u * TY;
        kd[u] = rr < r1 ? kind[rr] : (uint8_t)0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (kd[u]) v[u] = *reinterpret_cast<const float4*>(dy + (r + (int64_t)u * TY) * ld + 4 * cv);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kd[u] == 2) a_off = vadd(a_off, v[u]);
        else if (kd[u] == 1) a_diag = vadd(a_diag, v[u]);
      }
    }
  }
  red[0][threadIdx.x] = a_off;
  red[1][threadIdx.x] = a_diag;
  __syncthreads();
  if (ty == 0) {
    for (int k = 1; k < TY; ++k) {
      a_off = vadd(a_off, red[0][k * C4 + cv]);
      a_diag = vadd(a_diag, red[1][k * C4 + cv]);
    }
    float* o = part + (int64_t)blockIdx.x * 2 * 4 * C4;
    *reinterpret_cast<float4*>(o + 4 * cv) = a_off;
    *reinterpret_cast<float4*>(o + 4 * C4 + 4 * cv) = a_diag;
  }
}

// pass 2: a block per 16 channels of one kind, thread = (channel, slice): slice s adds the partial rows s, s + 16, ...
// in order (eight loads in flight), then the 16 slice sums are added in slice order -- a fixed order, bitwise
// reproducible.  (One thread per channel walking all ~4 000 partial rows took 0.23 ms per call: 12 waves on the chip.)
constexpr int BGC_SLICES = 16;
__global__ __launch_bounds__(256) void k_bias_grad_combine(const float* __restrict__ part, int64_t nb, int64_t C,
                                                           float* __restrict__ db_off, float* __restrict__ db_diag) {
  __shared__ float red[BGC_SLICES][16];
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t t = (int64_t)blockIdx.x * 16 + cx;       // channel index in [0, 2 C): off-diagonal sums, then diagonal
  const bool live = t < 2 * C;
  float acc = 0.f;
  if (live) {
    int64_t k = sl;
    for (; k + 7 * BGC_SLICES < nb; k += 8 * BGC_SLICES) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(k + u * BGC_SLICES) * 2 * C + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < nb; k += BGC_SLICES) acc += part[k * 2 * C + t];
  }
  red[sl][cx] = acc;
  __syncthreads();
  if (sl == 0 && live) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < BGC_SLICES; ++q) tot += red[q][cx];
    float* dst = t < C ? db_off : db_diag;
    if (dst) dst[t < C ? t : t - C] = tot;
  }
}

// ------------------------------------------------------------------ order-2 'light' basis
template <int V>
__global__ void k_basis22_light_fwd(const float* __restrict__ x, int64_t ld_x, const uint8_t* __restrict__ kind,
                                    const int32_t* __restrict__ rown, const int32_t* __restrict__ coln,
                                    const int32_t* __restrict__ tpos, const int32_t* __restrict__ diag_pos,
                                    float* __restrict__ x5, int64_t ld_x5, int64_t R, int64_t D) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / DV, c = (t - r * DV) * V;
    uint8_t kd = kind[r];
    T z = vzero(T());
    T v1 = z, v2 = z, v3 = z, v4 = z, v5 = z;
    if (kd == 1) {
      v1 = *reinterpret_cast<const T*>(x + r * ld_x + c);
    } else if (kd == 2) {
      v2 = *reinterpret_cast<const T*>(x + r * ld_x + c);
      int32_t tp = tpos[r];
      if (tp >= 0) v3 = *reinterpret_cast<const T*>(x + (int64_t)tp * ld_x + c);
      v4 = *reinterpret_cast<const T*>(x + (int64_t)diag_pos[rown[r]] * ld_x + c);
      v5 = *reinterpret_cast<const T*>(x + (int64_t)diag_pos[coln[r]] * ld_x + c);
    }
    float* o = x5 + r * ld_x5 + c;
    *reinterpret_cast<T*>(o) = v1;
    *reinterpret_cast<T*>(o + D) = v2;
    *reinterpret_cast<T*>(o + 2 * D) = v3;
    *reinterpret_cast<T*>(o + 3 * D) = v4;
    *reinterpret_cast<T*>(o + 4 * D) = v5;
  }
}

template <int V>
__global__ void k_basis22_light_bwd(const float* __restrict__ g, int64_t ld_g, const uint8_t* __restrict__ kind,
                                    const int32_t* __restrict__ rown, const int32_t* __restrict__ tpos,
                                    const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ row_perm,
                                    const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ col_perm,
                                    float* __restrict__ dx, int64_t ld_dx, int64_t R, int64_t D) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / DV, c = (t - r * DV) * V;
    uint8_t kd = kind[r];
    T acc = vzero(T());
    if (kd == 2) {
      acc = *reinterpret_cast<const T*>(g + r * ld_g + D + c);
      int32_t tp = tpos[r];  // x_r is the transposed input of row tp
      if (tp >= 0) acc = vadd(acc, *reinterpret_cast<const T*>(g + (int64_t)tp * ld_g + 2 * D + c));
    } else if (kd == 1) {
      acc = *reinterpret_cast<const T*>(g + r * ld_g + c);
      int32_t nd = rown[r];
      for (int32_t p = row_ptr[nd]; p < row_ptr[nd + 1]; ++p) {  // x_(i,i) broadcast along row i
        int32_t e = row_perm[p];
        if (kind[e] == 2) acc = vadd(acc, *reinterpret_cast<const T*>(g + (int64_t)e * ld_g + 3 * D + c));
      }
      for (int32_t p = col_ptr[nd]; p < col_ptr[nd + 1]; ++p) {  // x_(j,j) broadcast along column j
        int32_t e = col_perm[p];
        if (kind[e] == 2) acc = vadd(acc, *reinterpret_cast<const T*>(g + (int64_t)e * ld_g + 4 * D + c));
      }
    }
    *reinterpret_cast<T*>(dx + r * ld_dx + c) = acc;
  }
}

// ------------------------------------------------------------------ order-2 'light' basis, factored form
// The five-block basis [diag? x | off? x | off&tau? x_t | off? x_(i,i) | off? x_(j,j)] @ W is linear in its blocks, so
// the three node-valued blocks are projected once per NODE ([Nn,D] @ [W1|W4|W5]) and only the two entry-valued
// blocks go through the big GEMM:  x2[r] = off-diagonal ? [ x_r | tau ? x_t(r) : 0 ] : 0   ([R, 2D]).
template <int V>
__global__ void k_basis22_pair_fwd(const float* __restrict__ x, int64_t ld_x, const uint8_t* __restrict__ kind,
                                   const int32_t* __restrict__ tpos, float* __restrict__ x2, int64_t ld_x2, int64_t R,
                                   int64_t D, const int32_t* __restrict__ diag_pos, float* __restrict__ xd, int64_t ld_xd,
                                   int64_t Nn) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV, total_n = total + (diag_pos ? Nn * DV : 0);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_n; t += (int64_t)gridDim.x * blockDim.x) {
    if (t >= total) {      // (hotk_basis22_pair_fwd_diag: xd[n] = x on node n's diagonal entry, zeros for a padding node)
      const int64_t n = (t - total) / DV, c = (t - total - n * DV) * V;
      const int64_t s = diag_pos[n];
      T v = vzero(T());
      if (s >= 0) v = *reinterpret_cast<const T*>(x + s * ld_x + c);
      *reinterpret_cast<T*>(xd + n * ld_xd + c) = v;
      continue;
    }
    int64_t r = t / DV, c = (t - r * DV) * V;
    T a = vzero(T()), b = a;
    if (kind[r] == 2) {
      a = *reinterpret_cast<const T*>(x + r * ld_x + c);
      int32_t tp = tpos[r];
      if (tp >= 0) b = *reinterpret_cast<const T*>(x + (int64_t)tp * ld_x + c);
    }
    float* o = x2 + r * ld_x2 + c;
    *reinterpret_cast<T*>(o) = a;
    *reinterpret_cast<T*>(o + D) = b;
  }
}

// transpose map of k_basis22_pair_fwd plus the node-level gradient on the diagonal rows; add (optional, [R, D]): another
// gradient of the same input, added on valid rows (one pass instead of a separate elementwise sum over [R, D])
template <int V>
__global__ void k_basis22_pair_bwd(const float* __restrict__ g, int64_t ld_g, const uint8_t* __restrict__ kind,
                                   const int32_t* __restrict__ tpos, const int32_t* __restrict__ rown,
                                   const float* __restrict__ gd, int64_t ld_gd, const float* __restrict__ add,
                                   int64_t ld_add, float* __restrict__ dx, int64_t ld_dx, int64_t R, int64_t D) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / DV, c = (t - r * DV) * V;
    uint8_t kd = kind[r];
    T acc = vzero(T());
    if (kd == 2) {
      acc = *reinterpret_cast<const T*>(g + r * ld_g + c);
      int32_t tp = tpos[r];  // x_r is the transposed input of row tp
      if (tp >= 0) acc = vadd(acc, *reinterpret_cast<const T*>(g + (int64_t)tp * ld_g + D + c));
    } else if (kd == 1) {
      acc = *reinterpret_cast<const T*>(gd + (int64_t)rown[r] * ld_gd + c);
    }
    if (add && kd) acc = vadd(acc, *reinterpret_cast<const T*>(add + r * ld_add + c));
    *reinterpret_cast<T*>(dx + r * ld_dx + c) = acc;
  }
}

// y[r] (in place on the big GEMM's output z):  off-diagonal (i,j): z + yn[i][C..2C) + yn[j][2C..3C) + b_off
//                                               diagonal (v,v)   : yn[v][0..C) + b_diag          padding: 0
template <int V>
__global__ void k_assemble22(float* __restrict__ z, int64_t ld, const float* __restrict__ yn, int64_t ld_yn,
                             const uint8_t* __restrict__ kind, const int32_t* __restrict__ rown,
                             const int32_t* __restrict__ coln, const float* __restrict__ b_off,
                             const float* __restrict__ b_diag, int64_t R, int64_t C) {
  using T = typename VecT<V>::type;
  const int64_t CV = C / V, total = R * CV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / CV, c = (t - r * CV) * V;
    uint8_t kd = kind[r];
    T* p = reinterpret_cast<T*>(z + r * ld + c);
    T v = vzero(T());
    if (kd == 2) {
      v = *p;
      v = vadd(v, *reinterpret_cast<const T*>(yn + (int64_t)rown[r] * ld_yn + C + c));
      v = vadd(v, *reinterpret_cast<const T*>(yn + (int64_t)coln[r] * ld_yn + 2 * C + c));
      if (b_off) v = vadd(v, *reinterpret_cast<const T*>(b_off + c));
    } else if (kd == 1) {
      v = *reinterpret_cast<const T*>(yn + (int64_t)rown[r] * ld_yn + c);
      if (b_diag) v = vadd(v, *reinterpret_cast<const T*>(b_diag + c));
    }
    *p = v;
  }
}

inline bool vec4_ok(int64_t C, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  if (C % 4) return false;
  for (int64_t l : lds) if (l % 4) return false;
  for (const void* p : ptrs) if (p && !hotk_aligned16(p)) return false;
  return true;
}

}  // namespace

extern "C" const char* hotk_strerror(int code) {
  switch (code) {
    case HOTK_OK: return "ok";
    case HOTK_EINVAL: return "invalid argument";
    case HOTK_EWORKSPACE: return "workspace too small";
    case HOTK_ELAUNCH: return "HIP launch/runtime error";
    case HOTK_EUNSUPPORTED: return "unsupported configuration";
    case HOTK_ETOOLARGE: return "problem exceeds the fixed scratch capacity of this *_static entry point (use the workspace form)";
    default: return "unknown error";
  }
}

extern "C" int hotk_version(void) { return 100; }

extern "C" int hotk_gather_rows(const float* src, int64_t ld_src, const int32_t* idx, float* out, int64_t ld_out,
                                int64_t n_out, int64_t C, int accumulate, void* stream) {
  HOTK_CHECK_ARG(n_out >= 0 && C >= 0 && ld_src >= C && ld_out >= C);
  if (n_out == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(src && out);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(C, {ld_src, ld_out}, {src, out})) {
    hipLaunchKernelGGL(k_gather_rows<4>, dim3(hotk_grid(n_out * (C / 4), 256)), dim3(256), 0, st, src, ld_src, idx, out,
                       ld_out, n_out, C / 4, accumulate);
  } else {
    hipLaunchKernelGGL(k_gather_rows<1>, dim3(hotk_grid(n_out * C, 256)), dim3(256), 0, st, src, ld_src, idx, out,
                       ld_out, n_out, C, accumulate);
  }
  return hotk_launch_status();
}

extern "C" int hotk_mask_rows(const float* src, int64_t ld_src, const uint8_t* kind, unsigned keep, float fill,
                              float* out, int64_t ld_out, int64_t n, int64_t C, void* stream) {
  HOTK_CHECK_ARG(n >= 0 && C >= 0 && ld_src >= C && ld_out >= C);
  if (n == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(src && out && kind);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(C, {ld_src, ld_out}, {src, out})) {
    hipLaunchKernelGGL(k_mask_rows<4>, dim3(hotk_grid(n * (C / 4), 256)), dim3(256), 0, st, src, ld_src, kind, keep,
                       fill, out, ld_out, n, C / 4);
  } else {
    hipLaunchKernelGGL(k_mask_rows<1>, dim3(hotk_grid(n * C, 256)), dim3(256), 0, st, src, ld_src, kind, keep, fill,
                       out, ld_out, n, C);
  }
  return hotk_launch_status();
}

extern "C" int hotk_seg_sum_rows(const float* src, int64_t ld_src, const int32_t* seg_begin, const int32_t* seg_end,
                                 const int32_t* perm, const uint8_t* kind, unsigned keep, float* out, int64_t ld_out,
                                 int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream) {
  HOTK_CHECK_ARG(n_seg >= 0 && C >= 0 && ld_src >= C && ld_out >= C && max_seg_len >= 0);
  if (n_seg == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(src && out && seg_begin && seg_end);
  hipStream_t st = (hipStream_t)stream;
  // one thread walks a whole segment in order (bitwise reproducible); longer segments are split by the caller into
  // consecutive pieces whose partial rows are summed by a second call (hot_amd.ops.refine_seg), never with atomics
  if (max_seg_len > HOTK_SEG_CHUNK) return HOTK_EUNSUPPORTED;
  const int64_t n_chunk = 1;
  const bool v4 = vec4_ok(C, {ld_src, ld_out}, {src, out});
  const int64_t CV = v4 ? C / 4 : C;
  const dim3 grid(hotk_grid(n_seg * n_chunk * CV, 256)), block(256);
#define HOTK_SEGSUM_LAUNCH(VV, PP, KK)                                                                              \
  hipLaunchKernelGGL((k_seg_sum_rows<VV, PP, KK>), grid, block, 0, st, src, ld_src, seg_begin, seg_end, perm, kind, \
                     keep, out, ld_out, n_seg, CV, n_chunk)
#define HOTK_SEGSUM_PK(VV)                                                    \
  do {                                                                        \
    if (perm) {                                                               \
      if (kind) HOTK_SEGSUM_LAUNCH(VV, true, true); else HOTK_SEGSUM_LAUNCH(VV, true, false);   \
    } else {                                                                  \
      if (kind) HOTK_SEGSUM_LAUNCH(VV, false, true); else HOTK_SEGSUM_LAUNCH(VV, false, false); \
    }                                                                         \
  } while (0)
  if (v4) HOTK_SEGSUM_PK(4); else HOTK_SEGSUM_PK(1);
#undef HOTK_SEGSUM_PK
#undef HOTK_SEGSUM_LAUNCH
  return hotk_launch_status();
}

extern "C" int hotk_seg_sum_rows_pair_own(const float* src, int64_t ld_src, const int32_t* a_begin, const int32_t* a_end,
                                          const int32_t* a_perm, const int32_t* b_begin, const int32_t* b_end,
                                          const int32_t* b_perm, const uint8_t* kind, unsigned keep, float* out_a,
                                          float* out_b, const int32_t* own_row, float* out_own, int64_t ld_out,
                                          int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream) {
  HOTK_CHECK_ARG(n_seg >= 0 && C >= 0 && ld_src >= C && ld_out >= C && max_seg_len >= 0 && (!own_row == !out_own));
  if (n_seg == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(src && out_a && out_b && a_begin && a_end && a_perm && b_begin && b_end && b_perm && kind);
  if (max_seg_len > HOTK_SEG_CHUNK) return HOTK_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = vec4_ok(C, {ld_src, ld_out}, {src, out_a, out_b}) && (!out_own || hotk_aligned16(out_own));
  const int64_t CV = v4 ? C / 4 : C;
  const dim3 grid(hotk_grid(n_seg * CV, 256)), block(256);
  if (v4)
    hipLaunchKernelGGL(k_seg_sum_rows_pair<4>, grid, block, 0, st, src, ld_src, a_begin, a_end, a_perm, b_begin, b_end, b_perm,
                       kind, keep, out_a, out_b, ld_out, n_seg, (uint32_t)CV, own_row, out_own);
  else
    hipLaunchKernelGGL(k_seg_sum_rows_pair<1>, grid, block, 0, st, src, ld_src, a_begin, a_end, a_perm, b_begin, b_end, b_perm,
                       kind, keep, out_a, out_b, ld_out, n_seg, (uint32_t)CV, own_row, out_own);
  return hotk_launch_status();
}

extern "C" int hotk_seg_sum_rows_pair(const float* src, int64_t ld_src, const int32_t* a_begin, const int32_t* a_end,
                                      const int32_t* a_perm, const int32_t* b_begin, const int32_t* b_end,
                                      const int32_t* b_perm, const uint8_t* kind, unsigned keep, float* out_a, float* out_b,
                                      int64_t ld_out, int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream) {
  return hotk_seg_sum_rows_pair_own(src, ld_src, a_begin, a_end, a_perm, b_begin, b_end, b_perm, kind, keep, out_a, out_b,
                                    nullptr, nullptr, ld_out, n_seg, C, max_seg_len, stream);
}

extern "C" int hotk_bias_mask_rows(float* y, int64_t ld, const uint8_t* kind, const float* b_off,
                                   const float* b_diag, int64_t n, int64_t C, void* stream) {
  HOTK_CHECK_ARG(n >= 0 && C >= 0 && ld >= C);
  if (n == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(y && kind);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(C, {ld}, {y, b_off, b_diag})) {
    hipLaunchKernelGGL(k_bias_mask_rows<4>, dim3(hotk_grid(n * (C / 4), 256)), dim3(256), 0, st, y, ld, kind, b_off,
                       b_diag, n, C / 4);
  } else {
    hipLaunchKernelGGL(k_bias_mask_rows<1>, dim3(hotk_grid(n * C, 256)), dim3(256), 0, st, y, ld, kind, b_off, b_diag,
                       n, C);
  }
  return hotk_launch_status();
}

namespace {
// pass-1 geometry of the bias-gradient sums (shared by the workspace query and the launch)
struct BiasGradGeom { bool vec4; int C4, TY; int64_t nb, rpb; };
BiasGradGeom bias_grad_geom(const float* dy, int64_t ld, int64_t n, int64_t C) {
  BiasGradGeom g{};
  g.vec4 = C % 4 == 0 && C / 4 <= 256 && ld % 4 == 0 && hotk_aligned16(dy);
  if (g.vec4) {
    g.C4 = (int)(C / 4);
    g.TY = 256 / g.C4;
    g.nb = hotk_grid(n, 16 * g.TY, 4096);   // >= 16 rows per thread unless that needs more than 4096 blocks
  } else {
    g.nb = hotk_grid(n, 64, 2048);          // >= 64 rows per block
  }
  g.rpb = (n + g.nb - 1) / g.nb;
  g.nb = (n + g.rpb - 1) / g.rpb;
  return g;
}
}  // namespace

extern "C" int hotk_bias_grad_workspace_bytes(int64_t n, int64_t C, size_t* bytes) {
  HOTK_CHECK_ARG(bytes && n >= 0 && C >= 0);
  // an upper bound over both geometries: at most 4096 blocks, each leaving 2 x C floats
  int64_t nb = n < 4096 ? (n > 0 ? n : 1) : 4096;
  *bytes = (size_t)nb * 2 * (size_t)C * sizeof(float);
  return HOTK_OK;
}

extern "C" int hotk_bias_grad_rows(const float* dy, int64_t ld, const uint8_t* kind, float* db_off, float* db_diag,
                                   int64_t n, int64_t C, void* workspace, size_t ws_bytes, void* stream) {
  HOTK_CHECK_ARG(n >= 0 && C >= 0 && ld >= C);
  if (C == 0 || (!db_off && !db_diag)) return HOTK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipLaunchKernelGGL(k_bias_grad_combine, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, st, (const float*)nullptr,
                       (int64_t)0, C, db_off, db_diag);
    return hotk_launch_status();
  }
  HOTK_CHECK_ARG(dy && kind);
  const BiasGradGeom g = bias_grad_geom(dy, ld, n, C);
  if (!workspace || ws_bytes < (size_t)g.nb * 2 * (size_t)C * sizeof(float)) return HOTK_EWORKSPACE;
  float* part = (float*)workspace;
  if (g.vec4)
    hipLaunchKernelGGL(k_bias_grad_rows4, dim3((unsigned)g.nb), dim3(256), 0, st, dy, ld, kind, part, n, g.C4, g.TY, g.rpb);
  else
    hipLaunchKernelGGL(k_bias_grad_rows, dim3((unsigned)g.nb), dim3(256), 0, st, dy, ld, kind, part, n, C, g.rpb);
  hipLaunchKernelGGL(k_bias_grad_combine, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, st, part, g.nb, C, db_off,
                     db_diag);
  return hotk_launch_status();
}

extern "C" int hotk_basis22_light_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* rown,
                                      const int32_t* coln, const int32_t* tpos, const int32_t* diag_pos, float* x5,
                                      int64_t ld_x5, int64_t R, int64_t D, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D >= 0 && ld_x >= D && ld_x5 >= 5 * D);
  if (R == 0 || D == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && kind && rown && coln && tpos && diag_pos && x5);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(D, {ld_x, ld_x5}, {x, x5})) {
    hipLaunchKernelGGL(k_basis22_light_fwd<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, x, ld_x, kind, rown,
                       coln, tpos, diag_pos, x5, ld_x5, R, D);
  } else {
    hipLaunchKernelGGL(k_basis22_light_fwd<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, x, ld_x, kind, rown, coln,
                       tpos, diag_pos, x5, ld_x5, R, D);
  }
  return hotk_launch_status();
}

extern "C" int hotk_basis22_light_bwd(const float* dx5, int64_t ld_dx5, const uint8_t* kind, const int32_t* rown,
                                      const int32_t* tpos, const int32_t* row_ptr, const int32_t* row_perm,
                                      const int32_t* col_ptr, const int32_t* col_perm, float* dx, int64_t ld_dx,
                                      int64_t R, int64_t D, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D >= 0 && ld_dx >= D && ld_dx5 >= 5 * D);
  if (R == 0 || D == 0) return HOTK_OK;
  HOTK_CHECK_ARG(dx5 && kind && rown && tpos && row_ptr && row_perm && col_ptr && col_perm && dx);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(D, {ld_dx5, ld_dx}, {dx5, dx})) {
    hipLaunchKernelGGL(k_basis22_light_bwd<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, dx5, ld_dx5, kind,
                       rown, tpos, row_ptr, row_perm, col_ptr, col_perm, dx, ld_dx, R, D);
  } else {
    hipLaunchKernelGGL(k_basis22_light_bwd<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, dx5, ld_dx5, kind, rown,
                       tpos, row_ptr, row_perm, col_ptr, col_perm, dx, ld_dx, R, D);
  }
  return hotk_launch_status();
}

extern "C" int hotk_basis22_pair_fwd_diag(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* tpos, float* x2,
                                          int64_t ld_x2, int64_t R, int64_t D, const int32_t* diag_pos, float* xd,
                                          int64_t ld_xd, int64_t Nn, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D >= 0 && ld_x >= D && ld_x2 >= 2 * D && (!diag_pos == !xd) && (!diag_pos || (Nn >= 0 && ld_xd >= D)));
  if (!diag_pos) Nn = 0;
  if ((R == 0 && Nn == 0) || D == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && kind && tpos && x2);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(D, {ld_x, ld_x2}, {x, x2}) && (!xd || (hotk_aligned16(xd) && (ld_xd & 3) == 0))) {
    hipLaunchKernelGGL(k_basis22_pair_fwd<4>, dim3(hotk_grid((R + Nn) * (D / 4), 256)), dim3(256), 0, st, x, ld_x, kind, tpos,
                       x2, ld_x2, R, D, diag_pos, xd, ld_xd, Nn);
  } else {
    hipLaunchKernelGGL(k_basis22_pair_fwd<1>, dim3(hotk_grid((R + Nn) * D, 256)), dim3(256), 0, st, x, ld_x, kind, tpos, x2,
                       ld_x2, R, D, diag_pos, xd, ld_xd, Nn);
  }
  return hotk_launch_status();
}

extern "C" int hotk_basis22_pair_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* tpos, float* x2,
                                     int64_t ld_x2, int64_t R, int64_t D, void* stream) {
  return hotk_basis22_pair_fwd_diag(x, ld_x, kind, tpos, x2, ld_x2, R, D, nullptr, nullptr, 0, 0, stream);
}

extern "C" int hotk_basis22_pair_bwd(const float* dx2, int64_t ld_dx2, const uint8_t* kind, const int32_t* tpos,
                                     const int32_t* rown, const float* dxd, int64_t ld_dxd, const float* add,
                                     int64_t ld_add, float* dx, int64_t ld_dx, int64_t R, int64_t D, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D >= 0 && ld_dx >= D && ld_dx2 >= 2 * D && ld_dxd >= D && (!add || ld_add >= D));
  if (R == 0 || D == 0) return HOTK_OK;
  HOTK_CHECK_ARG(dx2 && kind && tpos && rown && dxd && dx);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(D, {ld_dx2, ld_dxd, ld_dx, add ? ld_add : 0}, {dx2, dxd, dx, add})) {
    hipLaunchKernelGGL(k_basis22_pair_bwd<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, dx2, ld_dx2, kind, tpos,
                       rown, dxd, ld_dxd, add, ld_add, dx, ld_dx, R, D);
  } else {
    hipLaunchKernelGGL(k_basis22_pair_bwd<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, dx2, ld_dx2, kind, tpos, rown,
                       dxd, ld_dxd, add, ld_add, dx, ld_dx, R, D);
  }
  return hotk_launch_status();
}

extern "C" int hotk_assemble22(float* z, int64_t ld, const float* yn, int64_t ld_yn, const uint8_t* kind,
                               const int32_t* rown, const int32_t* coln, const float* b_off, const float* b_diag,
                               int64_t R, int64_t C, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && C >= 0 && ld >= C && ld_yn >= 3 * C);
  if (R == 0 || C == 0) return HOTK_OK;
  HOTK_CHECK_ARG(z && yn && kind && rown && coln);
  hipStream_t st = (hipStream_t)stream;
  if (vec4_ok(C, {ld, ld_yn}, {z, yn, b_off, b_diag})) {
    hipLaunchKernelGGL(k_assemble22<4>, dim3(hotk_grid(R * (C / 4), 256)), dim3(256), 0, st, z, ld, yn, ld_yn, kind, rown,
                       coln, b_off, b_diag, R, C);
  } else {
    hipLaunchKernelGGL(k_assemble22<1>, dim3(hotk_grid(R * C, 256)), dim3(256), 0, st, z, ld, yn, ld_yn, kind, rown, coln,
                       b_off, b_diag, R, C);
  }
  return hotk_launch_status();
}
