// Fused attention core of the 2->2 sparse kernel-attention layer for the headline shape
// (4 heads x 16 random features = one 64-lane wavefront per slot; head width 8, value width 8,
// generalized / ReLU FAVOR+ map).  Reference: models/sparse/kernelattn.py:119-159 calling
// models/sparse/kernelattncoef.py:74-234 ten times; here the ten terms are produced by
//
//   graph sums   : one wave per row chunk keeps the four per-graph  sum phi(K) (x) [V|1]  in registers
//   node segments: one wave per node j keeps the six per-node       sum_i phi(K_i) (x) [V_ij|1]  (36+18 floats
//                  per lane) in registers while it walks column segment j, and immediately applies them to the
//                  node's own query, to the entries of column j and to the entries of row j.  No key-value
//                  tensor, no phi(Q)/phi(K) tensor and no transposed value copy ever reaches HBM.
//
// Lane l = (head h = l>>4, feature mm = l&15).  A lane computes phi for its (h,mm) from the raw 8-wide head
// slice (wave-broadcast 16-B loads), products with the 8-wide value / gradient slice of head h, and the
// 16-lane sums with xor-shuffles that stay inside the head (shuffle-transpose reduction: 8 results in 8 steps).
#include "common.h"

namespace {

constexpr int NH = 4, NM = 16, DH = 8, NC = 8;  // heads, features, head width, value width per head
constexpr int SLOT = 32;                        // floats per dim_qk / dim_v slot (NH*DH == NH*NC == 32)
constexpr float PHI_EPS = 1e-3f;

using A = hotk_attn22_args;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct Lane {
  int lane, h, mm;
  float ps[DH];  // P[mm][:] * d^-1/4
};

__device__ __forceinline__ Lane make_lane(const float* __restrict__ proj) {
  Lane L;
  L.lane = threadIdx.x & 63;
  L.h = L.lane >> 4;
  L.mm = L.lane & 15;
  const float scale = 0.59460355750136051f;  // 8^-1/4
#pragma unroll
  for (int j = 0; j < DH; ++j) L.ps[j] = proj[L.mm * DH + j] * scale;
  return L;
}

// pre-activation of feature (h,mm) for the raw 32-wide slot at x
__device__ __forceinline__ float pre_of(const Lane& L, const float* __restrict__ x) {
  float4 a = ld4(x + L.h * DH), b = ld4(x + L.h * DH + 4);
  float s = a.x * L.ps[0];
  s = fmaf(a.y, L.ps[1], s); s = fmaf(a.z, L.ps[2], s); s = fmaf(a.w, L.ps[3], s);
  s = fmaf(b.x, L.ps[4], s); s = fmaf(b.y, L.ps[5], s); s = fmaf(b.z, L.ps[6], s); s = fmaf(b.w, L.ps[7], s);
  return s;
}
__device__ __forceinline__ float phi_of(float pre) { return fmaxf(pre, 0.f) + PHI_EPS; }

// load the 8-wide slice of head h of a 32-wide slot (zeros if p == nullptr)
__device__ __forceinline__ void ld8(const Lane& L, const float* __restrict__ p, float (&o)[NC]) {
  if (p) {
    float4 a = ld4(p + L.h * NC), b = ld4(p + L.h * NC + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = 0.f;
  }
}

// ---- 16-lane (one DPP row = one head) reductions on the VALU: no LDS round trips ---------------------------
// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_mirror = 0x140, row_half_mirror = 0x141
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sum over the 16 lanes of a head, result in every lane
__device__ __forceinline__ float head_allsum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}

// element index held by lane mm after head_sum8, and whether this lane is the one that stores it
__device__ __forceinline__ int sum8_idx(int mm) { return 4 * (mm >> 3) + 2 * ((mm >> 2) & 1) + (mm & 1); }
__device__ __forceinline__ bool sum8_owner(int mm) { return (mm & 2) == 0; }

// t[0..7] per lane -> sum over the head's 16 lanes of t[sum8_idx(mm)] (shuffle-transpose reduction: each step
// halves the number of live values while adding the mirrored partner's half)
__device__ __forceinline__ float head_sum8(const Lane& L, const float (&t)[8]) {
  const bool hi8 = L.mm & 8, hi4 = L.mm & 4, odd = L.mm & 1;
  float w[4], u[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float give = hi8 ? t[k] : t[k + 4];
    const float keep = hi8 ? t[k + 4] : t[k];
    w[k] = keep + dpp<0x140>(give);          // partner 15 - mm
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float give = hi4 ? w[k] : w[k + 2];
    const float keep = hi4 ? w[k + 2] : w[k];
    u[k] = keep + dpp<0x141>(give);          // partner 7 - (mm & 7) within the half
  }
  const float give = odd ? u[0] : u[1];
  const float keep = odd ? u[1] : u[0];
  float r = keep + dpp<0xB1>(give);          // partner mm ^ 1
  r += dpp<0x4E>(r);                         // partner mm ^ 2
  return r;
}

// store the head_sum8 result: slot base p (32 floats), element h*8 + sum8_idx(mm)
__device__ __forceinline__ void st_sum8(const Lane& L, float* __restrict__ p, float r) {
  if (sum8_owner(L.mm)) p[L.h * 8 + sum8_idx(L.mm)] = r;
}

__device__ __forceinline__ void zero_slots(const Lane& L, float* __restrict__ p, int nslots) {
  for (int i = L.lane; i < nslots * SLOT; i += 64) p[i] = 0.f;
}

// KV accumulate: kv[0..7] += phi * val, kv[8] += phi
__device__ __forceinline__ void kv_acc(float (&kv)[NC + 1], float phi, const float (&val)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) kv[c] = fmaf(phi, val[c], kv[c]);
  kv[NC] += phi;
}

// O = phi(q).KV / guard(phi(q).ks) for one slot; writes 32 floats at out
__device__ __forceinline__ void apply_fwd(const Lane& L, const float* __restrict__ q, const float (&kv)[NC + 1],
                                          float* __restrict__ out) {
  const float phi = phi_of(pre_of(L, q));
  float t[8];
#pragma unroll
  for (int c = 0; c < NC; ++c) t[c] = phi * kv[c];
  float den = head_allsum(phi * kv[NC]);
  if (den == 0.f) den = 1e-5f;
  float r = head_sum8(L, t);
  st_sum8(L, out, r / den);
}

// backward of apply_fwd: writes d(raw q) (32 floats at dq) and accumulates dkv
__device__ __forceinline__ void apply_bwd(const Lane& L, const float* __restrict__ q, const float (&kv)[NC + 1],
                                          const float* __restrict__ dO, float* __restrict__ dq,
                                          float (&dkv)[NC + 1]) {
  const float pre = pre_of(L, q);
  const float phi = phi_of(pre);
  float g[NC];
  ld8(L, dO, g);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s = fmaf(kv[c], g[c], s);
  float den = head_allsum(phi * kv[NC]);
  const float dot = head_allsum(phi * s);
  const bool guarded = (den == 0.f);
  if (guarded) den = 1e-5f;
  const float inv = __builtin_amdgcn_rcpf(den);
  const float dden = guarded ? 0.f : -dot * inv * inv;
  const float dphi = fmaf(kv[NC], dden, s * inv);
#pragma unroll
  for (int c = 0; c < NC; ++c) dkv[c] = fmaf(phi, g[c] * inv, dkv[c]);
  dkv[NC] = fmaf(phi, dden, dkv[NC]);
  const float dpre = pre > 0.f ? dphi : 0.f;
  float t[8];
#pragma unroll
  for (int j = 0; j < DH; ++j) t[j] = dpre * L.ps[j];
  st_sum8(L, dq, head_sum8(L, t));
}

// backward of one KV accumulation term: dV (32 floats at dv, may be null) and d(raw key) (32 floats at dk)
__device__ __forceinline__ void kv_bwd(const Lane& L, const float* __restrict__ k, const float* __restrict__ val_p,
                                       const float (&dkv)[NC + 1], float* __restrict__ dv, float* __restrict__ dk) {
  const float pre = pre_of(L, k);
  const float phi = phi_of(pre);
  float val[NC];
  ld8(L, val_p, val);
  float t[8];
  float dphi = dkv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    t[c] = phi * dkv[c];
    dphi = fmaf(dkv[c], val[c], dphi);
  }
  float r = head_sum8(L, t);
  if (dv) st_sum8(L, dv, r);
  const float dpre = pre > 0.f ? dphi : 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) t[j] = dpre * L.ps[j];
  st_sum8(L, dk, head_sum8(L, t));
}

// ---- register-slice forms: the 8-wide head slices come out of the wave's LDS staging row --------------------
__device__ __forceinline__ float pre_x(const Lane& L, const float (&x)[DH]) {
  float s = x[0] * L.ps[0];
#pragma unroll
  for (int j = 1; j < DH; ++j) s = fmaf(x[j], L.ps[j], s);
  return s;
}

__device__ __forceinline__ void slice8(const Lane& L, const float* lds_slot, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(lds_slot + L.h * 8);
  const float4 b = *reinterpret_cast<const float4*>(lds_slot + L.h * 8 + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

template <int NR>
__device__ __forceinline__ void stage_put(const Lane& L, float* stage, const float (&r)[NR]) {
  // The staging row is private to the wave and the LDS executes one wave's instructions in order, so only the
  // COMPILER has to be kept from moving the cross-lane reads over the writes: wavefront-scope fences.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < NR; ++t) stage[t * 64 + L.lane] = r[t];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void apply_fwd_x(const Lane& L, const float (&q)[DH], const float (&kv)[NC + 1],
                                            float* __restrict__ out) {
  const float phi = phi_of(pre_x(L, q));
  float t[8];
#pragma unroll
  for (int c = 0; c < NC; ++c) t[c] = phi * kv[c];
  float den = head_allsum(phi * kv[NC]);
  if (den == 0.f) den = 1e-5f;
  st_sum8(L, out, head_sum8(L, t) * __builtin_amdgcn_rcpf(den));
}

__device__ __forceinline__ void apply_bwd_x(const Lane& L, const float (&q)[DH], const float (&kv)[NC + 1],
                                            const float (&g)[NC], float* __restrict__ dq, float (&dkv)[NC + 1]) {
  const float pre = pre_x(L, q);
  const float phi = phi_of(pre);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s = fmaf(kv[c], g[c], s);
  float den = head_allsum(phi * kv[NC]);
  const float dot = head_allsum(phi * s);
  const bool guarded = (den == 0.f);
  if (guarded) den = 1e-5f;
  const float inv = __builtin_amdgcn_rcpf(den);
  const float dden = guarded ? 0.f : -dot * inv * inv;
  const float dphi = fmaf(kv[NC], dden, s * inv);
#pragma unroll
  for (int c = 0; c < NC; ++c) dkv[c] = fmaf(phi, g[c] * inv, dkv[c]);
  dkv[NC] = fmaf(phi, dden, dkv[NC]);
  const float dpre = pre > 0.f ? dphi : 0.f;
  float t[8];
#pragma unroll
  for (int j = 0; j < DH; ++j) t[j] = dpre * L.ps[j];
  st_sum8(L, dq, head_sum8(L, t));
}

__device__ __forceinline__ void kv_bwd_x(const Lane& L, const float (&k)[DH], const float (&val)[NC],
                                         const float (&dkv)[NC + 1], float* __restrict__ dv, float* __restrict__ dk) {
  const float pre = pre_x(L, k);
  const float phi = phi_of(pre);
  float t[8];
  float dphi = dkv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    t[c] = phi * dkv[c];
    dphi = fmaf(dkv[c], val[c], dphi);
  }
  const float r = head_sum8(L, t);
  if (dv) st_sum8(L, dv, r);
  const float dpre = pre > 0.f ? dphi : 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) t[j] = dpre * L.ps[j];
  st_sum8(L, dk, head_sum8(L, t));
}

// Software pipeline over `cnt` items, DEPTH 4: the NR coalesced row loads of items k+1..k+3 are in flight while
// item k is consumed.  Four named register sets keep every access statically indexed; loads past the end are
// clamped to the last item (a cache hit) so that the load stream is branch-free and the compiler can count it.
template <int NR, class LoadF, class UseF>
__device__ __forceinline__ void pipeline4(int cnt, LoadF load, UseF use) {
  if (cnt <= 0) return;
  const int last = cnt - 1;
  float r0[NR], r1[NR], r2[NR], r3[NR];
  load(0, r0);
  load(1 < last ? 1 : last, r1);
  load(2 < last ? 2 : last, r2);
  for (int k = 0; k < cnt; k += 4) {
    load(k + 3 < last ? k + 3 : last, r3);
    use(k, r0);
    load(k + 4 < last ? k + 4 : last, r0);
    if (k + 1 < cnt) use(k + 1, r1);
    load(k + 5 < last ? k + 5 : last, r1);
    if (k + 2 < cnt) use(k + 2, r2);
    load(k + 6 < last ? k + 6 : last, r2);
    if (k + 3 < cnt) use(k + 3, r3);
  }
}

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

// per-graph KV registers <-> memory, layout [B][4][9][64]
__device__ __forceinline__ void kvg_load(const Lane& L, const float* __restrict__ base, float (&kv)[NC + 1]) {
#pragma unroll
  for (int c = 0; c <= NC; ++c) kv[c] = base[c * 64 + L.lane];
}
__device__ __forceinline__ void kvg_store(const Lane& L, float* __restrict__ base, const float (&kv)[NC + 1]) {
#pragma unroll
  for (int c = 0; c <= NC; ++c) base[c * 64 + L.lane] = kv[c];
}
template <int S>
__device__ __forceinline__ void kv_zero(float (&kv)[S][NC + 1]) {
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int c = 0; c <= NC; ++c) kv[s][c] = 0.f;
}

// h1 slots: q1 = 0..3, k1 = 4..11;  h2 slots: q2 = 0..5, k2 = 6..7, skip path after slot 7
// v slots 0..5 feed the node-segment terms, 6..9 the per-graph terms
constexpr int K1 = 4, K2 = 6;
constexpr int STAGE_FLOATS = 6 * 64;
constexpr int WPB = 1;            // waves per block: single-wave blocks release their slot as soon as the wave ends
constexpr int TPB = 64 * WPB;
#ifndef HOTK_GRID_CAP
#define HOTK_GRID_CAP (1 << 20)
#endif
constexpr int64_t GRID_CAP = HOTK_GRID_CAP;
#ifndef HOTK_MINW
#define HOTK_MINW 1
#endif

// ------------------------------------------------------------------------------------------------
// per-graph sums (forward): wave per (graph, row chunk); rows are consecutive, two coalesced loads per row
//   diagonal row (node n):  keys k1 slots 6,7 = h1[n][10*32 .. 12*32),  values v slots 6, 8
//   off-diagonal row     :  keys k2 slots 0,1 = h2[r][ 6*32 ..  8*32),  values v slots 7, 9
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chunk_rows(const A& a, int64_t w, int cpg, int rows_per_chunk, int64_t* b, int64_t* r0,
                                           int64_t* r1) {
  *b = w / cpg;
  const int64_t k = w - *b * cpg;
  *r0 = *b * a.Emax + k * rows_per_chunk;
  int64_t e = *r0 + rows_per_chunk;
  const int64_t rend = *b * a.Emax + a.n_entries[*b];
  *r1 = e < rend ? e : rend;
}

// rows [ra, rb) of graph b, all diagonal (DIAG) or all off-diagonal: two coalesced loads per row
template <bool DIAG>
__device__ __forceinline__ void graphsum_rows(const Lane& L, const A& a, float* stage, int64_t b, int64_t ra, int64_t rb,
                                              float (&kv)[4][NC + 1]) {
  const bool lo = L.lane < 32;
  auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    const float* vr = a.v + row * a.ld_v;
    if (DIAG) {
      const int64_t n = b * a.Nmax + (row - b * a.Emax);
      r[0] = a.h1[n * a.ld_h1 + 10 * SLOT + L.lane];
      r[1] = vr[(lo ? 6 * SLOT : 7 * SLOT) + L.lane];          // slot 6 | slot 8
    } else {
      r[0] = a.h2[row * a.ld_h2 + K2 * SLOT + L.lane];
      r[1] = vr[(lo ? 7 * SLOT : 8 * SLOT) + L.lane];          // slot 7 | slot 9
    }
  };
  auto use = [&](int, const float (&r)[2]) __attribute__((always_inline)) {
    stage_put<2>(L, stage, r);
    float x[8], val[8];
    slice8(L, stage, x); slice8(L, stage + 64, val);
    kv_acc(kv[DIAG ? 0 : 1], phi_of(pre_x(L, x)), val);
    slice8(L, stage + 32, x); slice8(L, stage + 96, val);
    kv_acc(kv[DIAG ? 2 : 3], phi_of(pre_x(L, x)), val);
  };
  pipeline4<2>((int)(rb - ra), load, use);
}

__global__ __launch_bounds__(TPB, HOTK_MINW) void k_graphsum_fwd(A a, float* __restrict__ dst, int cpg, int rows_per_chunk) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  const int64_t nwork = a.B * cpg;
  for (int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); w < nwork; w += (int64_t)gridDim.x * WPB) {
    int64_t b, r0, r1;
    chunk_rows(a, w, cpg, rows_per_chunk, &b, &r0, &r1);
    const int64_t nodes_end = b * a.Emax + a.n_nodes[b];   // rows below are the diagonal (layout invariant)
    const int64_t rm = r1 < nodes_end ? r1 : (r0 > nodes_end ? r0 : nodes_end);
    float kv[4][NC + 1];
    kv_zero<4>(kv);
    graphsum_rows<true>(L, a, stage, b, r0, rm, kv);
    graphsum_rows<false>(L, a, stage, b, rm, r1, kv);
#pragma unroll
    for (int s = 0; s < 4; ++s) kvg_store(L, dst + ((int64_t)w * 4 + s) * (NC + 1) * 64, kv[s]);
  }
}

// out[b][x] = sum_k part[b*cpg + k][x], x over 4*9*64 floats
__global__ void k_graph_reduce(const float* __restrict__ part, float* __restrict__ out, int64_t B, int cpg) {
  const int64_t X = 4 * (NC + 1) * 64, total = B * X;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = t / X, x = t - b * X;
    float s = 0.f;
    for (int k = 0; k < cpg; ++k) s += part[((int64_t)b * cpg + k) * X + x];
    out[t] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// node segments: one wave per (node, PART).  PART p owns the two column sums of slots 2p, 2p+1
//   p = 0: terms 1, 2  (values v0 | v1^T, applied to the node's own query; also the node-level graph terms 7, 8)
//   p = 1: terms 3, 5  (values v2 | v3^T, applied to the entries of column n; also the entry-level graph terms 9, 10)
//   p = 2: terms 4, 6  (values v4^T | v5, applied to the entries of ROW n)
// so a wave carries 18 accumulators instead of 54 and three times as many waves hide the gather latency.
// Segment entries come as packed records (plan): col_rec[p] = {entry row, row node, transposed row, kind},
// row_rec[p] = entry row (off-diagonal) or ~entry row (diagonal); 64 records are fetched per block, one per lane,
// and walked with wave-uniform readlane indices.
// ------------------------------------------------------------------------------------------------
struct ColBlock { int32_t e, i, te, kd; };

__device__ __forceinline__ ColBlock col_block(const Lane& L, const A& a, int32_t base, int cnt) {
  ColBlock s{0, 0, -1, 0};
  if (L.lane < cnt) {
    const int4 r = reinterpret_cast<const int4*>(a.col_rec)[base + L.lane];
    s.e = r.x; s.i = r.y; s.te = r.z; s.kd = r.w;
  }
  return s;
}

// two coalesced loads for entry e = (i, j) of a column segment: key slots {2p, 2p+1} of node i and the value
// slots {2p | 2p+1} where the transposed slot is read from row te (zeros when the transposed entry is absent)
template <int P>
__device__ __forceinline__ void load_kv2(const Lane& L, const A& a, int32_t e, int32_t i, int32_t te, float (&r)[2]) {
  r[0] = a.h1[(int64_t)i * a.ld_h1 + (K1 + 2 * P) * SLOT + L.lane];
  const bool lo = L.lane < 32;
  const int32_t row = (P == 2) ? (lo ? te : e) : (lo ? e : te);
  // a missing transposed entry reads the entry's own row instead (any valid address) and is masked at USE time
  // (t_scale): selecting on the loaded value here would make the prefetch synchronous
  r[1] = a.v[(int64_t)(row < 0 ? e : row) * a.ld_v + 2 * P * SLOT + L.lane];
}

// KV accumulate whose value channels are scaled by s (0 when the transposed entry is absent): the normaliser
// channel still counts the key (kernelattncoef.py:103-122 sums keys over every entry of the column)
__device__ __forceinline__ void kv_acc_scaled(float (&kv)[NC + 1], float phi, const float (&val)[NC], float s) {
  const float ps = phi * s;
#pragma unroll
  for (int c = 0; c < NC; ++c) kv[c] = fmaf(ps, val[c], kv[c]);
  kv[NC] += phi;
}

template <int P, class F>
__device__ __forceinline__ void for_col_entries(const Lane& L, const A& a, float* stage, int32_t n, F per_entry) {
  const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
  for (int32_t base = c0; base < c1; base += 64) {
    const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
    const ColBlock cb = col_block(L, a, base, cnt);
    auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) { load_kv2<P>(L, a, rl(cb.e, k), rl(cb.i, k), rl(cb.te, k), r); };
    auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
      stage_put<2>(L, stage, r);
      per_entry(rl(cb.e, k), rl(cb.te, k));
    };
    pipeline4<2>(cnt, load, use);
  }
}

template <int P>
__device__ __forceinline__ void seg_accumulate(const Lane& L, const A& a, float* stage, int32_t n, float (&kv)[2][NC + 1]) {
  kv_zero<2>(kv);
  for_col_entries<P>(L, a, stage, n, [&](int32_t, int32_t te) __attribute__((always_inline)) {
    float x[8], val[8];
    const float ts = te < 0 ? 0.f : 1.f;   // the transposed-value slot is slot 0 for P == 2, slot 1 otherwise
    slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, val);
    if (P == 2) kv_acc_scaled(kv[0], phi_of(pre_x(L, x)), val, ts);
    else kv_acc(kv[0], phi_of(pre_x(L, x)), val);
    slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, val);
    if (P == 2) kv_acc(kv[1], phi_of(pre_x(L, x)), val);
    else kv_acc_scaled(kv[1], phi_of(pre_x(L, x)), val, ts);
  });
}

template <int P>
__global__ __launch_bounds__(TPB, HOTK_MINW) void k_seg_fwd(A a) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  for (int64_t n = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); n < a.Nn; n += (int64_t)gridDim.x * WPB) {
    if (!a.nkind[n]) continue;
    float kv[2][NC + 1];
    seg_accumulate<P>(L, a, stage, (int32_t)n, kv);
    if (P == 0) {
      // node queries: terms 1, 2 (column sums) and 7, 8 (graph sums) go to the node's diagonal row
      const float* kvg = a.kvg + (n / a.Nmax) * 4 * (NC + 1) * 64;
      float g0[NC + 1], g1[NC + 1];
      const float* q = a.h1 + n * a.ld_h1;
      float* o = a.att + (int64_t)a.diag_pos[n] * a.ld_att;
      float r[2] = {q[L.lane], q[64 + L.lane]};
      kvg_load(L, kvg + 0 * (NC + 1) * 64, g0);
      kvg_load(L, kvg + 1 * (NC + 1) * 64, g1);
      stage_put<2>(L, stage, r);
      float x[8];
      slice8(L, stage + 0 * SLOT, x); apply_fwd_x(L, x, kv[0], o + 0 * SLOT);
      slice8(L, stage + 1 * SLOT, x); apply_fwd_x(L, x, kv[1], o + 1 * SLOT);
      slice8(L, stage + 2 * SLOT, x); apply_fwd_x(L, x, g0, o + 6 * SLOT);
      slice8(L, stage + 3 * SLOT, x); apply_fwd_x(L, x, g1, o + 7 * SLOT);
      zero_slots(L, o + 2 * SLOT, 4);
      zero_slots(L, o + 8 * SLOT, 2);
    } else if (P == 1) {
      // entries of column n: terms 3, 5 (this node's sums) and 9, 10 (graph sums)
      const float* kvg = a.kvg + (n / a.Nmax) * 4 * (NC + 1) * 64;
      float g0[NC + 1], g1[NC + 1];
      kvg_load(L, kvg + 2 * (NC + 1) * 64, g0);
      kvg_load(L, kvg + 3 * (NC + 1) * 64, g1);
      const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
      for (int32_t base = c0; base < c1; base += 64) {
        const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
        const ColBlock cb = col_block(L, a, base, cnt);
        auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
          const float* q = a.h2 + (int64_t)rl(cb.e, k) * a.ld_h2;
          r[0] = q[L.lane];               // query slots 0, 1
          r[1] = q[4 * SLOT + L.lane];    // query slots 4, 5
        };
        auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
          if (rl(cb.kd, k) != 2) return;
          float* o = a.att + (int64_t)rl(cb.e, k) * a.ld_att;
          stage_put<2>(L, stage, r);
          float x[8];
          slice8(L, stage + 0 * SLOT, x); apply_fwd_x(L, x, kv[0], o + 2 * SLOT);
          slice8(L, stage + 1 * SLOT, x); apply_fwd_x(L, x, kv[1], o + 4 * SLOT);
          slice8(L, stage + 2 * SLOT, x); apply_fwd_x(L, x, g0, o + 8 * SLOT);
          slice8(L, stage + 3 * SLOT, x); apply_fwd_x(L, x, g1, o + 9 * SLOT);
          o[L.lane] = 0.f;                 // terms 1, 2 live on the diagonal only
          o[6 * SLOT + L.lane] = 0.f;      // terms 7, 8 too
        };
        pipeline4<2>(cnt, load, use);
      }
    } else {
      // entries of row n: terms 4, 6 read the sums of their ROW node
      const int32_t c0 = a.row_ptr[n], c1 = a.row_ptr[n + 1];
      for (int32_t base = c0; base < c1; base += 64) {
        const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
        const int32_t rec = L.lane < cnt ? a.row_rec[base + L.lane] : -1;
        auto load = [&](int k, float (&r)[1]) __attribute__((always_inline)) {
          const int32_t e = rl(rec, k);
          r[0] = a.h2[(int64_t)(e < 0 ? ~e : e) * a.ld_h2 + 2 * SLOT + L.lane];
        };
        auto use = [&](int k, const float (&r)[1]) __attribute__((always_inline)) {
          const int32_t e = rl(rec, k);
          if (e < 0) return;
          float* o = a.att + (int64_t)e * a.ld_att;
          stage_put<1>(L, stage, r);
          float x[8];
          slice8(L, stage + 0 * SLOT, x); apply_fwd_x(L, x, kv[0], o + 3 * SLOT);
          slice8(L, stage + 1 * SLOT, x); apply_fwd_x(L, x, kv[1], o + 5 * SLOT);
        };
        pipeline4<1>(cnt, load, use);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-graph terms (backward): (1) query side + dKVg partials, (2) key/value side
// ------------------------------------------------------------------------------------------------
template <bool DIAG>
__device__ __forceinline__ void graph_apply_bwd_rows(const Lane& L, const A& a, float* stage, int64_t b, int64_t ra,
                                                     int64_t rb, const float (&kv)[4][NC + 1], float (&dkv)[4][NC + 1]) {
  auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    const float* g = a.datt + row * a.ld_datt;
    if (DIAG) {
      const int64_t n = b * a.Nmax + (row - b * a.Emax);
      r[0] = a.h1[n * a.ld_h1 + 2 * SLOT + L.lane];     // node query slots 2, 3
      r[1] = g[6 * SLOT + L.lane];                       // d term 7, 8
    } else {
      r[0] = a.h2[row * a.ld_h2 + 4 * SLOT + L.lane];   // entry query slots 4, 5
      r[1] = g[8 * SLOT + L.lane];                       // d term 9, 10
    }
  };
  auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    stage_put<2>(L, stage, r);
    float x[8], g[8];
    float* dq = DIAG ? a.dh1 + (b * a.Nmax + (row - b * a.Emax)) * a.ld_dh1 + 2 * SLOT
                     : a.dh2 + row * a.ld_dh2 + 4 * SLOT;
    slice8(L, stage, x); slice8(L, stage + 64, g);
    apply_bwd_x(L, x, kv[DIAG ? 0 : 2], g, dq, dkv[DIAG ? 0 : 2]);
    slice8(L, stage + 32, x); slice8(L, stage + 96, g);
    apply_bwd_x(L, x, kv[DIAG ? 1 : 3], g, dq + SLOT, dkv[DIAG ? 1 : 3]);
  };
  pipeline4<2>((int)(rb - ra), load, use);
}

__global__ __launch_bounds__(TPB, HOTK_MINW) void k_graph_apply_bwd(A a, float* __restrict__ dst, int cpg, int rows_per_chunk) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  const int64_t nwork = a.B * cpg;
  for (int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); w < nwork; w += (int64_t)gridDim.x * WPB) {
    int64_t b, r0, r1;
    chunk_rows(a, w, cpg, rows_per_chunk, &b, &r0, &r1);
    const int64_t nodes_end = b * a.Emax + a.n_nodes[b];
    const int64_t rm = r1 < nodes_end ? r1 : (r0 > nodes_end ? r0 : nodes_end);
    float kv[4][NC + 1], dkv[4][NC + 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) kvg_load(L, a.kvg + ((int64_t)b * 4 + s) * (NC + 1) * 64, kv[s]);
    kv_zero<4>(dkv);
    graph_apply_bwd_rows<true>(L, a, stage, b, r0, rm, kv, dkv);
    graph_apply_bwd_rows<false>(L, a, stage, b, rm, r1, kv, dkv);
#pragma unroll
    for (int s = 0; s < 4; ++s) kvg_store(L, dst + ((int64_t)w * 4 + s) * (NC + 1) * 64, dkv[s]);
  }
}

// wave per chunk of 16 consecutive rows of one graph
constexpr int GB_ROWS = 16;
__global__ __launch_bounds__(TPB, HOTK_MINW) void k_graphsum_bwd(A a) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  const bool lo = L.lane < 32;
  const int64_t cpg = (a.Emax + GB_ROWS - 1) / GB_ROWS;
  const int64_t nwork = a.B * cpg;
  for (int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); w < nwork; w += (int64_t)gridDim.x * WPB) {
    int64_t b, r0, r1;
    chunk_rows(a, w, (int)cpg, GB_ROWS, &b, &r0, &r1);
    if (r1 <= r0) continue;
    const int64_t nodes_end = b * a.Emax + a.n_nodes[b];
    const float* dkvg = a.dkvg + b * 4 * (NC + 1) * 64;
    float d[4][NC + 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) kvg_load(L, dkvg + s * (NC + 1) * 64, d[s]);
    auto load = [&](int k, float (&r)[3]) __attribute__((always_inline)) {
      const int64_t row = r0 + k;
      const float* vr = a.v + row * a.ld_v;
      if (row < nodes_end) {
        const int64_t n = b * a.Nmax + (row - b * a.Emax);
        r[0] = a.h1[n * a.ld_h1 + 10 * SLOT + L.lane];
        r[1] = vr[(lo ? 6 * SLOT : 7 * SLOT) + L.lane];
      } else {
        r[0] = a.h2[row * a.ld_h2 + K2 * SLOT + L.lane];
        r[1] = vr[(lo ? 7 * SLOT : 8 * SLOT) + L.lane];
      }
      r[2] = (a.dout && L.lane < a.D) ? a.dout[row * a.ld_dout + L.lane] : 0.f;
    };
    auto use = [&](int k, const float (&r)[3]) __attribute__((always_inline)) {
      const int64_t row = r0 + k;
      const bool diag = row < nodes_end;
      stage_put<2>(L, stage, reinterpret_cast<const float (&)[2]>(r));
      float* dvr = a.dv + row * a.ld_dv;
      float* dh2r = a.dh2 + row * a.ld_dh2;
      float x[8], val[8];
      if (diag) {
        float* dkn = a.dh1 + (b * a.Nmax + (row - b * a.Emax)) * a.ld_dh1 + 10 * SLOT;
        slice8(L, stage, x); slice8(L, stage + 64, val);
        kv_bwd_x(L, x, val, d[0], dvr + 6 * SLOT, dkn);
        slice8(L, stage + 32, x); slice8(L, stage + 96, val);
        kv_bwd_x(L, x, val, d[2], dvr + 8 * SLOT, dkn + SLOT);
        if (lo) { dvr[7 * SLOT + L.lane] = 0.f; dvr[9 * SLOT + L.lane] = 0.f; }
        zero_slots(L, dh2r, 8);   // entry-level queries / keys are masked on the diagonal
      } else {
        slice8(L, stage, x); slice8(L, stage + 64, val);
        kv_bwd_x(L, x, val, d[1], dvr + 7 * SLOT, dh2r + K2 * SLOT);
        slice8(L, stage + 32, x); slice8(L, stage + 96, val);
        kv_bwd_x(L, x, val, d[3], dvr + 9 * SLOT, dh2r + (K2 + 1) * SLOT);
        if (lo) { dvr[6 * SLOT + L.lane] = 0.f; dvr[8 * SLOT + L.lane] = 0.f; }
      }
      // skip path: d h2[:, 8 slots:] = d out (zero when the caller routes that gradient elsewhere)
      if (L.lane < a.D) dh2r[8 * SLOT + L.lane] = r[2];
      for (int i = 64 + L.lane; i < a.D; i += 64) dh2r[8 * SLOT + i] = a.dout ? a.dout[row * a.ld_dout + i] : 0.f;
    };
    pipeline4<3>((int)(r1 - r0), load, use);
  }
}

// ------------------------------------------------------------------------------------------------
// node segments (backward): one wave per (node, part), same split as the forward
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(TPB, HOTK_MINW) void k_seg_bwd(A a) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  const bool lo = L.lane < 32;
  for (int64_t n = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); n < a.Nn; n += (int64_t)gridDim.x * WPB) {
    if (!a.nkind[n]) continue;
    float kv[2][NC + 1], dkv[2][NC + 1];
    seg_accumulate<P>(L, a, stage, (int32_t)n, kv);
    kv_zero<2>(dkv);
    if (P == 0) {  // node queries (terms 1, 2)
      const int64_t dr = a.diag_pos[n];
      float r[2] = {a.h1[n * a.ld_h1 + L.lane], a.datt[dr * a.ld_datt + L.lane]};
      float* dq = a.dh1 + n * a.ld_dh1;
      stage_put<2>(L, stage, r);
      float x[8], g[8];
      slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd_x(L, x, kv[0], g, dq, dkv[0]);
      slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd_x(L, x, kv[1], g, dq + SLOT, dkv[1]);
    } else if (P == 1) {  // entries of column n (terms 3, 5)
      const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
      for (int32_t base = c0; base < c1; base += 64) {
        const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
        const ColBlock cb = col_block(L, a, base, cnt);
        auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
          const int64_t e = rl(cb.e, k);
          r[0] = a.h2[e * a.ld_h2 + L.lane];                                      // query slots 0, 1
          r[1] = a.datt[e * a.ld_datt + (lo ? 2 * SLOT : 3 * SLOT) + L.lane];      // d term 3 | d term 5
        };
        auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
          if (rl(cb.kd, k) != 2) return;
          float* dq = a.dh2 + (int64_t)rl(cb.e, k) * a.ld_dh2;
          stage_put<2>(L, stage, r);
          float x[8], g[8];
          slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd_x(L, x, kv[0], g, dq, dkv[0]);
          slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd_x(L, x, kv[1], g, dq + SLOT, dkv[1]);
        };
        pipeline4<2>(cnt, load, use);
      }
    } else {  // entries of row n (terms 4, 6)
      const int32_t c0 = a.row_ptr[n], c1 = a.row_ptr[n + 1];
      for (int32_t base = c0; base < c1; base += 64) {
        const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
        const int32_t rec = L.lane < cnt ? a.row_rec[base + L.lane] : -1;
        auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
          const int32_t e_ = rl(rec, k);
          const int64_t e = e_ < 0 ? ~e_ : e_;
          r[0] = a.h2[e * a.ld_h2 + 2 * SLOT + L.lane];                            // query slots 2, 3
          r[1] = a.datt[e * a.ld_datt + (lo ? 3 * SLOT : 4 * SLOT) + L.lane];      // d term 4 | d term 6
        };
        auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
          const int32_t e = rl(rec, k);
          if (e < 0) return;
          float* dq = a.dh2 + (int64_t)e * a.ld_dh2 + 2 * SLOT;
          stage_put<2>(L, stage, r);
          float x[8], g[8];
          slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd_x(L, x, kv[0], g, dq, dkv[0]);
          slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd_x(L, x, kv[1], g, dq + SLOT, dkv[1]);
        };
        pipeline4<2>(cnt, load, use);
      }
    }
    // key / value side of this part's two column sums: dV of the plain slot goes to row e, dV of the transposed
    // slot to row te; d(raw key) contributions go to ck[e] (summed per row node afterwards)
    for_col_entries<P>(L, a, stage, (int32_t)n, [&](int32_t e, int32_t te) __attribute__((always_inline)) {
      float* dve = a.dv + (int64_t)e * a.ld_dv;
      float* dvt = te >= 0 ? a.dv + (int64_t)te * a.ld_dv : nullptr;
      float* ck = a.ck + (int64_t)e * (6 * SLOT) + 2 * P * SLOT;
      constexpr int S0 = 2 * P, S1 = 2 * P + 1;
      float* d0 = (P == 2) ? (dvt ? dvt + S0 * SLOT : nullptr) : dve + S0 * SLOT;
      float* d1 = (P == 2) ? dve + S1 * SLOT : (dvt ? dvt + S1 * SLOT : nullptr);
      float x[8], val[8];
      slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, val);
      if (P == 2 && te < 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) val[c] = 0.f;
      }
      kv_bwd_x(L, x, val, dkv[0], d0, ck);
      slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, val);
      if (P != 2 && te < 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) val[c] = 0.f;
      }
      kv_bwd_x(L, x, val, dkv[1], d1, ck + SLOT);
      if (te < 0 && lo) dve[((P == 2) ? S0 : S1) * SLOT + L.lane] = 0.f;   // no transposed partner writes this slot
    });
  }
}

int check_args(const A* a) {
  if (!a) return HOTK_EINVAL;
  if (a->H != NH || a->m != NM || a->d_head != DH || a->c != NC) return HOTK_EUNSUPPORTED;
  if (a->B < 0 || a->Emax < 0 || a->Nmax < 0 || a->R != a->B * a->Emax || a->Nn != a->B * a->Nmax) return HOTK_EINVAL;
  if (!a->h1 || !a->h2 || !a->v || !a->proj || !a->kind || !a->nkind || !a->rown || !a->coln || !a->tpos ||
      !a->diag_pos || !a->col_ptr || !a->col_perm || !a->row_ptr || !a->row_perm || !a->n_entries || !a->n_nodes || !a->col_rec || !a->row_rec)
    return HOTK_EINVAL;
  if (a->ld_h1 < 12 * SLOT || a->ld_h2 < 8 * SLOT || a->ld_v < 10 * SLOT) return HOTK_EINVAL;
  // 16-byte vector loads of 8-float head slices
  if ((a->ld_h1 | a->ld_h2 | a->ld_v) & 3) return HOTK_EINVAL;
  if (!hotk_aligned16(a->h1) || !hotk_aligned16(a->h2) || !hotk_aligned16(a->v)) return HOTK_EINVAL;
  return HOTK_OK;
}

void chunking(const A* a, int* cpg, int* rows) {
  int64_t c = (a->Emax + 63) / 64;
  if (c > 4096) c = 4096;
  if (c < 1) c = 1;
  *cpg = (int)c;
  *rows = (int)((a->Emax + c - 1) / c);
}

}  // namespace

extern "C" int hotk_attn22_workspace_bytes(int64_t B, int64_t Emax, size_t* bytes) {
  HOTK_CHECK_ARG(bytes && B >= 0 && Emax >= 0);
  int64_t c = (Emax + 63) / 64;
  if (c > 4096) c = 4096;
  if (c < 1) c = 1;
  *bytes = (size_t)B * c * 4 * (NC + 1) * 64 * sizeof(float);
  return HOTK_OK;
}

extern "C" int hotk_attn22_forward(const hotk_attn22_args* a, void* workspace, size_t ws_bytes, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  HOTK_CHECK_ARG(a->att && a->kvg && a->ld_att >= 10 * SLOT && (a->ld_att & 3) == 0 && hotk_aligned16(a->att));
  if (a->R == 0 || a->Nn == 0) return HOTK_OK;
  hipStream_t st = (hipStream_t)stream;
  int cpg, rows;
  chunking(a, &cpg, &rows);
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  if (cpg > 1 && (!workspace || ws_bytes < need)) return HOTK_EWORKSPACE;
  float* part = cpg > 1 ? (float*)workspace : a->kvg;
  unsigned g1 = hotk_grid((int64_t)a->B * cpg, WPB, GRID_CAP);
  hipLaunchKernelGGL(k_graphsum_fwd, dim3(g1), dim3(TPB), 0, st, *a, part, cpg, rows);
  if (cpg > 1)
    hipLaunchKernelGGL(k_graph_reduce, dim3(hotk_grid(a->B * 4 * (NC + 1) * 64, 256)), dim3(256), 0, st, part, a->kvg,
                       a->B, cpg);
  const dim3 gseg(hotk_grid(a->Nn, WPB, GRID_CAP));
  hipLaunchKernelGGL(k_seg_fwd<1>, gseg, dim3(TPB), 0, st, *a);
  hipLaunchKernelGGL(k_seg_fwd<2>, gseg, dim3(TPB), 0, st, *a);
  hipLaunchKernelGGL(k_seg_fwd<0>, gseg, dim3(TPB), 0, st, *a);
  return hotk_launch_status();
}

extern "C" int hotk_attn22_backward(const hotk_attn22_args* a, void* workspace, size_t ws_bytes, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  HOTK_CHECK_ARG(a->kvg && a->datt && a->dh1 && a->dh2 && a->dv && a->ck && a->dkvg);
  HOTK_CHECK_ARG(a->ld_datt >= 10 * SLOT && a->ld_dh1 >= 12 * SLOT && a->ld_dh2 >= 8 * SLOT + a->D && a->D >= 0 &&
                 a->ld_dv >= 10 * SLOT);
  HOTK_CHECK_ARG(((a->ld_datt | a->ld_dh1 | a->ld_dh2 | a->ld_dv) & 3) == 0 && hotk_aligned16(a->datt));
  HOTK_CHECK_ARG(!a->dout || a->ld_dout >= a->D);
  if (a->R == 0 || a->Nn == 0) return HOTK_OK;
  hipStream_t st = (hipStream_t)stream;
  int cpg, rows;
  chunking(a, &cpg, &rows);
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  if (cpg > 1 && (!workspace || ws_bytes < need)) return HOTK_EWORKSPACE;
  float* part = cpg > 1 ? (float*)workspace : a->dkvg;
  hipLaunchKernelGGL(k_graph_apply_bwd, dim3(hotk_grid((int64_t)a->B * cpg, WPB, GRID_CAP)), dim3(TPB), 0, st, *a, part, cpg, rows);
  if (cpg > 1)
    hipLaunchKernelGGL(k_graph_reduce, dim3(hotk_grid(a->B * 4 * (NC + 1) * 64, 256)), dim3(256), 0, st, part, a->dkvg,
                       a->B, cpg);
  hipLaunchKernelGGL(k_graphsum_bwd, dim3(hotk_grid(a->R, WPB, GRID_CAP)), dim3(TPB), 0, st, *a);
  const dim3 gseg(hotk_grid(a->Nn, WPB, GRID_CAP));
  hipLaunchKernelGGL(k_seg_bwd<1>, gseg, dim3(TPB), 0, st, *a);
  hipLaunchKernelGGL(k_seg_bwd<2>, gseg, dim3(TPB), 0, st, *a);
  hipLaunchKernelGGL(k_seg_bwd<0>, gseg, dim3(TPB), 0, st, *a);
  return hotk_launch_status();
}
