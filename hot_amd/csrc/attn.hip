// K3 (general sizes): segmented key-value reductions and their application, any HH / m / c.
// Reference: models/sparse/kernelattncoef.py:74-91 (per-graph sums), :103-122 (per-column sums through
// torch.sparse coalesce), :124-175 (query application and 1e-5 normaliser guard).
//
// KV layout: [n_seg][F = HH*m][c+1] floats; channel c is the normaliser (sum of keys) in the forward
// pass and the normaliser gradient in the backward pass.
#include "common.h"
#include <type_traits>

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
    int normalize, const int32_t* __restrict__ order) {
  const int F = HH * m, c1 = c + 1, W = HH * c;
  const int64_t total = n * W;
  const bool q4 = (m & 3) == 0 && (ld_q & 3) == 0 && (((uintptr_t)Q) & 15u) == 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pos = t / W;
    int w = (int)(t - pos * W);
    const int64_t r = order ? (int64_t)order[pos] : pos;
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
    float* __restrict__ X, int64_t ld_x, float* __restrict__ XA, int64_t ld_xa, int64_t n, int HH, int m, int c,
    const int32_t* __restrict__ order) {
  const int F = HH * m, c1 = c + 1;
  const int64_t total = n * HH;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pos = t / HH;
    int hh = (int)(t - pos * HH);
    const int64_t r = order ? (int64_t)order[pos] : pos;
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
    float* __restrict__ X, int64_t ld_x, float* __restrict__ XA, int64_t ld_xa, int64_t n, int HH, int m,
    const int32_t* __restrict__ order) {
  constexpr int c1 = C + 1;
  const int F = HH * m, W = HH * C;
  const int64_t total = n * W;
  // every lane of a wave runs the same number of iterations (total is padded), so the shuffles see whole groups
  const int64_t padded = (total + 63) & ~(int64_t)63;
  const bool q4 = (m & 3) == 0 && (ld_q & 3) == 0 && (((uintptr_t)Q) & 15u) == 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < padded; t += (int64_t)gridDim.x * blockDim.x) {
    const bool live = t < total;
    const int64_t pos = live ? t / W : 0;
    const int w = (int)(t - pos * W);
    const int64_t r = (live && order) ? (int64_t)order[pos] : pos;
    const int hh = live ? w / C : 0, chn = w - hh * C;
    const int64_t s = live ? seg_of(sidx, seg_div, r) : -1;
    const bool on = live && s >= 0 && (!kind || hotk_keep(keep, kind[r]));
    float x = 0.f, part = 0.f, inv = 0.f;
    bool guarded = false;
    if (on) {
      const float* q = Q + r * ld_q + hh * m;
      const float* kv = KV + ((int64_t)s * F + hh * m) * c1;
      float num = 0.f, den = 0.f;
      if (q4) {   // m % 4 == 0, 16-byte aligned rows: the query slice as vector loads (same fmaf order)
        for (int mm = 0; mm < m; mm += 4) {
          const float4 qv = *reinterpret_cast<const float4*>(q + mm);
          const float* k0 = kv + mm * c1;
          num = fmaf(qv.x, k0[chn], num);          den = fmaf(qv.x, k0[C], den);
          num = fmaf(qv.y, k0[c1 + chn], num);     den = fmaf(qv.y, k0[c1 + C], den);
          num = fmaf(qv.z, k0[2 * c1 + chn], num); den = fmaf(qv.z, k0[2 * c1 + C], den);
          num = fmaf(qv.w, k0[3 * c1 + chn], num); den = fmaf(qv.w, k0[3 * c1 + C], den);
        }
      } else {
        for (int mm = 0; mm < m; ++mm) {
          const float qv = q[mm];
          num = fmaf(qv, kv[mm * c1 + chn], num);
          den = fmaf(qv, kv[mm * c1 + C], den);
        }
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

// The two query-side contractions (k_row_contract, k_apply_bwd_stats_c) with the key-value column of a lane kept in
// REGISTERS across the rows of one segment, for c in {8, 16}, m in {16, 44} (m <= 4 c) and 16-byte aligned query
// rows.  The thread-per-element kernels fetch a row's 2 m key-value floats per lane again for every row (~100 load
// instructions per wave and row at m = 44, each spread over 4 x 68 B); rows arrive sorted by their segment (or are
// visited through `order`), so a lane group that walks QC_ROWS consecutive visits reloads its column only where the
// segment changes (~1 row in 10).  lane = (row slot, head, channel): 64 / (HH c) rows side by side when a row needs
// fewer than 64 lanes, each slot walking its own run of consecutive visits; (HH c) / 64 waves per run otherwise.
// What a wave reads per row besides that: the row's queries -- ONE 16-byte load per lane, QC_PF rows ahead, restaged
// through a wave-private LDS ring so that every lane reads its head's m floats as broadcast ds_read_b128 -- and the
// head's normaliser column, which sits in LDS beside the ring (it is the same for the c lanes of a head).  Row ids,
// segments, kinds and gather indices of all QC_ROWS visits are fetched up front, one visit per lane, and parked in LDS:
// two dependent round trips per QC_ROWS rows instead of three per row.  Same fmaf chains per element as the thread
// kernels: same bits.
#ifndef HOTK_QC_WAVE
#define HOTK_QC_WAVE 1
#endif
constexpr int QC_ROWS = 16, QC_PF = 4, QC_WAVES = 4;
static_assert(QC_ROWS % QC_PF == 0, "the row loop runs in groups of QC_PF steps");
template <int C, int MM, bool STATS>
__global__ __launch_bounds__(64 * QC_WAVES) void k_query_contract_w(
    const float* __restrict__ Q, int64_t ld_q, const int32_t* __restrict__ qidx, const float* __restrict__ KV,
    const int32_t* __restrict__ sidx, int64_t seg_div, const uint8_t* __restrict__ kind, unsigned keep,
    const int32_t* __restrict__ oidx, float* __restrict__ out, int64_t ld_out, int normalize,
    const float* __restrict__ dO, int64_t ld_do, float* __restrict__ XA, int64_t ld_xa, int64_t n, int HH,
    const int32_t* __restrict__ order) {
  constexpr int c1 = C + 1, m = MM;       // (m is a template constant: a guard per feature costs more than the arithmetic)
  // per wave: query ring [QC_PF][256 floats], normaliser columns [256 floats], visit records [8 slots][QC_ROWS] int4
  __shared__ __align__(16) float ring_all[QC_WAVES][QC_PF][256];
  __shared__ __align__(16) float kvd_all[QC_WAVES][256];
  __shared__ int4 meta_all[QC_WAVES][8 * QC_ROWS];
  const int W = HH * C, F = HH * m;
  const int gw = W < 64 ? W : 64, side = 64 / gw, slices = (W + 63) >> 6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / gw, wl = lane - slot * gw;
  const int hl = wl / C, chn = wl & (C - 1);       // head within the wave's slice, channel
  const int nq4 = (gw / C) * m >> 2;               // 16-byte pieces of a slot's query floats (<= gw since m <= 4 c)
  float* ring = &ring_all[wave][0][0] + slot * gw * 4;
  float* kvd = kvd_all[wave] + slot * gw * 4;
  int4* meta = meta_all[wave] + slot * QC_ROWS;
  const int64_t span = (int64_t)side * QC_ROWS;
  const int64_t items = ((n + span - 1) / span) * slices;
  const int64_t nwave = (int64_t)gridDim.x * QC_WAVES;
  for (int64_t it = (int64_t)blockIdx.x * QC_WAVES + wave; it < items; it += nwave) {
    const int64_t ch = it / slices;
    const int h0 = (int)(it - ch * slices) * (64 / C);     // first head of this wave's slice
    const int w = h0 * C + wl, hh = h0 + hl;
    const int64_t pos0 = ch * span + (int64_t)slot * QC_ROWS;
    // ---- visit records: (row, segment or -1 = zero result, query row, -); visits past the end repeat the last row
    // (the same values stored to the same addresses once more)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = wl; i < QC_ROWS; i += gw) {
      const int64_t pos = pos0 + i < n ? pos0 + i : n - 1;
      const int64_t r = order ? (int64_t)order[pos] : pos;
      const int64_t qr = qidx ? (int64_t)qidx[r] : r;
      const int64_t sg = seg_of(sidx, seg_div, r);
      int4 t = make_int4((int)r, -1, (int)qr, 0);
      if (sg >= 0 && qr >= 0 && (!kind || hotk_keep(keep, kind[r]))) t.y = (int)sg;
      meta[i] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Every global load and store of the row loop is unconditional straight-line code (lanes and rows with nothing to
    // fetch read the start of KV): loads and stores share one in-order counter, and the compiler waits for "all but the
    // last k" only when it knows k -- behind a load or store under a branch or an exec mask it waits for everything,
    // the store of the row before and the prefetches just issued included (see k_row_expand_w).
    auto q_addr = [&](const int4& t) -> const float* {
      return (t.y >= 0 && wl < nq4) ? Q + (int64_t)t.z * ld_q + h0 * m + 4 * wl : KV;
    };
    auto g_addr = [&](const int4& t) -> const float* { return t.y >= 0 ? dO + (int64_t)t.x * ld_do + w : KV; };
    // (named registers, not arrays: captured by the step lambda an array stays in scratch memory)
    float4 qp0, qp1, qp2, qp3;
    float gp0 = 0.f, gp1 = 0.f, gp2 = 0.f, gp3 = 0.f;
    auto first = [&](float4& qq, float& gq, const int k) __attribute__((always_inline)) {
      const int4 t = meta[k];
      qq = *reinterpret_cast<const float4*>(q_addr(t));
      if (STATS) gq = *g_addr(t);
    };
    first(qp0, gp0, 0); first(qp1, gp1, 1); first(qp2, gp2, 2); first(qp3, gp3, 3);
    float kvn[MM];
#pragma unroll
    for (int mm = 0; mm < MM; ++mm) kvn[mm] = 0.f;
    int s_cur = -1;
    // every lane runs all QC_ROWS steps: the shuffles below see whole groups.  Groups of QC_PF steps: a step's ring slot
    // and prefetch registers are compile-time (rotating them through moves would wait for the load just issued); the
    // group loop stays rolled (unrolled, the scheduler parks the records of all 16 steps in registers)
    auto step = [&](float4& qq, float& gq, const int b, const int i) __attribute__((always_inline)) {
      const int4 t = meta[i];
      const bool on = t.y >= 0;
      float* qrow = ring + b * 256;
      if (wl < nq4) *reinterpret_cast<float4*>(qrow + 4 * wl) = qq;
      const float gv = gq;
      {                                       // the operands of the row QC_PF steps ahead (the last group re-reads the last row)
        const int4 u = meta[i + QC_PF < QC_ROWS ? i + QC_PF : QC_ROWS - 1];
        qq = *reinterpret_cast<const float4*>(q_addr(u));
        if (STATS) gq = *g_addr(u);
      }
      float num = 0.f, den = 0.f;
      if (on) {
        if (t.y != s_cur) {                   // another segment: this lane's key-value column, the heads' normaliser columns
          const float* kv = KV + ((int64_t)t.y * F + hh * m) * c1;
#pragma unroll
          for (int mm = 0; mm < MM; ++mm) kvn[mm] = kv[mm * c1 + chn];
          const float* kvs = KV + ((int64_t)t.y * F + h0 * m) * c1 + C;
          for (int f = wl; f < 4 * nq4; f += gw) kvd[f] = kvs[(int64_t)f * c1];
          // (an opaque use of the column INSIDE the branch: the wait for these loads then sits here, on the rare path;
          // otherwise it lands in front of the products below as vmcnt(0) and every step waits for its own prefetches)
#pragma unroll
          for (int mm = 0; mm < MM; ++mm) asm volatile("" : "+v"(kvn[mm]));
          s_cur = t.y;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on) {
        const float* qh = qrow + hl * m;
        const float* kd = kvd + hl * m;
#pragma unroll
        for (int mm = 0; mm < MM; mm += 4) {
          const float4 qv = *reinterpret_cast<const float4*>(qh + mm);
          const float4 dv = *reinterpret_cast<const float4*>(kd + mm);
          num = fmaf(qv.x, kvn[mm], num);         den = fmaf(qv.x, dv.x, den);
          num = fmaf(qv.y, kvn[mm + 1], num);     den = fmaf(qv.y, dv.y, den);
          num = fmaf(qv.z, kvn[mm + 2], num);     den = fmaf(qv.z, dv.z, den);
          num = fmaf(qv.w, kvn[mm + 3], num);     den = fmaf(qv.w, dv.w, den);
          // (an opaque use of den: otherwise the two chains are packed into v_pk_fma_f32, whose operand PAIRS put every
          // kvn[] into a register pair with a per-step copy of the normaliser value beside it -- 2 m registers and a
          // v_mov per product; a packed multiply-add takes two issue slots anyway)
          asm volatile("" : "+v"(den));
          // (without this the scheduler issues all 2 m / 4 LDS reads of the row first: 88 more registers at m = 44)
          if ((mm & 4) != 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!STATS) {
        float o = 0.f;
        if (on) {
          if (normalize) {
            if (den == 0.f) den = 1e-5f;
            o = num / den;
          } else {
            o = num;
          }
        }
        out[(int64_t)t.x * ld_out + w] = o;
      } else {
        // X = dO / den, XA = -(dO . O) / den (k_apply_bwd_stats_c): the channel sum meets through xor shuffles
        float x = 0.f, part = 0.f, inv = 0.f;
        bool guarded = false;
        if (on) {
          guarded = (den == 0.f);
          if (guarded) den = 1e-5f;
          inv = 1.f / den;
          x = gv * inv;
          part = gv * (num * inv);
        }
#pragma unroll
        for (int o = 1; o < C; o <<= 1) part += __shfl_xor(part, o, 64);
        out[(int64_t)t.x * ld_out + w] = x;
        XA[(int64_t)t.x * ld_xa + hh] = (on && !guarded) ? -part * inv : 0.f;     // (all c lanes of the head: same value)
      }
    };
    static_assert(QC_PF == 4, "the group below lists its steps");
#pragma unroll 1
    for (int i0 = 0; i0 < QC_ROWS; i0 += QC_PF) {
      step(qp0, gp0, 0, i0);
      step(qp1, gp1, 1, i0 + 1);
      step(qp2, gp2, 2, i0 + 2);
      step(qp3, gp3, 3, i0 + 3);
    }
  }
}

template <bool STATS>
bool launch_query_contract_w(hipStream_t st, const float* Q, int64_t ld_q, const int32_t* qidx, const float* KV,
                             const int32_t* sidx, int64_t seg_div, const uint8_t* kind, unsigned keep,
                             const int32_t* oidx, float* out, int64_t ld_out, int normalize, const float* dO,
                             int64_t ld_do, float* XA, int64_t ld_xa, int64_t n, int HH, int m, int c,
                             const int32_t* order) {
  const int W = HH * c;
  // the feature counts of the shipped head widths: m = int(d log d) = 16 (d = 8), 44 (d = 16); kernel.py:60-63
  if (!HOTK_QC_WAVE || (c != 8 && c != 16) || (m != 16 && m != 44) || m > 4 * c || (ld_q & 3) || !hotk_aligned16(Q)) return false;
  if (oidx || !hotk_aligned16(KV)) return false;   // (no caller scatters the output rows; KV's start is the dummy address)
  if (n >= (int64_t)0x7fffffff) return false;      // visit records hold 32-bit row ids
  if (!(W <= 64 ? 64 % W == 0 : W % 64 == 0)) return false;
  const int side = W < 64 ? 64 / W : 1, slices = (W + 63) / 64;
  const int64_t span = (int64_t)side * QC_ROWS, items = ((n + span - 1) / span) * slices;
  const dim3 grid(hotk_grid(items, QC_WAVES, 256 * 32)), block(64 * QC_WAVES);
#define HOTK_QC_LAUNCH(CC, MMX)                                                                                        \
  hipLaunchKernelGGL((k_query_contract_w<CC, MMX, STATS>), grid, block, 0, st, Q, ld_q, qidx, KV, sidx, seg_div, kind, \
                     keep, oidx, out, ld_out, normalize, dO, ld_do, XA, ld_xa, n, HH, order)
  if (c == 8) {
    HOTK_QC_LAUNCH(8, 16);
  } else {
    if (m == 16) HOTK_QC_LAUNCH(16, 16); else HOTK_QC_LAUNCH(16, 44);
  }
#undef HOTK_QC_LAUNCH
  return true;
}

// thread per (row, feature)
__global__ __launch_bounds__(AT_THREADS) void k_row_expand(
    const float* __restrict__ KV, const int32_t* __restrict__ sidx, int64_t seg_div, const float* __restrict__ X,
    int64_t ld_x, const int32_t* __restrict__ xidx, const float* __restrict__ XA, int64_t ld_xa,
    const uint8_t* __restrict__ kind, unsigned keep, float* __restrict__ out, int64_t ld_out, int64_t n, int HH, int m,
    int c, const int32_t* __restrict__ order) {
  const int F = HH * m, c1 = c + 1;
  const int64_t total = n * F;
  const bool x8 = c == 8 && (ld_x & 3) == 0 && (((uintptr_t)X) & 15u) == 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pos = t / F;
    int f = (int)(t - pos * F);
    const int64_t r = order ? (int64_t)order[pos] : pos;
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

// The same map with a wave per (row, 64 consecutive features) for c in {8, 16} and 16-byte aligned X rows: the wave's
// key-value rows KV[s][f0 .. f0 + 64)[0 .. c] are ONE contiguous block of 64 (c + 1) floats, fetched with c + 1 coalesced
// loads into a wave-private LDS image and read back one row per lane (stride c + 1, odd: no bank conflicts) -- the
// thread-per-element kernel reads them as c + 1 loads whose lanes are 4 (c + 1) bytes apart (~34 cache lines per
// instruction at c = 16).  Same fmaf order per element: identical bits.
#ifndef HOTK_WAVE_C8
#define HOTK_WAVE_C8 1
#endif
constexpr int RX_WAVES = 4;
constexpr int RX_ROWS = 16;  // consecutive visits a wave walks per feature group: rows that share a segment (sorted entry
                             // lists, graph-keyed sums) reuse the key-value rows already in registers
constexpr int RX_PF = 4;     // rows whose operands are in flight ahead of the arithmetic
constexpr int RX_XF = 256 + 64;   // floats per ring stage: the heads' X pieces (<= 64 x 16 B), then one XA value per head
static_assert(RX_ROWS % RX_PF == 0, "the row loop runs in groups of RX_PF steps");
// Row ids, segments, kinds and gather indices of the RX_ROWS visits are fetched up front, one visit per lane, and parked
// in LDS (two dependent round trips per RX_ROWS rows instead of three per row); a row's X / XA pieces are fetched RX_PF
// rows ahead by the first lanes (ONE 16-byte load per lane instead of c / 4 broadcast loads by every lane) and restaged
// through a wave-private LDS ring; the next segment's block is requested one row ahead.  Needs m >= 8 (the heads that
// a 64-feature group touches then fit the ring).
// Every global load and store of the row loop is UNCONDITIONAL straight-line code -- lanes and rows that have nothing to
// fetch read a dummy address (the start of KV), lanes past a partial feature group and visits past the end duplicate
// the last real one (same value to the same address) -- because gfx9 counts loads and stores in one in-order counter and
// the compiler can only wait for "all but the last k" when it knows k: with a load or store under a branch or an exec
// mask it waits for everything (vmcnt(0)) before staging a prefetched row, i.e. for the store of the row before and the
// prefetches just issued, and the ring hides nothing (measured: 1 us per row either way).  Only the block fetch at a
// segment change stays conditional.
template <int C>
__global__ __launch_bounds__(64 * RX_WAVES) void k_row_expand_w(
    const float* __restrict__ KV, const int32_t* __restrict__ sidx, int64_t seg_div, const float* __restrict__ X,
    int64_t ld_x, const int32_t* __restrict__ xidx, const float* __restrict__ XA, int64_t ld_xa,
    const uint8_t* __restrict__ kind, unsigned keep, float* __restrict__ out, int64_t ld_out, int64_t n, int HH, int m,
    const int32_t* __restrict__ order) {
  constexpr int c1 = C + 1;
  __shared__ float img_all[RX_WAVES][64 * c1];
  __shared__ __align__(16) float ring_all[RX_WAVES][RX_PF][RX_XF];
  __shared__ int4 meta_all[RX_WAVES][RX_ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* img = img_all[wave];
  float* ring = &ring_all[wave][0][0];
  int4* meta = meta_all[wave];
  const int F = HH * m, NG = (F + 63) >> 6;
  const int64_t chunks = (n + RX_ROWS - 1) / RX_ROWS;
  const int64_t items = chunks * NG;
  const bool small = items < (int64_t)0x7fffffff;
  for (int64_t w = (int64_t)blockIdx.x * RX_WAVES + wave; w < items; w += (int64_t)gridDim.x * RX_WAVES) {
    const int64_t ch = small ? (int64_t)((uint32_t)w / (uint32_t)NG) : w / NG;
    const int f0 = (int)(w - ch * NG) << 6;
    const int nf = F - f0 < 64 ? F - f0 : 64;
    const int len = nf * c1;
    const int ln = lane < nf ? lane : nf - 1;      // lanes past a partial group shadow its last feature
    const int f = f0 + ln;
    const int hA = f0 / m, nh = (f0 + nf - 1) / m - hA + 1;     // heads this feature group touches
    const int hl = f / m - hA;
    const int nx4 = nh * (C / 4);
    const int64_t r0 = ch * RX_ROWS;
    // ---- visit records: (row, segment or -1 = zero output, X row or -1, -); visits past the end repeat the last one
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < RX_ROWS) {
      const int64_t pos = r0 + lane < n ? r0 + lane : n - 1;
      const int64_t r = order ? (int64_t)order[pos] : pos;
      const int64_t sg = seg_of(sidx, seg_div, r);
      int4 t = make_int4((int)r, -1, xidx ? xidx[r] : (int)r, 0);
      if (sg >= 0 && (!kind || hotk_keep(keep, kind[r]))) t.y = (int)sg;
      meta[lane] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // one row per step, so the records are wave-uniform: scalar registers, scalar branches
    auto record = [&](int i) {
      const int4 v = meta[i];
      return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                       __builtin_amdgcn_readfirstlane(v.z), 0);
    };
    auto x_addr = [&](const int4& t) -> const float* {      // this lane's 16 bytes of the row's X piece, or the dummy
      return (t.y >= 0 && t.z >= 0 && lane < nx4) ? X + (int64_t)t.z * ld_x + hA * C + 4 * lane : KV;
    };
    auto a_addr = [&](const int4& t) -> const float* {
      return (XA && t.y >= 0 && lane < nh) ? XA + (int64_t)t.x * ld_xa + hA + lane : KV;
    };
    // (named registers, not an array: captured by the step lambda an array stays in scratch memory)
    float4 xp0, xp1, xp2, xp3;
    float ap0, ap1, ap2, ap3;
    auto first = [&](float4& xq, float& aq, const int k) __attribute__((always_inline)) {
      const int4 t = record(k);
      xq = *reinterpret_cast<const float4*>(x_addr(t));
      aq = *a_addr(t);
    };
    first(xp0, ap0, 0); first(xp1, ap1, 1); first(xp2, ap2, 2); first(xp3, ap3, 3);
    float kv[c1], pre[c1];
#pragma unroll
    for (int k = 0; k < c1; ++k) kv[k] = pre[k] = 0.f;
    auto fetch = [&](int64_t sg) {          // the block's 64 (c + 1) floats, coalesced, into registers
      const float* blk = KV + (sg * F + f0) * c1;
#pragma unroll
      for (int k = 0; k < c1; ++k) {      // (a partial group re-reads its last float instead of branching per load)
        const int i = lane + 64 * k;
        pre[k] = blk[i < len ? i : len - 1];
      }
    };
    int s_cur = -1;
    const int s_pre = __builtin_amdgcn_readfirstlane(meta[0].y);
    if (s_pre >= 0) fetch(s_pre);
    // groups of RX_PF steps: a step's ring slot and prefetch registers are compile-time indices (rotating them through
    // moves would wait for the load just issued)
    auto step = [&](float4& xq, float& aq, const int b, const int i) __attribute__((always_inline)) {
      const int4 t = record(i);
      float* xs = ring + b * RX_XF;
      *reinterpret_cast<float4*>(xs + 4 * lane) = xq;      // (64 x 16 B: the whole stage; lanes past the piece hold dummies)
      xs[256 + lane] = (XA && t.y >= 0) ? aq : 1.f;
      {                                       // the row RX_PF steps ahead (the last group re-reads the chunk's last row)
        const int4 u = record(i + RX_PF < RX_ROWS ? i + RX_PF : RX_ROWS - 1);
        xq = *reinterpret_cast<const float4*>(x_addr(u));
        aq = *a_addr(u);
      }
      if (t.y >= 0 && t.y != s_cur) {         // transpose through the wave's LDS image: lane <- row lane of the block
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < c1; ++k) img[lane + 64 * k] = pre[k];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < c1; ++k) kv[k] = img[ln * c1 + k];
        s_cur = t.y;
      }
      // the next row's key-value block: its latency hides behind this row
      if (i + 1 < RX_ROWS) {
        const int sn = __builtin_amdgcn_readfirstlane(meta[i + 1].y);
        if (sn >= 0 && sn != s_cur) fetch(sn);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float o = 0.f;
      if (t.y >= 0) {
        if (t.z >= 0) {
          const float* xh = xs + hl * C;
#pragma unroll
          for (int q = 0; q < C / 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xh + 4 * q);
            o = fmaf(kv[4 * q + 0], xv.x, o); o = fmaf(kv[4 * q + 1], xv.y, o);
            o = fmaf(kv[4 * q + 2], xv.z, o); o = fmaf(kv[4 * q + 3], xv.w, o);
          }
        }
        o = fmaf(kv[C], xs[256 + hl], o);
      }
      out[(int64_t)t.x * ld_out + f] = o;
    };
    static_assert(RX_PF == 4, "the group below lists its steps");
#pragma unroll 1
    for (int i0 = 0; i0 < RX_ROWS; i0 += RX_PF) {
      step(xp0, ap0, 0, i0);
      step(xp1, ap1, 1, i0 + 1);
      step(xp2, ap2, 2, i0 + 2);
      step(xp3, ap3, 3, i0 + 3);
    }
  }
}

// Key-value sums with a thread per (segment, feature) holding all c + 1 accumulators (c in {8, 16}: ONE walk of the
// segment) and the wave's 64 result rows -- one contiguous block of 64 (c + 1) floats -- leaving through an LDS image
// as c + 1 coalesced stores instead of c + 1 stores with lanes 4 (c + 1) bytes apart.  The walk is pipelined like
// k_row_expand_w: the entries' row / key / value indices are fetched 64 at a time, one entry per lane (two dependent
// round trips per 64 entries instead of three per entry) and handed out with v_readlane; an entry's key value (one float
// per lane), the heads' value pieces (ONE 16-byte load by the first lanes instead of c / 4 broadcast loads by every
// lane) and weights are fetched RX_PF entries ahead -- unconditional loads, dummy addresses for lanes and entries that
// have nothing to fetch, so that the compiler can count them -- and the value pieces pass through a wave-private LDS
// ring.  Needs m >= 8.  Same fmaf order per accumulator as k_seg_outer_sum: identical bits.
template <int C>
__global__ __launch_bounds__(64 * RX_WAVES) void k_seg_outer_sum_w(
    const float* __restrict__ K, int64_t ld_k, const int32_t* __restrict__ kidx, const float* __restrict__ V,
    int64_t ld_v, const int32_t* __restrict__ vidx, const float* __restrict__ aug, int64_t ld_aug,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, const int32_t* __restrict__ perm,
    float* __restrict__ KV, int64_t n_seg, int HH, int m) {
  constexpr int c1 = C + 1;
  __shared__ float img_all[RX_WAVES][64 * c1];
  __shared__ __align__(16) float ring_all[RX_WAVES][RX_PF][RX_XF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* img = img_all[wave];
  float* ring = &ring_all[wave][0][0];
  const int F = HH * m, NG = (F + 63) >> 6;
  const int64_t items = n_seg * NG;
  for (int64_t w = (int64_t)blockIdx.x * RX_WAVES + wave; w < items; w += (int64_t)gridDim.x * RX_WAVES) {
    const int64_t s = w / NG;
    const int f0 = (int)(w - s * NG) << 6;
    const int nf = F - f0 < 64 ? F - f0 : 64;
    const int f = f0 + (lane < nf ? lane : nf - 1);     // idle lanes shadow the last feature (no divergent loop bounds)
    const int hA = f0 / m, nh = (f0 + nf - 1) / m - hA + 1;     // heads this feature group touches
    const int hl = f / m - hA;
    const int nx4 = nh * (C / 4);
    const int lo = __builtin_amdgcn_readfirstlane(seg_begin[s]), hi = __builtin_amdgcn_readfirstlane(seg_end[s]);
    float acc[c1];
#pragma unroll
    for (int k = 0; k < c1; ++k) acc[k] = 0.f;
    for (int base = lo; base < hi; base += 64) {
      const int cnt = hi - base < 64 ? hi - base : 64;
      // this batch's entries, one per lane (lanes past the end repeat the last entry)
      const int p = base + (lane < cnt ? lane : cnt - 1);
      const int rv = perm ? perm[p] : p;
      const int krv = kidx ? kidx[rv] : rv;
      const int vrv = vidx ? vidx[rv] : rv;
      struct Rec { int r, kr, vr; };
      auto rec = [&](int i) {        // (i is wave-uniform)
        const int j = i < cnt ? i : cnt - 1;
        return Rec{__builtin_amdgcn_readlane(rv, j), __builtin_amdgcn_readlane(krv, j), __builtin_amdgcn_readlane(vrv, j)};
      };
      auto k_addr = [&](const Rec& t) -> const float* { return t.kr >= 0 ? K + (int64_t)t.kr * ld_k + f : K; };
      auto v_addr = [&](const Rec& t) -> const float* {
        return (t.kr >= 0 && t.vr >= 0 && lane < nx4) ? V + (int64_t)t.vr * ld_v + hA * C + 4 * lane : V;
      };
      auto a_addr = [&](const Rec& t) -> const float* {
        return (aug && t.kr >= 0 && lane < nh) ? aug + (int64_t)t.r * ld_aug + hA + lane : K;
      };
      // (named registers, not arrays: captured by the step lambda an array stays in scratch memory)
      float kq0, kq1, kq2, kq3, aq0, aq1, aq2, aq3;
      float4 vq0, vq1, vq2, vq3;
      auto first = [&](float& kq, float4& vq, float& aq, const int i) __attribute__((always_inline)) {
        const Rec t = rec(i);
        kq = *k_addr(t);
        vq = *reinterpret_cast<const float4*>(v_addr(t));
        aq = *a_addr(t);
      };
      first(kq0, vq0, aq0, 0); first(kq1, vq1, aq1, 1); first(kq2, vq2, aq2, 2); first(kq3, vq3, aq3, 3);
      auto step = [&](float& kq, float4& vq, float& aq, const int b, const int i) __attribute__((always_inline)) {
        const Rec t = rec(i);
        float* xs = ring + b * RX_XF;
        *reinterpret_cast<float4*>(xs + 4 * lane) = vq;      // (the whole stage; lanes past the piece hold dummies)
        xs[256 + lane] = aug ? aq : 1.f;
        const float kv = kq;
        first(kq, vq, aq, i + RX_PF);                         // the entry RX_PF steps ahead (clamped to the batch's last)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (i < cnt && t.kr >= 0) {
          if (t.vr >= 0) {
            const float* vh = xs + hl * C;
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
              const float4 v = *reinterpret_cast<const float4*>(vh + 4 * q);
              acc[4 * q + 0] = fmaf(kv, v.x, acc[4 * q + 0]); acc[4 * q + 1] = fmaf(kv, v.y, acc[4 * q + 1]);
              acc[4 * q + 2] = fmaf(kv, v.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(kv, v.w, acc[4 * q + 3]);
            }
          } else {
#pragma unroll
            for (int k = 0; k < C; ++k) acc[k] = fmaf(kv, 0.f, acc[k]);
          }
          acc[C] = fmaf(kv, xs[256 + hl], acc[C]);
        }
      };
      static_assert(RX_PF == 4, "the group below lists its steps");
#pragma unroll 1
      for (int i0 = 0; i0 < cnt; i0 += RX_PF) {
        step(kq0, vq0, aq0, 0, i0);
        step(kq1, vq1, aq1, 1, i0 + 1);
        step(kq2, vq2, aq2, 2, i0 + 2);
        step(kq3, vq3, aq3, 3, i0 + 3);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < c1; ++k) img[lane * c1 + k] = acc[k];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* blk = KV + ((int64_t)s * F + f0) * c1;
    const int len = nf * c1;
#pragma unroll
    for (int k = 0; k < c1; ++k) {
      const int i = lane + 64 * k;
      if (i < len) blk[i] = img[i];
    }
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
  const bool v16 = (ld_v & 3) == 0 && hotk_aligned16(V);
  // c in {8, 16} with 16-byte aligned value rows: the wave form
  if ((c == 16 || (c == 8 && HOTK_WAVE_C8)) && v16 && m >= 8) {
    const int64_t items = n_seg * ((HH * m + 63) / 64);
    const dim3 grid(hotk_grid(items, RX_WAVES, 256 * 32)), block(64 * RX_WAVES);
    if (c == 16)
      hipLaunchKernelGGL(k_seg_outer_sum_w<16>, grid, block, 0, (hipStream_t)stream, K, ld_k, kidx, V, ld_v, vidx, aug,
                         ld_aug, seg_begin, seg_end, perm, KV, n_seg, HH, m);
    else
      hipLaunchKernelGGL(k_seg_outer_sum_w<8>, grid, block, 0, (hipStream_t)stream, K, ld_k, kidx, V, ld_v, vidx, aug,
                         ld_aug, seg_begin, seg_end, perm, KV, n_seg, HH, m);
    return hotk_launch_status();
  }
  const bool c8 = c == 8 && v16;
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
                                 int normalize, const int32_t* order, void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(Q && KV && out && ld_q >= (int64_t)HH * m && ld_out >= (int64_t)HH * c);
  if (launch_query_contract_w<false>((hipStream_t)stream, Q, ld_q, qidx, KV, sidx, seg_div, kind, keep, oidx, out, ld_out,
                                     normalize, nullptr, 0, nullptr, 0, n, HH, m, c, order))
    return hotk_launch_status();
  hipLaunchKernelGGL(k_row_contract, dim3(hotk_grid(n * HH * c, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                     (hipStream_t)stream, Q, ld_q, qidx, KV, sidx, seg_div, kind, keep, oidx, out, ld_out, n, HH, m, c,
                     normalize, order);
  return hotk_launch_status();
}

extern "C" int hotk_apply_bwd_stats(const float* Q, int64_t ld_q, const float* KV, const int32_t* sidx,
                                    int64_t seg_div, const uint8_t* kind, unsigned keep, const float* dO, int64_t ld_do,
                                    float* X, int64_t ld_x, float* XA, int64_t ld_xa, int64_t n, int HH, int m, int c,
                                    const int32_t* order, void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(Q && KV && dO && X && XA && ld_q >= (int64_t)HH * m && ld_do >= (int64_t)HH * c &&
                 ld_x >= (int64_t)HH * c && ld_xa >= HH);
  hipStream_t st = (hipStream_t)stream;
  if (launch_query_contract_w<true>(st, Q, ld_q, nullptr, KV, sidx, seg_div, kind, keep, nullptr, X, ld_x, 1, dO, ld_do, XA,
                                    ld_xa, n, HH, m, c, order))
    return hotk_launch_status();
#define HOTK_ABS_CASE(CC)                                                                                              \
  case CC:                                                                                                             \
    hipLaunchKernelGGL(k_apply_bwd_stats_c<CC>, dim3(hotk_grid(n * HH * CC, AT_THREADS, 256 * 32)), dim3(AT_THREADS),  \
                       0, st, Q, ld_q, KV, sidx, seg_div, kind, keep, dO, ld_do, X, ld_x, XA, ld_xa, n, HH, m, order); \
    break;
  switch (c) {
    HOTK_ABS_CASE(1) HOTK_ABS_CASE(2) HOTK_ABS_CASE(4) HOTK_ABS_CASE(8) HOTK_ABS_CASE(16) HOTK_ABS_CASE(32)
    HOTK_ABS_CASE(64)
    default:
      hipLaunchKernelGGL(k_apply_bwd_stats, dim3(hotk_grid(n * HH, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0, st, Q,
                         ld_q, KV, sidx, seg_div, kind, keep, dO, ld_do, X, ld_x, XA, ld_xa, n, HH, m, c, order);
  }
#undef HOTK_ABS_CASE
  return hotk_launch_status();
}

extern "C" int hotk_row_expand(const float* KV, const int32_t* sidx, int64_t seg_div, const float* X, int64_t ld_x,
                               const int32_t* xidx, const float* XA, int64_t ld_xa, const uint8_t* kind, unsigned keep,
                               float* out, int64_t ld_out, int64_t n, int HH, int m, int c, const int32_t* order,
                               void* stream) {
  HOTK_CHECK_ARG(dims_ok(HH, m, c) && n >= 0 && (sidx || seg_div > 0));
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(KV && X && out && ld_x >= (int64_t)HH * c && ld_out >= (int64_t)HH * m && (!XA || ld_xa >= HH));
  if ((c == 16 || (c == 8 && HOTK_WAVE_C8)) && (ld_x & 3) == 0 && hotk_aligned16(X) && hotk_aligned16(KV) && m >= 8 && n < (int64_t)0x7fffffff) {
    const int64_t items = ((n + RX_ROWS - 1) / RX_ROWS) * ((HH * m + 63) / 64);
    const dim3 grid(hotk_grid(items, RX_WAVES, 256 * 32)), block(64 * RX_WAVES);
    if (c == 16)
      hipLaunchKernelGGL(k_row_expand_w<16>, grid, block, 0, (hipStream_t)stream, KV, sidx, seg_div, X, ld_x, xidx, XA,
                         ld_xa, kind, keep, out, ld_out, n, HH, m, order);
    else
      hipLaunchKernelGGL(k_row_expand_w<8>, grid, block, 0, (hipStream_t)stream, KV, sidx, seg_div, X, ld_x, xidx, XA,
                         ld_xa, kind, keep, out, ld_out, n, HH, m, order);
    return hotk_launch_status();
  }
  hipLaunchKernelGGL(k_row_expand, dim3(hotk_grid(n * HH * m, AT_THREADS, 256 * 32)), dim3(AT_THREADS), 0,
                     (hipStream_t)stream, KV, sidx, seg_div, X, ld_x, xidx, XA, ld_xa, kind, keep, out, ld_out, n, HH,
                     m, c, order);
  return hotk_launch_status();
}
