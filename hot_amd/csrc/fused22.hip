// Fused attention core of the 2->2 sparse kernel-attention layer for the headline shape
// (4 heads x 16 random features = one 64-lane wavefront per slot; head width 8, value width 8,
// generalized / ReLU FAVOR+ map).  Reference: models/sparse/kernelattn.py:119-159 calling
// models/sparse/kernelattncoef.py:74-234 ten times; here the ten terms are produced by
//
//   graph sums   : one wave per row chunk keeps the four per-graph  sum phi(K) (x) [V|1]  in registers
//   node segments: one wave per (node j, part) keeps two per-node   sum_i phi(K_i) (x) [V_ij|1]  (2 x 9 floats per
//                  lane) in registers while it walks column segment j, and immediately applies them to the node's
//                  own query, to the entries of column j or to the entries of row j.  No key-value tensor, no
//                  phi(Q)/phi(K) tensor and no transposed value copy ever reaches HBM.
//
// The kernels are VALU-bound (measured: HBM traffic within 1.2x of the algorithmic bytes at ~2 TB/s), so the
// arithmetic is laid out to need as few vector instructions as possible.  Two lane layouts of a 16-lane DPP row
// (= one head h = lane >> 4, x = lane & 15):
//   F layout: lane (h, x) owns random feature mm = x: phi, the 9 key-value accumulators KV[h][mm][0..8];
//             accumulation  KV[mm][c] += phi[mm] * v[c]  needs no cross-lane traffic at all.
//   C layout: lane (h, x) owns output channel c = x (0..7 values, 8..15 the normaliser) and holds the ROTATED
//             column  rot[k] = KV[h][(x - k) & 15][c];  a contraction over the 16 features is then
//             sum_k row_ror_k(phi) * rot[k]  = 16 DPP multiply-adds, with no reduction tree, select or LDS traffic.
// The same rotated form of the projection matrix turns  d(raw q)[j] = sum_mm dpre[mm] P[mm][j]  into 16 DPP FMAs.
#include "common.h"
#include <cstdlib>
#include <mutex>

namespace {

constexpr int NH = 4, NM = 16, DH = 8, NC = 8;  // heads, features, head width, value width per head
constexpr int SLOT = 32;                        // floats per dim_qk / dim_v slot (NH*DH == NH*NC == 32)
constexpr int KVG_C = 25;                       // per-graph sums in memory: [B][4][25][64] = 9 F-layout + 16 rotated
constexpr float PHI_EPS = 1e-3f;

using A = hotk_attn22_args;
// typed views of the wide tensors of an argument block
template <class ST>
struct SV {
  const ST *h2, *v, *datt;
  ST *att, *dh2, *dv, *ck;
  __device__ __forceinline__ explicit SV(const A& a)
      : h2((const ST*)a.h2), v((const ST*)a.v), datt((const ST*)a.datt), att((ST*)a.att), dh2((ST*)a.dh2), dv((ST*)a.dv),
        ck((ST*)a.ck) {}
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- storage type ST of the wide entry-level tensors (h2, v, att and the gradients datt, dh2, dv, ck): float, or bf16 with
// all arithmetic and every sum in fp32 (a->storage; BASELINE north_star: bf16 mode held to 2e-2).  Node-level tensors
// (h1, phik, the key-value sums and their gradients) are always fp32.
typedef __bf16 bf16;
__device__ __forceinline__ float ldS(const float* p) { return *p; }
__device__ __forceinline__ float ldS(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void stS(float* p, float v) { *p = v; }
__device__ __forceinline__ void stS(bf16* p, float v) { *p = (bf16)v; }
__device__ __forceinline__ float4 ld4(const bf16* p) {   // 4 consecutive elements (8-byte aligned)
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, const float4& v) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  const bf16x2 a = __builtin_convertvector((f2v){v.x, v.y}, bf16x2), b = __builtin_convertvector((f2v){v.z, v.w}, bf16x2);
  uint2 u;
  u.x = __builtin_bit_cast(uint32_t, a);
  u.y = __builtin_bit_cast(uint32_t, b);
  *reinterpret_cast<uint2*>(p) = u;
}

// ---- 16-lane (one DPP row = one head) data movement on the VALU ---------------------------------------------
// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_mirror = 0x140, row_half_mirror = 0x141,
// row_ror:k = 0x120 + k (lane x reads lane (x - k) & 15 of its row)
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int K>
__device__ __forceinline__ float ror(float v) { return dpp<0x120 + K>(v); }
// source lane of row_ror:k
#ifndef HOTK_ROR_DIR
#define HOTK_ROR_DIR (-1)
#endif
__device__ __host__ __forceinline__ int ror_src(int x, int k) { return (x + HOTK_ROR_DIR * k) & 15; }

// sum over the 16 lanes of a head, result in every lane
__device__ __forceinline__ float head_allsum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}

// acc += row_ror_K(v) * r as ONE instruction.  hipcc (ROCm 7.2) folds a DPP move into v_mul_f32 but not into
// v_fmac_f32, so the multiply-add form is written out; the asm statements are volatile (kept in program order) and
// the first one of a contraction carries the two wait states a DPP read needs after the VALU write of its source.
template <int K>
__device__ __forceinline__ void fmac_ror(float& acc, float v, float r);
#define HOTK_FMAC_ROR(K)                                                                                         \
  template <>                                                                                                    \
  __device__ __forceinline__ void fmac_ror<K>(float& acc, float v, float r) {                                    \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1"             \
                 : "+v"(acc) : "v"(v), "v"(r));                                                                  \
  }
HOTK_FMAC_ROR(2) HOTK_FMAC_ROR(3) HOTK_FMAC_ROR(4) HOTK_FMAC_ROR(5) HOTK_FMAC_ROR(6) HOTK_FMAC_ROR(7) HOTK_FMAC_ROR(8)
HOTK_FMAC_ROR(9) HOTK_FMAC_ROR(10) HOTK_FMAC_ROR(11) HOTK_FMAC_ROR(12) HOTK_FMAC_ROR(13) HOTK_FMAC_ROR(14) HOTK_FMAC_ROR(15)
#undef HOTK_FMAC_ROR

// sum_k v[(x - k) & 15] * rot[k]: the 16-feature contraction of the C layout (two accumulation chains).  One asm
// statement: between separate statements the compiler pads with s_nop (it cannot see that they carry no hazard),
// which costs an issue slot per pair.  The plain multiply and one wait state separate a VALU write of v from the
// first DPP read of it.
#define HOTK_DPP_ROR(K) " row_ror:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ float rot_dot(float v, const float (&rot)[NM]) {
  float a0, a1;
  asm volatile("v_mul_f32_e32 %0, %2, %3\n\t"
               "s_nop 0\n\t"
               "v_mul_f32_dpp %1, %2, %4" HOTK_DPP_ROR(1)
               "v_fmac_f32_dpp %0, %2, %5" HOTK_DPP_ROR(2)
               "v_fmac_f32_dpp %1, %2, %6" HOTK_DPP_ROR(3)
               "v_fmac_f32_dpp %0, %2, %7" HOTK_DPP_ROR(4)
               "v_fmac_f32_dpp %1, %2, %8" HOTK_DPP_ROR(5)
               "v_fmac_f32_dpp %0, %2, %9" HOTK_DPP_ROR(6)
               "v_fmac_f32_dpp %1, %2, %10" HOTK_DPP_ROR(7)
               "v_fmac_f32_dpp %0, %2, %11" HOTK_DPP_ROR(8)
               "v_fmac_f32_dpp %1, %2, %12" HOTK_DPP_ROR(9)
               "v_fmac_f32_dpp %0, %2, %13" HOTK_DPP_ROR(10)
               "v_fmac_f32_dpp %1, %2, %14" HOTK_DPP_ROR(11)
               "v_fmac_f32_dpp %0, %2, %15" HOTK_DPP_ROR(12)
               "v_fmac_f32_dpp %1, %2, %16" HOTK_DPP_ROR(13)
               "v_fmac_f32_dpp %0, %2, %17" HOTK_DPP_ROR(14)
               "v_fmac_f32_dpp %1, %2, %18" HOTK_DPP_ROR(15)
               "v_add_f32_e32 %0, %0, %1"
               : "=&v"(a0), "=&v"(a1)
               : "v"(v), "v"(rot[0]), "v"(rot[1]), "v"(rot[2]), "v"(rot[3]), "v"(rot[4]), "v"(rot[5]), "v"(rot[6]),
                 "v"(rot[7]), "v"(rot[8]), "v"(rot[9]), "v"(rot[10]), "v"(rot[11]), "v"(rot[12]), "v"(rot[13]),
                 "v"(rot[14]), "v"(rot[15]));
  return a0;
}

// The same contraction when only EIGHT outputs per head exist (d(raw q), d(raw k), dV: lanes x and x + 8 both own output
// x & 7): lane x sums the eight features {x, x-1, .., x-7} (mod 16), lane x + 8 the other eight, and one more rotation
// by 8 adds the two halves -- 10 instructions instead of 17 and tables of 8 instead of 16 registers.
constexpr int NR8 = 8;
__device__ __forceinline__ float rot_dot8(float v, const float (&rot)[NR8]) {
  float a0, a1;
  asm volatile("v_mul_f32_e32 %0, %2, %3\n\t"
               "s_nop 0\n\t"
               "v_mul_f32_dpp %1, %2, %4" HOTK_DPP_ROR(1)
               "v_fmac_f32_dpp %0, %2, %5" HOTK_DPP_ROR(2)
               "v_fmac_f32_dpp %1, %2, %6" HOTK_DPP_ROR(3)
               "v_fmac_f32_dpp %0, %2, %7" HOTK_DPP_ROR(4)
               "v_fmac_f32_dpp %1, %2, %8" HOTK_DPP_ROR(5)
               "v_fmac_f32_dpp %0, %2, %9" HOTK_DPP_ROR(6)
               "v_fmac_f32_dpp %1, %2, %10" HOTK_DPP_ROR(7)
               "v_add_f32_e32 %0, %0, %1"
               : "=&v"(a0), "=&v"(a1)
               : "v"(v), "v"(rot[0]), "v"(rot[1]), "v"(rot[2]), "v"(rot[3]), "v"(rot[4]), "v"(rot[5]), "v"(rot[6]),
                 "v"(rot[7]));
  return a0 + ror<8>(a0);
}

struct Lane {
  int lane, h, x;
  float ps[DH];  // F layout: P[x][:] * d^-1/4
};
#ifndef HOTK_LDS_DOT
#define HOTK_LDS_DOT 1
#endif
struct LaneB : Lane {
  // projection table of the contraction d(raw)[x & 7] = sum_mm dpre[mm] P[mm][x & 7] (backward kernels only):
  // HOTK_LDS_DOT: the lane's half of the features, P[(x & 8) + k][x & 7]; else rotated, P[(x - k) & 15][x & 7], k < 8
  float pr[NR8];
  float* cb;      // LDS form: 64 floats of wave-private LDS through which a head's 16 values meet
  bool lds;       // which form this kernel's contractions take (a compile-time constant after inlining)
};

constexpr float PSCALE = 0.59460355750136051f;  // 8^-1/4

__device__ __forceinline__ void init_lane(Lane& L, const float* __restrict__ proj) {
  L.lane = threadIdx.x & 63;
  L.h = L.lane >> 4;
  L.x = L.lane & 15;
#pragma unroll
  for (int j = 0; j < DH; ++j) L.ps[j] = proj[L.x * DH + j] * PSCALE;
}
__device__ __forceinline__ Lane make_lane(const float* __restrict__ proj) {
  Lane L;
  init_lane(L, proj);
  return L;
}
__device__ __forceinline__ LaneB make_lane_b(const float* __restrict__ proj, float* cb, bool lds) {
  LaneB L;
  init_lane(L, proj);
  L.cb = cb;
  L.lds = lds && HOTK_LDS_DOT;
#pragma unroll
  for (int k = 0; k < NR8; ++k)
    L.pr[k] = proj[(L.lds ? (L.x & 8) + k : ror_src(L.x, k)) * DH + (L.x & 7)] * PSCALE;
  return L;
}
// out[x & 7] = sum over the head's 16 lanes mm of v[mm] * T[mm][x & 7], on every lane of the head.  A DPP operand costs a
// second issue slot: the rotation form is 1 + 7 DPP multiply-adds + 2 adds = 18 slots.  In the LDS form the head's values
// meet in LDS instead: one ds_write_b32, two ds_read_b128 (the lane's half of the features, broadcast reads), 8 plain
// multiply-adds and the cross-half add = 11 vector slots + 3 LDS instructions (tab[k] = T[(x & 8) + k][x & 7]).  Measured
// with the slots full: the streaming per-graph kernels gain (k_graphsum_bwd 2.9 -> 2.7 ms, k_graph_apply_bwd 2.3 -> 2.1),
// the node-segment kernels lose a little (k_seg_bwd 5.65 -> 5.8 ms: their LDS staging is already busy) and keep DPP.
__device__ __forceinline__ float contract8(const LaneB& L, float v, const float (&tab)[NR8]) {
  if (!L.lds) return rot_dot8(v, tab);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  L.cb[L.lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float* p = L.cb + (L.lane & ~7);
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  float s = a.x * tab[0];
  s = fmaf(a.y, tab[1], s); s = fmaf(a.z, tab[2], s); s = fmaf(a.w, tab[3], s);
  s = fmaf(b.x, tab[4], s); s = fmaf(b.y, tab[5], s); s = fmaf(b.z, tab[6], s); s = fmaf(b.w, tab[7], s);
  return s + ror<8>(s);
}

__device__ __forceinline__ float phi_of(float pre) { return fmaxf(pre, 0.f) + PHI_EPS; }

// pre-activation of feature (h, x) from the 8-wide head slice
__device__ __forceinline__ float pre_x(const Lane& L, const float (&x)[DH]) {
  float s = x[0] * L.ps[0];
#pragma unroll
  for (int j = 1; j < DH; ++j) s = fmaf(x[j], L.ps[j], s);
  return s;
}

// d(raw q or k)[x & 7] = sum_mm dpre[mm] P[mm][x & 7] (all 16 lanes of the head hold it)
__device__ __forceinline__ float proj_back(const LaneB& L, float dpre) { return contract8(L, dpre, L.pr); }

// the 8-float slice of head h of a 32-float slot staged in the wave's LDS row
__device__ __forceinline__ void slice8(const Lane& L, const float* lds_slot, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(lds_slot + L.h * 8);
  const float4 b = *reinterpret_cast<const float4*>(lds_slot + L.h * 8 + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// The staging rows are private to the wave and the LDS executes one wave's instructions in order, so only the
// COMPILER has to be kept from moving the cross-lane reads over the writes: wavefront-scope fences.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int NR>
__device__ __forceinline__ void stage_put(const Lane& L, float* stage, const float (&r)[NR]) {
  wave_fence();
#pragma unroll
  for (int t = 0; t < NR; ++t) stage[t * 64 + L.lane] = r[t];
  wave_fence();
}

// KV accumulate (F layout): kv[0..7] += phi * val, kv[8] += phi
__device__ __forceinline__ void kv_acc(float (&kv)[NC + 1], float phi, const float (&val)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) kv[c] = fmaf(phi, val[c], kv[c]);
  kv[NC] += phi;
}
// the same with the value channels scaled by s (0 when the transposed entry is absent): the normaliser channel
// still counts the key (kernelattncoef.py:103-122 sums keys over every entry of the column)
__device__ __forceinline__ void kv_acc_scaled(float (&kv)[NC + 1], float phi, const float (&val)[NC], float s) {
  const float ps = phi * s;
#pragma unroll
  for (int c = 0; c < NC; ++c) kv[c] = fmaf(ps, val[c], kv[c]);
  kv[NC] += phi;
}

// F layout -> C layout through the wave's LDS scratch (4 heads x 32 rows x 9 floats): every feature row is written
// twice, 16 rows apart, so that row (x - k) & 15 is the plain row x + 16 - k and all 16 reads use immediate offsets.
// DUP: lanes 8..15 repeat channels 0..7 (gradient of the values) instead of holding the normaliser channel.
constexpr int KVBUF_FLOATS = NH * 32 * (NC + 1);
template <bool DUP, int NK = DUP ? NR8 : NM>   // DUP tables feed rot_dot8: rotations 0..7 only
__device__ __forceinline__ void kv_to_rot(const Lane& L, float* kvbuf, const float (&kv)[NC + 1], float (&rot)[NK]) {
  wave_fence();
  float* w = kvbuf + (L.h * 32 + L.x) * (NC + 1);
#pragma unroll
  for (int c = 0; c <= NC; ++c) {
    w[c] = kv[c];
    w[16 * (NC + 1) + c] = kv[c];
  }
  wave_fence();
  const int cidx = DUP ? (L.x & 7) : (L.x < NC ? L.x : NC);
  if (HOTK_ROR_DIR < 0) {
    const float* r = kvbuf + (L.h * 32 + L.x + 1) * (NC + 1) + cidx;
#pragma unroll
    for (int k = 0; k < NK; ++k) rot[k] = r[(15 - k) * (NC + 1)];
  } else {
    const float* r = kvbuf + (L.h * 32 + L.x) * (NC + 1) + cidx;
#pragma unroll
    for (int k = 0; k < NK; ++k) rot[k] = r[k * (NC + 1)];
  }
}

// O = phi(q).KV / guard(phi(q).ks) for one slot (C layout: lanes x < 8 hold the 8 outputs of head h, lanes x >= 8 the
// normaliser); writes 32 floats at out
template <class OT>
__device__ __forceinline__ void apply_fwd(const Lane& L, const float (&q)[DH], const float (&rot)[NM],
                                          OT* __restrict__ out) {
  const float phi = phi_of(pre_x(L, q));
  const float num = rot_dot(phi, rot);
  // The reference replaces a zero normaliser by 1e-5 (kernelattncoef.py:68-71).  phi > 0 and the key sums are >= 0,
  // so the normaliser is zero only when the key set is empty, and then the numerator is zero as well: clamping from
  // below gives the same quotient (0) with one instruction instead of a compare and a select.
  const float den = fmaxf(ror<8>(num), 1e-30f);
  const float o = num * __builtin_amdgcn_rcpf(den);
  if (L.x < NC) stS(out + L.h * NC + L.x, o);
}

// backward of apply_fwd: writes d(raw q) (32 floats at dq) and accumulates dkv (F layout)
template <class OT>
__device__ __forceinline__ void apply_bwd(const LaneB& L, const float (&q)[DH], const float (&kv)[NC + 1],
                                          const float (&g)[NC], OT* __restrict__ dq, float (&dkv)[NC + 1]) {
  const float pre = pre_x(L, q);
  const float phi = phi_of(pre);
  float s = kv[0] * g[0];
#pragma unroll
  for (int c = 1; c < NC; ++c) s = fmaf(kv[c], g[c], s);
  float den = head_allsum(phi * kv[NC]);
  const float dot = head_allsum(phi * s);
  const bool guarded = (den == 0.f);
  if (guarded) den = 1e-5f;
  const float inv = __builtin_amdgcn_rcpf(den);
  const float dden = guarded ? 0.f : -dot * inv * inv;
  const float dphi = fmaf(kv[NC], dden, s * inv);
  const float pi = phi * inv;
#pragma unroll
  for (int c = 0; c < NC; ++c) dkv[c] = fmaf(pi, g[c], dkv[c]);
  dkv[NC] = fmaf(phi, dden, dkv[NC]);
  const float dpre = pre > 0.f ? dphi : 0.f;
  const float d = proj_back(L, dpre);
  stS(dq + L.h * DH + (L.x & 7), d);   // all 16 lanes of a head hold output x & 7: lanes x and x + 8 store the same value
}

// backward of one KV accumulation term: dV (32 floats at dv, may be null; drot = C layout of dKV with DUP) and
// d(raw key) (32 floats at dk)
template <class VT, class KT>
__device__ __forceinline__ void kv_bwd(const LaneB& L, const float (&k)[DH], const float (&val)[NC],
                                       const float (&dkv)[NC + 1], const float (&drot)[NR8], VT* __restrict__ dv,
                                       KT* __restrict__ dk) {
  const float pre = pre_x(L, k);
  const float phi = phi_of(pre);
  float dphi = dkv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) dphi = fmaf(dkv[c], val[c], dphi);
  const float dvv = contract8(L, phi, drot);
  if (dv) stS(dv + L.h * NC + (L.x & 7), dvv);
  const float dpre = pre > 0.f ? dphi : 0.f;
  const float d = proj_back(L, dpre);
  stS(dk + L.h * DH + (L.x & 7), d);
}

// the same for a key whose feature phi comes from the per-node table (pre > 0  <=>  phi > eps)
template <class VT, class KT>
__device__ __forceinline__ void kv_bwd_phi(const LaneB& L, float phi, const float (&val)[NC], const float (&dkv)[NC + 1],
                                           const float (&drot)[NR8], VT* __restrict__ dv, KT* __restrict__ dk) {
  float dphi = dkv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) dphi = fmaf(dkv[c], val[c], dphi);
  const float dvv = contract8(L, phi, drot);
  if (dv) stS(dv + L.h * NC + (L.x & 7), dvv);
  const float dpre = phi > PHI_EPS ? dphi : 0.f;
  const float d = proj_back(L, dpre);
  stS(dk + L.h * DH + (L.x & 7), d);
}

// the same for an entry that enters the column sum with weight w (split-phase mode: the replicated diagonal entry)
template <class VT, class KT>
__device__ __forceinline__ void kv_bwd_phi_w(const LaneB& L, float phi, float w, const float (&val)[NC],
                                             const float (&dkv)[NC + 1], const float (&drot)[NR8], VT* __restrict__ dv,
                                             KT* __restrict__ dk) {
  float dphi = dkv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) dphi = fmaf(dkv[c], val[c], dphi);
  const float dvv = contract8(L, phi * w, drot);
  if (dv) stS(dv + L.h * NC + (L.x & 7), dvv);
  const float dpre = phi > PHI_EPS ? dphi * w : 0.f;
  const float d = proj_back(L, dpre);
  stS(dk + L.h * DH + (L.x & 7), d);
}

template <class OT>
__device__ __forceinline__ void zero_slots(const Lane& L, OT* __restrict__ p, int nslots) {
  for (int i = L.lane; i < nslots * SLOT; i += 64) stS(p + i, 0.f);
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

// depth-8 variant for light consumers (the MFMA accumulation: three loads and two instructions per item)
template <int NR, class LoadF, class UseF>
__device__ __forceinline__ void pipeline8(int cnt, LoadF load, UseF use) {
  if (cnt <= 0) return;
  const int last = cnt - 1;
  float r0[NR], r1[NR], r2[NR], r3[NR], r4[NR], r5[NR], r6[NR], r7[NR];
#define HOTK_PL(i, r) load((i) < last ? (i) : last, r)
  HOTK_PL(0, r0); HOTK_PL(1, r1); HOTK_PL(2, r2); HOTK_PL(3, r3); HOTK_PL(4, r4); HOTK_PL(5, r5); HOTK_PL(6, r6);
  for (int k = 0; k < cnt; k += 8) {
    HOTK_PL(k + 7, r7);
    use(k, r0);
    HOTK_PL(k + 8, r0);
    if (k + 1 < cnt) use(k + 1, r1);
    HOTK_PL(k + 9, r1);
    if (k + 2 < cnt) use(k + 2, r2);
    HOTK_PL(k + 10, r2);
    if (k + 3 < cnt) use(k + 3, r3);
    HOTK_PL(k + 11, r3);
    if (k + 4 < cnt) use(k + 4, r4);
    HOTK_PL(k + 12, r4);
    if (k + 5 < cnt) use(k + 5, r5);
    HOTK_PL(k + 13, r5);
    if (k + 6 < cnt) use(k + 6, r6);
    HOTK_PL(k + 14, r6);
    if (k + 7 < cnt) use(k + 7, r7);
  }
#undef HOTK_PL
}

// depth-2 variant for register-heavy consumers
template <int NR, class LoadF, class UseF>
__device__ __forceinline__ void pipeline2(int cnt, LoadF load, UseF use) {
  if (cnt <= 0) return;
  const int last = cnt - 1;
  float r0[NR], r1[NR];
  load(0, r0);
  for (int k = 0; k < cnt; k += 2) {
    load(k + 1 < last ? k + 1 : last, r1);
    use(k, r0);
    load(k + 2 < last ? k + 2 : last, r0);
    if (k + 1 < cnt) use(k + 1, r1);
  }
}

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
// element offset of row `row` (>= 0) for a leading dimension below 2^31 (checked on the host): a 32 x 32 -> 64 bit
// product is two scalar instructions, the generic 64 x 64 one seven
__device__ __forceinline__ int64_t rowoff(int32_t row, int64_t ld) {
  return (int64_t)((uint64_t)(uint32_t)row * (uint64_t)(uint32_t)ld);
}

// per-graph sums <-> memory: slot base = [KVG_C][64] floats, channels 0..8 F layout, 9..24 rotated
__device__ __forceinline__ const float* kvg_slot(const float* base, int64_t b, int s) {
  return base + ((b * 4 + s) * KVG_C) * 64;
}
__device__ __forceinline__ void kvg_load(const Lane& L, const float* __restrict__ slot, float (&kv)[NC + 1]) {
#pragma unroll
  for (int c = 0; c <= NC; ++c) kv[c] = slot[c * 64 + L.lane];
}
template <int NK>
__device__ __forceinline__ void kvg_load_rot(const Lane& L, const float* __restrict__ slot, float (&rot)[NK]) {
#pragma unroll
  for (int k = 0; k < NK; ++k) rot[k] = slot[(NC + 1 + k) * 64 + L.lane];
}
// the 8-entry table of contract8 for a per-graph sum in memory: the lane's half of the features from the feature-major
// channels (HOTK_LDS_DOT), else the rotated copies
__device__ __forceinline__ void kvg_load_tab8(const LaneB& L, const float* __restrict__ slot, float (&tab)[NR8]) {
  if (L.lds) {
    const float* p = slot + (L.x & 7) * 64 + L.h * 16 + (L.x & 8);
#pragma unroll
    for (int k = 0; k < NR8; ++k) tab[k] = p[k];
  } else {
    kvg_load_rot<NR8>(L, slot, tab);
  }
}
// partial sums of one chunk: [4][9][64]
__device__ __forceinline__ void part_store(const Lane& L, float* __restrict__ base, const float (&kv)[NC + 1]) {
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
// index of this wave in the grid, as a value the compiler knows to be wave-uniform (everything indexed by it --
// segment pointers, node rows, graph ids -- then lives in scalar registers and is fetched by scalar loads)
__device__ __forceinline__ int64_t wave_index() {
  return (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WPB + (threadIdx.x >> 6)));
}
// ---- work distribution --------------------------------------------------------------------------------------------
// One single-wave workgroup per node leaves the chip at ~2.4 of the 4 waves per SIMD the registers allow (measured:
// SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE; 978 k workgroups of ~20 us each: slots are refilled slower than waves retire).  A
// persistent grid with a static stride fills the slots but scatters the nodes in flight over the whole batch and loses
// the L2 reuse between neighbouring nodes (8.1 -> 14.6 ms).  Persistent waves that draw TICKETS -- TN consecutive items
// per atomic, in item order -- keep both: full slots and a moving window of neighbouring nodes (k_seg_bwd 8.1 -> 5.7 ms).
// One atomic per NODE on one address would run at ~86 per microsecond (11.4 ms per kernel): hence TN items per ticket.
// The next ticket is requested when the current one starts and read when it ends.  ticket == nullptr: static stride.
#ifndef HOTK_DYN
#define HOTK_DYN 1
#endif
#ifndef HOTK_TICKET_MIN_NODES
#define HOTK_TICKET_MIN_NODES 300000
#endif
#ifndef HOTK_TICKETS_PER_WAVE
#define HOTK_TICKETS_PER_WAVE 16
#endif
#ifndef HOTK_TICKET_NODES
#define HOTK_TICKET_NODES 8
#endif
#ifndef HOTK_DYN_GRID
#define HOTK_DYN_GRID (256 * 4 * 6)   /* persistent workgroups: 6 per SIMD cover every register budget of these kernels */
#endif
template <int TN>
struct WorkIter {
  unsigned* t;
  int64_t total, cur, end;
  unsigned tv;
  int tn;      // items per ticket: at most TN, fewer when that would leave a wave under ~16 tickets (few long segments:
               // one graph of 125 k nodes with 50 entries each ran 5 % slower with 8-node tickets than with a static stride)
  __device__ __forceinline__ WorkIter(unsigned* __restrict__ ticket, int64_t n_items) : t(ticket), total(n_items) {
    const int64_t per = n_items / ((int64_t)gridDim.x * WPB * HOTK_TICKETS_PER_WAVE);
    tn = per < 1 ? 1 : (per > TN ? TN : (int)per);
    if (t) {
      cur = end = 0;
      tv = request();
    } else {
      cur = wave_index() - (int64_t)gridDim.x * WPB;
      end = 0;
      tv = 0;
    }
  }
  __device__ __forceinline__ unsigned request() {
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(t, 1u);
    return v;
  }
  // the next item of this wave, or false when the work is exhausted (every wave gets there: tickets only grow)
  __device__ __forceinline__ bool next(int64_t& item) {
    if (!t) {
      cur += (int64_t)gridDim.x * WPB;
      item = cur;
      return cur < total;
    }
    if (cur == end) {
      cur = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tv) * tn;
      if (cur >= total) return false;
      end = cur + tn < total ? cur + tn : total;
      tv = request();
    }
    item = cur++;
    return true;
  }
};
constexpr size_t TICKET_STRIDE = 16;            // unsigned per counter: one 64-byte line each
constexpr size_t TICKET_BYTES = 16 * TICKET_STRIDE * sizeof(unsigned);

#ifndef HOTK_GRID_CAP
#define HOTK_GRID_CAP (1 << 20)
#endif
constexpr int64_t GRID_CAP = HOTK_GRID_CAP;

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

// rows [ra, rb) of graph b, all diagonal (DIAG) or all off-diagonal.  Diagonal rows take phi(key) from the per-node
// table (slots 6, 7), off-diagonal rows compute it from the raw entry-level key.
template <class ST, bool DIAG>
__device__ __forceinline__ void graphsum_rows(const Lane& L, const A& a, float* stage, int64_t b, int64_t ra, int64_t rb,
                                              float (&kv)[4][NC + 1]) {
  const bool lo = L.lane < 32;
  const SV<ST> S(a);
  auto load = [&](int k, float (&r)[3]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    const ST* vr = S.v + row * a.ld_v;
    if (DIAG) {
      const int64_t n = b * a.Nmax + (row - b * a.Emax);
      r[0] = a.phik[n * a.ld_phik + 6 * 64 + L.lane];
      r[1] = a.phik[n * a.ld_phik + 7 * 64 + L.lane];
      r[2] = ldS(vr + (lo ? 6 * SLOT : 7 * SLOT) + L.lane);          // slot 6 | slot 8
    } else {
      r[0] = ldS(S.h2 + row * a.ld_h2 + K2 * SLOT + L.lane);
      r[1] = 0.f;
      r[2] = ldS(vr + (lo ? 7 * SLOT : 8 * SLOT) + L.lane);          // slot 7 | slot 9
    }
  };
  auto use = [&](int, const float (&r)[3]) __attribute__((always_inline)) {
    float x[8], val[8];
    if (DIAG) {
      const float rv[1] = {r[2]};
      stage_put<1>(L, stage, rv);
      slice8(L, stage, val);
      kv_acc(kv[0], r[0], val);
      slice8(L, stage + 32, val);
      kv_acc(kv[2], r[1], val);
    } else {
      const float rv[2] = {r[0], r[2]};
      stage_put<2>(L, stage, rv);
      slice8(L, stage, x); slice8(L, stage + 64, val);
      kv_acc(kv[1], phi_of(pre_x(L, x)), val);
      slice8(L, stage + 32, x); slice8(L, stage + 96, val);
      kv_acc(kv[3], phi_of(pre_x(L, x)), val);
    }
  };
  pipeline4<3>((int)(rb - ra), load, use);
}

template <class ST>
__global__ __launch_bounds__(TPB) void k_graphsum_fwd(A a, float* __restrict__ dst, int cpg, int rows_per_chunk, unsigned* __restrict__ ticket) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  const int64_t nwork = a.B * cpg;
  WorkIter<1> it(ticket, nwork);
  int64_t w;
  while (it.next(w)) {
    int64_t b, r0, r1;
    chunk_rows(a, w, cpg, rows_per_chunk, &b, &r0, &r1);
    const int64_t nodes_end = b * a.Emax + a.n_nodes[b];   // rows below are the diagonal (layout invariant)
    const int64_t rm = r1 < nodes_end ? r1 : (r0 > nodes_end ? r0 : nodes_end);
    float kv[4][NC + 1];
    kv_zero<4>(kv);
    graphsum_rows<ST, true>(L, a, stage, b, r0, rm, kv);
    graphsum_rows<ST, false>(L, a, stage, b, rm, r1, kv);
#pragma unroll
    for (int s = 0; s < 4; ++s) part_store(L, dst + ((int64_t)w * 4 + s) * (NC + 1) * 64, kv[s]);
  }
}

// out[b][s][c][lane] = sum_k part[b*cpg + k][s][c][lane] for c < 9 (fixed order: deterministic), followed by the
// rotated copy  out[b][s][9 + k][(h, x)] = out[b][s][cidx(x)][(h, (x - k) & 15)]
__global__ void k_graph_reduce(const float* __restrict__ part, float* __restrict__ out, int64_t B, int cpg, int dup) {
  const int64_t X = (NC + 1) * 64;
  for (int64_t bs = blockIdx.x; bs < B * 4; bs += gridDim.x) {
    const int64_t b = bs >> 2;
    const int s = (int)(bs & 3);
    float* o = out + bs * KVG_C * 64;
    for (int64_t x = threadIdx.x; x < X; x += blockDim.x) {
      float acc = 0.f;
      for (int k = 0; k < cpg; ++k) acc += part[(((int64_t)b * cpg + k) * 4 + s) * X + x];
      o[x] = acc;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NM * 64; t += blockDim.x) {
      const int k = t >> 6, lane = t & 63, h = lane >> 4, xx = lane & 15;
      const int cidx = dup ? (xx & 7) : (xx < NC ? xx : NC);
      o[(NC + 1 + k) * 64 + lane] = o[cidx * 64 + h * 16 + ror_src(xx, k)];
    }
    __syncthreads();
  }
}

// The same for graphs with many chunks (one giant graph: B = 1, cpg = 4096): a block per (graph, slot, channel c), its
// RG wave-sized groups each add every RG-th partial of the channel's 64 lanes, the groups' sums are added in group
// order (deterministic), and the block writes the rotated copies that are taken from channel c.
constexpr int RG = 16;
__global__ __launch_bounds__(64 * RG) void k_graph_reduce_wide(const float* __restrict__ part, float* __restrict__ out,
                                                               int64_t B, int cpg, int dup) {
  __shared__ float sums[RG][64];
  const int64_t X = (NC + 1) * 64;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int64_t w = blockIdx.x; w < B * 4 * (NC + 1); w += gridDim.x) {
    const int64_t bs = w / (NC + 1);
    const int c = (int)(w - bs * (NC + 1));
    const int64_t b = bs >> 2;
    const int s = (int)(bs & 3);
    float acc = 0.f;
    for (int k = grp; k < cpg; k += RG) acc += part[((b * cpg + k) * 4 + s) * X + c * 64 + lane];
    sums[grp][lane] = acc;
    __syncthreads();
    if (grp == 0) {
      float t = sums[0][lane];
#pragma unroll
      for (int g = 1; g < RG; ++g) t += sums[g][lane];
      sums[0][lane] = t;
      out[bs * KVG_C * 64 + c * 64 + lane] = t;
    }
    __syncthreads();
    // rotated copies out[b][s][9 + k][(h, x)] = sum[cidx(x)][(h, (x - k) & 15)]: the lanes whose cidx is this channel
    {
      const int k = grp, h = lane >> 4, xx = lane & 15;
      const int cidx = dup ? (xx & 7) : (xx < NC ? xx : NC);
      if (cidx == c) out[bs * KVG_C * 64 + (NC + 1 + k) * 64 + lane] = sums[0][h * 16 + ror_src(xx, k)];
    }
    __syncthreads();
  }
}
static_assert(RG == NM, "k_graph_reduce_wide: one group per rotation");

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

// three coalesced loads for entry e = (i, j) of a column segment: phi of key slots {2p, 2p+1} of node i (per-node table,
// F layout: the lane's own feature, no staging) and the value slots {2p | 2p+1}, the transposed one read from row te
template <class ST, int P>
__device__ __forceinline__ void load_kv3(const Lane& L, const A& a, int32_t e, int32_t i, int32_t te, float (&r)[3]) {
  const float* pk = a.phik + rowoff(i, a.ld_phik) + 2 * P * 64 + L.lane;
  r[0] = pk[0];
  r[1] = pk[64];
  const bool lo = L.lane < 32;
  // a missing transposed entry reads the entry's own row instead (any valid address) and is masked at USE time:
  // selecting on the loaded value here would make the prefetch synchronous
  const int32_t row = (P == 2) ? (lo ? te : e) : (lo ? e : te);
  r[2] = ldS((const ST*)a.v + rowoff(row < 0 ? e : row, a.ld_v) + 2 * P * SLOT + L.lane);
}

// per_entry(e, te, phi0, phi1, w): the two value slices are staged at stage[0..32) and stage[32..64); w is the weight of
// the entry in the column sums: 1, or a.diag_w for the diagonal entry in split-phase (edge-shard) mode (W)
template <class ST, int P, bool W = false, class F>
__device__ __forceinline__ void for_col_entries(const Lane& L, const A& a, float* stage, int32_t n, F per_entry) {
  const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
  for (int32_t base = c0; base < c1; base += 64) {
    const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
    const ColBlock cb = col_block(L, a, base, cnt);
    auto load = [&](int k, float (&r)[3]) __attribute__((always_inline)) { load_kv3<ST, P>(L, a, rl(cb.e, k), rl(cb.i, k), rl(cb.te, k), r); };
    auto use = [&](int k, const float (&r)[3]) __attribute__((always_inline)) {
      const float rv[1] = {r[2]};
      stage_put<1>(L, stage, rv);
      per_entry(rl(cb.e, k), rl(cb.te, k), r[0], r[1], (W && rl(cb.kd, k) == 1) ? a.diag_w : 1.f);
    };
    pipeline4<3>(cnt, load, use);
  }
}

template <class ST, int P, bool W = false>
__device__ __forceinline__ void seg_accumulate(const Lane& L, const A& a, float* stage, int32_t n, float (&kv)[2][NC + 1]) {
  kv_zero<2>(kv);
  for_col_entries<ST, P, W>(L, a, stage, n, [&](int32_t, int32_t te, float phi0, float phi1, float w) __attribute__((always_inline)) {
    float val[8];
    if (W) { phi0 *= w; phi1 *= w; }   // weights values and normaliser alike
    const float ts = te < 0 ? 0.f : 1.f;   // the transposed-value slot is slot 0 for P == 2, slot 1 otherwise
    slice8(L, stage + 0 * SLOT, val);
    if (P == 2) kv_acc_scaled(kv[0], phi0, val, ts);
    else kv_acc(kv[0], phi0, val);
    slice8(L, stage + 1 * SLOT, val);
    if (P == 2) kv_acc(kv[1], phi1, val);
    else kv_acc_scaled(kv[1], phi1, val, ts);
  });
}

// split-phase mode: the two sums of (node n, part P) in memory, [9][64] floats each, feature-major like part_store
constexpr int64_t KVN_SLOT = (NC + 1) * 64;
template <int P>
__device__ __forceinline__ float* kvn_slot(const A& a, int64_t n, int s) { return a.kvn + ((P * a.Nn + n) * 2 + s) * KVN_SLOT; }
template <int P>   // d(sums) of parts 1 and 2 only: part 0 exchanges [d num | d den] of the node queries instead (dd0)
__device__ __forceinline__ float* dkvn_slot(const A& a, int64_t n, int s) {
  static_assert(P >= 1, "part 0 has no dkvn");
  return a.dkvn + (((P - 1) * a.Nn + n) * 2 + s) * KVN_SLOT;
}

// phase FWD_NODESUM: the weighted local column sums of every valid node.  Parts 1 and 2 are applied to ENTRIES, which
// live on any rank: their sums are all-reduced.  Part 0 is applied to the node's own query, which every rank holds:
// phi(q) . KV is linear in KV, so a rank applies its PARTIAL sums and only [numerator | normaliser] (nd0: 64 floats per
// slot in the C layout instead of 576) is all-reduced; the partial sums stay local for the backward pass.
template <int P>
__global__ __launch_bounds__(TPB) void k_seg_kvsum(A a, unsigned* __restrict__ ticket) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  __shared__ __align__(16) float kvbuf_all[WPB][KVBUF_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  float* kvbuf = kvbuf_all[threadIdx.x >> 6];
  WorkIter<HOTK_TICKET_NODES> it(ticket, a.Nn);
  int64_t n;
  while (it.next(n)) {
    if (!a.nkind[n]) continue;
    float kv[2][NC + 1];
    seg_accumulate<float, P, true>(L, a, stage, (int32_t)n, kv);
    part_store(L, kvn_slot<P>(a, n, 0), kv[0]);
    part_store(L, kvn_slot<P>(a, n, 1), kv[1]);
    if (P == 0) {
      float rot0[NM], rot1[NM];
      kv_to_rot<false>(L, kvbuf, kv[0], rot0);
      kv_to_rot<false>(L, kvbuf, kv[1], rot1);
      const float* q = a.h1 + n * a.ld_h1;
      const float r[1] = {q[L.lane]};
      stage_put<1>(L, stage, r);
      float x[8];
      slice8(L, stage + 0 * SLOT, x);
      a.nd0[(n * 2 + 0) * 64 + L.lane] = rot_dot(phi_of(pre_x(L, x)), rot0);
      slice8(L, stage + 1 * SLOT, x);
      a.nd0[(n * 2 + 1) * 64 + L.lane] = rot_dot(phi_of(pre_x(L, x)), rot1);
    }
  }
}

// phase FWD_APPLY, part 0: terms 1, 2 of node n from the all-reduced [numerator | normaliser] (same quotient as apply_fwd)
__global__ __launch_bounds__(TPB) void k_node_apply_nd(A a) {
  const int lane = threadIdx.x & 63, h = lane >> 4, x = lane & 15;
  for (int64_t n = wave_index(); n < a.Nn; n += (int64_t)gridDim.x * WPB) {
    if (!a.nkind[n]) continue;
    float* o = a.att + rowoff(a.diag_pos[n], a.ld_att);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float num = a.nd0[(n * 2 + s) * 64 + lane];
      const float den = fmaxf(ror<8>(num), 1e-30f);
      const float v = num * __builtin_amdgcn_rcpf(den);
      if (x < NC) o[s * SLOT + h * NC + x] = v;
    }
  }
}

// phase BWD_QUERY, part 0: [d num | d den] of the node queries of THIS rank's (weighted) output gradient, from the
// all-reduced nd0 -- the quotient rule of apply_bwd; the caller all-reduces dd0 (the rule is linear in the gradient)
__global__ __launch_bounds__(TPB) void k_node_dd(A a) {
  const int lane = threadIdx.x & 63, h = lane >> 4, x = lane & 15;
  for (int64_t n = wave_index(); n < a.Nn; n += (int64_t)gridDim.x * WPB) {
    if (!a.nkind[n]) continue;
    const float* g = a.datt + rowoff(a.diag_pos[n], a.ld_datt);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float num = a.nd0[(n * 2 + s) * 64 + lane];
      float den = ror<8>(num);                                  // lanes x < 8: the normaliser of their head
      const float gc = x < NC ? g[s * SLOT + h * NC + x] : 0.f;
      const float dot = head_allsum(x < NC ? gc * num : 0.f);
      const bool guarded = (den == 0.f);
      if (guarded) den = 1e-5f;
      const float inv = __builtin_amdgcn_rcpf(den);
      const float dden = guarded ? 0.f : -dot * inv * inv;
      // lanes x >= 8 read the normaliser of lane x - 8 through ror<8>, i.e. a numerator: their dden is recomputed
      // from lane 0's value below
      const float dden0 = __shfl(dden, h * 16, 64);
      a.dd0[(n * 2 + s) * 64 + lane] = x < NC ? gc * inv : dden0;
    }
  }
}

// GT: also the per-graph terms and the zero fill (false: k_graph_apply_fwd_t does them)
// LOADKV: the node's sums come from a.kvn (split-phase mode) instead of being accumulated here
template <class ST, int P, bool GT = true, bool LOADKV = false>
__global__ __launch_bounds__(TPB) void k_seg_fwd(A a, unsigned* __restrict__ ticket) {
  const SV<ST> S(a);
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  __shared__ __align__(16) float kvbuf_all[WPB][KVBUF_FLOATS];
  const Lane L = make_lane(a.proj);
  float* stage = stage_all[threadIdx.x >> 6];
  float* kvbuf = kvbuf_all[threadIdx.x >> 6];
  WorkIter<HOTK_TICKET_NODES> it(ticket, a.Nn);
  int64_t n;
  while (it.next(n)) {
    if (!a.nkind[n]) continue;
    float rot0[NM], rot1[NM];
    {
      float kv[2][NC + 1];
      if (LOADKV) {
        kvg_load(L, kvn_slot<P>(a, n, 0), kv[0]);
        kvg_load(L, kvn_slot<P>(a, n, 1), kv[1]);
      } else {
        seg_accumulate<ST, P>(L, a, stage, (int32_t)n, kv);
      }
      kv_to_rot<false>(L, kvbuf, kv[0], rot0);
      kv_to_rot<false>(L, kvbuf, kv[1], rot1);
    }
    if (P == 0) {
      // node queries: terms 1, 2 (column sums) and 7, 8 (graph sums) go to the node's diagonal row
      const int64_t b = n / a.Nmax;
      const float* q = a.h1 + n * a.ld_h1;
      ST* o = S.att + rowoff(a.diag_pos[n], a.ld_att);
      float r[2] = {q[L.lane], q[64 + L.lane]};
      stage_put<2>(L, stage, r);
      float x[8];
      slice8(L, stage + 0 * SLOT, x); apply_fwd(L, x, rot0, o + 0 * SLOT);
      slice8(L, stage + 1 * SLOT, x); apply_fwd(L, x, rot1, o + 1 * SLOT);
      if (GT) {
        float g0[NM], g1[NM];
        kvg_load_rot(L, kvg_slot(a.kvg, b, 0), g0);
        kvg_load_rot(L, kvg_slot(a.kvg, b, 1), g1);
        slice8(L, stage + 2 * SLOT, x); apply_fwd(L, x, g0, o + 6 * SLOT);
        slice8(L, stage + 3 * SLOT, x); apply_fwd(L, x, g1, o + 7 * SLOT);
        zero_slots(L, o + 2 * SLOT, 4);
        zero_slots(L, o + 8 * SLOT, 2);
      }
    } else if (P == 1) {
      // entries of column n: terms 3, 5 (this node's sums) and 9, 10 (graph sums)
      const int64_t b = n / a.Nmax;
      float g0[NM], g1[NM];
      if (GT) {
        kvg_load_rot(L, kvg_slot(a.kvg, b, 2), g0);
        kvg_load_rot(L, kvg_slot(a.kvg, b, 3), g1);
      }
      const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
      for (int32_t base = c0; base < c1; base += 64) {
        const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
        const ColBlock cb = col_block(L, a, base, cnt);
        auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
          const ST* q = S.h2 + rowoff(rl(cb.e, k), a.ld_h2);
          r[0] = ldS(q + L.lane);               // query slots 0, 1
          r[1] = ldS(q + 4 * SLOT + L.lane);    // query slots 4, 5
        };
        auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
          if (rl(cb.kd, k) != 2) return;
          ST* o = S.att + rowoff(rl(cb.e, k), a.ld_att);
          stage_put<2>(L, stage, r);
          float x[8];
          slice8(L, stage + 0 * SLOT, x); apply_fwd(L, x, rot0, o + 2 * SLOT);
          slice8(L, stage + 1 * SLOT, x); apply_fwd(L, x, rot1, o + 4 * SLOT);
          if (GT) {
            slice8(L, stage + 2 * SLOT, x); apply_fwd(L, x, g0, o + 8 * SLOT);
            slice8(L, stage + 3 * SLOT, x); apply_fwd(L, x, g1, o + 9 * SLOT);
            stS(o + L.lane, 0.f);                 // terms 1, 2 live on the diagonal only
            stS(o + 6 * SLOT + L.lane, 0.f);      // terms 7, 8 too
          }
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
          r[0] = ldS(S.h2 + rowoff(e < 0 ? ~e : e, a.ld_h2) + 2 * SLOT + L.lane);
        };
        auto use = [&](int k, const float (&r)[1]) __attribute__((always_inline)) {
          const int32_t e = rl(rec, k);
          if (e < 0) return;
          ST* o = S.att + rowoff(e, a.ld_att);
          stage_put<1>(L, stage, r);
          float x[8];
          slice8(L, stage + 0 * SLOT, x); apply_fwd(L, x, rot0, o + 3 * SLOT);
          slice8(L, stage + 1 * SLOT, x); apply_fwd(L, x, rot1, o + 5 * SLOT);
        };
        pipeline4<1>(cnt, load, use);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-graph terms (backward): (1) query side + dKVg partials, (2) key/value side
// ------------------------------------------------------------------------------------------------
// rows [ra, rb) of graph b, all diagonal or all off-diagonal, against the two per-graph sums kv0 / kv1 of that kind
template <class ST, bool DIAG>
__device__ __forceinline__ void graph_apply_bwd_rows(const LaneB& L, const A& a, float* stage, int64_t b, int64_t ra,
                                                     int64_t rb, float* __restrict__ dst) {
  const SV<ST> S(a);
  if (rb <= ra) {
    float z[NC + 1];
#pragma unroll
    for (int c = 0; c <= NC; ++c) z[c] = 0.f;
    part_store(L, dst + (DIAG ? 0 : 2) * (NC + 1) * 64, z);
    part_store(L, dst + (DIAG ? 1 : 3) * (NC + 1) * 64, z);
    return;
  }
  float kv[2][NC + 1], dkv[2][NC + 1];
  kvg_load(L, kvg_slot(a.kvg, b, DIAG ? 0 : 2), kv[0]);
  kvg_load(L, kvg_slot(a.kvg, b, DIAG ? 1 : 3), kv[1]);
  kv_zero<2>(dkv);
  auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    const ST* g = S.datt + row * a.ld_datt;
    if (DIAG) {
      const int64_t n = b * a.Nmax + (row - b * a.Emax);
      r[0] = a.h1[n * a.ld_h1 + 2 * SLOT + L.lane];     // node query slots 2, 3
      r[1] = ldS(g + 6 * SLOT + L.lane);                 // d term 7, 8
    } else {
      r[0] = ldS(S.h2 + row * a.ld_h2 + 4 * SLOT + L.lane);   // entry query slots 4, 5
      r[1] = ldS(g + 8 * SLOT + L.lane);                       // d term 9, 10
    }
  };
  auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    stage_put<2>(L, stage, r);
    float x[8], g[8];
    if (DIAG) {
      float* dq = a.dh1 + (b * a.Nmax + (row - b * a.Emax)) * a.ld_dh1 + 2 * SLOT;
      slice8(L, stage, x); slice8(L, stage + 64, g);
      apply_bwd(L, x, kv[0], g, dq, dkv[0]);
      slice8(L, stage + 32, x); slice8(L, stage + 96, g);
      apply_bwd(L, x, kv[1], g, dq + SLOT, dkv[1]);
    } else {
      ST* dq = S.dh2 + row * a.ld_dh2 + 4 * SLOT;
      slice8(L, stage, x); slice8(L, stage + 64, g);
      apply_bwd(L, x, kv[0], g, dq, dkv[0]);
      slice8(L, stage + 32, x); slice8(L, stage + 96, g);
      apply_bwd(L, x, kv[1], g, dq + SLOT, dkv[1]);
    }
  };
  pipeline4<2>((int)(rb - ra), load, use);
  part_store(L, dst + (DIAG ? 0 : 2) * (NC + 1) * 64, dkv[0]);
  part_store(L, dst + (DIAG ? 1 : 3) * (NC + 1) * 64, dkv[1]);
}

template <class ST>
__global__ __launch_bounds__(TPB) void k_graph_apply_bwd(A a, float* __restrict__ dst, int cpg, int rows_per_chunk, unsigned* __restrict__ ticket) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  float* stage = stage_all[threadIdx.x >> 6];
  const LaneB L = make_lane_b(a.proj, stage + 4 * 64, true);
  const int64_t nwork = a.B * cpg;
  WorkIter<1> it(ticket, nwork);
  int64_t w;
  while (it.next(w)) {
    int64_t b, r0, r1;
    chunk_rows(a, w, cpg, rows_per_chunk, &b, &r0, &r1);
    const int64_t nodes_end = b * a.Emax + a.n_nodes[b];
    const int64_t rm = r1 < nodes_end ? r1 : (r0 > nodes_end ? r0 : nodes_end);
    float* d = dst + (int64_t)w * 4 * (NC + 1) * 64;
    graph_apply_bwd_rows<ST, true>(L, a, stage, b, r0, rm, d);
    graph_apply_bwd_rows<ST, false>(L, a, stage, b, rm, r1, d);
  }
}

// key / value side of the per-graph sums: rows [ra, rb) of one kind against dKVg slots (s0, s1)
#ifndef HOTK_GB_ROWS
#define HOTK_GB_ROWS 64   /* rows per wave: the two d(KV) sums (50 registers) are loaded once per chunk; 16: 4.4 ms, 32: 3.8, 64: 3.6 */
#endif
constexpr int GB_ROWS = HOTK_GB_ROWS, GB_ROWS_SMALL = 16, GB_ROWS_TINY = 4;   // SMALL: batches with too few 64-row chunks to fill the chip
template <class ST, bool DIAG>
__device__ __forceinline__ void graphsum_bwd_rows(const LaneB& L, const A& a, float* stage, int64_t b, int64_t ra,
                                                  int64_t rb) {
  if (rb <= ra) return;
  const bool lo = L.lane < 32;
  const SV<ST> S(a);
  float d0[NC + 1], d1[NC + 1], t0[NR8], t1[NR8];
  kvg_load(L, kvg_slot(a.dkvg, b, DIAG ? 0 : 1), d0);
  kvg_load(L, kvg_slot(a.dkvg, b, DIAG ? 2 : 3), d1);
  kvg_load_tab8(L, kvg_slot(a.dkvg, b, DIAG ? 0 : 1), t0);
  kvg_load_tab8(L, kvg_slot(a.dkvg, b, DIAG ? 2 : 3), t1);
  auto load = [&](int k, float (&r)[4]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    const ST* vr = S.v + row * a.ld_v;
    if (DIAG) {
      const int64_t n = b * a.Nmax + (row - b * a.Emax);
      r[0] = a.phik[n * a.ld_phik + 6 * 64 + L.lane];
      r[3] = a.phik[n * a.ld_phik + 7 * 64 + L.lane];
      r[1] = ldS(vr + (lo ? 6 * SLOT : 7 * SLOT) + L.lane);
    } else {
      r[0] = ldS(S.h2 + row * a.ld_h2 + K2 * SLOT + L.lane);
      r[3] = 0.f;
      r[1] = ldS(vr + (lo ? 7 * SLOT : 8 * SLOT) + L.lane);
    }
    r[2] = (a.dout && L.lane < a.D) ? a.dout[row * a.ld_dout + L.lane] : 0.f;
  };
  auto use = [&](int k, const float (&r)[4]) __attribute__((always_inline)) {
    const int64_t row = ra + k;
    ST* dvr = S.dv + row * a.ld_dv;
    ST* dh2r = S.dh2 + row * a.ld_dh2;
    float x[8], val[8];
    if (DIAG) {
      const float rv[1] = {r[1]};
      stage_put<1>(L, stage, rv);
      float* dkn = a.dh1 + (b * a.Nmax + (row - b * a.Emax)) * a.ld_dh1 + 10 * SLOT;
      slice8(L, stage, val);
      kv_bwd_phi(L, r[0], val, d0, t0, dvr + 6 * SLOT, dkn);
      slice8(L, stage + 32, val);
      kv_bwd_phi(L, r[3], val, d1, t1, dvr + 8 * SLOT, dkn + SLOT);
      if (lo) { stS(dvr + 7 * SLOT + L.lane, 0.f); stS(dvr + 9 * SLOT + L.lane, 0.f); }
      zero_slots(L, dh2r, 8);   // entry-level queries / keys are masked on the diagonal
    } else {
      stage_put<2>(L, stage, reinterpret_cast<const float (&)[2]>(r));
      slice8(L, stage, x); slice8(L, stage + 64, val);
      kv_bwd(L, x, val, d0, t0, dvr + 7 * SLOT, dh2r + K2 * SLOT);
      slice8(L, stage + 32, x); slice8(L, stage + 96, val);
      kv_bwd(L, x, val, d1, t1, dvr + 9 * SLOT, dh2r + (K2 + 1) * SLOT);
      if (lo) { stS(dvr + 6 * SLOT + L.lane, 0.f); stS(dvr + 8 * SLOT + L.lane, 0.f); }
    }
    // skip path: d h2[:, 8 slots:] = d out (zero when the caller routes that gradient elsewhere)
    if (L.lane < a.D) stS(dh2r + 8 * SLOT + L.lane, r[2]);
    for (int i = 64 + L.lane; i < a.D; i += 64) stS(dh2r + 8 * SLOT + i, a.dout ? a.dout[row * a.ld_dout + i] : 0.f);
  };
  pipeline4<4>((int)(rb - ra), load, use);
}

// wave per chunk of 16 consecutive rows of one kind of one graph; one instantiation per kind, so that only that kind's
// two dKVg slots (F layout + rotated) are live
template <class ST, bool DIAG, int ROWS = GB_ROWS>
__global__ __launch_bounds__(TPB) void k_graphsum_bwd(A a, unsigned* __restrict__ ticket) {
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  float* stage = stage_all[threadIdx.x >> 6];
  const LaneB L = make_lane_b(a.proj, stage + 4 * 64, true);
  const int64_t span = DIAG ? a.Nmax : a.Emax;
  const int64_t cpg = (span + ROWS - 1) / ROWS;
  const int64_t nwork = a.B * cpg;
  WorkIter<1> it(ticket, nwork);
  int64_t w;
  while (it.next(w)) {
    const int64_t b = w / cpg, k = w - b * cpg;
    const int64_t nn = a.n_nodes[b];
    const int64_t lo = b * a.Emax + (DIAG ? 0 : nn), hi = b * a.Emax + (DIAG ? nn : (int64_t)a.n_entries[b]);
    const int64_t r0 = lo + k * ROWS;
    const int64_t r1 = r0 + ROWS < hi ? r0 + ROWS : hi;
    graphsum_bwd_rows<ST, DIAG>(L, a, stage, b, r0, r1);
  }
}

// ------------------------------------------------------------------------------------------------
// node segments (backward): one wave per (node, part), same split as the forward
// ------------------------------------------------------------------------------------------------
#ifndef HOTK_SEGB_WAVES
#define HOTK_SEGB_WAVES 4
#endif
// MODE 0: the whole backward of a (node, part).  Split-phase mode: MODE 1 = query side only (the node's sums come from
// a.kvn, d(sums) go to a.dkvn for the caller to all-reduce), MODE 2 = key / value side only (d(sums) from a.dkvn),
// MODE 3 = key / value side of part 0 (d(sums) rebuilt from the all-reduced [d num | d den] of the node queries)
template <class ST, int P, int MODE = 0>
__global__ __launch_bounds__(TPB, HOTK_SEGB_WAVES) void k_seg_bwd(A a, unsigned* __restrict__ ticket) {
  const SV<ST> S(a);
  __shared__ __align__(16) float stage_all[WPB][STAGE_FLOATS];
  __shared__ __align__(16) float kvbuf_all[WPB][KVBUF_FLOATS];
  float* stage = stage_all[threadIdx.x >> 6];
  const LaneB L = make_lane_b(a.proj, stage + 4 * 64, false);
  float* kvbuf = kvbuf_all[threadIdx.x >> 6];
  const bool lo = L.lane < 32;
  WorkIter<HOTK_TICKET_NODES> it(ticket, a.Nn);
  int64_t n;
  while (it.next(n)) {
    if (!a.nkind[n]) continue;
    float dkv[2][NC + 1];
    if constexpr (MODE == 3) {
      // part 0, key / value side in split-phase mode: d(sums) = phi(q) (x) [d num | d den] from the all-reduced dd0; the
      // query gradient takes this rank's PARTIAL sums (the caller adds node gradients over ranks)
      static_assert(P == 0 || MODE != 3, "MODE 3 is part 0 only");
      float kv[2][NC + 1];
      kvg_load(L, kvn_slot<P>(a, n, 0), kv[0]);
      kvg_load(L, kvn_slot<P>(a, n, 1), kv[1]);
      const float r[3] = {a.h1[n * a.ld_h1 + L.lane], a.dd0[(n * 2 + 0) * 64 + L.lane], a.dd0[(n * 2 + 1) * 64 + L.lane]};
      stage_put<3>(L, stage, r);
      float* dq = a.dh1 + n * a.ld_dh1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float x[8];
        slice8(L, stage + s * SLOT, x);
        const float* dp = stage + 64 * (1 + s) + L.h * 16;     // [d num 0..7 | d den] of head h
        const float4 d0 = *reinterpret_cast<const float4*>(dp), d1 = *reinterpret_cast<const float4*>(dp + 4);
        const float dn[NC] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float dd = dp[NC];
        const float pre = pre_x(L, x);
        const float phi = phi_of(pre);
        float dphi = kv[s][NC] * dd;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          dphi = fmaf(kv[s][c], dn[c], dphi);
          dkv[s][c] = phi * dn[c];
        }
        dkv[s][NC] = phi * dd;
        const float d = proj_back(L, pre > 0.f ? dphi : 0.f);
        dq[s * SLOT + L.h * DH + (L.x & 7)] = d;
      }
    } else if constexpr (MODE == 2) {
      kvg_load(L, dkvn_slot<P>(a, n, 0), dkv[0]);
      kvg_load(L, dkvn_slot<P>(a, n, 1), dkv[1]);
    } else {
      kv_zero<2>(dkv);
      float kv[2][NC + 1];
      if (MODE == 1) {
        kvg_load(L, kvn_slot<P>(a, n, 0), kv[0]);
        kvg_load(L, kvn_slot<P>(a, n, 1), kv[1]);
      } else {
        seg_accumulate<ST, P>(L, a, stage, (int32_t)n, kv);
      }
      if (P == 0) {  // node queries (terms 1, 2)
        const int64_t dr = a.diag_pos[n];
        float r[2] = {a.h1[n * a.ld_h1 + L.lane], ldS(S.datt + rowoff((int32_t)dr, a.ld_datt) + L.lane)};
        float* dq = a.dh1 + n * a.ld_dh1;
        stage_put<2>(L, stage, r);
        float x[8], g[8];
        slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd(L, x, kv[0], g, dq, dkv[0]);
        slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd(L, x, kv[1], g, dq + SLOT, dkv[1]);
      } else if (P == 1) {  // entries of column n (terms 3, 5)
        const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
        for (int32_t base = c0; base < c1; base += 64) {
          const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
          const ColBlock cb = col_block(L, a, base, cnt);
          auto load = [&](int k, float (&r)[2]) __attribute__((always_inline)) {
            const int64_t e = rl(cb.e, k);
            r[0] = ldS(S.h2 + rowoff((int32_t)e, a.ld_h2) + L.lane);                                      // query slots 0, 1
            r[1] = ldS(S.datt + rowoff((int32_t)e, a.ld_datt) + (lo ? 2 * SLOT : 3 * SLOT) + L.lane);      // d term 3 | d term 5
          };
          auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
            if (rl(cb.kd, k) != 2) return;
            ST* dq = S.dh2 + rowoff(rl(cb.e, k), a.ld_dh2);
            stage_put<2>(L, stage, r);
            float x[8], g[8];
            slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd(L, x, kv[0], g, dq, dkv[0]);
            slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd(L, x, kv[1], g, dq + SLOT, dkv[1]);
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
            r[0] = ldS(S.h2 + rowoff((int32_t)e, a.ld_h2) + 2 * SLOT + L.lane);                            // query slots 2, 3
            r[1] = ldS(S.datt + rowoff((int32_t)e, a.ld_datt) + (lo ? 3 * SLOT : 4 * SLOT) + L.lane);      // d term 4 | d term 6
          };
          auto use = [&](int k, const float (&r)[2]) __attribute__((always_inline)) {
            const int32_t e = rl(rec, k);
            if (e < 0) return;
            ST* dq = S.dh2 + rowoff(e, a.ld_dh2) + 2 * SLOT;
            stage_put<2>(L, stage, r);
            float x[8], g[8];
            slice8(L, stage + 0 * SLOT, x); slice8(L, stage + 2 * SLOT, g); apply_bwd(L, x, kv[0], g, dq, dkv[0]);
            slice8(L, stage + 1 * SLOT, x); slice8(L, stage + 3 * SLOT, g); apply_bwd(L, x, kv[1], g, dq + SLOT, dkv[1]);
          };
          pipeline4<2>(cnt, load, use);
        }
      }
    }
    if constexpr (MODE == 1) {
      part_store(L, dkvn_slot<P>(a, n, 0), dkv[0]);
      part_store(L, dkvn_slot<P>(a, n, 1), dkv[1]);
      continue;
    }
    // key / value side of this part's two column sums: dV of the plain slot goes to row e, dV of the transposed
    // slot to row te; d(raw key) contributions go to ck[e] (summed per row node afterwards)
    float drot0[NR8], drot1[NR8];
    kv_to_rot<true>(L, kvbuf, dkv[0], drot0);
    kv_to_rot<true>(L, kvbuf, dkv[1], drot1);
    for_col_entries<ST, P, (MODE >= 2)>(L, a, stage, (int32_t)n, [&](int32_t e, int32_t te, float phi0, float phi1, float w) __attribute__((always_inline)) {
      ST* dve = S.dv + rowoff(e, a.ld_dv);
      ST* dvt = te >= 0 ? S.dv + rowoff(te, a.ld_dv) : nullptr;
      ST* ck = S.ck + rowoff(e, 6 * SLOT) + 2 * P * SLOT;
      constexpr int S0 = 2 * P, S1 = 2 * P + 1;
      ST* d0 = (P == 2) ? (dvt ? dvt + S0 * SLOT : nullptr) : dve + S0 * SLOT;
      ST* d1 = (P == 2) ? dve + S1 * SLOT : (dvt ? dvt + S1 * SLOT : nullptr);
      float val[8];
      slice8(L, stage + 0 * SLOT, val);
      if (P == 2 && te < 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) val[c] = 0.f;
      }
      if (MODE >= 2) kv_bwd_phi_w(L, phi0, w, val, dkv[0], drot0, d0, ck);
      else kv_bwd_phi(L, phi0, val, dkv[0], drot0, d0, ck);
      slice8(L, stage + 1 * SLOT, val);
      if (P != 2 && te < 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) val[c] = 0.f;
      }
      if (MODE >= 2) kv_bwd_phi_w(L, phi1, w, val, dkv[1], drot1, d1, ck + SLOT);
      else kv_bwd_phi(L, phi1, val, dkv[1], drot1, d1, ck + SLOT);
      if (te < 0 && lo) stS(dve + ((P == 2) ? S0 : S1) * SLOT + L.lane, 0.f);   // no transposed partner writes this slot
    });
  }
}

// ================================================================================================
// MFMA tile kernels (parts 1 and 2 of the node segments).  All small contractions of an entry -- the 8 -> 16 random
// feature projection and the 16-feature x 9-channel key-value products -- run on the matrix cores in exact f32
// (v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x1_4b_f32), 16 entries of a segment at a time:
//   "T layout" of a 16 x 16 matrix M[p][q]: lane (g, x) = (lane >> 4, lane & 15), register r holds M[4g + r][x].
//   One MFMA group computes  OUT[i][j] = sum_p TA[p][i] * TB[p][j]  from two T-layout operands and leaves OUT in the
//   T layout again, so a result feeds the next product that sums over its ROW index with no lane movement:
//     accumulate  KV[mm][c]  = sum_e phiK[e][mm] * [V | 1][e][c]   (4-block form: one instruction per entry, all heads)
//     project     PRE[mm][t] = sum_j P[mm][j] * q[t][j]            (entries t on the lanes)
//     apply       O[t][c]    = sum_mm phi(PRE)[mm][t] * KV[mm][c]  (channels on the lanes, entries in (g, r))
//   Key-value sums are accumulated in entry order by a k-ordered fmaf chain, as the VALU kernels do.
// Two slots are processed as a pair: slot A keeps its 8 value channels on lanes x < 8 and the normaliser on x >= 8,
// slot B the other way round, so that one full-width store writes both results.
// ================================================================================================
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct LaneT {
  int lane, g, x;
  float pt0, pt1;  // projection as the TA operand of PRE: P[x][2g], P[x][2g + 1] (scaled)
};
__device__ __forceinline__ LaneT make_lane_t(const float* __restrict__ proj) {
  LaneT L;
  L.lane = threadIdx.x & 63;
  L.g = L.lane >> 4;
  L.x = L.lane & 15;
  L.pt0 = proj[L.x * DH + 2 * L.g] * PSCALE;
  L.pt1 = proj[L.x * DH + 2 * L.g + 1] * PSCALE;
  return L;
}
__device__ __forceinline__ f4v mfma4(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v mfma4b(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v zero16() {
  f16v z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// the two column sums of part P for node n in the T layout (register 4h + r: feature 4g + r of head h, channel on x):
// kv0 with values on x < 8 / key sum on x >= 8, kv1 with values on x >= 8 / key sum on x < 8
template <class ST, int P>
__device__ __forceinline__ void seg_accumulate_t(const LaneT& L, const A& a, int32_t n, f16v& kv0, f16v& kv1) {
  kv0 = zero16();
  kv1 = zero16();
  const bool lo8 = L.x < 8;
  const float one0 = lo8 ? 0.f : 1.f, one1 = lo8 ? 1.f : 0.f;
  const int vcol = (lo8 ? 2 * P * SLOT : (2 * P + 1) * SLOT) + L.g * 8 + (L.x & 7);
  const int32_t c0 = a.col_ptr[n], c1 = a.col_ptr[n + 1];
  Lane L0;
  L0.lane = L.lane;
  for (int32_t base = c0; base < c1; base += 64) {
    const int cnt = (c1 - base) < 64 ? (c1 - base) : 64;
    const ColBlock cb = col_block(L0, a, base, cnt);
    auto load = [&](int k, float (&r)[3]) __attribute__((always_inline)) {
      const int32_t e = rl(cb.e, k), i = rl(cb.i, k), te = rl(cb.te, k);
      const float* pk = a.phik + rowoff(i, a.ld_phik) + 2 * P * 64 + L.lane;
      r[0] = pk[0];
      r[1] = pk[64];
      const int32_t tr = te < 0 ? e : te;   // a missing transposed entry is masked at use time
      const int64_t o0 = rowoff(P == 2 ? tr : e, a.ld_v), o1 = rowoff(P == 2 ? e : tr, a.ld_v);
      r[2] = ldS((const ST*)a.v + (lo8 ? o0 : o1) + vcol);
    };
    auto use = [&](int k, const float (&r)[3]) __attribute__((always_inline)) {
      const bool ok = rl(cb.te, k) >= 0;
      const float b0 = (lo8 && (P != 2 || ok)) ? r[2] : one0;
      const float b1 = (!lo8 && (P == 2 || ok)) ? r[2] : one1;
      kv0 = mfma4b(r[0], b0, kv0);
      kv1 = mfma4b(r[1], b1, kv1);
    };
    pipeline8<3>(cnt, load, use);
  }
}

// per-graph sum `slot` (feature-major part of kvg) in the T layout; HI: values on x >= 8
template <bool HI>
__device__ __forceinline__ f16v kvg_load_t(const LaneT& L, const float* __restrict__ slot) {
  const int c = HI ? (L.x >= 8 ? L.x - 8 : NC) : (L.x < 8 ? L.x : NC);
  f16v kv;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float4 t = ld4(slot + c * 64 + h * 16 + 4 * L.g);
    kv[4 * h + 0] = t.x; kv[4 * h + 1] = t.y; kv[4 * h + 2] = t.z; kv[4 * h + 3] = t.w;
  }
  return kv;
}

// max(a, b) for b > 0 and a not NaN as an integer max (negative floats are negative integers): fmaxf would put a
// canonicalising self-max in front, and an asm v_max would hide the MFMA -> VALU hazard from the compiler
__device__ __forceinline__ float max_pos(float a, float b) {
  const int ia = __float_as_int(a), ib = __float_as_int(b);
  return __int_as_float(ia > ib ? ia : ib);
}

// Tile I/O goes through a wave-private LDS image of 16 rows x 64 floats (+4 floats of padding per row), so that global
// memory only sees full 128-B / 256-B row pieces (fragment-shaped accesses -- 16 rows x 32 B per instruction -- keep
// the address path busy several times longer for the same bytes): lane (g, x) moves float4 number x of rows 4g + r.
constexpr int TS = 68;               // floats per staged row: b64 fragment reads of 32 lanes hit 64 distinct banks
constexpr int TILE_FLOATS = 16 * TS;
struct TileRaw { float4 r[4]; };
struct TileQ { float2 a[NH], b[NH]; };

// rows[r]: row of entry 4g + r; col: first of the 64 columns (two adjacent slots)
template <class QT>
__device__ __forceinline__ void tile_raw_load(const LaneT& L, const QT* const (&rows)[4], int col, TileRaw& t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t.r[r] = ld4(rows[r] + col + 4 * L.x);
}
// registers -> LDS image -> fragments: entry t = x, head h, elements 2g, 2g + 1 of the head slice of slots a | b
__device__ __forceinline__ void tile_fragments(const LaneT& L, float* tile, const TileRaw& t, TileQ& q) {
  wave_fence();
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(tile + (4 * L.g + r) * TS + 4 * L.x) = t.r[r];
  wave_fence();
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    q.a[h] = *reinterpret_cast<const float2*>(tile + L.x * TS + h * 8 + 2 * L.g);
    q.b[h] = *reinterpret_cast<const float2*>(tile + L.x * TS + SLOT + h * 8 + 2 * L.g);
  }
}

// 16 entries x 2 slots: out = phi(q).KV / (phi(q).ks).  q: the tile's raw queries (entry t = x); orow[r]: output row of
// entry t = 4g + r; oa / ob: output columns of the two slots.  Straight-line code: every lane stores (positions
// without an entry of their own repeat another entry of the segment, see tile_rows).  Two heads at a time: four
// independent MFMA chains cover the 40-cycle dependent latency and the live set stays small.
template <class OT>
__device__ __forceinline__ void tile_apply_pair_fwd(const LaneT& L, float* tile, const TileQ& q, const f16v& kvA,
                                                    const f16v& kvB, OT* const (&orow)[4], int oa, int ob) {
  const bool lo8 = L.x < 8;
  const f4v eps4 = {PHI_EPS, PHI_EPS, PHI_EPS, PHI_EPS};
  const f4v z4 = {0.f, 0.f, 0.f, 0.f};
  wave_fence();   // the fragment reads of this tile are done before its results overwrite the image
  float* tw = tile + 4 * L.g * TS + (lo8 ? L.x : SLOT + L.x - 8);
#pragma unroll
  for (int h0 = 0; h0 < NH; h0 += 2) {
    f4v pA[2], pB[2], oA[2], oB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // phi = max(pre, 0) + eps = max(pre + eps, eps)
      pA[i] = mfma4(L.pt0, q.a[h0 + i].x, eps4);
      pB[i] = mfma4(L.pt0, q.b[h0 + i].x, eps4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pA[i] = mfma4(L.pt1, q.a[h0 + i].y, pA[i]);
      pB[i] = mfma4(L.pt1, q.b[h0 + i].y, pB[i]);
      oA[i] = z4;
      oB[i] = z4;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        oA[i] = mfma4(max_pos(pA[i][s], PHI_EPS), kvA[4 * (h0 + i) + s], oA[i]);
        oB[i] = mfma4(max_pos(pB[i][s], PHI_EPS), kvB[4 * (h0 + i) + s], oB[i]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float num = lo8 ? oA[i][r] : oB[i][r];
        const float den = max_pos(ror<8>(lo8 ? oB[i][r] : oA[i][r]), 1e-30f);   // see apply_fwd: empty key set -> 0
        tw[r * TS + (h0 + i) * 8] = num * __builtin_amdgcn_rcpf(den);
      }
  }
  wave_fence();
  const int ocol = lo8 ? oa + 4 * L.x : ob + 4 * (L.x - 8);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    st4(orow[r] + ocol, *reinterpret_cast<const float4*>(tile + (4 * L.g + r) * TS + 4 * L.x));
}

// zero `nf4` float4s (a multiple of 16) at column `col` of the four rows of this lane group (16 lanes per row)
template <class OT>
__device__ __forceinline__ void tile_zero_rows(const LaneT& L, OT* const (&orow)[4], int col, int nf4) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    for (int j = L.x; j < nf4; j += 16) st4(orow[r] + col + 4 * j, z);
}

// Entry rows of one 16-position tile [t0, t0 + 16) of a non-empty segment [.., c1) (column segment for P == 1, row
// segment for P == 2): eb[r] = row of position t0 + 4g + r.  Positions past the end of the segment and diagonal
// entries are replaced by `sub`, an off-diagonal entry of the same segment: they recompute and rewrite that entry's
// results (same values, same addresses), which keeps the tile code free of branches.  FIRST: also finds sub (a
// segment holds at most one diagonal entry, so one of its first two positions is off-diagonal if any is; -1: none).
template <int P, bool FIRST>
__device__ __forceinline__ void tile_rows(const LaneT& L, const A& a, int32_t t0, int32_t c1, int32_t& sub, int32_t (&eb)[4]) {
  const int32_t last = c1 - 1;
  int32_t pb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) pb[r] = t0 + 4 * L.g + r < last ? t0 + 4 * L.g + r : last;
  int32_t rb[4];   // entry row, or a negative value for a diagonal entry
  if (P == 1) {
    const int4* rec = reinterpret_cast<const int4*>(a.col_rec);
    int4 qb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) qb[r] = rec[pb[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r] = qb[r].w == 2 ? qb[r].x : -1;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r] = a.row_rec[pb[r]];
  }
  if (FIRST) {
    const int32_t s0 = rl(rb[0], 0), s1 = rl(rb[1], 0);   // positions t0 and t0 + 1 (lane group 0)
    sub = s0 >= 0 ? s0 : (t0 + 1 < c1 ? s1 : -1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) eb[r] = (rb[r] >= 0 && t0 + 4 * L.g + r <= last) ? rb[r] : sub;
}

template <class ST, int P>
__global__ __launch_bounds__(TPB, 4) void k_segt_fwd(A a, unsigned* __restrict__ ticket) {
  const SV<ST> S(a);
  __shared__ __align__(16) float tile_all[WPB][TILE_FLOATS];
  float* tile = tile_all[threadIdx.x >> 6];
  const LaneT L = make_lane_t(a.proj);
  constexpr int QC = P == 1 ? 0 : 2 * SLOT;                                             // raw query slots 0, 1 / 2, 3
  constexpr int OA = P == 1 ? 2 * SLOT : 3 * SLOT, OB = P == 1 ? 4 * SLOT : 5 * SLOT;   // terms 3, 5 / 4, 6
  WorkIter<HOTK_TICKET_NODES> it(ticket, a.Nn);
  int64_t n;
  while (it.next(n)) {
    // P == 1: entries of column n, terms 3, 5;  P == 2: entries of row n, terms 4, 6 (sums of their ROW node)
    const int32_t c0 = P == 1 ? a.col_ptr[n] : a.row_ptr[n];
    int32_t c1 = P == 1 ? a.col_ptr[n + 1] : a.row_ptr[n + 1];
    if (c1 <= c0) continue;   // padding node
    int32_t sub, eb[4];
    tile_rows<P, true>(L, a, c0, c1, sub, eb);
    if (sub < 0) {   // no off-diagonal entry: nothing to write; keep the loads below on valid addresses
      c1 = c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) eb[r] = 0;
    }
    TileRaw raw;
    {
      const ST* rows[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rows[r] = S.h2 + rowoff(eb[r], a.ld_h2);
      tile_raw_load(L, rows, QC, raw);   // in flight during the accumulation
    }
    f16v kv0, kv1;
    seg_accumulate_t<ST, P>(L, a, (int32_t)n, kv0, kv1);
    for (int32_t t0 = c0; t0 < c1; t0 += 16) {
      if (t0 != c0) {
        tile_rows<P, false>(L, a, t0, c1, sub, eb);
        const ST* rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[r] = S.h2 + rowoff(eb[r], a.ld_h2);
        tile_raw_load(L, rows, QC, raw);
      }
      ST* orow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) orow[r] = S.att + rowoff(eb[r], a.ld_att);
      TileQ q;
      tile_fragments(L, tile, raw, q);
      tile_apply_pair_fwd(L, tile, q, kv0, kv1, orow, OA, OB);
    }
  }
}

// Per-graph terms, application side: consecutive rows of one graph share the graph's two key-value sums, so this is a
// streaming kernel over 16-row tiles.  Diagonal rows (node queries q1 slots 2, 3 against kvg slots 0, 1 -> terms 7, 8)
// and off-diagonal rows (entry queries q2 slots 4, 5 against kvg slots 2, 3 -> terms 9, 10); also zero-fills the terms
// that do not exist on a row kind (1, 2, 7, 8 off the diagonal; 3..6, 9, 10 on it).
constexpr int GA_TILES = 8;   // 16-row tiles per wave
#ifndef HOTK_GA_WAVES
#define HOTK_GA_WAVES 4
#endif
template <class ST>
__global__ __launch_bounds__(TPB, HOTK_GA_WAVES) void k_graph_apply_fwd_t(A a, unsigned* __restrict__ ticket) {
  const SV<ST> S(a);
  __shared__ __align__(16) float tile_all[WPB][TILE_FLOATS];
  float* tile = tile_all[threadIdx.x >> 6];
  const LaneT L = make_lane_t(a.proj);
  const int64_t nd = (a.Nmax + 16 * GA_TILES - 1) / (16 * GA_TILES), no = (a.Emax + 16 * GA_TILES - 1) / (16 * GA_TILES);
  const int64_t nwork = a.B * (nd + no);
  WorkIter<1> it(ticket, nwork);
  int64_t w;
  while (it.next(w)) {
    const int64_t b = w / (nd + no), k = w - b * (nd + no);
    const bool diag = k < nd;
    const int32_t nn = a.n_nodes[b], ne = a.n_entries[b];
    // rows [r0, r1) relative to the graph's first row
    const int32_t lo = diag ? 0 : nn, hi = diag ? nn : ne;
    const int32_t r0 = lo + (int32_t)(diag ? k : k - nd) * 16 * GA_TILES;
    if (r0 >= hi) continue;
    const int32_t r1 = r0 + 16 * GA_TILES < hi ? r0 + 16 * GA_TILES : hi;
    const int32_t last = r1 - 1;   // rows past the end repeat the last row
    const int64_t ebase = b * a.Emax, nbase = b * a.Nmax;
    // diagonal rows take their queries from the node table h1 (always fp32), off-diagonal rows from h2 (storage type)
    const float* qnode = a.h1 + nbase * a.ld_h1;
    const ST* qentry = S.h2 + ebase * a.ld_h2;
    const int qc = diag ? 2 * SLOT : 4 * SLOT;
    const int oa = diag ? 6 * SLOT : 8 * SLOT, ob = oa + SLOT;
    auto load_tile = [&](int32_t t0, TileRaw& raw) __attribute__((always_inline)) {
      int32_t tb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) tb[r] = t0 + 4 * L.g + r < last ? t0 + 4 * L.g + r : last;
      if (diag) {
        const float* rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[r] = qnode + tb[r] * a.ld_h1;
        tile_raw_load(L, rows, qc, raw);
      } else {
        const ST* rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[r] = qentry + tb[r] * a.ld_h2;
        tile_raw_load(L, rows, qc, raw);
      }
    };
    TileRaw raw;
    load_tile(r0, raw);
    const f16v g0 = kvg_load_t<false>(L, kvg_slot(a.kvg, b, diag ? 0 : 2));
    const f16v g1 = kvg_load_t<true>(L, kvg_slot(a.kvg, b, diag ? 1 : 3));
    for (int32_t t0 = r0; t0 < r1; t0 += 16) {
      ST* orow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int32_t tb = t0 + 4 * L.g + r < last ? t0 + 4 * L.g + r : last;
        orow[r] = S.att + (ebase + tb) * a.ld_att;
      }
      TileQ q;
      tile_fragments(L, tile, raw, q);
      load_tile(t0 + 16, raw);   // next tile in flight during this tile's products (the last one re-reads the last row)
      tile_apply_pair_fwd(L, tile, q, g0, g1, orow, oa, ob);
      if (diag) {
        tile_zero_rows(L, orow, 2 * SLOT, 32);
        tile_zero_rows(L, orow, 8 * SLOT, 16);
      } else {
        tile_zero_rows(L, orow, 0, 16);
        tile_zero_rows(L, orow, 6 * SLOT, 16);
      }
    }
  }
}

int check_args(const A* a) {
  if (!a) return HOTK_EINVAL;
  if (a->H != NH || a->m != NM || a->d_head != DH || a->c != NC) return HOTK_EUNSUPPORTED;
  if (a->storage != HOTK_STORAGE_F32 && a->storage != HOTK_STORAGE_BF16) return HOTK_EINVAL;
  if (a->B < 0 || a->Emax < 0 || a->Nmax < 0 || a->R != a->B * a->Emax || a->Nn != a->B * a->Nmax) return HOTK_EINVAL;
  if (!a->phik || a->ld_phik < 8 * 64) return HOTK_EINVAL;
  if (!a->h1 || !a->h2 || !a->v || !a->proj || !a->kind || !a->nkind || !a->rown || !a->coln || !a->tpos ||
      !a->diag_pos || !a->col_ptr || !a->col_perm || !a->row_ptr || !a->row_perm || !a->n_entries || !a->n_nodes || !a->col_rec || !a->row_rec)
    return HOTK_EINVAL;
  if (a->ld_h1 < 12 * SLOT || a->ld_h2 < 8 * SLOT || a->ld_v < 10 * SLOT) return HOTK_EINVAL;
  // 16-byte vector loads of 8-float head slices
  if ((a->ld_h1 | a->ld_h2 | a->ld_v) & 3) return HOTK_EINVAL;
  // row offsets are formed as 32 x 32 -> 64 bit products (rowoff)
  if (a->R > INT32_MAX || a->Nn > INT32_MAX) return HOTK_EUNSUPPORTED;
  for (int64_t ld : {a->ld_h1, a->ld_h2, a->ld_v, a->ld_phik, a->ld_att, a->ld_datt, a->ld_dout, a->ld_dh1, a->ld_dh2, a->ld_dv})
    if (ld < 0 || ld > INT32_MAX) return HOTK_EINVAL;
  if (!hotk_aligned16(a->h1) || !hotk_aligned16(a->h2) || !hotk_aligned16(a->v)) return HOTK_EINVAL;
  return HOTK_OK;
}

// HOTK_CORE=valu selects the VALU/DPP node-segment kernels instead of the MFMA tile kernels (A/B measurements)
bool core_valu() {
  static const int v = [] { const char* e = getenv("HOTK_CORE"); return (e && e[0] == 'v') ? 1 : 0; }();
  return v != 0;
}

// rows of one graph per wave in the per-graph sum kernels: fewer, longer chunks mean fewer partial sums to write and
// reduce (each partial is 4 x 9 x 64 floats)
#ifndef HOTK_CHUNK_ROWS
#define HOTK_CHUNK_ROWS 256
#endif
void chunking(int64_t B, int64_t Emax, int* cpg, int* rows) {
  // small batches (C2: 128 molecules of <= 62 rows): shorter chunks, so that a few thousand waves share the rows
  int64_t cr = HOTK_CHUNK_ROWS;
  while (cr > 16 && B * ((Emax + cr - 1) / cr) < 2048) cr >>= 1;
  int64_t c = (Emax + cr - 1) / cr;
  if (c > 4096) c = 4096;
  if (c < 1) c = 1;
  *cpg = (int)c;
  *rows = (int)((Emax + c - 1) / c);
}

// grid of a kernel that draws tickets (tk != nullptr: persistent workgroups) or walks with a static stride
inline dim3 work_grid(int64_t items, const unsigned* tk) {
  return dim3(hotk_grid(items, WPB, tk ? (int64_t)HOTK_DYN_GRID : GRID_CAP));
}
inline unsigned* tk_slot(unsigned* tickets, int i) { return tickets ? tickets + i * TICKET_STRIDE : nullptr; }

template <class ST>
void launch_graphsum_bwd(hipStream_t st, const A* a, unsigned* tickets) {
  if (a->B * ((a->Emax + GB_ROWS_SMALL - 1) / GB_ROWS_SMALL) < 1024) {
    // a few hundred chunks of 16: the kernel's time is one wave's chain of rows (128 PCQM molecules: 512 waves x 16 rows,
    // 27 us) -- four rows per wave
    constexpr int RW = GB_ROWS_TINY;
    hipLaunchKernelGGL((k_graphsum_bwd<ST, true, RW>), work_grid(a->B * ((a->Nmax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 4));
    hipLaunchKernelGGL((k_graphsum_bwd<ST, false, RW>), work_grid(a->B * ((a->Emax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 5));
  } else if (a->B * ((a->Emax + GB_ROWS - 1) / GB_ROWS) < 2048) {
    constexpr int RW = GB_ROWS_SMALL;
    hipLaunchKernelGGL((k_graphsum_bwd<ST, true, RW>), work_grid(a->B * ((a->Nmax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 4));
    hipLaunchKernelGGL((k_graphsum_bwd<ST, false, RW>), work_grid(a->B * ((a->Emax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 5));
  } else {
    constexpr int RW = GB_ROWS;
    hipLaunchKernelGGL((k_graphsum_bwd<ST, true, RW>), work_grid(a->B * ((a->Nmax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 4));
    hipLaunchKernelGGL((k_graphsum_bwd<ST, false, RW>), work_grid(a->B * ((a->Emax + RW - 1) / RW), tickets), dim3(TPB), 0, st, *a, tk_slot(tickets, 5));
  }
}

void launch_graph_reduce(hipStream_t st, const float* part, float* out, int64_t B, int cpg, int dup) {
  if (cpg > 16)
    hipLaunchKernelGGL(k_graph_reduce_wide, dim3(hotk_grid(B * 4 * (NC + 1), 1, 256 * 16)), dim3(64 * RG), 0, st, part, out, B, cpg, dup);
  else
    hipLaunchKernelGGL(k_graph_reduce, dim3(hotk_grid(B * 4, 1, 256 * 16)), dim3(256), 0, st, part, out, B, cpg, dup);
}

}  // namespace

extern "C" int hotk_attn22_workspace_bytes(int64_t B, int64_t Emax, size_t* bytes) {
  HOTK_CHECK_ARG(bytes && B >= 0 && Emax >= 0);
  int cpg, rows;
  chunking(B, Emax, &cpg, &rows);
  // per-chunk partial graph sums, then the work tickets of the persistent kernels (one 64-byte line per kernel)
  *bytes = (size_t)B * cpg * 4 * (NC + 1) * 64 * sizeof(float) + TICKET_BYTES;
  return HOTK_OK;
}

namespace {
// forward launches; `part` holds the per-chunk partial graph sums (hotk_attn22_workspace_bytes): every chunk of it is
// written by k_graphsum_fwd before k_graph_reduce reads it, so it needs no clearing
template <class ST>
int attn22_forward_launch_t(const A* a, float* part, unsigned* tickets, hipStream_t st) {
  int cpg, rows;
  chunking(a->B, a->Emax, &cpg, &rows);
  if (tickets && hipMemsetAsync(tickets, 0, TICKET_BYTES, st) != hipSuccess) return HOTK_ELAUNCH;
  hipLaunchKernelGGL(k_graphsum_fwd<ST>, work_grid((int64_t)a->B * cpg, nullptr), dim3(TPB), 0, st, *a, part, cpg, rows, (unsigned*)nullptr);
  launch_graph_reduce(st, part, a->kvg, a->B, cpg, 0);
  const dim3 gseg = work_grid(a->Nn, tickets);
  if (core_valu()) {
    hipLaunchKernelGGL((k_seg_fwd<ST, 1, true>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
    hipLaunchKernelGGL((k_seg_fwd<ST, 2, true>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
    hipLaunchKernelGGL((k_seg_fwd<ST, 0, true>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 3));
  } else {
    const int64_t per = 16 * GA_TILES;
    const int64_t nga = a->B * ((a->Nmax + per - 1) / per + (a->Emax + per - 1) / per);
    hipLaunchKernelGGL(k_graph_apply_fwd_t<ST>, work_grid(nga, nullptr), dim3(TPB), 0, st, *a, (unsigned*)nullptr);
    hipLaunchKernelGGL((k_segt_fwd<ST, 1>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
    hipLaunchKernelGGL((k_segt_fwd<ST, 2>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
    hipLaunchKernelGGL((k_seg_fwd<ST, 0, false>), work_grid(a->Nn, nullptr), dim3(TPB), 0, st, *a, (unsigned*)nullptr);
  }
  return hotk_launch_status();
}
// workspace = [partial graph sums | tickets]: the counters sit in the last TICKET_BYTES of hotk_attn22_workspace_bytes
// Tickets pay where the nodes are many and their segments short (C1 / C2: ~10 entries per node, a node-wave lives ~20 us
// and the dispatcher cannot keep the slots full); on ONE graph of 125 k nodes with 50 entries each (the C5 shard) the
// static stride with a workgroup per node is faster (phases 17.6 ms against 19.2-22.6 ms with tickets of 8 / 1 nodes).
// And where there are enough nodes for 8-node tickets: one counter serves ~86 atomics per microsecond, which a batch of
// 82 k small nodes (4 096 molecules) already exceeds with one node per ticket (the C2-shaped step went 167 -> 200 ms).
// Eight counters per kernel (one per XCD) lift that limit but let the XCDs drift apart, and the shared window of
// neighbouring nodes is what the tickets are for (k_seg_bwd 5.6 -> 9.0 ms at C1): one counter, large batches only.
unsigned* ws_tickets(const A* a, void* workspace, size_t need) {
  const bool short_segments = a->R < 24 * a->Nn;
  const bool many_nodes = a->Nn >= HOTK_TICKET_MIN_NODES;
  return (HOTK_DYN && short_segments && many_nodes) ? reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + need - TICKET_BYTES)
                                                   : nullptr;
}
int attn22_forward_launch(const A* a, float* part, unsigned* tickets, hipStream_t st) {
  return a->storage == HOTK_STORAGE_BF16 ? attn22_forward_launch_t<bf16>(a, part, tickets, st)
                                        : attn22_forward_launch_t<float>(a, part, tickets, st);
}
}  // namespace

extern "C" int hotk_attn22_forward(const hotk_attn22_args* a, void* workspace, size_t ws_bytes, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  HOTK_CHECK_ARG(a->att && a->kvg && a->ld_att >= 10 * SLOT && (a->ld_att & 3) == 0 && hotk_aligned16(a->att));
  if (a->R == 0 || a->Nn == 0) return HOTK_OK;
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  if (!workspace || ws_bytes < need) return HOTK_EWORKSPACE;
  return attn22_forward_launch(a, (float*)workspace, ws_tickets(a, workspace, need), (hipStream_t)stream);
}

// ---- the forward core for callers that cannot allocate (include/hotk.h: hotk_attn22_forward_static) ----------
// The per-chunk partial graph sums live in a file-scope device array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy of the array at its own address: the address is looked up once per device with
// hipGetSymbolAddress (the array's NAME is only a host-side stand-in and must never reach a kernel argument) and the
// kernels take the looked-up pointer exactly as they take the workspace of hotk_attn22_forward.
namespace {
constexpr size_t STATIC_SCRATCH_BYTES = HOTK_ATTN22_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
std::mutex scratch_mu;
void* scratch_ptr[MAX_DEVICES];   // per device, filled on the first call with that device current

// the array's address on `dev`: looked up once per device, with that device current
int static_scratch_on(int dev, float** out) {
  if (dev < 0 || dev >= MAX_DEVICES) return HOTK_ELAUNCH;
  std::lock_guard<std::mutex> lock(scratch_mu);
  if (!scratch_ptr[dev]) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return HOTK_ELAUNCH;
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return HOTK_ELAUNCH;
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (cur != dev && hipSetDevice(cur) != hipSuccess) return HOTK_ELAUNCH;
    if (e != hipSuccess || !ptr) return HOTK_ELAUNCH;
    scratch_ptr[dev] = ptr;
  }
  if (out) *out = (float*)scratch_ptr[dev];
  return HOTK_OK;
}
// the device a launch on `stream` runs on: the stream's own device, or the current one for the null stream
int stream_device(hipStream_t st, int* dev) {
  if (st) {
    hipDevice_t d;
    if (hipStreamGetDevice(st, &d) == hipSuccess) {
      *dev = (int)d;
      return HOTK_OK;
    }
    (void)hipGetLastError();
  }
  return hipGetDevice(dev) == hipSuccess ? HOTK_OK : HOTK_ELAUNCH;
}
}  // namespace

extern "C" int hotk_attn22_static_init(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return HOTK_ELAUNCH;
  return static_scratch_on(dev, nullptr);
}

extern "C" int hotk_attn22_forward_static(const hotk_attn22_args* a, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  HOTK_CHECK_ARG(a->att && a->kvg && a->ld_att >= 10 * SLOT && (a->ld_att & 3) == 0 && hotk_aligned16(a->att));
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  if (need > STATIC_SCRATCH_BYTES) return HOTK_ETOOLARGE;
  if (a->R == 0 || a->Nn == 0) return HOTK_OK;
  float* part = nullptr;
  int dev = -1;
  rc = stream_device((hipStream_t)stream, &dev);
  if (rc) return rc;
  rc = static_scratch_on(dev, &part);   // a cached pointer after hotk_attn22_static_init / the first call on this device
  if (rc) return rc;
  // static stride, no work tickets: a batch that fits the array is far below the size where tickets pay (ws_tickets), and
  // without them the array needs no clearing between calls
  return attn22_forward_launch(a, part, nullptr, (hipStream_t)stream);
}

namespace {
template <class ST>
int attn22_backward_launch_t(const A* a, float* part, unsigned* tickets, int cpg, int rows, hipStream_t st) {
  if (tickets && hipMemsetAsync(tickets, 0, TICKET_BYTES, st) != hipSuccess) return HOTK_ELAUNCH;
  hipLaunchKernelGGL(k_graph_apply_bwd<ST>, work_grid((int64_t)a->B * cpg, nullptr), dim3(TPB), 0, st, *a, part, cpg, rows, (unsigned*)nullptr);
  launch_graph_reduce(st, part, a->dkvg, a->B, cpg, 1);
  launch_graphsum_bwd<ST>(st, a, nullptr);      // (the streaming kernels gain nothing from tickets: measured)
  const dim3 gseg = work_grid(a->Nn, tickets);
  hipLaunchKernelGGL((k_seg_bwd<ST, 1>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
  hipLaunchKernelGGL((k_seg_bwd<ST, 2>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
  hipLaunchKernelGGL((k_seg_bwd<ST, 0>), gseg, dim3(TPB), 0, st, *a, tk_slot(tickets, 3));
  return hotk_launch_status();
}
}  // namespace

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
  chunking(a->B, a->Emax, &cpg, &rows);
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  if (!workspace || ws_bytes < need) return HOTK_EWORKSPACE;
  float* part = (float*)workspace;
  unsigned* tickets = ws_tickets(a, workspace, need);
  return a->storage == HOTK_STORAGE_BF16 ? attn22_backward_launch_t<bf16>(a, part, tickets, cpg, rows, st)
                                        : attn22_backward_launch_t<float>(a, part, tickets, cpg, rows, st);
}

// The core in phases for one graph sharded by entries (include/hotk.h): the caller all-reduces kvg / kvn after the
// sum phases and dkvg / dkvn after BWD_QUERY.  Node-segment application runs on the VALU/DPP kernels (their sums are
// feature-major in memory, as the all-reduce leaves them); the streaming per-graph kernels are shared with the
// single-device path.
extern "C" int hotk_attn22_phase(const hotk_attn22_args* a, int phase, void* workspace, size_t ws_bytes, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  HOTK_CHECK_ARG(phase >= HOTK_ATTN22_FWD_GRAPHSUM && phase <= HOTK_ATTN22_BWD_KEYVAL_NODE);
  if (a->storage != HOTK_STORAGE_F32) return HOTK_EUNSUPPORTED;   // the sums exchanged between the phases are fp32 tables
  HOTK_CHECK_ARG(a->kvg && a->kvn && a->nd0 && hotk_aligned16(a->kvn) && a->diag_w > 0.f);
  if (a->R == 0 || a->Nn == 0) return HOTK_OK;
  hipStream_t st = (hipStream_t)stream;
  int cpg, rows;
  chunking(a->B, a->Emax, &cpg, &rows);
  size_t need;
  hotk_attn22_workspace_bytes(a->B, a->Emax, &need);
  float* part = (float*)workspace;
  // the node-segment kernels draw tickets when the caller's workspace has room for them (it does when it was sized by
  // hotk_attn22_workspace_bytes); the light per-node kernels and the streaming kernels keep the static stride
  unsigned* tickets = (workspace && ws_bytes >= need) ? ws_tickets(a, workspace, need) : nullptr;
  if (tickets && hipMemsetAsync(tickets, 0, TICKET_BYTES, st) != hipSuccess) return HOTK_ELAUNCH;
  const dim3 gseg(hotk_grid(a->Nn, WPB, GRID_CAP));
  const dim3 gdyn = work_grid(a->Nn, tickets);
  const dim3 gchunk(hotk_grid((int64_t)a->B * cpg, WPB, GRID_CAP));
  switch (phase) {
    case HOTK_ATTN22_FWD_GRAPHSUM:
      if (!workspace || ws_bytes < need) return HOTK_EWORKSPACE;
      hipLaunchKernelGGL(k_graphsum_fwd<float>, gchunk, dim3(TPB), 0, st, *a, part, cpg, rows, (unsigned*)nullptr);
      launch_graph_reduce(st, part, a->kvg, a->B, cpg, 0);
      break;
    case HOTK_ATTN22_FWD_NODESUM:
      hipLaunchKernelGGL(k_seg_kvsum<0>, gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 0));
      hipLaunchKernelGGL(k_seg_kvsum<1>, gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
      hipLaunchKernelGGL(k_seg_kvsum<2>, gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
      break;
    case HOTK_ATTN22_FWD_APPLY:
    case HOTK_ATTN22_FWD_APPLY_GRAPH:
    case HOTK_ATTN22_FWD_APPLY_NODE: {
      HOTK_CHECK_ARG(a->att && a->ld_att >= 10 * SLOT && (a->ld_att & 3) == 0 && hotk_aligned16(a->att));
      if (phase != HOTK_ATTN22_FWD_APPLY_NODE) {      // terms 7-10: per-graph sums only (kvg)
        const int64_t per = 16 * GA_TILES;
        const int64_t nga = a->B * ((a->Nmax + per - 1) / per + (a->Emax + per - 1) / per);
        hipLaunchKernelGGL(k_graph_apply_fwd_t<float>, dim3(hotk_grid(nga, WPB, GRID_CAP)), dim3(TPB), 0, st, *a, (unsigned*)nullptr);
      }
      if (phase != HOTK_ATTN22_FWD_APPLY_GRAPH) {     // terms 1-6: per-node sums (kvn, nd0)
        hipLaunchKernelGGL((k_seg_fwd<float, 1, false, true>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
        hipLaunchKernelGGL((k_seg_fwd<float, 2, false, true>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
        hipLaunchKernelGGL(k_node_apply_nd, gseg, dim3(TPB), 0, st, *a);
      }
      break;
    }
    case HOTK_ATTN22_BWD_QUERY:
    case HOTK_ATTN22_BWD_KEYVAL:
    case HOTK_ATTN22_BWD_KEYVAL_GRAPH:
    case HOTK_ATTN22_BWD_KEYVAL_NODE:
      HOTK_CHECK_ARG(a->dkvn && a->dd0 && hotk_aligned16(a->dkvn) && a->datt && a->dh1 && a->dh2 && a->dv && a->ck && a->dkvg);
      HOTK_CHECK_ARG(a->ld_datt >= 10 * SLOT && a->ld_dh1 >= 12 * SLOT && a->ld_dh2 >= 8 * SLOT + a->D && a->D >= 0 &&
                     a->ld_dv >= 10 * SLOT);
      HOTK_CHECK_ARG(((a->ld_datt | a->ld_dh1 | a->ld_dh2 | a->ld_dv) & 3) == 0 && hotk_aligned16(a->datt));
      HOTK_CHECK_ARG(!a->dout || a->ld_dout >= a->D);
      if (phase == HOTK_ATTN22_BWD_QUERY) {
        if (!workspace || ws_bytes < need) return HOTK_EWORKSPACE;
        hipLaunchKernelGGL(k_graph_apply_bwd<float>, gchunk, dim3(TPB), 0, st, *a, part, cpg, rows, (unsigned*)nullptr);
        launch_graph_reduce(st, part, a->dkvg, a->B, cpg, 1);
        hipLaunchKernelGGL((k_seg_bwd<float, 1, 1>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
        hipLaunchKernelGGL((k_seg_bwd<float, 2, 1>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
        hipLaunchKernelGGL(k_node_dd, gseg, dim3(TPB), 0, st, *a);
      } else {
        if (phase != HOTK_ATTN22_BWD_KEYVAL_NODE) launch_graphsum_bwd<float>(st, a, nullptr);      // from dkvg only
        if (phase != HOTK_ATTN22_BWD_KEYVAL_GRAPH) {                                             // from dkvn / dd0
          hipLaunchKernelGGL((k_seg_bwd<float, 1, 2>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 1));
          hipLaunchKernelGGL((k_seg_bwd<float, 2, 2>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 2));
          hipLaunchKernelGGL((k_seg_bwd<float, 0, 3>), gdyn, dim3(TPB), 0, st, *a, tk_slot(tickets, 0));
        }
      }
      break;
  }
  return hotk_launch_status();
}
