// The full (15-operator) order-2 -> order-2 equivariant basis of Linear(2, 2, cfg='default') as ONE expand kernel and its
// transpose map (reference: models/sparse/linear.py:143-238, which builds [B,E,E] boolean masks and pools with bmm's;
// here every pooled operator is a closed form of per-node and per-graph sums).
//
// With A = values, At = tau ? A_ji : 0, diag / Rs / Cs the per-node tables (A_vv, off-diagonal row sum, off-diagonal
// column sum), T / S the per-graph trace and off-diagonal total, od / idg the off-diagonal out / in degrees, n / Eo the
// node / off-diagonal counts of the graph, and  nrm(x, c) = normalize ? (c == 0 ? 0 : x / c) : x :
//   off-diagonal entry (i, j):  V2 = A   V3 = At   V4 = diag_i   V5 = diag_j
//       V8  = nrm(Cs_j - A,  idg_j - 1)      V9  = nrm(Rs_i - A,  od_i - 1)
//       V10 = nrm(Rs_j - At, od_j - tau)     V11 = nrm(Cs_i - At, idg_i - tau)
//       V14 = nrm(T - diag_i - diag_j, n - 2)
//       V15 = nrm(S - Rs_i - Cs_i - Rs_j - Cs_j + A + At,  Eo - od_i - idg_i - od_j - idg_j + 1 + tau)
//   diagonal entry (v, v):      V1 = A   V6 = nrm(Rs_v, od_v)   V7 = nrm(Cs_v, idg_v)
//       V12 = nrm(T - A, n - 1)              V13 = nrm(S - Rs_v - Cs_v, Eo - od_v - idg_v)
//   every other block of a row, and padding rows, are zero.  Output [R, 15 D] in the reference's block order V1..V15.
// The map is linear in A, so the backward needs no saved values: kernel 1 turns dV [R, 15 D] into the entry's own
// gradient, the gradient of its transposed partner and its contributions to the node / graph tables ([R, 10 D]); the
// caller sums the contributions per row node, column node and graph (hotk_seg_sum_rows: fixed order, no atomics) and
// kernel 2 gathers everything back:  dA_ij = self + partner's dAt + dRs_i + dCs_j + dS,  dA_vv = self + d diag_v + dT.
#include "common.h"

namespace {

template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

__device__ __forceinline__ float vz(float) { return 0.f; }
__device__ __forceinline__ float4 vz(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
__device__ __forceinline__ float4 vsub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float vdiv(float a, float c) { return a / c; }
__device__ __forceinline__ float4 vdiv(float4 a, float c) { return make_float4(a.x / c, a.y / c, a.z / c, a.w / c); }

template <class T>
__device__ __forceinline__ T nrm(T x, float cnt, int normalize) {
  if (!normalize) return x;
  return cnt == 0.f ? vz(T()) : vdiv(x, cnt);
}

template <class T>
__device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const T*>(p); }
template <class T>
__device__ __forceinline__ void st(float* p, T v) { *reinterpret_cast<T*>(p) = v; }

struct Counts {
  const float *od, *idg;   // [Nn] off-diagonal out / in degree of the node row
  const float *ng, *eog;   // [B]  nodes / off-diagonal entries of the graph
};

template <int V>
__global__ void k_basis22_full_fwd(const float* __restrict__ x, int64_t ld_x, const uint8_t* __restrict__ kind,
                                   const int32_t* __restrict__ rown, const int32_t* __restrict__ coln,
                                   const int32_t* __restrict__ tpos, const float* __restrict__ nt, int64_t ld_nt,
                                   const float* __restrict__ gt, int64_t ld_gt, Counts cn, float* __restrict__ out,
                                   int64_t ld_out, int64_t R, int64_t D, int64_t Emax, int normalize) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / DV, c = (t - r * DV) * V;
    const uint8_t kd = kind[r];
    const T z = vz(T());
    T v[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) v[k] = z;
    if (kd) {
      const int64_t b = r / Emax;
      const T A = ld<T>(x + r * ld_x + c);
      const T Tg = ld<T>(gt + b * ld_gt + c), Sg = ld<T>(gt + b * ld_gt + D + c);
      const float n = cn.ng[b], eo = cn.eog[b];
      const int64_t i = rown[r];
      const float* ni = nt + i * ld_nt + c;
      const T Ri = ld<T>(ni + D), Ci = ld<T>(ni + 2 * D);
      const float odi = cn.od[i], idi = cn.idg[i];
      if (kd == 1) {
        v[0] = A;
        v[5] = nrm(Ri, odi, normalize);
        v[6] = nrm(Ci, idi, normalize);
        v[11] = nrm(vsub(Tg, A), n - 1.f, normalize);
        v[12] = nrm(vsub(vsub(Sg, Ri), Ci), eo - odi - idi, normalize);
      } else {
        const int64_t j = coln[r];
        const int32_t tp = tpos[r];
        const float tau = tp >= 0 ? 1.f : 0.f;
        const T At = tp >= 0 ? ld<T>(x + (int64_t)tp * ld_x + c) : z;
        const float* nj = nt + j * ld_nt + c;
        const T Di = ld<T>(ni), Dj = ld<T>(nj), Rj = ld<T>(nj + D), Cj = ld<T>(nj + 2 * D);
        const float odj = cn.od[j], idj = cn.idg[j];
        v[1] = A;
        v[2] = At;
        v[3] = Di;
        v[4] = Dj;
        v[7] = nrm(vsub(Cj, A), idj - 1.f, normalize);
        v[8] = nrm(vsub(Ri, A), odi - 1.f, normalize);
        v[9] = nrm(vsub(Rj, At), odj - tau, normalize);
        v[10] = nrm(vsub(Ci, At), idi - tau, normalize);
        v[13] = nrm(vsub(vsub(Tg, Di), Dj), n - 2.f, normalize);
        const T s15 = vadd(vadd(vsub(vsub(vsub(vsub(Sg, Ri), Ci), Rj), Cj), A), At);
        v[14] = nrm(s15, eo - odi - idi - odj - idj + 1.f + tau, normalize);
      }
    }
    float* o = out + r * ld_out + c;
#pragma unroll
    for (int k = 0; k < 15; ++k) st<T>(o + k * D, v[k]);
  }
}

// kernel 1 of the transpose map.  w [R, 10 D] = [ self | dAt | node i: d diag, dRs, dCs | node j: d diag, dRs, dCs | dT, dS ]
template <int V>
__global__ void k_basis22_full_bwd1(const float* __restrict__ g, int64_t ld_g, const uint8_t* __restrict__ kind,
                                    const int32_t* __restrict__ rown, const int32_t* __restrict__ coln,
                                    const int32_t* __restrict__ tpos, Counts cn, float* __restrict__ w, int64_t ld_w,
                                    int64_t R, int64_t D, int64_t Emax, int normalize) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / DV, c = (t - r * DV) * V;
    const uint8_t kd = kind[r];
    const T z = vz(T());
    T o[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) o[k] = z;
    if (kd) {
      const int64_t b = r / Emax;
      const float n = cn.ng[b], eo = cn.eog[b];
      const int64_t i = rown[r];
      const float odi = cn.od[i], idi = cn.idg[i];
      const float* gr = g + r * ld_g + c;
      if (kd == 1) {
        const T n6 = nrm(ld<T>(gr + 5 * D), odi, normalize), n7 = nrm(ld<T>(gr + 6 * D), idi, normalize);
        const T n12 = nrm(ld<T>(gr + 11 * D), n - 1.f, normalize);
        const T n13 = nrm(ld<T>(gr + 12 * D), eo - odi - idi, normalize);
        o[0] = vsub(ld<T>(gr), n12);       // V1 = A, V12 = (T - A) / (n - 1)
        o[3] = vsub(n6, n13);              // Rs_v
        o[4] = vsub(n7, n13);              // Cs_v
        o[8] = n12;                       // T
        o[9] = n13;                       // S
      } else {
        const int64_t j = coln[r];
        const float tau = tpos[r] >= 0 ? 1.f : 0.f;
        const float odj = cn.od[j], idj = cn.idg[j];
        const T n8 = nrm(ld<T>(gr + 7 * D), idj - 1.f, normalize), n9 = nrm(ld<T>(gr + 8 * D), odi - 1.f, normalize);
        const T n10 = nrm(ld<T>(gr + 9 * D), odj - tau, normalize), n11 = nrm(ld<T>(gr + 10 * D), idi - tau, normalize);
        const T n14 = nrm(ld<T>(gr + 13 * D), n - 2.f, normalize);
        const T n15 = nrm(ld<T>(gr + 14 * D), eo - odi - idi - odj - idj + 1.f + tau, normalize);
        o[0] = vadd(vsub(vsub(ld<T>(gr + D), n8), n9), n15);
        if (tau != 0.f) o[1] = vadd(vsub(vsub(ld<T>(gr + 2 * D), n10), n11), n15);
        o[2] = vsub(ld<T>(gr + 3 * D), n14);   // diag_i
        o[3] = vsub(n9, n15);                  // Rs_i
        o[4] = vsub(n11, n15);                 // Cs_i
        o[5] = vsub(ld<T>(gr + 4 * D), n14);   // diag_j
        o[6] = vsub(n10, n15);                 // Rs_j
        o[7] = vsub(n8, n15);                  // Cs_j
        o[8] = n14;
        o[9] = n15;
      }
    }
    float* ow = w + r * ld_w + c;
#pragma unroll
    for (int k = 0; k < 10; ++k) st<T>(ow + k * D, o[k]);
  }
}

// kernel 2: dn [Nn, 3 D] = d diag | dRs | dCs (row-node plus column-node contributions), dg [B, 2 D] = dT | dS
template <int V>
__global__ void k_basis22_full_bwd2(const float* __restrict__ w, int64_t ld_w, const uint8_t* __restrict__ kind,
                                    const int32_t* __restrict__ rown, const int32_t* __restrict__ coln,
                                    const int32_t* __restrict__ tpos, const float* __restrict__ dn, int64_t ld_dn,
                                    const float* __restrict__ dg, int64_t ld_dg, float* __restrict__ dx, int64_t ld_dx,
                                    int64_t R, int64_t D, int64_t Emax) {
  using T = typename VecT<V>::type;
  const int64_t DV = D / V, total = R * DV;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / DV, c = (t - r * DV) * V;
    const uint8_t kd = kind[r];
    T acc = vz(T());
    if (kd) {
      const int64_t b = r / Emax, i = rown[r];
      acc = ld<T>(w + r * ld_w + c);
      if (kd == 1) {
        acc = vadd(acc, vadd(ld<T>(dn + i * ld_dn + c), ld<T>(dg + b * ld_dg + c)));
      } else {
        const int64_t j = coln[r];
        const int32_t tp = tpos[r];
        if (tp >= 0) acc = vadd(acc, ld<T>(w + (int64_t)tp * ld_w + D + c));   // this entry is the partner's At
        acc = vadd(acc, vadd(ld<T>(dn + i * ld_dn + D + c), ld<T>(dn + j * ld_dn + 2 * D + c)));
        acc = vadd(acc, ld<T>(dg + b * ld_dg + D + c));
      }
    }
    st<T>(dx + r * ld_dx + c, acc);
  }
}

bool v4(int64_t D, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  if (D % 4) return false;
  for (int64_t l : lds) if (l % 4) return false;
  for (const void* p : ptrs) if (!hotk_aligned16(p)) return false;
  return true;
}

}  // namespace

extern "C" int hotk_basis22_full_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* rown,
                                     const int32_t* coln, const int32_t* tpos, const float* node_tab, int64_t ld_nt,
                                     const float* graph_tab, int64_t ld_gt, const float* od, const float* idg,
                                     const float* n_graph, const float* eo_graph, float* out, int64_t ld_out, int64_t R,
                                     int64_t D, int64_t Emax, int normalize, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D > 0 && Emax > 0 && ld_x >= D && ld_nt >= 3 * D && ld_gt >= 2 * D && ld_out >= 15 * D);
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(x && kind && rown && coln && tpos && node_tab && graph_tab && od && idg && n_graph && eo_graph && out);
  const Counts cn{od, idg, n_graph, eo_graph};
  hipStream_t st = (hipStream_t)stream;
  if (v4(D, {ld_x, ld_nt, ld_gt, ld_out}, {x, node_tab, graph_tab, out}))
    hipLaunchKernelGGL(k_basis22_full_fwd<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, x, ld_x, kind, rown, coln,
                       tpos, node_tab, ld_nt, graph_tab, ld_gt, cn, out, ld_out, R, D, Emax, normalize);
  else
    hipLaunchKernelGGL(k_basis22_full_fwd<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, x, ld_x, kind, rown, coln, tpos,
                       node_tab, ld_nt, graph_tab, ld_gt, cn, out, ld_out, R, D, Emax, normalize);
  return hotk_launch_status();
}

extern "C" int hotk_basis22_full_bwd_entries(const float* g, int64_t ld_g, const uint8_t* kind, const int32_t* rown,
                                             const int32_t* coln, const int32_t* tpos, const float* od, const float* idg,
                                             const float* n_graph, const float* eo_graph, float* w, int64_t ld_w,
                                             int64_t R, int64_t D, int64_t Emax, int normalize, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D > 0 && Emax > 0 && ld_g >= 15 * D && ld_w >= 10 * D);
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(g && kind && rown && coln && tpos && od && idg && n_graph && eo_graph && w);
  const Counts cn{od, idg, n_graph, eo_graph};
  hipStream_t st = (hipStream_t)stream;
  if (v4(D, {ld_g, ld_w}, {g, w}))
    hipLaunchKernelGGL(k_basis22_full_bwd1<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, g, ld_g, kind, rown, coln,
                       tpos, cn, w, ld_w, R, D, Emax, normalize);
  else
    hipLaunchKernelGGL(k_basis22_full_bwd1<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, g, ld_g, kind, rown, coln, tpos,
                       cn, w, ld_w, R, D, Emax, normalize);
  return hotk_launch_status();
}

extern "C" int hotk_basis22_full_bwd_gather(const float* w, int64_t ld_w, const uint8_t* kind, const int32_t* rown,
                                            const int32_t* coln, const int32_t* tpos, const float* dnode, int64_t ld_dn,
                                            const float* dgraph, int64_t ld_dg, float* dx, int64_t ld_dx, int64_t R,
                                            int64_t D, int64_t Emax, void* stream) {
  HOTK_CHECK_ARG(R >= 0 && D > 0 && Emax > 0 && ld_w >= 10 * D && ld_dn >= 3 * D && ld_dg >= 2 * D && ld_dx >= D);
  if (R == 0) return HOTK_OK;
  HOTK_CHECK_ARG(w && kind && rown && coln && tpos && dnode && dgraph && dx);
  hipStream_t st = (hipStream_t)stream;
  if (v4(D, {ld_w, ld_dn, ld_dg, ld_dx}, {w, dnode, dgraph, dx}))
    hipLaunchKernelGGL(k_basis22_full_bwd2<4>, dim3(hotk_grid(R * (D / 4), 256)), dim3(256), 0, st, w, ld_w, kind, rown, coln,
                       tpos, dnode, ld_dn, dgraph, ld_dg, dx, ld_dx, R, D, Emax);
  else
    hipLaunchKernelGGL(k_basis22_full_bwd2<1>, dim3(hotk_grid(R * D, 256)), dim3(256), 0, st, w, ld_w, kind, rown, coln, tpos,
                       dnode, ld_dn, dgraph, ld_dg, dx, ld_dx, R, D, Emax);
  return hotk_launch_status();
}
