// K1: edge-index sort / segment construction for the padded sparse batch.
// Replaces hot_pytorch/utils/sparse.py:102-127 (O(E^2) transpose match) and the grouping that
// torch.sparse coalesce() performs in models/sparse/kernelattncoef.py:103-122.
//
// Data flow (all on one stream, no host sync, no allocation):
//   init      : padded node rows of (i,j) per entry, kinds, degree histograms, compact key/value list
//   radix sort: stable LSD, 8 bits per pass, wave64 ballot ranking -> entries grouped by column node
//               (ties keep entry-row order), then by row node (ties keep column order)
//   scan      : exclusive prefix of the degree histograms -> CSR pointers
//   transpose : binary search of (j,i) inside row segment j -> t_indices / t_mask / tpos
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// ---------------------------------------------------------------------------------------------
// exclusive scan of uint32, three kernels (tile sums -> scan of sums -> tile scan + base)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *tot
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds /*[8]*/, uint32_t* tot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < SCAN_THREADS / 64; ++k) {
    uint32_t s = lds[k];
    if (k < w) base += s;
    total += s;
  }
  __syncthreads();
  *tot = total;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(const uint32_t* __restrict__ in, int64_t n,
                                                                 uint32_t* __restrict__ sums) {
  __shared__ uint32_t lds[8];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += in[base + k];
  uint32_t tot;
  block_excl_scan(s, lds, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of `nb` tile sums in place
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(uint32_t* __restrict__ sums, int64_t nb) {
  __shared__ uint32_t lds[8];
  uint32_t carry = 0;
  for (int64_t s0 = 0; s0 < nb; s0 += SCAN_THREADS) {
    int64_t i = s0 + threadIdx.x;
    uint32_t v = i < nb ? sums[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan(v, lds, &tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t* __restrict__ in, int64_t n,
                                                             const uint32_t* __restrict__ sums,
                                                             uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[8];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    s += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, lds, &tot) + sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
}

// in-place capable (in == out allowed: each tile reads its items before writing them)
int scan_exclusive(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* tile_sums, hipStream_t st) {
  if (n <= 0) return HOTK_OK;
  int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, tile_sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, tile_sums, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, tile_sums, out);
  return hotk_launch_status();
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort, 8-bit digits
// ---------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 2048 keys per block
constexpr int RS_WAVES = RS_THREADS / 64;

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t* __restrict__ hist, int64_t nb) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    int64_t i = base + k * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];  // digit-major
}

// Item order inside a tile (defines stability): wave w owns items [w*512, (w+1)*512), round k, lane l
// -> tile offset w*512 + k*64 + l.
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                              const uint32_t* __restrict__ hist_scanned, int64_t nb) {
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ uint32_t gbase[256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RS_WAVES; ++k) cnt[k][threadIdx.x] = 0;
  gbase[threadIdx.x] = hist_scanned[(int64_t)threadIdx.x * nb + blockIdx.x];
  __syncthreads();

  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * (64 * RS_ITEMS);
  uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    int64_t i = base + k * 64 + lane;
    bool ok = i < n;
    key[k] = ok ? keys_in[i] : 0u;
    val[k] = ok ? vals_in[i] : 0u;
    uint32_t d = (key[k] >> shift) & 255u;
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      bool bit = (d >> b) & 1u;
      unsigned long long mb = __ballot(bit);
      peers &= bit ? mb : ~mb;
    }
    // peers: lanes (valid) holding the same digit; lowest lane is the leader
    uint32_t r = (uint32_t)__popcll(peers & lt);
    uint32_t old = 0;
    if (ok && r == 0) {
      old = cnt[w][d];
      cnt[w][d] = old + (uint32_t)__popcll(peers);
    }
    int leader = ok ? (__ffsll((long long)peers) - 1) : lane;
    old = __shfl(old, leader, 64);
    rank[k] = old + r;
  }
  __syncthreads();
  {  // exclusive prefix over waves for digit = threadIdx.x
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) {
      uint32_t t = cnt[k][threadIdx.x];
      cnt[k][threadIdx.x] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    int64_t i = base + k * 64 + lane;
    if (i < n) {
      uint32_t d = (key[k] >> shift) & 255u;
      uint32_t pos = gbase[d] + cnt[w][d] + rank[k];
      keys_out[pos] = key[k];
      vals_out[pos] = val[k];
    }
  }
}

struct SortWs {
  uint32_t* keys_alt;
  uint32_t* vals_alt;
  uint32_t* hist;       // 256 * nb
  uint32_t* tile_sums;  // ceil(256*nb / SCAN_TILE)
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_ws_bytes(int64_t n) {
  int64_t nb = (n + RS_TILE - 1) / RS_TILE;
  if (nb < 1) nb = 1;
  int64_t nh = 256 * nb;
  int64_t nts = (nh + SCAN_TILE - 1) / SCAN_TILE;
  return 2 * align256((size_t)(n > 0 ? n : 1) * 4) + align256((size_t)nh * 4) + align256((size_t)nts * 4);
}

SortWs carve_sort_ws(void* ws, int64_t n) {
  int64_t nb = (n + RS_TILE - 1) / RS_TILE;
  if (nb < 1) nb = 1;
  int64_t nh = 256 * nb;
  char* p = (char*)ws;
  SortWs s;
  s.keys_alt = (uint32_t*)p; p += align256((size_t)(n > 0 ? n : 1) * 4);
  s.vals_alt = (uint32_t*)p; p += align256((size_t)(n > 0 ? n : 1) * 4);
  s.hist = (uint32_t*)p; p += align256((size_t)nh * 4);
  s.tile_sums = (uint32_t*)p;
  return s;
}

// sorts in place (result ends in keys/vals); passes = ceil(bits/8)
int radix_sort_pairs(uint32_t* keys, uint32_t* vals, int64_t n, int bits, SortWs ws, hipStream_t st) {
  if (n <= 1 || bits <= 0) return HOTK_OK;
  int passes = (bits + 7) / 8;
  int64_t nb = (n + RS_TILE - 1) / RS_TILE;
  uint32_t *ki = keys, *vi = vals, *ko = ws.keys_alt, *vo = ws.vals_alt;
  for (int p = 0; p < passes; ++p) {
    int shift = 8 * p;
    hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, ki, n, shift, ws.hist, nb);
    int rc = scan_exclusive(ws.hist, ws.hist, 256 * nb, ws.tile_sums, st);
    if (rc != HOTK_OK) return rc;
    hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, shift,
                       ws.hist, nb);
    uint32_t* t;
    t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  if (ki != keys) {
    hipMemcpyAsync(keys, ki, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(vals, vi, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  }
  return hotk_launch_status();
}

// ---------------------------------------------------------------------------------------------
// plan kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_plan_init(const int64_t* __restrict__ indices, const int32_t* __restrict__ n_nodes,
                            const int32_t* __restrict__ n_entries, const int64_t* __restrict__ entry_ofs, int64_t B,
                            int64_t Emax, int64_t Nmax, int32_t* __restrict__ rown, int32_t* __restrict__ coln,
                            uint8_t* __restrict__ kind, uint32_t* __restrict__ deg_col, uint32_t* __restrict__ deg_row,
                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int32_t* __restrict__ status) {
  const int64_t R = B * Emax;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = r / Emax, e = r - b * Emax;
    int32_t nn = n_nodes[b], ne = n_entries[b];
    int32_t rn = -1, cn = -1;
    uint8_t kd = 0;
    if (e < ne) {
      int64_t i = indices[2 * r], j = indices[2 * r + 1];
      if (i < 0 || j < 0 || i >= nn || j >= nn) {
        status[1] = 1;  // out of range: treated as padding so that no later kernel indexes out of bounds
      } else {
        rn = (int32_t)(b * Nmax + i);
        cn = (int32_t)(b * Nmax + j);
        kd = (i == j) ? 1 : 2;
        // layout invariant (batch/sparse.py:38, utils/sparse.py:43-48): rows 0..n-1 are (v,v) in order,
        // no other diagonal entries
        if ((e < nn) != (i == j) || (e < nn && i != e)) status[0] = 1;
        int64_t p = entry_ofs[b] + e;
        keys[p] = (uint32_t)cn;
        vals[p] = (uint32_t)r;
        atomicAdd(&deg_col[cn], 1u);
        atomicAdd(&deg_row[rn], 1u);
      }
    }
    rown[r] = rn;
    coln[r] = cn;
    kind[r] = kd;
  }
}

// entries that failed the range check leave holes in the compact list: mark them so they sort last
__global__ void k_fill_u32(uint32_t* __restrict__ p, int64_t n, uint32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_plan_nodes(const int32_t* __restrict__ n_nodes, int64_t B, int64_t Emax, int64_t Nmax,
                             int32_t* __restrict__ diag_pos, uint8_t* __restrict__ nkind) {
  const int64_t Nn = B * Nmax;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < Nn; n += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = n / Nmax, v = n - b * Nmax;
    bool ok = v < n_nodes[b];
    diag_pos[n] = ok ? (int32_t)(b * Emax + v) : -1;
    nkind[n] = ok ? 1 : 0;
  }
}

// keys[p] = table[vals[p]]  (re-key the column-sorted list by row node)
__global__ void k_rekey(const uint32_t* __restrict__ vals, const int32_t* __restrict__ table,
                        uint32_t* __restrict__ keys, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    keys[i] = (uint32_t)table[vals[i]];
}

__global__ void k_copy_i32(const uint32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (int32_t)src[i];
}

__global__ void k_plan_transpose(const int32_t* __restrict__ rown, const int32_t* __restrict__ coln,
                                 const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ row_perm,
                                 int64_t B, int64_t Emax, int64_t* __restrict__ t_indices, uint8_t* __restrict__ t_mask,
                                 int32_t* __restrict__ tpos) {
  const int64_t R = B * Emax;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    int32_t i = rown[r], j = coln[r];
    int64_t ti = 0;
    uint8_t tm = 0;
    int32_t tp = -1;
    if (i >= 0) {
      // row segment of node j is ordered by column node, ties by entry row: lower bound = first match
      int32_t lo = row_ptr[j], hi = row_ptr[j + 1];
      const int32_t end = hi;
      while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (coln[row_perm[mid]] < i) lo = mid + 1; else hi = mid;
      }
      if (lo < end) {
        int32_t q = row_perm[lo];
        if (coln[q] == i) {
          tp = q;
          tm = 1;
          ti = (int64_t)q - (r / Emax) * Emax;
        }
      }
    }
    t_indices[r] = ti;
    t_mask[r] = tm;
    tpos[r] = tp;
  }
}

// status[2] / status[3]: maximum column / row degree (host sizes the long-segment split from them)
__global__ void k_plan_maxdeg(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ row_ptr, int64_t Nn,
                              int32_t* __restrict__ status) {
  int mc = 0, mr = 0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < Nn; n += (int64_t)gridDim.x * blockDim.x) {
    mc = max(mc, col_ptr[n + 1] - col_ptr[n]);
    mr = max(mr, row_ptr[n + 1] - row_ptr[n]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mc = max(mc, __shfl_xor(mc, o, 64));
    mr = max(mr, __shfl_xor(mr, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&status[2], mc);
    atomicMax(&status[3], mr);
  }
}

int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n && b < 32) ++b;
  return b;
}

}  // namespace

extern "C" int hotk_sort_workspace_bytes(int64_t n, size_t* bytes) {
  HOTK_CHECK_ARG(bytes && n >= 0);
  *bytes = sort_ws_bytes(n);
  return HOTK_OK;
}

extern "C" int hotk_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int bits, void* workspace,
                                   size_t ws_bytes, void* stream) {
  HOTK_CHECK_ARG(n >= 0 && bits >= 0 && bits <= 32);
  if (n == 0) return HOTK_OK;
  HOTK_CHECK_ARG(keys && vals && workspace);
  HOTK_CHECK_ARG(n < ((int64_t)1 << 31));
  if (ws_bytes < sort_ws_bytes(n)) return HOTK_EWORKSPACE;
  return radix_sort_pairs(keys, vals, n, bits, carve_sort_ws(workspace, n), (hipStream_t)stream);
}

extern "C" int hotk_plan_workspace_bytes(int64_t B, int64_t Emax, int64_t Nmax, int64_t Ev, size_t* bytes) {
  HOTK_CHECK_ARG(bytes && B >= 0 && Emax >= 0 && Nmax >= 0 && Ev >= 0);
  int64_t Nn = B * Nmax;
  int64_t nts = (Nn + 1 + SCAN_TILE - 1) / SCAN_TILE + 1;
  *bytes = 2 * align256((size_t)(Ev > 0 ? Ev : 1) * 4) /* keys, vals */ + sort_ws_bytes(Ev) +
           align256((size_t)nts * 4);
  return HOTK_OK;
}

extern "C" int hotk_plan_build(const int64_t* indices, const int32_t* n_nodes, const int32_t* n_entries,
                               const int64_t* entry_ofs, int64_t B, int64_t Emax, int64_t Nmax, int64_t Ev,
                               int64_t* t_indices, uint8_t* t_mask, int32_t* rown, int32_t* coln, uint8_t* kind,
                               int32_t* tpos, int32_t* diag_pos, uint8_t* nkind, int32_t* col_ptr, int32_t* col_perm,
                               int32_t* row_ptr, int32_t* row_perm, int32_t* status, void* workspace, size_t ws_bytes,
                               void* stream) {
  HOTK_CHECK_ARG(B >= 0 && Emax >= 0 && Nmax >= 0 && Ev >= 0);
  const int64_t R = B * Emax, Nn = B * Nmax;
  HOTK_CHECK_ARG(R < ((int64_t)1 << 31) && Nn < ((int64_t)1 << 31) - 1 && Ev <= R);
  HOTK_CHECK_ARG(status && col_ptr && row_ptr);
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st);
  hipMemsetAsync(col_ptr, 0, (size_t)(Nn + 1) * 4, st);
  hipMemsetAsync(row_ptr, 0, (size_t)(Nn + 1) * 4, st);
  if (R == 0 || Nn == 0) return hotk_launch_status();
  HOTK_CHECK_ARG(indices && n_nodes && n_entries && entry_ofs && t_indices && t_mask && rown && coln && kind &&
                 tpos && diag_pos && nkind && col_perm && row_perm && workspace);
  size_t need;
  hotk_plan_workspace_bytes(B, Emax, Nmax, Ev, &need);
  if (ws_bytes < need) return HOTK_EWORKSPACE;
  char* p = (char*)workspace;
  uint32_t* keys = (uint32_t*)p; p += align256((size_t)(Ev > 0 ? Ev : 1) * 4);
  uint32_t* vals = (uint32_t*)p; p += align256((size_t)(Ev > 0 ? Ev : 1) * 4);
  SortWs sws = carve_sort_ws(p, Ev);
  p += sort_ws_bytes(Ev);
  uint32_t* tile_sums = (uint32_t*)p;

  const int T = 256;
  if (Ev > 0) {
    // holes left by rejected entries sort to the end (key 0xFFFFFFFF) and point at row 0
    hipLaunchKernelGGL(k_fill_u32, dim3(hotk_grid(Ev, T)), dim3(T), 0, st, keys, Ev, 0xFFFFFFFFu);
    hipMemsetAsync(vals, 0, (size_t)Ev * 4, st);
  }
  hipLaunchKernelGGL(k_plan_init, dim3(hotk_grid(R, T)), dim3(T), 0, st, indices, n_nodes, n_entries, entry_ofs, B,
                     Emax, Nmax, rown, coln, kind, (uint32_t*)col_ptr, (uint32_t*)row_ptr, keys, vals, status);
  hipLaunchKernelGGL(k_plan_nodes, dim3(hotk_grid(Nn, T)), dim3(T), 0, st, n_nodes, B, Emax, Nmax, diag_pos, nkind);
  const int bits = bits_for(Nn);
  int rc;
  if (Ev > 0) {
    // rejected entries keep key 0xFFFFFFFF only if we sort all 32 bits; they are an error condition that
    // the host turns into an exception (status[1]), so ordering among them does not matter.
    rc = radix_sort_pairs(keys, vals, Ev, bits, sws, st);  // by column node, stable in entry row
    if (rc != HOTK_OK) return rc;
    hipLaunchKernelGGL(k_copy_i32, dim3(hotk_grid(Ev, T)), dim3(T), 0, st, vals, col_perm, Ev);
    hipLaunchKernelGGL(k_rekey, dim3(hotk_grid(Ev, T)), dim3(T), 0, st, vals, rown, keys, Ev);
    rc = radix_sort_pairs(keys, vals, Ev, bits, sws, st);  // by row node, stable in (column node, entry row)
    if (rc != HOTK_OK) return rc;
    hipLaunchKernelGGL(k_copy_i32, dim3(hotk_grid(Ev, T)), dim3(T), 0, st, vals, row_perm, Ev);
  }
  rc = scan_exclusive((uint32_t*)col_ptr, (uint32_t*)col_ptr, Nn + 1, tile_sums, st);
  if (rc != HOTK_OK) return rc;
  rc = scan_exclusive((uint32_t*)row_ptr, (uint32_t*)row_ptr, Nn + 1, tile_sums, st);
  if (rc != HOTK_OK) return rc;
  hipLaunchKernelGGL(k_plan_transpose, dim3(hotk_grid(R, T)), dim3(T), 0, st, rown, coln, row_ptr, row_perm, B, Emax,
                     t_indices, t_mask, tpos);
  hipLaunchKernelGGL(k_plan_maxdeg, dim3(hotk_grid(Nn, T, 1024)), dim3(T), 0, st, col_ptr, row_ptr, Nn, status);
  return hotk_launch_status();
}
