/*
 * hotk.h -- C ABI of libhotk.so: hand-written gfx950 (MI355X) HIP kernels for the second-order sparse
 * kernel-attention path of Higher-Order Transformers (reference: jw9730/hot, pure PyTorch; the reference
 * has no FFI of its own, so every entry point cites the reference Python it replaces, file:line relative
 * to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (contiguous, 4-byte aligned; 16-byte alignment
 *     enables the vector paths); nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns;
 *   - return value: HOTK_OK (0) or a negative HOTK_E* code; hotk_strerror() names it;
 *   - "rows": order-2 tensors are [R = B*Emax, C] (padded entry rows), order-1 tensors are
 *     [Nn = B*Nmax, C] (padded node rows); `ld*` is the row stride in floats, so a column slice of a wider
 *     tensor is passed as (ptr + col0, ld) without a copy;
 *   - kind[r] (uint8): 0 = padding, 1 = diagonal entry / valid node, 2 = off-diagonal entry;
 *     `keep` is a bit set over kinds (bit k set = rows of kind k pass, others are written as zero).
 */
#ifndef HOTK_H
#define HOTK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOTK_OK 0
#define HOTK_EINVAL (-1)   /* null pointer / bad size / bad mode          */
#define HOTK_EWORKSPACE (-2) /* workspace too small                        */
#define HOTK_ELAUNCH (-3)  /* hipLaunch / runtime error                    */
#define HOTK_EUNSUPPORTED (-4)
#define HOTK_ETOOLARGE (-5) /* problem exceeds the fixed capacity of a *_static entry point */

#define HOTK_KEEP_PAD 1u
#define HOTK_KEEP_DIAG 2u
#define HOTK_KEEP_OFFDIAG 4u

const char* hotk_strerror(int code);
int hotk_version(void);

/* ------------------------------------------------------------------------------------------------
 * K1  edge-index sort / segment construction
 * replaces hot_pytorch/utils/sparse.py:102-127 (get_transpose_info, O(E^2) all-pairs match) and the
 * implicit grouping done by torch.sparse coalesce() in models/sparse/kernelattncoef.py:103-122.
 *
 * in : indices [B,Emax,2] int64 (local node ids, padding rows (0,0)); n_nodes[B], n_entries[B]
 *      (= n_b + e_b) int32; entry_ofs[B] int64 = exclusive cumsum of n_entries; Ev = sum(n_entries)
 * out: t_indices [B,Emax] int64, t_mask [B,Emax] bool -- bit-exact with the reference;
 *      rown/coln [R] int32 padded node row of i / j (-1 on padding), kind [R] uint8,
 *      tpos [R] int32 padded entry row of the transposed entry (-1 if none),
 *      diag_pos [Nn] int32 entry row of node's diagonal (-1 on padded nodes), nkind [Nn] uint8,
 *      col_ptr/row_ptr [Nn+1], col_perm/row_perm [Ev] int32: entries grouped by column / row node,
 *      column groups ordered by entry row, row groups ordered by (column node, entry row);
 *      status [4] int32: [0] diagonal-prefix layout invariant broken (batch/sparse.py:38),
 *      [1] index out of range, [2] max column degree, [3] max row degree.
 * ---------------------------------------------------------------------------------------------- */
int hotk_plan_workspace_bytes(int64_t B, int64_t Emax, int64_t Nmax, int64_t Ev, size_t* bytes);
int hotk_plan_build(const int64_t* indices, const int32_t* n_nodes, const int32_t* n_entries,
                    const int64_t* entry_ofs, int64_t B, int64_t Emax, int64_t Nmax, int64_t Ev,
                    int64_t* t_indices, uint8_t* t_mask, int32_t* rown, int32_t* coln, uint8_t* kind,
                    int32_t* tpos, int32_t* diag_pos, uint8_t* nkind, int32_t* col_ptr, int32_t* col_perm,
                    int32_t* row_ptr, int32_t* row_perm, int32_t* status, void* workspace, size_t ws_bytes,
                    void* stream);
/* stable LSD radix sort of (key,value) uint32 pairs on the low `bits` bits of key; used by the plan and
 * exported for tests.  workspace: hotk_sort_workspace_bytes(n). keys/vals are sorted in place. */
int hotk_sort_workspace_bytes(int64_t n, size_t* bytes);
int hotk_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int bits, void* workspace, size_t ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4  coalesced row gather / mask / segment sums
 * ---------------------------------------------------------------------------------------------- */
/* out[r] = idx[r] >= 0 ? src[idx[r]] : 0   (idx NULL = identity); accumulate != 0 adds into out.
 * replaces do_transpose (utils/sparse.py:130-134), get_diag / to_diag (utils/sparse.py:34-87) and their
 * autograd mirrors. */
int hotk_gather_rows(const float* src, int64_t ld_src, const int32_t* idx, float* out, int64_t ld_out,
                     int64_t n_out, int64_t C, int accumulate, void* stream);
/* out[r] = (keep >> kind[r]) & 1 ? src[r] : fill.  replaces masked_fill (utils/set.py:47-48), get_nondiag
 * (utils/sparse.py:90-99), Batch.apply_mask (batch/sparse.py:77-80). */
int hotk_mask_rows(const float* src, int64_t ld_src, const uint8_t* kind, unsigned keep, float fill,
                   float* out, int64_t ld_out, int64_t n, int64_t C, void* stream);
/* out[s] = sum over p in [seg_begin[s], seg_end[s]) of src[perm[p]] restricted to kinds in `keep`
 * (perm NULL = identity, kind NULL = no filter).  One thread adds the rows of a segment in segment order: bitwise
 * reproducible, no atomics.  max_seg_len > HOTK_SEG_CHUNK returns HOTK_EUNSUPPORTED: the caller splits longer segments
 * into consecutive pieces and sums the partial rows with a second call (hot_amd.ops.refine_seg does this).
 * replaces the M.float().bmm(values) poolings of models/sparse/linear.py:187-231 and scatter_add_
 * backward of the gathers. */
#define HOTK_SEG_CHUNK 1024
int hotk_seg_sum_rows(const float* src, int64_t ld_src, const int32_t* seg_begin, const int32_t* seg_end,
                      const int32_t* perm, const uint8_t* kind, unsigned keep, float* out, int64_t ld_out,
                      int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream);
/* Two segment sums of the SAME source in one pass: out_a[s] / out_b[s] = the hotk_seg_sum_rows result for segment s of
 * list A / list B (both with a position -> row map and the kind filter), bit for bit.  For the row and column CSR of one
 * plan -- the two node gradients of Linear(2, 2, 'light'), models/sparse/linear.py:157-159 under autograd (scatter_add_
 * of the two gathers) -- every source row is then fetched from HBM once and the second time from the last-level cache,
 * instead of crossing HBM in two launches.  n_seg segments in both lists, out_a / out_b share ld_out. */
int hotk_seg_sum_rows_pair(const float* src, int64_t ld_src, const int32_t* a_begin, const int32_t* a_end,
                           const int32_t* a_perm, const int32_t* b_begin, const int32_t* b_end, const int32_t* b_perm,
                           const uint8_t* kind, unsigned keep, float* out_a, float* out_b, int64_t ld_out, int64_t n_seg,
                           int64_t C, int64_t max_seg_len, void* stream);
/* The same plus out_own[s] = src[own_row[s]] (zeros where own_row[s] < 0) in the same launch: with own_row = the plan's
 * diag_pos this is hotk_gather_rows of the diagonal entries, i.e. all three node gradient blocks of Linear(2, 2, 'light')
 * (models/sparse/linear.py:152-159 under autograd) from one launch -- small batches are bound by launches, not bytes.
 * out_own shares ld_out; own_row / out_own both NULL: hotk_seg_sum_rows_pair. */
int hotk_seg_sum_rows_pair_own(const float* src, int64_t ld_src, const int32_t* a_begin, const int32_t* a_end,
                               const int32_t* a_perm, const int32_t* b_begin, const int32_t* b_end, const int32_t* b_perm,
                               const uint8_t* kind, unsigned keep, float* out_a, float* out_b, const int32_t* own_row,
                               float* out_own, int64_t ld_out, int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream);
/* y[r] = kind 2: y + b_off, kind 1: y + b_diag, kind 0: 0   (in place).  order-2 _Bias + apply_mask
 * (models/sparse/linear.py:297-306,345-346); order 1: pass nkind and b_diag (b_off may be NULL). */
int hotk_bias_mask_rows(float* y, int64_t ld, const uint8_t* kind, const float* b_off, const float* b_diag,
                        int64_t n, int64_t C, void* stream);
/* db_off[c] = sum over kind-2 rows of dy, db_diag[c] = sum over kind-1 rows (either may be NULL): gradients of the two
 * _Bias vectors (models/sparse/linear.py:293-306).  Two passes (per-block partial rows in the workspace, then added in
 * a fixed order: 16 interleaved slices of the block sequence, then the slices): bitwise reproducible, no atomics;
 * outputs are overwritten.
 * workspace: hotk_bias_grad_workspace_bytes(n, C). */
int hotk_bias_grad_workspace_bytes(int64_t n, int64_t C, size_t* bytes);
int hotk_bias_grad_rows(const float* dy, int64_t ld, const uint8_t* kind, float* db_off, float* db_diag,
                        int64_t n, int64_t C, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4  order-2 equivariant basis, 'light' configuration (models/sparse/linear.py:143-161):
 * X5[r] = [ diag? x_r | off? x_r | off&tau? x_t(r) | off? x_(i,i) | off? x_(j,j) ]   ([R, 5*D])
 * and its transpose (gradient) map.
 * ---------------------------------------------------------------------------------------------- */
int hotk_basis22_light_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* rown,
                           const int32_t* coln, const int32_t* tpos, const int32_t* diag_pos, float* x5,
                           int64_t ld_x5, int64_t R, int64_t D, void* stream);
int hotk_basis22_light_bwd(const float* dx5, int64_t ld_dx5, const uint8_t* kind, const int32_t* rown,
                           const int32_t* tpos, const int32_t* row_ptr, const int32_t* row_perm,
                           const int32_t* col_ptr, const int32_t* col_perm, float* dx, int64_t ld_dx,
                           int64_t R, int64_t D, void* stream);

/* Factored form of the same layer (the basis is linear in its five blocks): the node-valued blocks x_(v,v), x_(i,i),
 * x_(j,j) are projected once per node, yn = x_diag @ [W1 | W4 | W5]  ([Nn, 3C]), and only the entry-valued pair
 *   x2[r] = off-diagonal ? [ x_r | tau ? x_t(r) : 0 ] : 0   ([R, 2D])
 * goes through the big GEMM z = x2 @ [W2; W3].  hotk_assemble22 finishes y in place on z:
 *   off-diagonal (i,j): z + yn[i][C..2C) + yn[j][2C..3C) + b_off;  diagonal (v,v): yn[v][0..C) + b_diag;  padding: 0
 * (models/sparse/linear.py:143-161 basis, :46-48 projection, :297-306 bias, :345-346 mask).  pair_bwd is the transpose
 * map: dx[r] = off-diagonal ? dx2[r][0..D) + (tau ? dx2[t(r)][D..2D) : 0) : diagonal ? dxd[node] : 0, plus add[r] on valid
 * rows when add != NULL (another gradient of the same input -- fc_v's -- folded into the pass). */
int hotk_basis22_pair_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* tpos, float* x2,
                          int64_t ld_x2, int64_t R, int64_t D, void* stream);
/* pair_fwd plus xd[n] = x[diag_pos[n]] (zeros for a padding node, diag_pos[n] < 0) in the same launch: the node rows
 * x_(v,v) that models/sparse/linear.py:152-153 gathers (= hotk_gather_rows over diag_pos).  xd [Nn, D], row stride ld_xd. */
int hotk_basis22_pair_fwd_diag(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* tpos, float* x2,
                               int64_t ld_x2, int64_t R, int64_t D, const int32_t* diag_pos, float* xd, int64_t ld_xd,
                               int64_t Nn, void* stream);
int hotk_basis22_pair_bwd(const float* dx2, int64_t ld_dx2, const uint8_t* kind, const int32_t* tpos,
                          const int32_t* rown, const float* dxd, int64_t ld_dxd, const float* add, int64_t ld_add,
                          float* dx, int64_t ld_dx, int64_t R, int64_t D, void* stream);
int hotk_assemble22(float* z, int64_t ld, const float* yn, int64_t ld_yn, const uint8_t* kind, const int32_t* rown,
                    const int32_t* coln, const float* b_off, const float* b_diag, int64_t R, int64_t C, void* stream);

/* The full 15-operator order-2 -> order-2 basis of Linear(2, 2, cfg='default') (models/sparse/linear.py:143-238; the
 * reference pools with [B,E,E] boolean masks and bmm's, O(E^2)) as one expand kernel over closed forms of per-node and
 * per-graph sums:  out[r] = [V1 | ... | V15]  ([R, 15 D], the reference's block order).
 *   node_tab  [Nn, 3 D] = A_vv | off-diagonal row sum | off-diagonal column sum      (hotk_gather_rows / hotk_seg_sum_rows)
 *   graph_tab [B, 2 D]  = trace | off-diagonal total;   od / idg [Nn], n_graph / eo_graph [B]: element counts as floats
 *   normalize: every pooled operator divided by its element count, a pool over an empty set is 0 (linear.py:162-238).
 * Transpose map (the basis is linear in the values, nothing is saved): _bwd_entries turns the gradient g [R, 15 D] into
 *   w [R, 10 D] = [ own | transposed partner's | row node: d diag, d rowsum, d colsum | column node: same | d trace, d total ];
 * the caller sums w[:, 2D:5D] per row node, w[:, 5D:8D] per column node and w[:, 8D:10D] per graph (hotk_seg_sum_rows) into
 * dnode [Nn, 3 D] / dgraph [B, 2 D]; _bwd_gather finishes dx [R, D]. */
int hotk_basis22_full_fwd(const float* x, int64_t ld_x, const uint8_t* kind, const int32_t* rown, const int32_t* coln,
                          const int32_t* tpos, const float* node_tab, int64_t ld_nt, const float* graph_tab, int64_t ld_gt,
                          const float* od, const float* idg, const float* n_graph, const float* eo_graph, float* out,
                          int64_t ld_out, int64_t R, int64_t D, int64_t Emax, int normalize, void* stream);
int hotk_basis22_full_bwd_entries(const float* g, int64_t ld_g, const uint8_t* kind, const int32_t* rown,
                                  const int32_t* coln, const int32_t* tpos, const float* od, const float* idg,
                                  const float* n_graph, const float* eo_graph, float* w, int64_t ld_w, int64_t R,
                                  int64_t D, int64_t Emax, int normalize, void* stream);
int hotk_basis22_full_bwd_gather(const float* w, int64_t ld_w, const uint8_t* kind, const int32_t* rown,
                                 const int32_t* coln, const int32_t* tpos, const float* dnode, int64_t ld_dn,
                                 const float* dgraph, int64_t ld_dg, float* dx, int64_t ld_dx, int64_t R, int64_t D,
                                 int64_t Emax, void* stream);

/* Thin GEMMs of the dense projections with the weight resident in LDS (fp32 MFMA) and the row epilogue fused:
 *   Y[R, C] = A[R, K] @ W[K, C] (+ bias[C]) (+ res[R, C])        W row-major [K, C]
 * stands in for nn.Linear / `V @ weight` and their data gradients over the ~1e7 entry rows of a batch
 * (models/sparse/kernelattn.py:46-51,65-82; models/sparse/linear.py:46-48).  Shapes: (K, C) with K in {32, 64, 288, 320}
 * and C a multiple of {160 or 96, 96, 64, 32} resp. (the C1 layer; K = 32 with C % 96 == 0: the node-level products 32 -> 384
 * and, as column blocks 480 + 384 with Y / bias offset by the caller, 32 -> 864), K*C*4 <= 150 KB; hotk_thin_gemm_supported tells, other
 * shapes return HOTK_EUNSUPPORTED (the Python layer then uses the library GEMM).  A rows 16-byte aligned.
 * hotk_linear22_entry_fwd is the entry GEMM of the factored Linear(2,2,'light') with hotk_assemble22's epilogue applied
 * to the accumulators (one pass over Y instead of three); Nn = rows of the node table yn (Nn * ld_yn < 2^32, else
 * HOTK_EUNSUPPORTED: the caller then runs the library GEMM followed by hotk_assemble22). */
int hotk_thin_gemm_supported(int64_t K, int64_t C);
int hotk_thin_gemm_bias(const float* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                        const float* res, int64_t ld_res, float* Y, int64_t ldy, int64_t R, void* stream);
/* The K = 32 product with the result rounded to bf16 at the store (Y: bf16 rows, 8-byte aligned, ldy % 4 == 0): fc_v
 * and d att = d y . W_o of the bf16 storage mode (models/sparse/kernelattn.py:51,65-72 at the north_star's 2e-2). */
int hotk_thin_gemm_bias_bf16(const float* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                             void* Y, int64_t ldy, int64_t R, void* stream);
/* Y[R, C] (fp32) = A[R, K] (bf16 rows, lda in bf16 elements, a multiple of 8) @ W[K, C] (fp32) (+ bias) (+ res, fp32): the wide-input
 * products of the bf16 storage mode (fc_o: K = 320; the entry blocks' data gradient of Linear(2,2,'light'): K = 288 -> 64; no
 * counterpart in the reference).  A bf16 value is its own plane: three exact products per step against the split weight. */
int hotk_thin_gemm_bias_bf16in(const void* A, int64_t lda, const float* W, int64_t K, int64_t C, const float* bias,
                               const float* res, int64_t ld_res, float* Y, int64_t ldy, int64_t R, void* stream);
int hotk_linear22_entry_fwd(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C, const float* yn,
                            int64_t ld_yn, const uint8_t* kind, const int32_t* rown, const int32_t* coln,
                            const float* b_off, const float* b_diag, float* Y, int64_t ldy, int64_t R, int64_t Nn, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  FAVOR+ random-feature maps (models/common/kernel.py:81-138 fused with the head split, loop mask
 * and padding mask of models/sparse/kernelattncoef.py:19-28).
 * x [n, HH*d] -> phi [n, HH*m]; proj [m,d].  mode: 0 generalized ReLU (+1e-3), 1 softmax query,
 * 2 softmax key (needs gstat from hotk_phi_key_stats).  Rows whose kind is not in `keep` are zero.
 * ---------------------------------------------------------------------------------------------- */
#define HOTK_PHI_RELU 0
#define HOTK_PHI_SOFTMAX_Q 1
#define HOTK_PHI_SOFTMAX_K 2
int hotk_phi_fwd(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind, unsigned keep,
                 float* phi, int64_t ld_phi, int64_t n, int HH, int d, int m, int mode, const float* gstat,
                 void* stream);
int hotk_phi_bwd(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind, unsigned keep,
                 const float* dphi, int64_t ld_dphi, float* dx, int64_t ld_dx, int64_t n, int HH, int d,
                 int m, int mode, const float* gstat, void* stream);
/* gstat[0] = max over ALL n rows (padding included, as the reference does, kernel.py:112) and heads of the
 * projected data; workspace >= hotk_phi_stats_workspace_bytes(n, HH, m). */
int hotk_phi_stats_workspace_bytes(int64_t n, int HH, int m, size_t* bytes);
int hotk_phi_key_stats(const float* x, int64_t ld_x, const float* proj, int64_t n, int HH, int d, int m,
                       float* gstat, void* workspace, size_t ws_bytes, void* stream);
/* gstat[1] = sum over rows in keep of dphi * ratio * exp(.) (d loss / d(-max)), gstat[2] = number of
 * elements equal to the max; needed by the key-branch backward.
 * Edge-shard mode (one graph over several ranks): gstat[0] is MAX-all-reduced after hotk_phi_key_stats and
 * gstat[1], gstat[2] are SUM-all-reduced after this call; gstat[3] = 1/world (0 = not sharded) weights maxima that sit on
 * rows every rank holds (kind != 2), in the count here and in hotk_phi_bwd's share of the maximum's gradient. */
int hotk_phi_key_bwd_stats(const float* x, int64_t ld_x, const float* proj, const uint8_t* kind, unsigned keep,
                           const float* dphi, int64_t ld_dphi, int64_t n, int HH, int d, int m, float* gstat,
                           void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  segmented key-value reductions and their application
 * (models/sparse/kernelattncoef.py:74-91 full / :103-122 partial sums, :124-175 query application).
 * F = HH*m features, c = value channels per head; KV is [n_seg, F, c+1]: channels 0..c-1 hold
 * sum phiK (x) V, channel c holds sum phiK * aug (aug = 1 in the forward pass: the normaliser).
 * ---------------------------------------------------------------------------------------------- */
/* KV[s][f][ch] = sum_{p in seg s} K[kidx[e]][f] * (ch<c ? V[vidx[e]][hh*c+ch] : AUG[e][hh] or 1), e = perm[p].
 * kidx/vidx/perm NULL = identity; vidx[e] < 0 contributes zeros to the value channels only.
 * Segment order, no atomics; max_seg_len > HOTK_SEG_CHUNK returns HOTK_EUNSUPPORTED (split the segments and add the
 * partial sums with hotk_seg_sum_rows, as hot_amd.ops.seg_outer_sum does). */
int hotk_seg_outer_sum(const float* K, int64_t ld_k, const int32_t* kidx, const float* V, int64_t ld_v,
                       const int32_t* vidx, const float* aug, int64_t ld_aug, const int32_t* seg_begin,
                       const int32_t* seg_end, const int32_t* perm, float* KV, int64_t n_seg, int HH, int m,
                       int c, int64_t max_seg_len, void* stream);
/* The three row-wise calls below take `order` (NULL = natural): the n rows are VISITED as r = order[0..n); rows that
 * are not listed are left untouched.  With the rows of one segment adjacent in it (a CSR permutation, e.g. the plan's
 * col_perm when sidx is the column node) a segment's [F, c+1] block is fetched once per segment instead of once per row.
 * Results do not depend on it. */
/* out[orow][hh*c+ch] = sum_mm Q[qidx[r]][hh*m+mm] * KV[s][hh*m+mm][ch]  (/ guard(sum_mm Q*KV[..][c]) if
 * normalize), s = sidx ? sidx[r] : r / seg_div, orow = oidx ? oidx[r] : r.  Rows with kind not in keep or
 * s < 0 write zeros (skipped when oidx is given and orow < 0). */
int hotk_row_contract(const float* Q, int64_t ld_q, const int32_t* qidx, const float* KV, const int32_t* sidx,
                      int64_t seg_div, const uint8_t* kind, unsigned keep, const int32_t* oidx, float* out,
                      int64_t ld_out, int64_t n, int HH, int m, int c, int normalize, const int32_t* order,
                      void* stream);
/* backward statistics of the normalised application: X[r][hh*c+ch] = dO/den, XA[r][hh] = -(dO.O)/den
 * (0 where the 1e-5 guard was taken or the row is masked). */
int hotk_apply_bwd_stats(const float* Q, int64_t ld_q, const float* KV, const int32_t* sidx, int64_t seg_div,
                         const uint8_t* kind, unsigned keep, const float* dO, int64_t ld_do, float* X,
                         int64_t ld_x, float* XA, int64_t ld_xa, int64_t n, int HH, int m, int c,
                         const int32_t* order, void* stream);
/* out[r][f] = sum_ch KV[s][f][ch] * X[xidx[r]][hh*c+ch] + KV[s][f][c] * (XA ? XA[r][hh] : 1) */
int hotk_row_expand(const float* KV, const int32_t* sidx, int64_t seg_div, const float* X, int64_t ld_x,
                    const int32_t* xidx, const float* XA, int64_t ld_xa, const uint8_t* kind, unsigned keep,
                    float* out, int64_t ld_out, int64_t n, int HH, int m, int c, const int32_t* order, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Order-0 softmax readout: segmented softmax pooling with a per-segment (per-graph) query vector
 * (models/sparse/attncoef.py:25-59 `_vector_query`, apply_attn with a zeroth-order query :165-173; callers
 * SelfAttn._1_to_0 / _2_to_0, models/sparse/attn.py:82-131).
 *   s[r][h] = q[b][h*d..] . k[r][h*d..] / sqrt(d) over the rows r of segment b whose kind is in `keep`;
 *   alpha = softmax over those rows (0 elsewhere, all 0 for an empty set); out[b][h*c+ch] = sum_r alpha[r][h] v[r][h*c+ch].
 * alpha [n_rows, H] is written for the rows inside segments (caller zero-fills) and feeds the backward pass, which
 * writes dq [n_seg, H*d] and the kept rows of dk / dv (caller zero-fills those too).
 * One wavefront walks a segment, so a very long set (one giant graph) is split by the caller into pieces: q_row[s] (may
 * be NULL = s) names the row of q / out / dout that piece s belongs to, stats [n_seg, H, 2] (may be NULL) receives each
 * piece's (maximum score, sum of exponentials) from which the caller merges the pieces' out and alpha exactly
 * (hot_amd.ops.SoftmaxPool); the backward pass takes the MERGED alpha / out and yields per-piece partial dq rows.
 * ---------------------------------------------------------------------------------------------- */
int hotk_seg_softmax_pool_fwd(const float* q, int64_t ld_q, const float* k, int64_t ld_k, const float* v, int64_t ld_v,
                              const int32_t* seg_begin, const int32_t* seg_end, const uint8_t* kind, unsigned keep,
                              float* out, int64_t ld_out, float* alpha, float* stats, const int32_t* q_row,
                              int64_t n_seg, int H, int d, int c, void* stream);
int hotk_seg_softmax_pool_bwd(const float* q, int64_t ld_q, const float* k, int64_t ld_k, const float* v, int64_t ld_v,
                              const int32_t* seg_begin, const int32_t* seg_end, const float* alpha, const float* out,
                              int64_t ld_out, const float* dout, int64_t ld_dout, float* dq, int64_t ld_dq, float* dk,
                              int64_t ld_dk, float* dv, int64_t ld_dv, const int32_t* q_row, int64_t n_seg, int H, int d,
                              int c, void* stream);

/* Backward of a narrow-input linear layer y[R, C] = x[R, K] @ W[K, C] + b in one pass over g = dy [R, C]
 * (nn.Linear fc_v of models/sparse/kernelattn.py:51,65-68 under autograd; the reference gets the three gradients from
 * aten's addmm backward):  dx[R, K] = g @ W^T;  partial[n][K + 2][C]: per-workgroup partial sums of x^T @ g (rows 0..K-1)
 * and of the column sums of g (rows K, K + 1).  The caller sums the n = hotk_linear_bwd_partials(R) blocks in order:
 * dW = sum[0:K], db = sum[K] + sum[K + 1].  dx = NULL: only the partial sums (W is not read) -- the weight gradient of a
 * wide-INPUT layer y = a @ W^T is the same product with x := dy and g := a (fc_o, kernelattn.py:52).  Shapes: K = 32 and C a multiple of 32 up to 1 024 (hotk_linear_bwd_supported; launches of at most
 * five 32-column chunks side by side: fc_v 320, the node-level products of Linear(2,1/2,'light') 384 / 864), others
 * HOTK_EUNSUPPORTED.  g, x, dx rows 16-byte aligned. */
int hotk_linear_bwd_supported(int64_t K, int64_t C);
int hotk_linear_bwd_partials(int64_t R, int64_t* n_partials);
int hotk_linear_bwd_fused(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t K, int64_t C,
                          float* dx, int64_t lddx, float* partial, int64_t R, void* stream);
/* the same with g and x as bf16 tensors (the bf16 storage mode; no counterpart in the reference): leading dimensions in bf16
 * elements (multiples of 4, rows 8-byte aligned); values are widened to fp32 on the way in, W / dx / partial are fp32. */
int hotk_linear_bwd_fused_bf16(const void* g, int64_t ldg, const void* x, int64_t ldx, const float* W, int64_t K, int64_t C,
                               float* dx, int64_t lddx, float* partial, int64_t R, void* stream);

/* Entry-level reductions of the backward pass of the factored Linear(2, 2, 'light') (models/sparse/linear.py:143-161,
 * 293-306 under autograd: the scatter_add_ of the diagonal / row / column gathers and the weight gradient of the entry
 * blocks) in ONE pass over g = dy [R, C], graph by graph:
 *   dyn[v][0..C) = g at the diagonal entry (v, v);  dyn[v][C..2C) / [2C..3C) = sums of g over the off-diagonal entries of
 *   row / column v;  partial[n][K2][C] = per-workgroup parts of x2^T @ g (x2 = [x_ij | x_ji] [R, K2], zero on diagonal
 *   rows), summed by the caller in order (n = hotk_linear22_bwd_reduce_partials(B)).
 * All as bf16 matrix-core products on the exact three-way split of g (the sums against one-hot operands): fp32-exact
 * (2e-7), no atomics, bitwise reproducible.  Rows of graph b are [g_ebeg[b], g_eend[b]); node rows b * Nmax + local id;
 * n_nodes[b] nodes are real, the other node rows of dyn leave as zeros.  Every dyn row is written.  Every row of a graph's range
 * is taken as an entry, diagonal iff its row node equals its column node (`kind` is not read and may be NULL).  Shapes: K2 = 64,
 * C = 288, Nmax <= 64 (hotk_linear22_bwd_reduce_supported), others HOTK_EUNSUPPORTED.  g rows 16-byte aligned. */
int hotk_linear22_bwd_reduce_supported(int64_t K2, int64_t C, int64_t Nmax);
int hotk_linear22_bwd_reduce_partials(int64_t B, int64_t* n_partials);
int hotk_linear22_bwd_reduce(const float* g, int64_t ldg, const float* x2, int64_t ld_x2, int64_t K2, int64_t C,
                             const uint8_t* kind, const int32_t* rown, const int32_t* coln, const int32_t* g_ebeg,
                             const int32_t* g_eend, const int32_t* n_nodes, int64_t B, int64_t Nmax, int64_t R, float* dyn,
                             int64_t ld_dyn, float* partial, void* stream);
/* the same with g and x2 as bf16 tensors (the bf16 storage mode; no counterpart in the reference): leading dimensions in
 * bf16 elements, multiples of 8; a bf16 value is its own single plane, so every product is exact and dyn / partial (fp32)
 * are fp32 sums of the stored bf16 values. */
int hotk_linear22_bwd_reduce_bf16(const void* g, int64_t ldg, const void* x2, int64_t ld_x2, int64_t K2, int64_t C,
                                  const uint8_t* kind, const int32_t* rown, const int32_t* coln, const int32_t* g_ebeg,
                                  const int32_t* g_eend, const int32_t* n_nodes, int64_t B, int64_t Nmax, int64_t R,
                                  float* dyn, int64_t ld_dyn, float* partial, void* stream);

/* Wide dense projections as fp32-exact tiled GEMMs on the bf16 matrix pipe (csrc/gemm6.hip): every fp32 operand is split
 * exactly into three bf16 numbers and the six products of relative weight >= 2^-16 are summed in fp32 (1.8e-7 against
 * fp64, as the library's fp32 GEMM).  replaces the aten mm / addmm calls under nn.Linear and _Weight.get_output for the
 * shapes the LDS-resident thin GEMM does not take (models/sparse/linear.py:46-48, kernelattn.py:65-72, and their
 * autograd): C2's hidden widths 256 / 512, the dim_qk = 64 head shape.
 *   hotk_gemm6_split_weight: planes [3][K / 32][N][32] bf16 (hi | mid | lo; 3 N K elements) of the B operand of C = A . B:
 *       transposed = 1: B = W with W [K, N] row-major (y = x @ W);  0: B = W^T with W [N, K] row-major (nn.Linear, dx = g @ W^T)
 *   hotk_gemm6_nn: C[M, N] = A[M, K] . B (+ bias[N]);  K % 32 == 0, A rows 16-byte aligned
 *   hotk_gemm6_tn: partial[s][K][N] = A[rows of slab s]^T . G[rows of slab s], s < hotk_gemm6_tn_partials(M, K, N); the
 *       caller adds the slabs in order (weight gradient x^T . dy);  K, N % 4 == 0, rows 16-byte aligned */
int hotk_gemm6_nn_supported(int64_t M, int64_t K, int64_t N);
int hotk_gemm6_tn_supported(int64_t M, int64_t K, int64_t N);
int hotk_gemm6_split_weight(const float* W, int64_t ldw, int64_t K, int64_t N, int transposed, void* planes, void* stream);
int hotk_gemm6_nn(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc, int64_t M,
                  int64_t K, int64_t N, void* stream);
int hotk_gemm6_tn_partials(int64_t M, int64_t K, int64_t N, int64_t* n_partials);
int hotk_gemm6_tn(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t M, int64_t K, int64_t N,
                  void* stream);
/* The same products over a LIST of rows: rows[0 .. n_rows) ascending, each < M -- the valid rows of the reference's padded
 * [B, Emax, D] / [B, Nmax, D] batch (models/sparse/linear.py:46-48 multiplies the padding rows too and apply_mask(0)
 * zeroes them afterwards, :345-346; ~30 % of a PCQM4M batch's rows).  _nn_rows: C[rows[i]] = A[rows[i]] . B + bias, every
 * other row of C[M, N] is written as zeros (the mask is part of the product);  _tn_rows: the sum runs over the listed
 * rows only, partial blocks: hotk_gemm6_tn_partials(n_rows, K, N).  rows == NULL: all rows, as the calls above. */
int hotk_gemm6_nn_rows(const float* A, int64_t lda, const void* planes, const float* bias, float* C, int64_t ldc, int64_t M,
                       int64_t K, int64_t N, const int32_t* rows, int64_t n_rows, void* stream);
int hotk_gemm6_tn_rows(const float* A, int64_t lda, const float* G, int64_t ldg, float* partial, int64_t M, int64_t K, int64_t N,
                       const int32_t* rows, int64_t n_rows, void* stream);
/* The entry GEMM of the factored Linear(2, 2, 'light') with its assembly as the epilogue, over row RECORDS (the tiled-GEMM
 * form of hotk_linear22_entry_fwd, for widths the LDS-resident thin GEMM does not take): rec[i] = {row, next listed row
 * (M after the last), row node, column node or -1 on the diagonal}, i < n_rows, rows ascending;
 *   off-diagonal row: Y = A . B + yn[row node][N..2N) + yn[column node][2N..3N) + b_off
 *   diagonal row    : Y = yn[node][0..N) + b_diag                   every row that is not listed: zeros
 * replaces get_output's matmul + _Bias._2 + apply_mask(0) (models/sparse/linear.py:46-48,297-306,345-346) and
 * hotk_assemble22's pass over Y.  N % 4 == 0, Y / yn / biases 16-byte aligned. */
int hotk_gemm6_entry22(const float* A, int64_t lda, const void* planes, float* Y, int64_t ldy, int64_t M, int64_t K, int64_t N,
                       const int32_t* rec, int64_t n_rows, const float* yn, int64_t ld_yn, const float* b_off,
                       const float* b_diag, void* stream);

/* LayerNorm over the rows of a batch with the padding mask folded in: y[r] = (keep >> kind[r]) & 1 ? LN(x[r]) * gamma +
 * beta : 0 (kind NULL: every row), stats[r] = {mean, rstd} for the backward.  replaces Apply(nn.LayerNorm) + batch_like's
 * masked_fill clone in EncLayer (models/encoder.py:36,40-41, sparse/f.py:20-27, utils/set.py:47-48) and, in _bwd, aten's
 * three layer-norm backward kernels plus the mask's: dx (zero on rows not kept) and per-workgroup partial sums
 * partial[n][2][D] of dgamma | dbeta, n = hotk_layernorm_partials(R), summed by the caller in order.  biased variance,
 * eps inside the root (torch.nn.LayerNorm).  D % 4 == 0, D <= 1024 (hotk_layernorm_supported), rows 16-byte aligned. */
int hotk_layernorm_supported(int64_t D);
int hotk_layernorm_partials(int64_t R, int64_t* n_partials);
int hotk_layernorm_fwd(const float* x, int64_t ld_x, const float* gamma, const float* beta, const uint8_t* kind,
                       unsigned keep, float eps, float* y, int64_t ld_y, float* stats, int64_t R, int64_t D, void* stream);
int hotk_layernorm_bwd(const float* g, int64_t ld_g, const float* x, int64_t ld_x, const float* stats, const float* gamma,
                       const uint8_t* kind, unsigned keep, float* dx, int64_t ld_dx, float* partial, int64_t R, int64_t D,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention core of KernelSelfAttn(2,2) (models/sparse/kernelattn.py:119-159): all ten kernel-attention
 * terms from the raw projections, for the shape 4 heads x 16 features (one wavefront per slot), head width 8,
 * value width 8, generalized (ReLU) feature map.  Other shapes: HOTK_EUNSUPPORTED (use the general kernels).
 *   h1  [Nn, >=12*32]  fc_1 output: raw query slots 0..3, raw key slots 4..11 (node level)
 *   h2  [R,  >= 8*32]  fc_2 output: raw query slots 0..5, raw key slots 6..7 (entry level), skip path after them
 *   v   [R,  >=10*32]  fc_v output (padding rows are ignored, they need not be zero)
 *   phik[Nn, >= 8*64]  feature map of the eight node-level key slots, computed ONCE per node by hotk_phi_fwd
 *                      (x = h1 + 4*32, HH = 32, d = 8, m = 16, ReLU): every node is the key of ~deg(i) entries
 *   att [R,  >=10*32]  the ten attention terms in the reference's concatenation order; valid rows are fully
 *                      written, padding rows are left untouched (caller zero-fills when the batch has padding)
 *   kvg [B,4,25,64]    per-graph sums (saved for the backward pass): per slot 9 feature-major channels + 16 rotated
 *                      channel-major copies (see csrc/fused22.hip)
 * backward: datt [R,10*32] -> dh1 [Nn,12*32] (key slots 4..9 are NOT written here: they are the row-segment sum of
 * ck [R,6*32], hotk_seg_sum_rows over the row CSR), dh2 [R, 8*32 (+D)] (skip-path columns = dout when given,
 * zero otherwise), dv [R,10*32]; padding rows of dh1/dh2/dv are left untouched.  dkvg [B,4,25,64] scratch.
 * workspace: hotk_attn22_workspace_bytes(B, Emax): per-chunk partial graph sums, then 1 KB of work tickets -- on batches
 * of many short segments the node kernels run as persistent waves that draw runs of consecutive nodes from counters
 * there (cleared by a memset node at the start of every call; calls sharing one workspace must not overlap).
 * ---------------------------------------------------------------------------------------------- */
#define HOTK_STORAGE_F32 0
#define HOTK_STORAGE_BF16 1
typedef struct hotk_attn22_args {
  int64_t B, Emax, Nmax, R, Nn, D;
  int32_t H, m, d_head, c;
  const uint8_t *kind, *nkind;
  const int32_t *rown, *coln, *tpos, *diag_pos, *col_ptr, *col_perm, *row_ptr, *row_perm, *n_entries, *n_nodes;
  const int32_t *col_rec; /* [Ev][4] = {entry row, row node, transposed row or -1, kind} in column-CSR order */
  const int32_t *row_rec; /* [Ev] entry row in row-CSR order, bit-complemented for diagonal entries */
  const float* h1; int64_t ld_h1;
  const float* h2; int64_t ld_h2;
  const float* v; int64_t ld_v;
  const float* proj;
  const float* phik; int64_t ld_phik; /* [Nn, >= 8*64] phi of the node-level key slots (h1 slots 4..11), hotk_phi_fwd */
  float* att; int64_t ld_att;
  float* kvg;
  const float* datt; int64_t ld_datt;
  const float* dout; int64_t ld_dout;
  float* dh1; int64_t ld_dh1;
  float* dh2; int64_t ld_dh2;
  float* dv; int64_t ld_dv;
  float* ck;
  float* dkvg;
  /* split-phase (edge-shard) mode only, see hotk_attn22_phase; ignored by hotk_attn22_forward / _backward */
  float* kvn;   /* [3 parts][Nn][2 slots][9][64] per-node key-value sums (feature-major, as the 9 first kvg channels);
                   part 0 (applied to the node's own query) stays rank-local, parts 1 and 2 are all-reduced */
  float* dkvn;  /* [2][Nn][2][9][64]: gradients of the sums of parts 1 and 2 */
  float* nd0;   /* [Nn][2 slots][64] part 0: phi(q) . sums = [8 numerators | normaliser] per head (16 lanes), all-reduced */
  float* dd0;   /* [Nn][2][64] part 0: [d numerators | d normaliser] of the node queries, all-reduced */
  float diag_w; /* weight of a diagonal entry in the per-node sums (1/world: every rank holds all diagonal entries) */
  /* storage type of the WIDE entry-level tensors h2, v, att, datt, dh2, dv, ck (their ld_* count elements of that type):
   * HOTK_STORAGE_F32, or HOTK_STORAGE_BF16 = bf16 in memory with every product and sum in fp32 (BASELINE north_star's
   * bf16 mode, held to 2e-2; the reference trains in fp32 only, --precision 32).  Node-level tensors (h1, phik, dh1, kvg,
   * dout) stay fp32.  hotk_attn22_phase takes HOTK_STORAGE_F32 only. */
  int32_t storage;
} hotk_attn22_args;
int hotk_attn22_workspace_bytes(int64_t B, int64_t Emax, size_t* bytes);
int hotk_attn22_forward(const hotk_attn22_args* args, void* workspace, size_t ws_bytes, void* stream);
int hotk_attn22_backward(const hotk_attn22_args* args, void* workspace, size_t ws_bytes, void* stream);

/* hotk_attn22_forward for callers that cannot allocate device memory.  NO COUNTERPART IN THE REFERENCE (which allocates
 * freely through PyTorch): it exists for callers under HIP graph capture, tight serving loops and harnesses that forbid
 * hipMalloc.  Same inputs, outputs and results (bit for bit) as hotk_attn22_forward, but the per-chunk partial graph
 * sums live in a file-scope device array of HOTK_ATTN22_STATIC_SCRATCH_BYTES inside the library (one copy per GPU the
 * library runs on, held until the process ends) instead of a caller workspace; the call allocates and frees nothing.
 * Limit: hotk_attn22_workspace_bytes(B, Emax) <= HOTK_ATTN22_STATIC_SCRATCH_BYTES, i.e. at most 3 640 row chunks
 * (C2-sized minibatches: 128 molecules need 0.6-4.7 MB); larger batches return HOTK_ETOOLARGE and launch nothing.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on
 * one stream are ordered and need no clearing in between (each call rewrites every partial it reads; the kernels walk
 * with a static stride, so the array holds no counters).
 * The array's device address is looked up once per GPU (hipGetSymbolAddress: lazy module and symbol resolution, a mutex)
 * and cached.  hotk_attn22_static_init() does that for the CURRENT device; call it once per GPU at set-up, outside any
 * graph capture.  Without it the first hotk_attn22_forward_static call on a GPU does the lookup itself -- for the device
 * that owns `stream` (the current device for the null stream) -- and that first call must then not be under capture. */
#define HOTK_ATTN22_STATIC_SCRATCH_BYTES (32u << 20)
int hotk_attn22_static_init(void);
int hotk_attn22_forward_static(const hotk_attn22_args* args, void* stream);

/* bf16-storage forms of two row primitives around the core (hotk_attn22_args.storage = HOTK_STORAGE_BF16; no reference
 * counterpart beyond the fp32 forms they mirror): hotk_assemble22 in place on a bf16 z (yn and biases fp32, one rounding),
 * hotk_seg_sum_rows over a bf16 source with fp32 sums.  C and the leading dimensions multiples of 4, rows 8-byte
 * aligned (others: HOTK_EUNSUPPORTED). */
int hotk_linear22_entry_fwd_bf16(const float* x2, int64_t ld_x2, const float* W, int64_t K, int64_t C, const float* yn,
                                 int64_t ld_yn, const uint8_t* kind, const int32_t* rown, const int32_t* coln,
                                 const float* b_off, const float* b_diag, void* Y, int64_t ldy, int64_t R, int64_t Nn,
                                 void* stream);   /* hotk_linear22_entry_fwd with a bf16 Y: fp32 products, one rounding */
int hotk_assemble22_bf16(void* z, int64_t ld, const float* yn, int64_t ld_yn, const uint8_t* kind, const int32_t* rown,
                         const int32_t* coln, const float* b_off, const float* b_diag, int64_t R, int64_t C,
                         void* stream);
int hotk_seg_sum_rows_bf16(const void* src, int64_t ld_src, const int32_t* seg_begin, const int32_t* seg_end,
                           const int32_t* perm, const uint8_t* kind, unsigned keep, float* out, int64_t ld_out,
                           int64_t n_seg, int64_t C, int64_t max_seg_len, void* stream);

/* The same core in phases, for ONE graph whose entries are sharded over ranks (SURVEY 8e edge-shard; the reference has
 * no multi-device form of kernelattncoef.py:103-175, the sums are what its coalesce() builds on one device).  Sums over
 * ENTRIES are partial on a rank: the caller all-reduces them between the phases (RCCL SUM):
 *   FWD_GRAPHSUM : kvg  <- per-graph sums of the local rows            -> all-reduce kvg slots 1 and 3 (entry keys)
 *   FWD_NODESUM  : kvn  <- per-node column sums of the local entries, diagonal entries weighted by diag_w;
 *                  nd0  <- part 0's partial sums applied to the node queries (linear in the sums, and every rank
 *                          holds every node query: 128 floats per node cross the ranks instead of 1 152)
 *                                                                       -> all-reduce kvn parts 1, 2 and nd0
 *   FWD_APPLY    : att  <- all ten terms from kvg / kvn / nd0
 *   BWD_QUERY    : dh1 (query slots 2, 3), dh2 (query slots), dkvg, dkvn, dd0 from datt, kvg, kvn, nd0
 *                                                                       -> all-reduce dkvg slots 1 and 3, dkvn, dd0
 *   BWD_KEYVAL   : dv, ck, dh2 (key slots, skip path), dh1 (query slots 0, 1 from this rank's partial part-0 sums;
 *                  key slots 10, 11) from dkvg / dkvn / dd0
 * Buffers and shapes as for hotk_attn22_forward / _backward; kvn / dkvn rows of padding nodes are not touched.
 * FWD_APPLY and BWD_KEYVAL in two halves each, for callers that overlap the large per-node all-reduce with the half that
 * does not need it: FWD_APPLY_GRAPH (terms 7-10, from kvg) | FWD_APPLY_NODE (terms 1-6, from kvn / nd0);
 * BWD_KEYVAL_GRAPH (the per-graph key / value side, from dkvg) | BWD_KEYVAL_NODE (the per-node one, from dkvn / dd0).
 * The halves write disjoint outputs or accumulate in a fixed order: same results as the whole phase. */
enum { HOTK_ATTN22_FWD_GRAPHSUM = 0, HOTK_ATTN22_FWD_NODESUM = 1, HOTK_ATTN22_FWD_APPLY = 2,
       HOTK_ATTN22_BWD_QUERY = 3, HOTK_ATTN22_BWD_KEYVAL = 4,
       HOTK_ATTN22_FWD_APPLY_GRAPH = 5, HOTK_ATTN22_FWD_APPLY_NODE = 6,
       HOTK_ATTN22_BWD_KEYVAL_GRAPH = 7, HOTK_ATTN22_BWD_KEYVAL_NODE = 8 };
int hotk_attn22_phase(const hotk_attn22_args* args, int phase, void* workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOTK_H */
