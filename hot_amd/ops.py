"""torch.autograd.Function wrappers around the C-ABI kernels of libhotk.so.

Tensors are torch tensors (device memory, streams: plumbing); all arithmetic on the path happens in the HIP
kernels.  Each wrapper states which reference code it stands in for.
"""
from ctypes import byref, c_size_t, c_float, c_int64 as ctypes_long

import os

import torch
from torch.autograd import Function

from . import _lib
from ._lib import lib, p, check, call, stream_ptr, rows2d, KEEP_DIAG, KEEP_OFFDIAG, KEEP_VALID, SEG_CHUNK
from .plan import Seg


# ------------------------------------------------------------------------------------------------
# raw launchers (no autograd)
# ------------------------------------------------------------------------------------------------
def gather_rows(src, idx, n_out, out=None, accumulate=False):
    s, ld, n, C = rows2d(src)
    if out is None:
        out = torch.empty(n_out, C, dtype=src.dtype, device=src.device)
    o, ldo, no, Co = rows2d(out)
    assert o.data_ptr() == out.data_ptr() and no == n_out and Co == C
    call('hotk_gather_rows', 8 * n_out * C + 4 * n_out, p(s), ld, p(idx), p(o), ldo, n_out, C, int(accumulate))
    return out


def mask_rows(src, kind, keep, fill=0.0):
    s, ld, n, C = rows2d(src)
    assert kind.numel() == n, (kind.numel(), n)
    out = torch.empty(src.shape, dtype=src.dtype, device=src.device)
    call('hotk_mask_rows', 8 * n * C + n, p(s), ld, p(kind), keep, c_float(fill), p(out), C, n, C)
    return out


SEG_SPLIT = 256        # rows one thread adds at most (<= the ABI's HOTK_SEG_CHUNK)
POOL_SPLIT = 2048      # rows one wavefront of the softmax-pooling kernel walks at most


class _Refined:
    """pieces of a segment list: `fine` (the pieces, a segment list over the same positions), `group` (per original
    segment the range of its pieces), `owner` (piece -> original segment); rows()/piece_of_row() on demand"""

    def __init__(self, seg, chunk):
        beg, end, perm, maxlen = seg
        n_seg = beg.numel()
        length = (end.long() - beg.long()).clamp_min(0)
        pieces = (length + chunk - 1) // chunk
        gend = torch.cumsum(pieces, 0)
        gbeg = gend - pieces
        # the two host reads of this bookkeeping (piece count, most pieces of one segment): once per plan and segment list
        total, most = (int(v) for v in torch.stack([gend[-1], pieces.max()]).tolist()) if n_seg else (0, 0)
        owner = torch.repeat_interleave(torch.arange(n_seg, device=beg.device), pieces, output_size=total)
        k = torch.arange(total, device=beg.device) - gbeg[owner]
        vbeg = beg.long()[owner] + k * chunk
        vend = torch.minimum(vbeg + chunk, end.long()[owner])
        i32 = lambda t: t.to(torch.int32).contiguous()
        self.fine = (i32(vbeg), i32(vend), perm, min(int(maxlen), int(chunk)))
        self.group = (i32(gbeg), i32(gend), None, most)
        self.owner = i32(owner)
        self._rows = None

    def rows(self):
        """(row ids covered by the pieces, piece after piece; the piece of each of them)"""
        if self._rows is None:
            fb, fe = self.fine[0], self.fine[1]
            length = (fe - fb).long()
            start = torch.cumsum(length, 0) - length
            total = int(length.sum())
            piece = torch.repeat_interleave(torch.arange(fb.numel(), device=fb.device), length, output_size=total)
            self._rows = (fb.long()[piece] + (torch.arange(total, device=fb.device) - start[piece]), piece)
        return self._rows


def refined(seg, chunk=SEG_CHUNK):
    """Segments longer than `chunk` cut into consecutive pieces of at most `chunk` positions (-> _Refined).  Summing
    `fine` and then `group` adds every segment in a fixed order without atomics; a segment with more than SEG_CHUNK pieces
    is refined again by the caller.  Index bookkeeping (torch ops + one host read), built once per plan for the plan's own
    segment lists (plan.Seg) and kept on the plan; any other segment list is refined afresh on every call."""
    plan = getattr(seg, 'plan', None)
    if plan is None:
        return _Refined(seg, chunk)
    cache = plan.__dict__.setdefault('_refined', {})
    key = (seg.name, int(chunk))
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = _Refined(seg, chunk)
        # the pieces are segment lists of the same plan: a second refinement (one graph of millions of rows) is kept too
        hit.fine = Seg(*hit.fine, plan, f'{seg.name}/fine{chunk}')
        hit.group = Seg(*hit.group, plan, f'{seg.name}/group{chunk}')
    return hit


def refine_seg(seg, chunk=SEG_CHUNK):
    r = refined(seg, chunk)
    return r.fine, r.group


def seg_sum_rows(src, seg, kind=None, keep=KEEP_VALID, out=None):
    """out[s] = sum of the kept rows of segment s, added in segment order (bitwise reproducible).  out: optional
    destination (a column slice of a wider row-major tensor is fine)."""
    beg, end, perm, maxlen = seg
    n_seg = beg.numel()
    if maxlen > SEG_SPLIT:
        # one thread walks a segment: hub nodes and whole-graph segments are cut into pieces of SEG_SPLIT rows (more
        # threads, same fixed order: piece sums, then the pieces of a segment in piece order)
        fine, group = refine_seg(seg, SEG_SPLIT)
        part = seg_sum_rows(src, fine, kind, keep)
        return seg_sum_rows(part, group, None, KEEP_VALID, out)
    s, ld, n, C = rows2d(src)
    if out is None:
        out = torch.empty(n_seg, C, dtype=src.dtype, device=src.device)
    o, ldo, no, Co = rows2d(out)
    assert o.data_ptr() == out.data_ptr() and no == n_seg and Co == C, 'seg_sum_rows: destination is not row-major'
    call('hotk_seg_sum_rows', 4 * C * (n + n_seg), p(s), ld, p(beg), p(end), p(perm), p(kind), keep, p(o), ldo, n_seg, C, maxlen)
    return out


_SEG_PAIR = os.environ.get('HOTK_SEG_PAIR', '1') == '1'


def seg_sum_rows_pair(src, seg_a, seg_b, kind, keep, out_a, out_b, own=None):
    """out_a / out_b = seg_sum_rows(src, seg_a / seg_b, kind, keep), the source read from HBM once (hotk_seg_sum_rows_pair);
    lists with long segments, or without a row map, take the two separate calls.
    own = (rows, out_own): out_own[s] = src[rows[s]] as well (zeros where rows[s] < 0), in the same launch"""
    (ab, ae, ap, am), (bb, be, bp, bm) = seg_a, seg_b
    s, ld, n, C = rows2d(src)
    oa, lda, na, Ca = rows2d(out_a)
    ob, ldb, nb, Cb = rows2d(out_b)
    ok = (_SEG_PAIR and max(am, bm) <= SEG_SPLIT and ap is not None and bp is not None and kind is not None
          and ab.numel() == bb.numel() == na == nb and lda == ldb and Ca == Cb == C
          and oa.data_ptr() == out_a.data_ptr() and ob.data_ptr() == out_b.data_ptr())
    own_ok = False
    if own is not None:
        oo, ldo, no, Co = rows2d(own[1])
        own_ok = ok and ldo == lda and no == na and Co == C and oo.data_ptr() == own[1].data_ptr() and own[0].numel() == na
        if not own_ok:
            assert oo.data_ptr() == own[1].data_ptr()
            call('hotk_gather_rows', 8 * no * Co, p(s), ld, p(own[0]), p(oo), ldo, no, Co, 0)
    if not ok:
        seg_sum_rows(src, seg_a, kind, keep, out=out_a)
        seg_sum_rows(src, seg_b, kind, keep, out=out_b)
        return
    if own_ok:
        call('hotk_seg_sum_rows_pair_own', 4 * C * (n + 3 * na), p(s), ld, p(ab), p(ae), p(ap), p(bb), p(be), p(bp), p(kind),
             keep, p(oa), p(ob), p(own[0]), p(oo), lda, na, C, max(am, bm))
        return
    call('hotk_seg_sum_rows_pair', 4 * C * (n + 2 * na), p(s), ld, p(ab), p(ae), p(ap), p(bb), p(be), p(bp), p(kind), keep,
         p(oa), p(ob), lda, na, C, max(am, bm))


def bias_grads(g, kind, want_off, want_diag):
    """(db_off, db_diag) = column sums of g over the off-diagonal / diagonal rows (hotk_bias_grad_rows: two passes,
    bitwise reproducible)"""
    gs, ld, n, C = rows2d(g)
    db_off = torch.empty(C, dtype=g.dtype, device=g.device) if want_off else None
    db_diag = torch.empty(C, dtype=g.dtype, device=g.device) if want_diag else None
    if db_off is None and db_diag is None:
        return None, None
    need = c_size_t(0)
    check(lib().hotk_bias_grad_workspace_bytes(n, C, byref(need)), 'hotk_bias_grad_workspace_bytes')
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=g.device)
    call('hotk_bias_grad_rows', 4 * n * C, p(gs), ld, p(kind), p(db_off), p(db_diag), n, C, p(ws), need.value)
    return db_off, db_diag


def seg_outer_sum(K, kidx, V, vidx, aug, seg, HH, m, c):
    beg, end, perm, maxlen = seg
    if maxlen > SEG_SPLIT:      # per-piece partial sums, then added per segment in piece order: no atomics
        fine, group = refine_seg(seg, SEG_SPLIT)
        part = seg_outer_sum(K, kidx, V, vidx, aug, fine, HH, m, c)
        return seg_sum_rows(part.view(part.size(0), -1), group).view(beg.numel(), HH * m, c + 1)
    k, ldk, _, Ck = rows2d(K)
    v, ldv, _, Cv = rows2d(V)
    assert Ck == HH * m and Cv == HH * c, (Ck, Cv, HH, m, c)
    if aug is not None:
        a, lda, _, Ca = rows2d(aug)
        assert Ca == HH
    else:
        a, lda = None, 0
    n_seg = beg.numel()
    KV = torch.empty(n_seg, HH * m, c + 1, dtype=K.dtype, device=K.device)
    call('hotk_seg_outer_sum', 4 * (v.numel() // max(Cv, 1)) * (Ck + Cv) + 4 * KV.numel(), p(k), ldk, p(kidx), p(v), ldv, p(vidx), p(a), lda, p(beg), p(end), p(perm), p(KV),
                                   n_seg, HH, m, c, maxlen)
    return KV


def _visit(order, n):
    """(order, rows visited, allocator): an order that lists fewer than n rows leaves the others untouched -> zeros"""
    if order is None:
        return None, n, torch.empty
    return order, order.numel(), (torch.empty if order.numel() == n else torch.zeros)


def row_contract(Q, qidx, KV, sidx, seg_div, kind, keep, n, HH, m, c, normalize, oidx=None, n_out=None, order=None):
    q, ldq, _, Cq = rows2d(Q)
    assert Cq == HH * m
    n_out = n if n_out is None else n_out
    order, nv, alloc = _visit(order, n)
    if oidx is not None:
        alloc = torch.zeros
    out = alloc(n_out, HH * c, dtype=Q.dtype, device=Q.device)
    call('hotk_row_contract', 4 * nv * HH * (m + c) + 4 * KV.numel(), p(q), ldq, p(qidx), p(KV), p(sidx), seg_div, p(kind), keep, p(oidx), p(out), HH * c, nv,
                                  HH, m, c, int(normalize), p(order))
    return out


def apply_bwd_stats(Q, KV, sidx, seg_div, kind, keep, dO, n, HH, m, c, order=None):
    q, ldq, _, _ = rows2d(Q)
    g, ldg, ng, Cg = rows2d(dO)
    assert ng == n and Cg == HH * c
    order, nv, alloc = _visit(order, n)
    X = alloc(n, HH * c, dtype=Q.dtype, device=Q.device)
    XA = alloc(n, HH, dtype=Q.dtype, device=Q.device)
    call('hotk_apply_bwd_stats', 4 * nv * HH * (m + 2 * c + 1) + 4 * KV.numel(), p(q), ldq, p(KV), p(sidx), seg_div, p(kind), keep, p(g), ldg, p(X), HH * c, p(XA), HH,
                                     nv, HH, m, c, p(order))
    return X, XA


def row_expand(KV, sidx, seg_div, X, xidx, XA, kind, keep, n, HH, m, c, order=None):
    x, ldx, _, Cx = rows2d(X)
    assert Cx == HH * c
    if XA is not None:
        a, lda, _, _ = rows2d(XA)
    else:
        a, lda = None, 0
    order, nv, alloc = _visit(order, n)
    out = alloc(n, HH * m, dtype=X.dtype, device=X.device)
    call('hotk_row_expand', 4 * nv * HH * (m + c) + 4 * KV.numel(), p(KV), p(sidx), seg_div, p(x), ldx, p(xidx), p(a), lda, p(kind), keep, p(out), HH * m, nv,
                                HH, m, c, p(order))
    return out


THIN_MIN_ROWS = 1 << 15


def thin_ok(R, K, C):
    """the hand-written LDS-resident-weight MFMA GEMM (csrc/gemm.hip) covers this product.  It is used where an
    epilogue is fused (entry GEMM + assembly of Linear(2,2,'light'): 6.2 ms vs 5.4 + 5.8 ms; fc_o + bias + skip path)
    and for the narrow-to-wide plain products (thin_wide_ok); elsewhere it runs on par with the library GEMM, which
    stays."""
    return R >= THIN_MIN_ROWS and lib().hotk_thin_gemm_supported(int(K), int(C)) == 1


def thin_wide_ok(R, K, C):
    """narrow-to-wide product [R, 32] x [32, 320] (fc_v, d att = d y . W_o): the result store dominates and the
    row-major staged epilogue of csrc/gemm.hip writes full 128-byte row pieces: 4.15 ms vs 4.64 ms (library) at 10 M rows"""
    return K == 32 and C >= 8 * K and thin_ok(R, K, C)


def linear_bwd_fused(g, x, w):
    """(dx [R,K], dW [K,C], db [C]) of y = x @ w + b from g = dy [R,C] in one pass (hotk_linear_bwd_fused); the per-wave
    partial blocks are summed in a fixed order (deterministic).  w = None: (None, x^T @ g, column sums of g)."""
    R, C = g.shape
    K = x.size(1)
    n = ctypes_long()
    check(lib().hotk_linear_bwd_partials(R, byref(n)), 'hotk_linear_bwd_partials')
    part = torch.empty(n.value, K + 2, C, dtype=torch.float32, device=g.device)
    dx = torch.empty(R, K, dtype=torch.float32, device=g.device) if w is not None else None
    wc = None if w is None else w.float().contiguous()       # (bound to a name: alive until the launch is queued)
    if g.dtype == torch.bfloat16:
        # bf16 storage mode: g and x as stored (hotk_linear_bwd_fused_bf16), fp32 weight and results
        assert x.dtype == torch.bfloat16 and g.is_contiguous() and x.is_contiguous()
        call('hotk_linear_bwd_fused_bf16', 2 * R * (C + K) + 4 * R * K, p(g), C, p(x), K, p(wc), K, C, p(dx), K, p(part), R)
    else:
        call('hotk_linear_bwd_fused', 4 * R * (C + 2 * K), p(g), C, p(x), K, p(wc), K, C, p(dx), K, p(part), R)
    tot = part.sum(0)
    return dx, tot[:K], tot[K] + tot[K + 1]


_DWO_FUSED = os.environ.get('HOTK_DWO_FUSED', '1') == '1'


def linear_bwd_ok(R, K, C):
    return R >= THIN_MIN_ROWS and lib().hotk_linear_bwd_supported(int(K), int(C)) == 1


def n_rows(t):
    return t.numel() // t.size(-1)


def thin_gemm(A, W, bias=None, res=None):
    """Y[R, C] = A[R, K] @ W[K, C] (+ bias) (+ res) in one pass (hotk_thin_gemm_bias); call thin_ok first"""
    a, lda, R, K = rows2d(A)
    W = W.contiguous()
    C = W.size(1)
    assert W.size(0) == K
    Y = torch.empty(R, C, dtype=A.dtype, device=A.device)
    r_, ldr = None, 0
    if res is not None:
        r_, ldr, Rr, Cr = rows2d(res)
        assert Rr == R and Cr == C
    call('hotk_thin_gemm_bias', 4 * R * (K + C), p(a), lda, p(W), K, C, p(bias), p(r_), ldr, p(Y), C, R)
    return Y


_THIN_NODE = os.environ.get('HOTK_THIN_NODE', '1') == '1'      # (A/B switch)


def _thin_blocks(K, C):
    """column blocks of a [., K] x [K, C] product that hotk_thin_gemm_bias takes one by one (the weight block and the result
    images share LDS: 32 x 864 does not fit, 480 + 384 do), widest first; None when C does not split"""
    if lib().hotk_thin_gemm_supported(int(K), int(C)) == 1:
        return [C]
    for b in (480, 384, 320, 288, 192, 160, 96):
        if b < C and lib().hotk_thin_gemm_supported(int(K), b) == 1:
            rest = _thin_blocks(K, C - b)
            if rest is not None:
                return [b] + rest
    return None


def thin_blocks_ok(A, K, C):
    """narrow-to-wide product of a tall fp32 operand with K = 32 (the node-level products of the light basis and fc_1 at the
    headline widths: [0.98 M, 32] x [32, 864 | 384]): store-bound, the thin GEMM's row-major staged epilogue against
    the library's tiles (0.79 ms per call for 3.4 / 1.5 GB of results)"""
    return (_THIN_NODE and A.is_cuda and A.dtype == torch.float32 and K == 32 and C >= 8 * K and n_rows(A) >= THIN_MIN_ROWS
            and _thin_blocks(K, C) is not None)


def thin_gemm_blocks(A, W, bias=None):
    """Y[R, C] = A[R, K] @ W[K, C] (+ bias), one hotk_thin_gemm_bias launch per column block of _thin_blocks"""
    a, lda, R, K = rows2d(A)
    C = W.size(1)
    Y = torch.empty(R, C, dtype=A.dtype, device=A.device)
    c0 = 0
    for b in _thin_blocks(K, C):
        wb = W[:, c0:c0 + b].contiguous()
        bb = None if bias is None else bias[c0:c0 + b]
        yb = Y[:, c0:c0 + b]
        call('hotk_thin_gemm_bias', 4 * R * (K + b), p(a), lda, p(wb), K, b, p(bb), p(None), 0, p(yb), C, R)
        c0 += b
    return Y


def thin_in_ok(R, K, C):
    """hotk_thin_gemm_bias_bf16in covers this product (bf16 rows in, fp32 out: the wide-input stream forms)"""
    return R >= THIN_MIN_ROWS and ((K == 288 and C % 64 == 0) or (K == 320 and C % 32 == 0)) and thin_ok(R, K, C)


def thin_gemm_bf16in(A, W, bias=None, res=None):
    """Y[R, C] (fp32) = A[R, K] (bf16) @ W[K, C] (fp32) (+ bias) (+ res, fp32 rows), exact products, fp32 accumulation; call
    thin_in_ok first"""
    assert A.dtype == torch.bfloat16 and A.dim() == 2 and A.is_contiguous()
    R, K = A.shape
    W = W.float().contiguous()
    C = W.size(1)
    assert W.size(0) == K
    Y = torch.empty(R, C, dtype=torch.float32, device=A.device)
    r_, ldr = None, 0
    if res is not None:
        r_, ldr, Rr, Cr = rows2d(res)
        assert Rr == R and Cr == C and res.dtype == torch.float32
    call('hotk_thin_gemm_bias_bf16in', R * (2 * K + 4 * C), p(A), K, p(W), K, C, p(bias), p(r_), ldr, p(Y), C, R)
    return Y


def thin_gemm_bf16(A, W, bias=None):
    """Y[R, C] (bf16) = A[R, 32] @ W[32, C] (+ bias), fp32 operands and accumulation, rounded at the store
    (hotk_thin_gemm_bias_bf16); call thin_wide_ok first"""
    a, lda, R, K = rows2d(A)
    W = W.contiguous()
    C = W.size(1)
    assert W.size(0) == K and A.dtype == torch.float32
    Y = torch.empty(R, C, dtype=torch.bfloat16, device=A.device)
    call('hotk_thin_gemm_bias_bf16', R * (4 * K + 2 * C), p(a), lda, p(W), K, C, p(bias), p(Y), C, R)
    return Y


# ------------------------------------------------------------------------------------------------
# wide dense projections: fp32-exact tiled GEMMs on the bf16 matrix pipe (csrc/gemm6.hip)
# ------------------------------------------------------------------------------------------------
_G6 = os.environ.get('HOTK_GEMM6', '1') == '1'
G6_MIN_ROWS = 16384       # below: launch-bound either way, the library call is one launch instead of two


def _row_major(t):
    """(tensor, leading dimension) of a 2-D operand readable as rows (a column slice of a wider row-major tensor is fine)"""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.size(1) or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t, t.stride(0) if t.size(0) > 1 else t.size(1)


def mm6_ok(A, K, N):
    return (_G6 and A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.size(0) >= G6_MIN_ROWS and K % 32 == 0
            and K >= 128 and N >= 96)


_G6_ROWS = os.environ.get('HOTK_GEMM_ROWS', '1') == '1'


def valid_rows(plan, level):
    """ascending int32 list of the plan's valid entry rows (level 'e') or node rows ('n') for the GEMMs that skip padding
    rows (hotk_gemm6_nn_rows / _tn_rows); None when no row is padding.  Built once per plan."""
    if not _G6_ROWS:
        return None
    key = '_valid_' + level
    d = plan.__dict__
    if key not in d:
        k = plan.kind if level == 'e' else plan.nkind
        idx = torch.nonzero(k).view(-1).to(torch.int32)
        d[key] = None if idx.numel() == k.numel() else idx
    return d[key]


def _ones4(plan, like):
    """the constant block [1 0 0 0] per node row, kept on the plan (two fills per use otherwise)"""
    key = '_ones4_' + str(like.dtype)
    d = plan.__dict__
    if key not in d:
        one = torch.zeros(like.size(0), 4, dtype=like.dtype, device=like.device)
        one[:, 0] = 1.
        d[key] = one
    return d[key]


def entry_records(plan):
    """int32 [n, 4] per valid entry row, ascending: {row, next valid row (R after the last), row node, column node or -1 on
    the diagonal} -- what hotk_gemm6_entry22's epilogue reads per stored row.  Built once per plan."""
    d = plan.__dict__
    if '_entry_rec' not in d:
        rows = valid_rows(plan, 'e')
        if rows is None:
            rows = torch.arange(plan.R, dtype=torch.int32, device=plan.kind.device)
        r = rows.long()
        nxt = torch.cat([rows[1:], torch.tensor([plan.R], dtype=torch.int32, device=rows.device)])
        cn = torch.where(plan.kind[r] == 1, torch.full_like(rows, -1), plan.coln[r])
        d['_entry_rec'] = torch.stack([rows, nxt, plan.rown[r], cn], dim=1).contiguous()
    return d['_entry_rec']


def entry22_ok(x2, K, N, b_off, b_diag, plan):
    return (_G6_ROWS and _G6_ENTRY and mm6_ok(x2, K, N) and N % 4 == 0 and plan.R < (1 << 31) and plan.Ev > 0
            and all(b is None or b.data_ptr() % 16 == 0 for b in (b_off, b_diag)))


_G6_ENTRY = os.environ.get('HOTK_GEMM6_ENTRY', '1') == '1'
_L22_SKIP_MASK = os.environ.get('HOTK_L22_SKIP_MASK', '1') == '1'      # (A/B switch: see _l22_backward)


def mm(A, W, bias=None, trans_w=False, rows=None):
    """A[M, K] @ W[K, N] (+ bias), or A[M, K] @ W[N, K]^T with trans_w: the split-bf16 tiled GEMM for tall, wide products
    (fp32-exact: six bf16 MFMAs per product on three-way exact splits, weight split once per call), the library otherwise.
    rows (valid_rows of the plan): the tiled GEMM computes the listed rows only and writes zeros to the others; the library
    path computes every row (callers mask afterwards as if rows were None: see mm_masks)."""
    K = A.size(1)
    N = W.size(0) if trans_w else W.size(1)
    assert (W.size(1) if trans_w else W.size(0)) == K
    if not mm6_ok(A, K, N) or W.dtype != torch.float32:
        Wm = W.t() if trans_w else W
        return torch.addmm(bias, A, Wm) if bias is not None else A @ Wm
    if rows is not None and rows.numel() == 0:      # (an empty list: a NULL pointer would mean "all rows" to the library)
        return torch.zeros(A.size(0), N, dtype=torch.float32, device=A.device)
    a, lda = _row_major(A)
    M = a.size(0)
    if W.stride(1) != 1 and W.stride(0) == 1:      # a transposed VIEW of a row-major weight: read the storage as it lies
        W, trans_w = W.t(), not trans_w
    w, ldw = _row_major(W)
    planes = torch.empty(3, N, K, dtype=torch.bfloat16, device=A.device)
    call('hotk_gemm6_split_weight', 10 * K * N, p(w), ldw, K, N, 0 if trans_w else 1, p(planes))
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    if rows is None or not _rows_ok(N, bias):
        call('hotk_gemm6_nn', 4 * M * (K + N), p(a), lda, p(planes), p(bias), p(C), N, M, K, N)
    else:
        call('hotk_gemm6_nn_rows', 4 * rows.numel() * (K + N), p(a), lda, p(planes), p(bias), p(C), N, M, K, N, p(rows),
             rows.numel())
    return C


def _rows_ok(N, bias):
    """hotk_gemm6_nn_rows stores 16-byte pieces only"""
    return N % 4 == 0 and (bias is None or bias.data_ptr() % 16 == 0)


def mm_masks(A, W, rows, trans_w=False, bias=None):
    """whether mm(A, W, bias, trans_w, rows) leaves the padding rows of its result zero by itself"""
    N = W.size(0) if trans_w else W.size(1)
    return rows is not None and W.dtype == torch.float32 and mm6_ok(A, A.size(1), N) and _rows_ok(N, bias)


def wgrad6_ok(A, B, rows=None):
    """tall operands whose product fills the tiles (256 x 256, and 256 x 128 for the tail of N = 320 / 384; measured at
    254 k rows: [., 512]^T [., 256] 181 against 134 TFLOP/s for the library's split-K batch, [., 256]^T [., 864] 166 / 121;
    below 256 columns on either side the tile is mostly padding: [., 128]^T [., 576] 75 / 118).  The tiles' padded area
    may exceed K N by a quarter."""
    K, N = A.size(1), B.size(1)
    if not (_G6 and A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32 and A.size(0) >= G6_MIN_ROWS
            and K >= 256 and N >= 256 and K % 4 == 0 and N % 4 == 0):
        return False
    tail = N % 256
    n_pad = (N - tail) + (0 if tail == 0 else 128 if (tail <= 128 and N < 512) else 256)      # (as g6_tn_tail, csrc/gemm6.hip)
    return -(-K // 256) * 256 * n_pad <= 1.25 * K * N


def wgrad6(A, B, rows=None):
    """A^T @ B over the rows (hotk_gemm6_tn): per-slab partial products summed in slab order (reproducible); rows: over the
    listed rows only (hotk_gemm6_tn_rows)"""
    if rows is not None and rows.numel() == 0:
        return torch.zeros(A.size(1), B.size(1), dtype=torch.float32, device=A.device)
    a, lda = _row_major(A)
    b, ldb = _row_major(B)
    M, K, N = a.size(0), a.size(1), b.size(1)
    Ms = M if rows is None else rows.numel()
    n = ctypes_long()
    check(lib().hotk_gemm6_tn_partials(Ms, K, N, byref(n)), 'hotk_gemm6_tn_partials')
    part = torch.empty(n.value, K, N, dtype=torch.float32, device=A.device)
    if rows is None:
        call('hotk_gemm6_tn', 4 * M * (K + N), p(a), lda, p(b), ldb, p(part), M, K, N)
    else:
        call('hotk_gemm6_tn_rows', 4 * Ms * (K + N), p(a), lda, p(b), ldb, p(part), M, K, N, p(rows), Ms)
    return part[0] if n.value == 1 else part.sum(0)


# ------------------------------------------------------------------------------------------------
# autograd functions
# ------------------------------------------------------------------------------------------------
class LayerNormRows(Function):
    """y = LayerNorm(x) over the last dimension with the padding rows zeroed in the same pass (kind None: every row is
    normalised).  Stands in for Apply(nn.LayerNorm) + batch_like's masking clone (models/encoder.py:36,40-41,
    sparse/f.py:20-27): one kernel forward, one backward (csrc/norm.hip) instead of aten's layer norm + a mask copy and
    three backward kernels + the mask's."""

    @staticmethod
    def forward(ctx, x, gamma, beta, kind, eps):
        xs, ld, n, D = rows2d(x)
        y = torch.empty(n, D, dtype=x.dtype, device=x.device)
        stats = torch.empty(n, 2, dtype=torch.float32, device=x.device)
        call('hotk_layernorm_fwd', 8 * n * D, p(xs), ld, p(gamma), p(beta), p(kind), KEEP_VALID, c_float(eps), p(y), D, p(stats),
             n, D)
        ctx.save_for_backward(xs, stats, gamma, kind)
        ctx.cfg = (ld, n, D, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        xs, stats, gamma, kind = ctx.saved_tensors
        ld, n, D, shape = ctx.cfg
        gs, ldg, ng, Dg = rows2d(g.contiguous())
        dx = torch.empty(n, D, dtype=g.dtype, device=g.device)
        npart = ctypes_long()
        check(lib().hotk_layernorm_partials(n, byref(npart)), 'hotk_layernorm_partials')
        part = torch.empty(npart.value, 2, D, dtype=torch.float32, device=g.device)
        call('hotk_layernorm_bwd', 12 * n * D, p(gs), ldg, p(xs), ld, p(stats), p(gamma), p(kind), KEEP_VALID, p(dx), D, p(part),
             n, D)
        tot = part.sum(0)
        return dx.view(shape), tot[0], tot[1], None, None


def layernorm_ok(x, ln):
    return (x.is_cuda and x.dtype == torch.float32 and ln.elementwise_affine and ln.bias is not None
            and len(ln.normalized_shape) == 1 and lib().hotk_layernorm_supported(int(x.size(-1))) == 1
            and x.size(-1) == ln.normalized_shape[0])


class GatherRows(Function):
    """out[r] = src[idx[r]] (zeros where idx < 0); idx_bwd is the inverse map (the maps used on this path
    are injective: transpose positions, diagonal positions).
    Stands in for do_transpose / get_diag / to_diag (hot_pytorch/utils/sparse.py:34-87,130-134)."""

    @staticmethod
    def forward(ctx, src, idx, idx_bwd, n_out):
        ctx.save_for_backward(idx_bwd)
        ctx.n_src = src.numel() // src.size(-1)
        ctx.src_shape = src.shape
        return gather_rows(src, idx, n_out)

    @staticmethod
    def backward(ctx, g):
        (idx_bwd,) = ctx.saved_tensors
        d = gather_rows(g.contiguous(), idx_bwd, ctx.n_src)
        return d.view(ctx.src_shape), None, None, None


class MaskRows(Function):
    """Zero the rows whose kind is not kept.  Stands in for masked_fill / apply_mask / get_nondiag
    (utils/set.py:47-48, batch/sparse.py:77-80, utils/sparse.py:90-99)."""

    @staticmethod
    def forward(ctx, src, kind, keep, fill):
        ctx.save_for_backward(kind)
        ctx.keep = keep
        return mask_rows(src, kind, keep, fill)

    @staticmethod
    def backward(ctx, g):
        (kind,) = ctx.saved_tensors
        return mask_rows(g, kind, ctx.keep, 0.0), None, None, None


class SegSumRows(Function):
    """out[s] = sum of the rows of segment s (row kinds filtered); backward broadcasts the segment gradient
    back to its rows through `row_seg` (segment id per row, -1 = none)."""

    @staticmethod
    def forward(ctx, src, seg, kind, keep, row_seg):
        ctx.save_for_backward(kind, row_seg)
        ctx.keep = keep
        ctx.src_shape = src.shape
        return seg_sum_rows(src, seg, kind, keep)

    @staticmethod
    def backward(ctx, g):
        kind, row_seg = ctx.saved_tensors
        n = row_seg.numel()
        d = gather_rows(g.contiguous(), row_seg, n)
        if kind is not None:
            d = mask_rows(d, kind, ctx.keep, 0.0)
        return d.view(ctx.src_shape), None, None, None, None


class GatherSeg(Function):
    """out[r] = src[idx[r]] (zeros where idx < 0) for a many-to-one index: node or graph rows broadcast to the
    entries / nodes that reference them.  Backward is the matching segment sum (`seg` groups the output rows by
    source row).  Stands in for torch.gather(diagonal, 1, row_idx / col_idx) and the broadcast halves of the
    masked bmm poolings (models/sparse/linear.py:157-159,187-231)."""

    @staticmethod
    def forward(ctx, src, idx, seg, n_out):
        ctx.seg = seg
        ctx.src_shape = src.shape
        return gather_rows(src, idx, n_out)

    @staticmethod
    def backward(ctx, g):
        d = seg_sum_rows(g.contiguous(), ctx.seg)
        return d.view(ctx.src_shape), None, None, None


class GatherSum(Function):
    """y[b] = sum_j P[idx_j[b], j*C:(j+1)*C] for the k index columns of a UniformPlan: the 1 -> k broadcast of the
    uniform extension after the per-node products (models/uniform/linear.py:33-40, factored).  Backward of each
    gather = segment sum over the hyperedges that hold the node in slot j."""

    @staticmethod
    def forward(ctx, P, plan, C):
        assert P.size(-1) == plan.k * C
        ctx.plan, ctx.C, ctx.N = plan, C, P.size(0)
        y = None
        for j in range(plan.k):
            y = gather_rows(P[:, j * C:(j + 1) * C], plan.idx32[j], plan.B_rows, out=y, accumulate=j > 0)
        return y

    @staticmethod
    def backward(ctx, g):
        plan, C = ctx.plan, ctx.C
        g = g.contiguous()
        dP = torch.cat([seg_sum_rows(g, plan.seg[j]) for j in range(plan.k)], dim=1)
        return dP, None, None


class Phi(Function):
    """Fused FAVOR+ feature map: head split -> projection -> ReLU/exp -> loop + padding masks.
    Stands in for KernelFeatureMapWrapper.forward (models/sparse/kernelattncoef.py:19-28) and
    KernelFeatureMap.{generalized_map,softmax_map} (models/common/kernel.py:81-138)."""

    @staticmethod
    def forward(ctx, x, proj, kind, keep, d, mode, shard=None):
        xs, ld, n, C = rows2d(x)
        m = proj.size(0)
        assert proj.size(1) == d and C % d == 0
        HH = C // d
        proj = proj.contiguous().float()
        gstat = None
        if mode == _lib.PHI_SOFTMAX_K:
            gstat = torch.zeros(4, dtype=torch.float32, device=x.device)
            need = c_size_t(0)
            check(lib().hotk_phi_stats_workspace_bytes(n, HH, m, byref(need)), 'hotk_phi_stats_workspace_bytes')
            ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
            call('hotk_phi_key_stats', 4 * n * HH * d, p(xs), ld, p(proj), n, HH, d, m, p(gstat), p(ws), need.value)
            if shard is not None and shard.world > 1:
                # the key map subtracts ONE maximum over every row of the batch (common/kernel.py:112): across the
                # shards of a graph that is the MAX over ranks; replicated rows weigh 1/world in the backward pass
                shard.all_reduce_max_(gstat[0:1])
                gstat[3] = 1.0 / shard.world
        out = torch.empty(*x.shape[:-1], HH * m, dtype=x.dtype, device=x.device)
        call('hotk_phi_fwd', 4 * n * HH * (d + m), p(xs), ld, p(proj), p(kind), keep, p(out), HH * m, n, HH, d, m, mode, p(gstat))
        ctx.save_for_backward(xs, proj, kind, gstat)
        ctx.shard = shard
        ctx.cfg = (ld, n, HH, d, m, mode, keep, x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        xs, proj, kind, gstat = ctx.saved_tensors
        ld, n, HH, d, m, mode, keep, xshape = ctx.cfg
        gs, ldg, ng, Cg = rows2d(g)
        assert ng == n and Cg == HH * m
        if mode == _lib.PHI_SOFTMAX_K:
            gstat = gstat.clone()
            need = c_size_t(0)
            check(lib().hotk_phi_stats_workspace_bytes(n, HH, m, byref(need)), 'hotk_phi_stats_workspace_bytes')
            ws = torch.empty(need.value, dtype=torch.uint8, device=g.device)
            call('hotk_phi_key_bwd_stats', 4 * n * HH * (d + m), p(xs), ld, p(proj), p(kind), keep, p(gs), ldg, n, HH, d, m, p(gstat),
                                               p(ws), need.value)
            if ctx.shard is not None and ctx.shard.world > 1:
                ctx.shard.all_reduce_(gstat[1:3])      # total gradient of the maximum and the number of maxima
        dx = torch.empty(xshape, dtype=g.dtype, device=g.device)
        call('hotk_phi_bwd', 4 * n * HH * (2 * d + m), p(xs), ld, p(proj), p(kind), keep, p(gs), ldg, p(dx), HH * d, n, HH, d, m, mode,
                                 p(gstat))
        return dx, None, None, None, None, None, None


class Basis22Light(Function):
    """[R,D] -> [R,5D] order-2 'light' basis (models/sparse/linear.py:143-161)."""

    @staticmethod
    def forward(ctx, x, plan):
        xs, ld, n, D = rows2d(x)
        assert n == plan.R
        out = torch.empty(*x.shape[:-1], 5 * D, dtype=x.dtype, device=x.device)
        call('hotk_basis22_light_fwd', 24 * n * D, p(xs), ld, p(plan.kind), p(plan.rown), p(plan.coln), p(plan.tpos),
                                           p(plan.diag_pos), p(out), 5 * D, n, D)
        ctx.plan = plan
        ctx.xshape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        gs, ldg, n, C5 = rows2d(g)
        D = C5 // 5
        dx = torch.empty(ctx.xshape, dtype=g.dtype, device=g.device)
        call('hotk_basis22_light_bwd', 24 * n * D, p(gs), ldg, p(plan.kind), p(plan.rown), p(plan.tpos), p(plan.row_ptr),
                                           p(plan.row_perm), p(plan.col_ptr), p(plan.col_perm), p(dx), D, n, D)
        return dx, None


class Basis22Full(Function):
    """V = the 15 order-2 -> order-2 basis tensors of Linear(2, 2, cfg='default') concatenated, [R, 15 D]
    (models/sparse/linear.py:143-238), from closed forms of the per-node / per-graph sums: hotk_basis22_full_fwd after
    five segment sums / gathers; the backward is the transpose map (hotk_basis22_full_bwd_entries, three segment sums,
    hotk_basis22_full_bwd_gather).  counts = (od, idg [Nn], n_graph, eo_graph [B]) as fp32.  Edge-shard mode: the
    off-diagonal row / column / graph sums and their gradients are all-reduced (the diagonal is replicated)."""

    @staticmethod
    def _entry_sums(nt, gt, D, shard):
        """all-reduce of the sums over off-diagonal entries inside the node / graph tables (in place)"""
        if shard is None:
            return
        flat = torch.cat([nt[:, D:].reshape(-1), gt[:, D:].reshape(-1)])
        shard.all_reduce_(flat)
        k = nt.size(0) * 2 * D
        nt[:, D:] = flat[:k].view(-1, 2 * D)
        gt[:, D:] = flat[k:].view(-1, D)

    @staticmethod
    def forward(ctx, x, plan, counts, normalize):
        xs, ld, n, D = rows2d(x)
        assert n == plan.R and xs.dtype == torch.float32
        _lib.require_device(xs)
        dev = xs.device
        od, idg, ng, eog = counts
        nt = torch.empty(plan.Nn, 3 * D, dtype=torch.float32, device=dev)
        gt = torch.empty(plan.B, 2 * D, dtype=torch.float32, device=dev)
        gather_rows(xs, plan.diag_pos, plan.Nn, out=nt[:, 0:D])
        seg_sum_rows(xs, plan.row_seg(), plan.kind, KEEP_OFFDIAG, out=nt[:, D:2 * D])
        seg_sum_rows(xs, plan.col_seg(), plan.kind, KEEP_OFFDIAG, out=nt[:, 2 * D:])
        seg_sum_rows(xs, plan.graph_seg(2), plan.kind, KEEP_DIAG, out=gt[:, 0:D])
        seg_sum_rows(xs, plan.graph_seg(2), plan.kind, KEEP_OFFDIAG, out=gt[:, D:])
        shard = shard_of(plan)
        Basis22Full._entry_sums(nt, gt, D, shard)
        V = torch.empty(n, 15 * D, dtype=torch.float32, device=dev)
        call('hotk_basis22_full_fwd', 4 * n * (2 * D + 15 * D), p(xs), ld, p(plan.kind), p(plan.rown), p(plan.coln),
             p(plan.tpos), p(nt), 3 * D, p(gt), 2 * D, p(od), p(idg), p(ng), p(eog), p(V), 15 * D, n, D, max(plan.Emax, 1),
             int(normalize))
        ctx.plan, ctx.counts, ctx.normalize, ctx.xshape, ctx.shard = plan, counts, int(normalize), x.shape, shard
        return V.view(*x.shape[:-1], 15 * D)

    @staticmethod
    def backward(ctx, g):
        plan, (od, idg, ng, eog) = ctx.plan, ctx.counts
        D = ctx.xshape[-1]
        n = plan.R
        g = g.contiguous().view(n, 15 * D)
        dev = g.device
        w = torch.empty(n, 10 * D, dtype=torch.float32, device=dev)
        call('hotk_basis22_full_bwd_entries', 4 * n * 25 * D, p(g), 15 * D, p(plan.kind), p(plan.rown), p(plan.coln),
             p(plan.tpos), p(od), p(idg), p(ng), p(eog), p(w), 10 * D, n, D, max(plan.Emax, 1), ctx.normalize)
        dn = seg_sum_rows(w[:, 2 * D:5 * D], plan.row_seg(), plan.kind, KEEP_VALID)
        dn += seg_sum_rows(w[:, 5 * D:8 * D], plan.col_seg(), plan.kind, KEEP_OFFDIAG)
        dg = seg_sum_rows(w[:, 8 * D:], plan.graph_seg(2), plan.kind, KEEP_VALID)
        Basis22Full._entry_sums(dn, dg, D, ctx.shard)      # gradients of the all-reduced sums
        dx = torch.empty(n, D, dtype=torch.float32, device=dev)
        call('hotk_basis22_full_bwd_gather', 4 * n * 4 * D, p(w), 10 * D, p(plan.kind), p(plan.rown), p(plan.coln),
             p(plan.tpos), p(dn), 3 * D, p(dg), 2 * D, p(dx), D, n, D, max(plan.Emax, 1))
        return dx.view(ctx.xshape), None, None, None


def _l22_forward(xs, ld, W, b_off, b_diag, plan):
    """factored Linear(2, 2, 'light') on rows xs [R, D] -> (y [R, C], xd [Nn, D], x2 [R, 2D], Wn [D, 3C]: the node-level weight
    blocks side by side, kept for the backward pass)"""
    n, D = plan.R, W.size(0) // 5
    C = W.size(1)
    dev = xs.device
    # the node rows xd [Nn, D] and the entry pairs x2 [R, 2D] from one launch
    xd = torch.empty(plan.Nn, D, dtype=xs.dtype, device=dev)
    x2 = torch.empty(plan.R, 2 * D, dtype=xs.dtype, device=dev)
    call('hotk_basis22_pair_fwd_diag', 4 * n * 4 * D + 8 * plan.Nn * D, p(xs), ld, p(plan.kind), p(plan.tpos), p(x2), 2 * D, n, D,
         p(plan.diag_pos), p(xd), D, plan.Nn)
    Wn = torch.cat([W[0:D], W[3 * D:4 * D], W[4 * D:5 * D]], dim=1)               # [D, 3C]
    yn = thin_gemm_blocks(xd, Wn) if thin_blocks_ok(xd, D, 3 * C) else mm(xd, Wn, rows=valid_rows(plan, 'n'))
    if thin_ok(n, 2 * D, C) and plan.Nn * 3 * C < (1 << 32):
        # entry GEMM with the assembly as its epilogue: y is written once
        y = torch.empty(n, C, dtype=xs.dtype, device=dev)
        w2 = W[D:3 * D].contiguous()
        call('hotk_linear22_entry_fwd', 4 * n * (2 * D + C), p(x2), 2 * D, p(w2), 2 * D, C, p(yn), 3 * C,
             p(plan.kind), p(plan.rown), p(plan.coln), p(b_off), p(b_diag), p(y), C, n, plan.Nn)
    elif entry22_ok(x2, 2 * D, C, b_off, b_diag, plan):
        # the same on the tiled GEMM (wide layers): products over the valid rows, assembly in the epilogue, padding rows zero
        w2, ldw = _row_major(W[D:3 * D])
        planes = torch.empty(3, C, 2 * D, dtype=torch.bfloat16, device=dev)
        call('hotk_gemm6_split_weight', 10 * 2 * D * C, p(w2), ldw, 2 * D, C, 1, p(planes))
        rec = entry_records(plan)
        y = torch.empty(n, C, dtype=xs.dtype, device=dev)
        call('hotk_gemm6_entry22', 4 * rec.size(0) * (2 * D + C), p(x2), 2 * D, p(planes), p(y), C, n, 2 * D, C, p(rec),
             rec.size(0), p(yn), 3 * C, p(b_off), p(b_diag))
    else:
        y = mm(x2, W[D:3 * D], rows=valid_rows(plan, 'e'))
        call('hotk_assemble22', 8 * n * C, p(y), C, p(yn), 3 * C, p(plan.kind), p(plan.rown), p(plan.coln), p(b_off),
             p(b_diag), n, C)
    return y, xd, x2, Wn


_L22_REDUCE = os.environ.get('HOTK_L22_REDUCE', '1') == '1'      # (A/B switch)


def l22_reduce_ok(g, x2, plan):
    """the one-pass entry-level reductions of the light basis' backward (hotk_linear22_bwd_reduce) cover this batch: a
    sparse plan with its graphs' rows side by side (not an edge shard, whose sums are all-reduced elsewhere), fp32, the
    headline widths"""
    return (_L22_REDUCE and g.dtype == torch.float32 and x2.dtype == torch.float32 and n_rows(g) >= THIN_MIN_ROWS
            and type(plan).__name__ == 'Plan' and getattr(plan, 'shard', None) is None and plan.R < (1 << 31)
            and lib().hotk_linear22_bwd_reduce_supported(int(x2.size(1)), int(g.size(1)), int(plan.Nmax)) == 1)


def _l22_backward(g, xd, x2, W, plan, has_b, want_dx=True, want_dw=True, dxd_extra=None, dx_add=None, Wn=None):
    """gradients of the factored Linear(2, 2, 'light') from g = dy [R, C] -> (dx [R, D], dW [5D, C], db_off, db_diag).
    dxd_extra [Nn, D]: another gradient of the diagonal input rows (fc_1's), dx_add [R, D]: another gradient of all input
    rows (fc_v's); both are folded into hotk_basis22_pair_bwd's single pass over dx."""
    D, C = xd.size(1), W.size(1)
    dev = g.device
    vn, ve = valid_rows(plan, 'n'), valid_rows(plan, 'e')
    # Padding rows of an upstream gradient may hold anything.  They are zeroed first -- unless nothing below reads them:
    # the gather and the segment sums go through the plan's indices (valid rows only), and the two products skip them
    # when both run on the tiled GEMMs over the row list (then the pass over [R, C] is saved: 25 of them per C2 step).
    uses_thin = g.dtype == torch.float32 and thin_ok(n_rows(g), C, 2 * D)      # (the thin GEMM reads every row)
    skips_padding = (_L22_SKIP_MASK and ve is not None
                     and (not want_dx or (not uses_thin and mm_masks(g, W[D:3 * D], ve, True)))
                     and (not want_dw or wgrad6_ok(x2, g, ve)))
    if plan.Ev != plan.R and not skips_padding:
        g = mask_rows(g, plan.kind, KEEP_VALID, 0.0)
    # node gradients: [ dy on the node's diagonal | sum over off-diagonal row entries | ... column entries ]
    dyn = torch.empty(plan.Nn, 3 * C, dtype=g.dtype, device=dev)
    use_reduce = l22_reduce_ok(g, x2, plan)
    dW2e = None
    if use_reduce:
        # one pass over g for the three node blocks AND the entry-level weight gradient (csrc/l22bwd.hip)
        n = ctypes_long()
        check(lib().hotk_linear22_bwd_reduce_partials(plan.B, byref(n)), 'hotk_linear22_bwd_reduce_partials')
        part = torch.empty(n.value, 2 * D, C, dtype=g.dtype, device=dev)
        call('hotk_linear22_bwd_reduce', 4 * plan.Ev * (C + 2 * D) + 12 * plan.Nn * C, p(g), C, p(x2), 2 * D, 2 * D, C,
             p(plan.kind), p(plan.rown), p(plan.coln), p(plan.g_ebeg), p(plan.g_eend), p(plan.n_nodes_dev), plan.B,
             plan.Nmax, plan.R, p(dyn), 3 * C, p(part))
        dW2e = part[0] if n.value == 1 else part.sum(0)
    else:
        seg_sum_rows_pair(g, plan.row_seg(), plan.col_seg(), plan.kind, KEEP_OFFDIAG, dyn[:, C:2 * C], dyn[:, 2 * C:],
                          own=(plan.diag_pos, dyn[:, 0:C]))
    # every off-diagonal entry sits in exactly one row segment and every diagonal entry on one node, so the two
    # bias gradients are column sums of the node table (N rows) instead of another pass over the E entry rows -- and
    # those column sums are one more row of the node-level weight gradient: [xd | 1]^T . dyn (no pass of their own)
    db_diag = db_off = None
    # (not when the node-level weight gradient runs on the tiled GEMM: a 257th row would cost it a whole second row of tiles)
    # narrow node input, wide node gradient (the headline widths): data gradient, weight gradient and the column sums
    # (= both bias gradients) from ONE pass over dyn [Nn, 3C] (hotk_linear_bwd_fused) instead of three
    node_fused = want_dx and want_dw and dyn.dtype == torch.float32 and linear_bwd_ok(plan.Nn, D, 3 * C)
    ones_row = want_dw and (has_b[0] or has_b[1]) and not node_fused and not wgrad6_ok(xd, dyn, vn)
    if not ones_row and not node_fused and (has_b[0] or has_b[1]):
        db = dyn[:, 0:2 * C].sum(0)      # (one pass over both column blocks)
        db_diag = db[0:C] if has_b[1] else None
        db_off = db[C:2 * C] if has_b[0] else None
    if Wn is None:
        Wn = torch.cat([W[0:D], W[3 * D:4 * D], W[4 * D:5 * D]], dim=1)
    dx = dW = None
    dWn = None
    if node_fused:
        dxd, dWn, dbn = linear_bwd_fused(dyn, xd, Wn)
        db_diag = dbn[0:C] if has_b[1] else None
        db_off = dbn[C:2 * C] if has_b[0] else None
        if dxd_extra is not None:
            dxd += dxd_extra
    # the weight gradients read g and dyn only: on the side stream (small batches) next to the data-gradient chain.
    # (Measured on the 128-molecule C2 step, 10.62 ms without: this form 10.24; with the entry-level one forked earlier,
    # under the gather and the segment sums, fc_v's forward forked too and Proj22's joins deferred to its end: 10.32-10.37
    # in every combination -- a fork / join pair costs about what a short overlap gains.)
    with fork(g, want_dx and want_dw) as side:
        if want_dw:
            if dW2e is None:
                dW2e = wgrad(x2, g, ve)
            if dWn is not None:
                pass
            elif ones_row:
                # (the ones column padded to four: rows of [xd | 1 0 0 0] stay 16-byte aligned for the tiled weight-gradient GEMM)
                dWn = wgrad(torch.cat([xd, _ones4(plan, xd)], dim=1), dyn, vn)
                db_diag = dWn[D, 0:C] if has_b[1] else None
                db_off = dWn[D, C:2 * C] if has_b[0] else None
                dWn = dWn[:D]
            else:
                dWn = wgrad(xd, dyn, vn)
    if want_dx:
        if not node_fused:
            dxd = mm(dyn, Wn, trans_w=True, rows=vn)
            if dxd_extra is not None:
                dxd += dxd_extra
        # wide-input data gradient [R, C] x [C, 2D]: the thin kernel's row stream since its two-register-set form
        # (same-box A/B at 10 M rows, C = 288 -> 64: 3.08 ms against 3.46 ms for the library GEMM)
        dx2 = thin_gemm(g, W[D:3 * D].t()) if (g.dtype == torch.float32 and thin_ok(n_rows(g), C, 2 * D)) else mm(g, W[D:3 * D], trans_w=True, rows=ve)
        dx = torch.empty(plan.R, D, dtype=g.dtype, device=dev)
        call('hotk_basis22_pair_bwd', 4 * plan.R * 4 * D, p(dx2), 2 * D, p(plan.kind), p(plan.tpos), p(plan.rown),
             p(dxd), D, p(dx_add), D, p(dx), D, plan.R, D)
    side.join()
    if want_dw:
        dW = torch.cat([dWn[:, 0:C], dW2e, dWn[:, C:2 * C], dWn[:, 2 * C:]], dim=0)
    return dx, dW, db_off, db_diag


class Linear22Light(Function):
    """y = Linear(2, 2, cfg='light')(x) incl. both biases and the padding mask, in factored form.

    The five-block basis of models/sparse/linear.py:143-161 is linear in its blocks, so
        y[(i,j)] = x_ij W2 + tau x_ji W3 + (x_ii W4) + (x_jj W5) + b_off,      y[(v,v)] = x_vv W1 + b_diag
    with the three node-valued products computed once per node (N << E) and gathered back by hotk_assemble22; the big
    GEMM contracts 2D instead of 5D inputs per entry (2.5x fewer FLOPs, no [R,5D] tensor).  Backward mirrors it: the
    node gradients are row / column segment sums of dy (hotk_seg_sum_rows over the plan's CSR views)."""

    @staticmethod
    def forward(ctx, x, W, b_off, b_diag, plan):
        xs, ld, n, D = rows2d(x)
        assert n == plan.R and W.size(0) == 5 * D
        y, xd, x2, Wn = _l22_forward(xs, ld, W, b_off, b_diag, plan)
        ctx.save_for_backward(xd, x2, W, Wn)
        ctx.plan, ctx.has_b, ctx.xshape = plan, (b_off is not None, b_diag is not None), x.shape
        return y.view(*x.shape[:-1], W.size(1))

    @staticmethod
    def backward(ctx, g):
        xd, x2, W, Wn = ctx.saved_tensors
        plan = ctx.plan
        g = g.contiguous().view(plan.R, W.size(1))
        dx, dW, db_off, db_diag = _l22_backward(g, xd, x2, W, plan, ctx.has_b, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                                Wn=Wn)
        return (None if dx is None else dx.view(ctx.xshape)), dW, db_off, db_diag, None


class Proj22(Function):
    """The three input projections of KernelSelfAttn(2, 2) (models/sparse/kernelattn.py:119-123: h_1 = fc_1(G),
    h_2 = fc_2(G), v = fc_v(G.values)) as ONE autograd node: the layer input feeds all three, so autograd would
    otherwise materialise three [R, D] gradients (one of them a scatter of the diagonal rows into zeros) and add them
    in two more passes.  Here fc_1's node-level gradient joins fc_2's before hotk_basis22_pair_bwd and fc_v's is added
    inside that kernel: dx is written once.  Forward is the same kernels as Linear(2,1,'light') (diagonal gather + GEMM
    + bias / padding mask), ops.Linear22Light and ops.RowsMatmul."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2_off, b2_diag, Wv, bv, plan, nodes_padded):
        xs, ld, n, D = rows2d(x)
        assert n == plan.R and W2.size(0) == 5 * D and W1.size(0) == D and Wv.size(1) == D
        vn, ve = valid_rows(plan, 'n'), valid_rows(plan, 'e')
        xc = xs.reshape(n, D)       # (a copy only when x is a column slice of a wider tensor)
        wv = Wv.t()
        v = thin_gemm(xc, wv, bv) if thin_wide_ok(n, D, wv.size(1)) else mm(xc, Wv, bv, trans_w=True, rows=ve)
        h2, xd, x2, Wn = _l22_forward(xs, ld, W2, b2_off, b2_diag, plan)
        if b1 is not None and not nodes_padded and thin_blocks_ok(xd, D, W1.size(1)):
            h1 = thin_gemm_blocks(xd, W1, b1)
        elif b1 is not None and (not nodes_padded or mm_masks(xd, W1, vn, False, b1)):
            h1 = mm(xd, W1, b1, rows=vn)   # the bias rides on the GEMM: no padding node rows, or the GEMM leaves them zero
        else:
            h1 = mm(xd, W1, rows=vn)
            call('hotk_bias_mask_rows', 8 * plan.Nn * h1.size(1), p(h1), h1.size(1), p(plan.nkind), p(None), p(b1), plan.Nn,
                 h1.size(1))
        ctx.save_for_backward(xc, xd, x2, W1, W2, Wv, Wn)
        ctx.plan, ctx.xshape, ctx.nodes_padded = plan, x.shape, nodes_padded
        ctx.has_b = (b1 is not None, b2_off is not None, b2_diag is not None)
        return h1.view(plan.B, plan.Nmax, -1), h2.view(plan.B, plan.Emax, -1), v.view(plan.B, plan.Emax, -1)

    @staticmethod
    def backward(ctx, dh1, dh2, dv):
        xc, xd, x2, W1, W2, Wv, Wn = ctx.saved_tensors
        plan = ctx.plan
        D = xc.size(1)
        # fc_v: dx, dW and db from one pass over dv
        dv2 = dv.contiguous().view(plan.R, -1)
        wv = Wv.t()
        if linear_bwd_ok(plan.R, D, dv2.size(1)):
            dxv, dwv, dbv = linear_bwd_fused(dv2, xc, wv)
        else:
            ve = valid_rows(plan, 'e')
            with fork(dv2) as side_v:
                dwv, dbv = wgrad(xc, dv2, ve), dv2.sum(0)
            dxv = mm(dv2, Wv, rows=ve)
            side_v.join()
        # fc_1 (node rows)
        g1 = dh1.contiguous().view(plan.Nn, -1)
        if ctx.nodes_padded:
            g1 = mask_rows(g1, plan.nkind, KEEP_VALID, 0.0)
        vn = valid_rows(plan, 'n')
        if g1.dtype == torch.float32 and linear_bwd_ok(plan.Nn, xd.size(1), g1.size(1)):
            dxd1, dW1, db1 = linear_bwd_fused(g1, xd, W1)      # one pass over g1 [Nn, 12 dqk]
            if not ctx.has_b[0]:
                db1 = None
        else:
            db1 = bias_grads(g1, plan.nkind, False, True)[1] if ctx.has_b[0] else None
            with fork(g1) as side_1:
                dW1 = wgrad(xd, g1, vn)
            dxd1 = mm(g1, W1, trans_w=True, rows=vn)
            side_1.join()
        # fc_2, with the other two gradients of x folded in
        g2 = dh2.contiguous().view(plan.R, -1)
        dx, dW2, db_off, db_diag = _l22_backward(g2, xd, x2, W2, plan, ctx.has_b[1:], True, True, dxd_extra=dxd1, dx_add=dxv,
                                                 Wn=Wn)
        return dx.view(ctx.xshape), dW1, db1, dW2, db_off, db_diag, dwv.t(), dbv, None, None


class Assemble12(Function):
    """y[(i,j)] = yn[i][C:2C] + yn[j][2C:3C] + b_off off the diagonal, y[(v,v)] = yn[v][0:C] + b_diag, padding rows zero:
    the broadcast step of an order-1 -> order-2 linear layer (models/dense/linear.py:92-111 `_1_to_2`, get_output and
    the order-2 bias :214-225) in factored form -- the node-level blocks are projected per node ([Nn, 3C]) and this
    kernel (hotk_assemble22 on a zero tensor) gathers them back.  Backward: the node gradients are dy on the node's
    diagonal entry and the row / column segment sums of dy over the off-diagonal entries."""

    @staticmethod
    def forward(ctx, yn, b_off, b_diag, plan):
        _lib.require_device(yn)
        C = yn.size(-1) // 3
        assert yn.size(-1) == 3 * C and yn.numel() == plan.Nn * 3 * C
        yn = yn.contiguous()
        y = torch.zeros(plan.R, C, dtype=yn.dtype, device=yn.device)
        call('hotk_assemble22', 8 * plan.R * C, p(y), C, p(yn), 3 * C, p(plan.kind), p(plan.rown), p(plan.coln), p(b_off),
             p(b_diag), plan.R, C)
        ctx.plan, ctx.has_b, ctx.C = plan, (b_off is not None, b_diag is not None), C
        return y

    @staticmethod
    def backward(ctx, g):
        plan, C = ctx.plan, ctx.C
        g = g.contiguous().view(plan.R, C)
        if plan.Ev != plan.R:
            g = mask_rows(g, plan.kind, KEEP_VALID, 0.0)
        dyn = torch.empty(plan.Nn, 3 * C, dtype=g.dtype, device=g.device)
        seg_sum_rows_pair(g, plan.row_seg(), plan.col_seg(), plan.kind, KEEP_OFFDIAG, dyn[:, C:2 * C], dyn[:, 2 * C:],
                          own=(plan.diag_pos, dyn[:, 0:C]))
        db_off = dyn[:, C:2 * C].sum(0) if ctx.has_b[0] else None
        db_diag = dyn[:, 0:C].sum(0) if ctx.has_b[1] else None
        return dyn, db_off, db_diag, None


class BiasMask(Function):
    """y + bias by row kind, padding rows zeroed, in place on y (a fresh GEMM output).
    Stands in for _Bias._1/_2 followed by Linear's apply_mask(0) (models/sparse/linear.py:293-306,345-346)."""

    @staticmethod
    def forward(ctx, y, kind, b_off, b_diag, no_pad=False):
        ys, ld, n, C = rows2d(y)
        assert ys.data_ptr() == y.data_ptr(), 'BiasMask expects a row-contiguous tensor'
        call('hotk_bias_mask_rows', 8 * n * C, p(ys), ld, p(kind), p(b_off), p(b_diag), n, C)
        ctx.mark_dirty(y)
        ctx.save_for_backward(kind)
        ctx.has = (b_off is not None, b_diag is not None)
        ctx.no_pad = no_pad
        return y

    @staticmethod
    def backward(ctx, g):
        (kind,) = ctx.saved_tensors
        gs, ld, n, C = rows2d(g)
        db_off, db_diag = bias_grads(g, kind, ctx.has[0], ctx.has[1])
        return (g if ctx.no_pad else mask_rows(g, kind, KEEP_VALID, 0.0)), None, db_off, db_diag, None


class KernelAttn(Function):
    """Key-value-first kernel attention for one (already channel-concatenated) query/key/value triple.
    Stands in for KernelAttnCoef.forward (models/sparse/kernelattncoef.py:198-234): per-graph sums when
    diagonal is None, per-column sums (keys gathered by row node) otherwise; the output is zero on padding
    and, for order-2 queries, on diagonal rows.  KV is recomputed nowhere: it is saved for the backward."""

    @staticmethod
    def forward(ctx, q, k, v, plan, ord_q, ord_k, m, c, diagonal):
        HH = q.size(-1) // m
        assert q.size(-1) == HH * m and k.size(-1) == HH * m and v.size(-1) == HH * c, (q.shape, k.shape, v.shape, m, c)
        nq = q.numel() // q.size(-1)
        nk = k.numel() // k.size(-1)
        nv = v.numel() // v.size(-1)
        if diagonal is None:
            assert nk == nv
            seg = plan.graph_seg(ord_k)
            kidx = None
            sidx, seg_div = None, (plan.Nmax if ord_q == 1 else plan.Emax)
            vsidx, vseg_div = None, (plan.Nmax if ord_k == 1 else plan.Emax)
        else:
            dim1 = diagonal[0]
            assert ord_k == 1 and nv == plan.R and nk == plan.Nn
            seg = plan.col_seg()
            kidx = plan.rown
            if ord_q == 1:
                sidx, seg_div = None, 1
            else:
                sidx, seg_div = (plan.coln if dim1 == 2 else plan.rown), 1
            vsidx, vseg_div = plan.coln, 1
        # rows keyed by their COLUMN node are visited column by column (the plan's CSR permutation), so that a node's
        # key-value block is fetched once per node instead of once per entry (it is m (c + 1) floats per head: 24 KB at
        # d = 16, m = 44); rows keyed by their row node are adjacent already
        by_col = plan.col_perm[:plan.Ev] if diagonal is not None else None
        qorder = by_col if (diagonal is not None and ord_q == 2 and diagonal[0] == 2) else None
        vorder = by_col
        kind_q, keep_q = (plan.nkind, KEEP_DIAG) if ord_q == 1 else (plan.kind, KEEP_OFFDIAG)
        # edge-shard mode (hot_amd.dist): sums over ENTRIES are partial on each rank and are all-reduced; diagonal
        # entries are replicated on every rank and enter with weight 1/world.  Sums over nodes are replicated work.
        shard = getattr(plan, 'shard', None)
        reduce_kv = shard is not None and (diagonal is not None or ord_k == 2)
        wv = aug = None
        vw = v
        if shard is not None and diagonal is not None:
            wv = shard.entry_weight.view(-1, 1)
            vw = v.reshape(nv, HH * c) * wv
            aug = wv.expand(nv, HH).contiguous()
        KV = seg_outer_sum(k, kidx, vw, None, aug, seg, HH, m, c)
        if reduce_kv:
            shard.all_reduce_(KV)
        out = row_contract(q, None, KV, sidx, seg_div, kind_q, keep_q, nq, HH, m, c, True, order=qorder)
        ctx.save_for_backward(q, k, vw, KV, aug, wv)
        ctx.cfg = (plan, ord_q, ord_k, HH, m, c, diagonal, nq, nk, nv, kidx, sidx, seg_div, vsidx, vseg_div, kind_q, keep_q)
        ctx.orders = (qorder, vorder)
        ctx.vshape, ctx.reduce_kv = v.shape, reduce_kv
        return out.view(*q.shape[:-1], HH * c)

    @staticmethod
    def backward(ctx, dO):
        q, k, v, KV, aug, wv = ctx.saved_tensors
        plan, ord_q, ord_k, HH, m, c, diagonal, nq, nk, nv, kidx, sidx, seg_div, vsidx, vseg_div, kind_q, keep_q = ctx.cfg
        # (dO may be a column slice of the gradient of the concatenated attention terms: read in place, rows2d)
        qorder, vorder = ctx.orders
        X, XA = apply_bwd_stats(q, KV, sidx, seg_div, kind_q, keep_q, dO, nq, HH, m, c, order=qorder)
        dq = row_expand(KV, sidx, seg_div, X, None, XA, kind_q, keep_q, nq, HH, m, c, order=qorder)
        # dKV[s] = sum over the queries that read segment s of q (x) [dnum | dden]
        if diagonal is None:
            qseg = plan.graph_seg(ord_q)
        elif ord_q == 1:
            qseg = plan.unit_seg(nq)
        else:
            qseg = plan.col_seg() if diagonal[0] == 2 else plan.row_seg()
        dKV = seg_outer_sum(q, None, X, None, XA, qseg, HH, m, c)
        if ctx.reduce_kv:
            plan.shard.all_reduce_(dKV)       # backward of the forward all-reduce: every rank needs the full dKV
        kind_v = plan.nkind if (diagonal is None and ord_k == 1) else plan.kind
        dv = row_contract(k, kidx, dKV, vsidx, vseg_div, kind_v, KEEP_VALID, nv, HH, m, c, False, order=vorder)
        if wv is not None:
            dv = dv * wv
        dkc = row_expand(dKV, vsidx, vseg_div, v, None, aug, kind_v, KEEP_VALID, nv, HH, m, c, order=vorder)
        if diagonal is None:
            dk = dkc
        else:
            dk = seg_sum_rows(dkc, plan.row_seg())
        return dq.view(q.shape), dk.view(k.shape), dv.view(ctx.vshape), None, None, None, None, None, None


class AllReduceSum(Function):
    """Edge-shard mode (hot_amd.dist): a sum over ENTRIES is partial on a rank; forward adds the partial sums
    of all ranks (RCCL SUM), backward adds the gradients that the ranks' local losses send into the sum -- the
    all-reduce is its own adjoint.  Replicated quantities (node tables, diagonal entries, graph-level vectors) never
    pass through it: their gradients stay partial per rank and are added over ranks by the caller, like the parameter
    gradients."""

    @staticmethod
    def forward(ctx, x, shard):
        ctx.shard = shard
        return shard.all_reduce_(x.contiguous().clone())

    @staticmethod
    def backward(ctx, g):
        return ctx.shard.all_reduce_(g.contiguous().clone()), None


def shard_of(plan):
    """the edge-shard context of a plan when more than one rank holds the graph, else None"""
    shard = getattr(plan, 'shard', None)
    return shard if shard is not None and shard.world > 1 else None


def entry_sum(x, plan):
    """x = this rank's sum over (off-diagonal) entries -> the sum over every rank's entries"""
    shard = shard_of(plan)
    return x if shard is None else AllReduceSum.apply(x, shard)


class SoftmaxPool(Function):
    """Per-graph softmax attention with a zeroth-order query: out[b] = sum_r softmax_r(q_b . k_r / sqrt(d)) v_r over the
    kept rows of graph b, per head.  Stands in for AttnCoef._vector_query + apply_attn(k = 0)
    (models/sparse/attncoef.py:25-59,165-173).  One wavefront walks a segment; graphs with more than POOL_SPLIT rows
    are cut into pieces whose results are merged exactly (piece maximum and sum of exponentials: the usual split
    softmax), so a single giant graph spreads over the chip."""

    @staticmethod
    def forward(ctx, q, k, v, seg, kind, keep, H, shard=None):
        """shard: edge-shard context when the pooled rows are ENTRIES spread over ranks (the rows of a rank are then pieces
        of the graph's softmax like the pieces of a long segment: piece maxima MAX-all-reduced, the sums of exponentials and
        the weighted piece outputs SUM-all-reduced; backward all-reduces the incoming gradient, since every rank's local
        loss sees the replicated output)"""
        beg, end, _, maxlen = seg
        qs, ldq, nb, Cq = rows2d(q)
        ks, ldk, n, Ck = rows2d(k)
        vs, ldv, nv, Cv = rows2d(v)
        assert Cq == Ck and nv == n and nb == beg.numel() and Cq % H == 0 and Cv % H == 0 and kind.numel() == n
        d, c = Cq // H, Cv // H
        dev = q.device
        alpha = torch.zeros(n, H, dtype=q.dtype, device=dev)
        split = maxlen > POOL_SPLIT or shard is not None
        if not split:
            out = torch.empty(nb, Cv, dtype=q.dtype, device=dev)
            call('hotk_seg_softmax_pool_fwd', 4 * n * (Ck + Cv), p(qs), ldq, p(ks), ldk, p(vs), ldv, p(beg), p(end), p(kind),
                 keep, p(out), Cv, p(alpha), p(None), p(None), nb, H, d, c)
            owner = fine = group = None
        else:
            ref = refined(seg, POOL_SPLIT)
            fine, group, owner = ref.fine, ref.group, ref.owner
            fb, fe = fine[0], fine[1]
            npc = fb.numel()
            outp = torch.empty(npc, Cv, dtype=q.dtype, device=dev)
            stats = torch.empty(npc, H, 2, dtype=q.dtype, device=dev)
            call('hotk_seg_softmax_pool_fwd', 4 * n * (Ck + Cv), p(qs), ldq, p(ks), ldk, p(vs), ldv, p(fb), p(fe), p(kind),
                 keep, p(outp), Cv, p(alpha), p(stats), p(owner), npc, H, d, c)
            # merge: weight of piece p in its graph = s_p e^(m_p - M) / sum_q s_q e^(m_q - M)   (tiny [pieces, H] tensors)
            own = owner.long()
            m_p, s_p = stats[..., 0], stats[..., 1]
            M = torch.full((nb, H), float('-inf'), dtype=q.dtype, device=dev).scatter_reduce(
                0, own.unsqueeze(1).expand(-1, H), m_p, 'amax', include_self=True)
            if shard is not None:
                shard.all_reduce_max_(M)
            w = torch.where(s_p > 0, s_p * torch.exp(m_p - M[own]), torch.zeros_like(s_p))
            Z = seg_sum_rows(w, group)
            if shard is not None:
                shard.all_reduce_(Z)
            wn = torch.where(Z[own] > 0, w / Z[own], torch.zeros_like(w))                       # [pieces, H]
            out = seg_sum_rows((outp.view(npc, H, c) * wn.unsqueeze(-1)).view(npc, Cv), group)
            if shard is not None:
                shard.all_reduce_(out)
            # alpha of a row = its within-piece softmax weight x the weight of its piece
            row_ids, piece_of_row = ref.rows()
            alpha[row_ids] = alpha[row_ids] * wn[piece_of_row]
        ctx.save_for_backward(qs, ks, vs, alpha, out, beg, end)
        ctx.cfg = (ldq, ldk, ldv, nb, n, H, d, c, q.shape, k.shape, v.shape)
        ctx.split = (fine, group, owner) if split else None
        ctx.shard = shard
        return out

    @staticmethod
    def backward(ctx, g):
        qs, ks, vs, alpha, out, beg, end = ctx.saved_tensors
        ldq, ldk, ldv, nb, n, H, d, c, qshape, kshape, vshape = ctx.cfg
        g = g.contiguous()
        if ctx.shard is not None:
            g = ctx.shard.all_reduce_(g.clone())
        dk = torch.zeros(n, H * d, dtype=g.dtype, device=g.device)
        dv = torch.zeros(n, H * c, dtype=g.dtype, device=g.device)
        if ctx.split is None:
            dq = torch.zeros(nb, H * d, dtype=g.dtype, device=g.device)
            call('hotk_seg_softmax_pool_bwd', 4 * n * 2 * H * (d + c), p(qs), ldq, p(ks), ldk, p(vs), ldv, p(beg), p(end),
                 p(alpha), p(out), H * c, p(g), H * c, p(dq), H * d, p(dk), H * d, p(dv), H * c, p(None), nb, H, d, c)
        else:
            fine, group, owner = ctx.split
            npc = fine[0].numel()
            dqp = torch.zeros(npc, H * d, dtype=g.dtype, device=g.device)
            call('hotk_seg_softmax_pool_bwd', 4 * n * 2 * H * (d + c), p(qs), ldq, p(ks), ldk, p(vs), ldv, p(fine[0]), p(fine[1]),
                 p(alpha), p(out), H * c, p(g), H * c, p(dqp), H * d, p(dk), H * d, p(dv), H * c, p(owner), npc, H, d, c)
            dq = seg_sum_rows(dqp, group)
        return dq.view(qshape), dk.view(kshape), dv.view(vshape), None, None, None, None, None


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# below 64 k rows the batches are 256 rows each (measured on the 128-molecule C2 step, 7 936 rows: 11.30 ms with one GEMM
# per weight gradient, 10.66 with 31 batches of 256 rows, 10.8-11.7 with 128 / 192 / 384 / 512 / 1 024 / 2 048: the
# library's kernel choice likes the power-of-two k extent)
_WGRAD_SPLIT_MIN = int(os.environ.get('HOTK_WGRAD_SPLIT_MIN', 1 << 12))
_WGRAD_SPLIT_ROWS = int(os.environ.get('HOTK_WGRAD_SPLIT_ROWS', 256))


# Small batches (the 128-molecule C2 step: 7 936 rows) leave most of the chip idle in every launch: a weight gradient has
# no consumer inside the backward pass, so it runs on a side stream next to the data-gradient chain of the same autograd
# node.  fork() makes the side stream wait for what is queued on the current one; join() makes the current stream wait
# for the side stream before the node returns -- so every tensor the side work read or wrote is ordered before any later
# use or reuse on the caller's stream (no record_stream needed), and a stream capture that contains a fork contains
# its join.
_SIDE_WGRAD = os.environ.get('HOTK_SIDE_WGRAD', '1') == '1'
_SIDE_MAX_ROWS = int(os.environ.get('HOTK_SIDE_MAX_ROWS', 1 << 14))
_side_streams = {}


class fork:
    """`with fork(ref) as f: ...` (body on the side stream when ref is a small CUDA batch), then `f.join()`"""

    def __init__(self, ref, enabled=True):
        self.side = None
        if enabled and _SIDE_WGRAD and ref.is_cuda and n_rows(ref) <= _SIDE_MAX_ROWS:
            dev = ref.device.index
            if dev not in _side_streams:
                _side_streams[dev] = torch.cuda.Stream(device=ref.device)
            self.side, self.cur = _side_streams[dev], torch.cuda.current_stream(ref.device)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.cur)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.side is not None:
            self.cur.wait_stream(self.side)


def wgrad(A, B, rows=None):
    """A^T @ B for tall operands A [R,K], B [R,N] (a reduction over R rows into a small K x N result).
    A single library GEMM has only K*N/tile workgroups to run on, so the row dimension is split into S batches
    (torch.bmm -> rocBLAS/hipBLASLt batched GEMM) whose partial products are summed: split-K over 256 CUs.
    rows: the rows that are not padding (the tiled GEMM skips the others; they are zero in A or B either way)."""
    R = A.size(0)
    if wgrad6_ok(A, B, rows):
        return wgrad6(A, B, rows)
    if R < _WGRAD_SPLIT_MIN:
        return (A.t() @ B).float()
    S = 256 if R >= (1 << 16) else max(2, R // _WGRAD_SPLIT_ROWS)
    rs = R // S
    main = rs * S
    out = torch.bmm(A[:main].view(S, rs, A.size(1)).transpose(1, 2), B[:main].view(S, rs, B.size(1))).float().sum(0)
    if main < R:
        out = out + (A[main:].t() @ B[main:]).float()
    return out


class RowsMatmul(Function):
    """y = x @ W (+ b) over rows [R,K] x [K,N]; backward uses the split-K weight gradient.  The dense projections
    stay library GEMMs (models/sparse/linear.py:46-48, kernelattn.py:65-72)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        x2 = x.reshape(-1, x.size(-1))
        # 2-D result on purpose (a fresh tensor, not a view): callers may finish it in place (BiasMask)
        if x2.is_cuda and x2.dtype == torch.float32 and thin_wide_ok(x2.size(0), x2.size(1), w.size(1)):
            return thin_gemm(x2.contiguous(), w, b)
        return mm(x2, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.size(-1))
        x2 = x.reshape(-1, x.size(-1))
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if (g2.is_cuda and g2.dtype == torch.float32 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and g2.size(0) >= THIN_MIN_ROWS and lib().hotk_linear_bwd_supported(x2.size(1), g2.size(1)) == 1):
            # narrow input, wide gradient: dx, dW and db from ONE pass over g (csrc/linbwd.hip) instead of a data-gradient
            # GEMM, a split-K batched GEMM over [x | 1] and the concatenation that builds it
            dx, dw, db = linear_bwd_fused(g2.contiguous(), x2.contiguous(), w)
            return dx.view(x.shape), dw, (db if want_b else None)
        dw = db = None
        with fork(g2, ctx.needs_input_grad[0]) as f:
            if ctx.needs_input_grad[1] and want_b and g2.size(1) > 2 * x2.size(1) and x2.size(0) >= (1 << 16):
                # g is the wide operand: [x | 1]^T @ g yields dW and db in ONE pass over g
                xa = torch.cat([x2, torch.ones(x2.size(0), 1, dtype=x2.dtype, device=x2.device)], dim=1)
                dwa = wgrad(xa, g2)
                dw, db = dwa[:-1], dwa[-1]
            else:
                dw = wgrad(x2, g2) if ctx.needs_input_grad[1] else None
                db = g2.sum(0) if want_b else None
        dx = mm(g2, w, trans_w=True).view(x.shape) if ctx.needs_input_grad[0] else None
        f.join()
        return dx, dw, db


def linear_rows(x, lin):
    """nn.Linear over the rows of a batch tensor through RowsMatmul: aten's own backward forms the weight gradient
    x^T . dy as ONE GEMM over all rows (32 x 640 outputs over 1 M rows: 81 workgroups, 1.9 ms), RowsMatmul splits the
    rows (fused / split-K / tiled forms above)"""
    if not x.is_cuda or x.numel() == 0:
        return lin(x)
    y = RowsMatmul.apply(x, lin.weight.t(), lin.bias)
    return y.view(*x.shape[:-1], lin.out_features)


class AttnCore22(Function):
    """out = skip + mask(fc_o(cat(att_1..att_10))) for KernelSelfAttn(2,2) at the fused shape, straight from the raw
    projections h1 = fc_1(G), h2 = fc_2(G), v = fc_v(values): the feature maps, the six per-node and four per-graph
    key-value sums, their application, the transposes and the diagonal scatters all happen inside
    hotk_attn22_forward / _backward (models/sparse/kernelattn.py:119-159 and everything it calls in
    kernelattncoef.py).  The two dense projections around it stay library GEMMs."""

    @staticmethod
    def _args(plan, h1, h2, v, proj, D):
        a = _lib.Attn22Args()
        a.B, a.Emax, a.Nmax, a.R, a.Nn, a.D = plan.B, plan.Emax, plan.Nmax, plan.R, plan.Nn, D
        a.H, a.m, a.d_head, a.c = 4, 16, 8, 8
        for n in ('kind', 'nkind', 'rown', 'coln', 'tpos', 'diag_pos', 'col_ptr', 'col_perm', 'row_ptr', 'row_perm'):
            setattr(a, n, getattr(plan, n).data_ptr())
        a.n_entries = plan.n_entries_dev.data_ptr()
        a.n_nodes = plan.n_nodes_dev.data_ptr()
        col_rec, row_rec = plan.records()
        a.col_rec, a.row_rec = col_rec.data_ptr(), row_rec.data_ptr()
        a.h1, a.ld_h1 = h1.data_ptr(), h1.size(-1)
        a.h2, a.ld_h2 = h2.data_ptr(), h2.size(-1)
        a.v, a.ld_v = v.data_ptr(), v.size(-1)
        a.proj = proj.data_ptr()
        return a

    @staticmethod
    def _phik(plan, h1, proj):
        """phi of the eight node-level key slots, once per node (each node is the key of ~deg entries)"""
        phik = torch.empty(plan.Nn, 512, dtype=torch.float32, device=h1.device)
        call('hotk_phi_fwd', 4 * plan.Nn * (256 + 512), c_void_p_of(h1, 128), 384, p(proj), p(plan.nkind), KEEP_DIAG,
             p(phik), 512, plan.Nn, 32, 8, 16, _lib.PHI_RELU, p(None))
        return phik

    @staticmethod
    def _workspace(plan, dev):
        need = c_size_t(0)
        check(lib().hotk_attn22_workspace_bytes(plan.B, plan.Emax, byref(need)), 'hotk_attn22_workspace_bytes')
        return torch.empty(max(need.value, 16), dtype=torch.uint8, device=dev), need.value

    @staticmethod
    def _drop_masks(plan, dev, drop_p, mu_p):
        """keep masks of the two dropouts that sit between cat(att_1..att_10) and fc_o (kernelattn.py:70-72): MuDropout
        keeps or drops a whole term of a whole graph (common/mudrop.py:5-20: mask [n_terms, B, 1, 1]), nn.Dropout single
        elements.  Returned as (mu [B, 1, 10, 1] float or None, keep [R, 320] bool or None); scales are applied apart."""
        mu = keep = None
        if mu_p > 0.:
            mu = (torch.rand(10, plan.B, device=dev) > mu_p).float().t().reshape(plan.B, 1, 10, 1) / (1. - mu_p)
        if drop_p > 0.:
            keep = torch.rand(plan.R, 320, device=dev) > drop_p
        return mu, keep

    @staticmethod
    def _drop_apply(x, plan, mu, keep, drop_p):
        """x [R, 320] *= mu-dropout and dropout multipliers, in place (x is a fresh tensor)"""
        if mu is not None:
            x.view(plan.B, plan.Emax, 10, 32).mul_(mu)
        if keep is not None:
            x.mul_(keep).mul_(1. / (1. - drop_p))
        return x

    @staticmethod
    def forward(ctx, h1, h2, v, proj, w_o, b_o, plan, drop_p=0., mu_p=0.):
        for t_ in (h1, h2, v):
            _lib.require_device(t_)
            assert t_.is_contiguous() and t_.dtype == torch.float32
        D = h2.size(-1) - 256
        assert h1.size(-1) == 384 and v.size(-1) == 320 and D >= 0 and w_o.shape == (D, 320)
        dev = h2.device
        proj = proj.contiguous().float()
        padded = plan.Ev != plan.R
        att = (torch.zeros if padded else torch.empty)(plan.R, 320, dtype=torch.float32, device=dev)
        kvg = torch.empty(plan.B, 4, 25, 64, dtype=torch.float32, device=dev)
        a = AttnCore22._args(plan, h1, h2, v, proj, D)
        phik = AttnCore22._phik(plan, h1.view(plan.Nn, 384), proj)
        a.phik, a.ld_phik = phik.data_ptr(), 512
        a.att, a.ld_att, a.kvg = att.data_ptr(), 320, kvg.data_ptr()
        ws, nws = AttnCore22._workspace(plan, dev)
        shard = getattr(plan, 'shard', None)
        kvn = None
        if shard is None:
            call('hotk_attn22_forward', 4 * plan.Ev * (8 * 32 + 10 * 32 + 10 * 32) + 4 * plan.Nn * 384,
                 byref(a), p(ws), nws)
        else:
            # edge-shard mode (hot_amd.dist): the sums over ENTRIES are partial on a rank; the core runs in phases and
            # the sums are all-reduced in between (per-graph sums of entry keys = kvg slots 1, 3; all per-node sums)
            # kvn = [part 0 sums (rank-local) | part 1, 2 sums | nd0]: everything after part 0 is ONE all-reduce
            P1 = plan.Nn * 1152
            kvn = (torch.empty if plan.B == 1 else torch.zeros)(3 * P1 + plan.Nn * 128, dtype=torch.float32, device=dev)
            a.kvn, a.nd0, a.diag_w = kvn.data_ptr(), kvn.data_ptr() + 4 * 3 * P1, 1.0 / shard.world
            AttnCore22._phase(a, _lib.ATTN22_FWD_GRAPHSUM, ws, nws, 4 * plan.Ev * 4 * 32)
            AttnCore22._reduce_entry_slots(shard, kvg)
            AttnCore22._phase(a, _lib.ATTN22_FWD_NODESUM, ws, nws, 4 * plan.Ev * 6 * 32 + 4 * kvn.numel())
            # the exchange of the per-node sums (the large one: 2 432 floats per node) runs under the half of the
            # application that needs the per-graph sums only
            pending = shard.all_reduce_start_(kvn[P1:])
            AttnCore22._phase(a, _lib.ATTN22_FWD_APPLY_GRAPH, ws, nws, 4 * plan.Ev * (2 * 32 + 4 * 32))
            pending.wait()
            AttnCore22._phase(a, _lib.ATTN22_FWD_APPLY_NODE, ws, nws, 4 * plan.Ev * (4 * 32 + 6 * 32) + 4 * kvn.numel())
        # training-mode MuDropout / Dropout act on the concatenated terms, outside the core: the fused kernels stay on
        # the path (the reference's shipped training configs use dropout 0.1, enc-small.sh:3)
        mu, keep = AttnCore22._drop_masks(plan, dev, drop_p, mu_p)
        AttnCore22._drop_apply(att, plan, mu, keep, drop_p)      # att now holds what fc_o sees
        ctx.drop = (mu, keep, drop_p)
        skip = h2.view(plan.R, -1)[:, 256:]
        if not padded and thin_ok(plan.R, 320, D):
            out = thin_gemm(att, w_o.t(), b_o, skip).view(*h2.shape[:-1], D)      # projection + bias + skip path in one pass
        else:
            ve = valid_rows(plan, 'e')
            y = mm(att, w_o, b_o, trans_w=True, rows=ve)
            if padded and not mm_masks(att, w_o, ve, True, b_o):
                y = mask_rows(y, plan.kind, KEEP_VALID, 0.0)
            out = y.add_(skip).view(*h2.shape[:-1], D)
        ctx.save_for_backward(h1, h2, v, proj, w_o, att, kvg, phik, kvn)
        ctx.plan = plan
        return out

    @staticmethod
    def _phase(a, phase, ws, nws, alg_bytes):
        call('hotk_attn22_phase', alg_bytes, byref(a), phase, p(ws), nws)

    @staticmethod
    def _reduce_entry_slots(shard, kvg):
        """per-graph sums whose keys are entries (terms 8, 10 = slots 1, 3) are partial on a rank; the node-keyed
        slots 0, 2 are replicated work.  The rotated copies inside a slot are linear in it, so they reduce along."""
        t = kvg[:, 1::2].contiguous()
        shard.all_reduce_(t)
        kvg[:, 1::2] = t

    @staticmethod
    def backward(ctx, d_out):
        h1, h2, v, proj, w_o, att, kvg, phik, kvn = ctx.saved_tensors
        plan = ctx.plan
        dev = h2.device
        D = w_o.size(0)
        d_out = d_out.contiguous().view(plan.R, D)
        padded = plan.Ev != plan.R
        ve = valid_rows(plan, 'e')
        use_thin = thin_wide_ok(plan.R, D, 320)
        use_fused_dwo = _DWO_FUSED and d_out.dtype == torch.float32 and linear_bwd_ok(plan.R, D, 320)
        # the padding rows of the incoming gradient are zeroed first -- unless nothing below reads them: both products on
        # the tiled GEMMs over the row list, the bias gradient as a sum by row kind (wide layers: C2)
        skips_padding = (_L22_SKIP_MASK and padded and ve is not None and not use_thin and not use_fused_dwo
                         and mm_masks(d_out, w_o, ve) and wgrad6_ok(d_out, att, ve))
        dy = mask_rows(d_out, plan.kind, KEEP_VALID, 0.0) if (padded and not skips_padding) else d_out
        datt = thin_gemm(dy, w_o) if use_thin else mm(dy, w_o, rows=ve)
        # att as saved = the (dropped-out) input of fc_o.  dW_o = dy^T . att in the weight-gradient-only mode of
        # hotk_linear_bwd_fused (x = dy, g = att): 3.65 ms against 3.41 ms for the split-K batched GEMM before that
        # kernel's loads were made unconditional, measured again after (profiles/NOTES_r01_r02.md, old section 5)
        # (small batches: fc_o's weight and bias gradients on the side stream, under the core's backward kernels)
        with fork(dy, getattr(plan, 'shard', None) is None) as side:
            if use_fused_dwo:
                dw_o = linear_bwd_fused(att, dy.contiguous(), None)[1]
            else:
                dw_o = wgrad(dy, att, ve)
            if skips_padding:
                db = bias_grads(dy, plan.kind, True, True)
                db_o = db[0] + db[1]
            else:
                db_o = dy.sum(0)
        mu, keep, drop_p = ctx.drop
        AttnCore22._drop_apply(datt, plan, mu, keep, drop_p)
        padded = plan.Ev != plan.R
        # padding rows of dh1 / dh2 are never read (the BiasMask backward of fc_1 / fc_2 selects valid rows only);
        # dv feeds a plain GEMM, so its padding rows must be exact zeros
        dh1 = torch.empty(plan.Nn, 384, dtype=torch.float32, device=dev)
        dh2 = torch.empty(plan.R, 256 + D, dtype=torch.float32, device=dev)
        dv = (torch.zeros if padded else torch.empty)(plan.R, 320, dtype=torch.float32, device=dev)
        ck = torch.empty(plan.R, 192, dtype=torch.float32, device=dev)
        dkvg = torch.empty_like(kvg)
        a = AttnCore22._args(plan, h1, h2, v, proj, D)
        a.phik, a.ld_phik = phik.data_ptr(), 512
        a.kvg = kvg.data_ptr()
        a.datt, a.ld_datt = datt.data_ptr(), 320
        a.dout, a.ld_dout = d_out.data_ptr(), D
        a.dh1, a.ld_dh1 = dh1.data_ptr(), 384
        a.dh2, a.ld_dh2 = dh2.data_ptr(), 256 + D
        a.dv, a.ld_dv = dv.data_ptr(), 320
        a.ck, a.dkvg = ck.data_ptr(), dkvg.data_ptr()
        ws, nws = AttnCore22._workspace(plan, dev)
        shard = getattr(plan, 'shard', None)
        if shard is None:
            # algorithmic bytes as SURVEY 8d counts them: read h2, v, datt, write dh2, dv (+ indices); the per-entry key
            # gradients ck are this implementation's scratch, not algorithmic traffic
            call('hotk_attn22_backward', plan.Ev * (4 * (8 * 32 * 2 + 10 * 32 * 3) + 25) + 4 * plan.Nn * 384 * 2,
                 byref(a), p(ws), nws)
        else:
            # backward of the forward all-reduces: every rank needs the full d(sums) before its key / value side
            # dkvn = [d(part 1, 2 sums) | dd0]: one all-reduce
            P1 = plan.Nn * 1152
            dkvn = (torch.empty if plan.B == 1 else torch.zeros)(2 * P1 + plan.Nn * 128, dtype=torch.float32, device=dev)
            a.kvn, a.nd0, a.diag_w = kvn.data_ptr(), kvn.data_ptr() + 4 * 3 * P1, 1.0 / shard.world
            a.dkvn, a.dd0 = dkvn.data_ptr(), dkvn.data_ptr() + 4 * 2 * P1
            AttnCore22._phase(a, _lib.ATTN22_BWD_QUERY, ws, nws, 4 * plan.Ev * (6 * 32 * 2 + 10 * 32) + 8 * kvn.numel())
            AttnCore22._reduce_entry_slots(shard, dkvg)      # (small, and first: the per-graph half below waits for it)
            pending = shard.all_reduce_start_(dkvn)
            AttnCore22._phase(a, _lib.ATTN22_BWD_KEYVAL_GRAPH, ws, nws, 4 * plan.Ev * (4 * 32 * 2 + 2 * 32))
            pending.wait()
            AttnCore22._phase(a, _lib.ATTN22_BWD_KEYVAL_NODE, ws, nws, 4 * plan.Ev * (6 * 32 * 2 + 6 * 32 + 2 * 32) + 4 * kvn.numel())
        # d(raw key slots 0..5 of node i) = sum over the entries of row i of their per-entry contributions
        seg_sum_rows(ck, plan.row_seg(), None, KEEP_VALID, out=dh1[:, 128:320])
        side.join()
        return dh1.view(h1.shape), dh2.view(h2.shape), dv.view(v.shape), None, dw_o, db_o, None, None, None


def seg_sum_rows_bf16(src, seg, kind=None, keep=KEEP_VALID, out=None):
    """seg_sum_rows for a contiguous bf16 source [n, C]: fp32 sums in segment order (hotk_seg_sum_rows_bf16)"""
    beg, end, perm, maxlen = seg
    n_seg = beg.numel()
    if maxlen > SEG_SPLIT:
        fine, group = refine_seg(seg, SEG_SPLIT)
        part = seg_sum_rows_bf16(src, fine, kind, keep)
        return seg_sum_rows(part, group, None, KEEP_VALID, out)
    assert src.dtype == torch.bfloat16 and src.is_contiguous() and src.dim() == 2
    n, C = src.shape
    if out is None:
        out = torch.empty(n_seg, C, dtype=torch.float32, device=src.device)
    o, ldo, no, Co = rows2d(out)
    assert o.data_ptr() == out.data_ptr() and no == n_seg and Co == C
    call('hotk_seg_sum_rows_bf16', 2 * C * n + 4 * C * n_seg, p(src), C, p(beg), p(end), p(perm), p(kind), keep, p(o), ldo,
         n_seg, C, maxlen)
    return out


class AttnLayer22BF16(Function):
    """KernelSelfAttn(2,2) at the fused shape in the bf16 STORAGE mode (BASELINE north_star: bf16 held to 2e-2; the
    reference trains in fp32 only): out = skip + mask(fc_o(cat(att_1..att_10))) from the layer input x [R, D] and the
    node-level projection h1 = fc_1(G) (fp32).  The wide entry-level tensors -- h2 = fc_2(G) [R, 8 dqk + D], v = fc_v(x)
    [R, 10 dv], att [R, 10 dv] and their gradients datt, dh2, dv, ck -- live in bf16; every product accumulates in fp32
    (library bf16 GEMMs, hotk_attn22_forward / _backward with storage = bf16, fp32 segment sums), node-level tensors,
    parameters, the layer input and output stay fp32.  Same factored Linear(2,2,'light') as ops.Linear22Light
    (models/sparse/linear.py:143-161) and the same core as ops.AttnCore22 (kernelattn.py:119-159)."""

    @staticmethod
    def forward(ctx, x, h1, W2, b_off, b_diag, Wv, bv, Wo, bo, proj, plan, drop_p=0., mu_p=0.):
        _lib.require_device(x)
        _lib.require_unsharded(plan, 'the bf16 storage mode')
        bf = torch.bfloat16
        dev = x.device
        R, Nn = plan.R, plan.Nn
        D = x.size(-1)
        C = W2.size(1)
        assert C == 256 + D and W2.size(0) == 5 * D and Wv.shape == (320, D) and Wo.shape == (D, 320)
        assert D % 4 == 0, 'bf16 rows are moved four channels at a time'
        xs = x.reshape(R, D).contiguous()
        padded = plan.Ev != plan.R
        # fc_2 = Linear(2, 2, 'light') in factored form: node blocks per node in fp32, entry blocks as ONE bf16 GEMM
        xd = gather_rows(xs, plan.diag_pos, Nn)
        Wn = torch.cat([W2[0:D], W2[3 * D:4 * D], W2[4 * D:5 * D]], dim=1)
        yn = xd @ Wn
        x2 = torch.empty(R, 2 * D, dtype=torch.float32, device=dev)
        call('hotk_basis22_pair_fwd', 4 * R * 4 * D, p(xs), D, p(plan.kind), p(plan.tpos), p(x2), 2 * D, R, D)
        x2b = x2.to(bf)          # saved for the weight gradient (half the bytes of x2)
        if thin_ok(R, 2 * D, C) and Nn * 3 * C < (1 << 32):
            # entry GEMM (fp32 operands, split-bf16 MFMAs) with the assembly as its epilogue, rounded to bf16 at the store
            h2 = torch.empty(R, C, dtype=bf, device=dev)
            w2 = W2[D:3 * D].contiguous()
            call('hotk_linear22_entry_fwd_bf16', R * (4 * 2 * D + 2 * C), p(x2), 2 * D, p(w2), 2 * D, C,
                 p(yn), 3 * C, p(plan.kind), p(plan.rown), p(plan.coln), p(b_off), p(b_diag), p(h2), C, R, Nn)
        else:
            h2 = x2b @ W2[D:3 * D].to(bf)
            call('hotk_assemble22_bf16', 4 * R * C, p(h2), C, p(yn), 3 * C, p(plan.kind), p(plan.rown), p(plan.coln),
                 p(b_off), p(b_diag), R, C)
        del x2
        xb = xs.to(bf)
        if thin_wide_ok(R, D, 320):
            # fp32 operands and accumulation, rounded to bf16 at the store (the library's bf16 GEMM wrote this tensor at
            # ~2 TB/s: 3.2 ms against 1.6 ms)
            v = thin_gemm_bf16(xs, Wv.t(), bv)
        else:
            v = torch.addmm(bv.to(bf), xb, Wv.t().to(bf))
        # the core
        proj = proj.contiguous().float()
        h1n = h1.reshape(Nn, 384).contiguous()
        att = (torch.zeros if padded else torch.empty)(R, 320, dtype=bf, device=dev)
        kvg = torch.empty(plan.B, 4, 25, 64, dtype=torch.float32, device=dev)
        a = AttnCore22._args(plan, h1n, h2, v, proj, D)
        a.storage = _lib.STORAGE_BF16
        phik = AttnCore22._phik(plan, h1n, proj)
        a.phik, a.ld_phik = phik.data_ptr(), 512
        a.att, a.ld_att, a.kvg = att.data_ptr(), 320, kvg.data_ptr()
        ws, nws = AttnCore22._workspace(plan, dev)
        call('hotk_attn22_forward', plan.Ev * (2 * (8 * 32 + 10 * 32 + 10 * 32)) + 4 * Nn * 384, byref(a), p(ws), nws)
        mu, keep = AttnCore22._drop_masks(plan, dev, drop_p, mu_p)
        AttnCore22._drop_apply(att, plan, mu, keep, drop_p)
        # fc_o + bias + skip path
        if thin_in_ok(R, 320, D):
            y = thin_gemm_bf16in(att, Wo.t(), bo)       # bf16 rows in, fp32 out, bias in the epilogue
        else:
            y = (att @ Wo.t().to(bf)).float()
            y += bo
        y += h2[:, 256:]
        if padded:
            y = mask_rows(y, plan.kind, KEEP_VALID, 0.0)
        ctx.save_for_backward(xb, xd, x2b, h1n, h2, v, att, kvg, phik, proj, W2, Wv, Wo)
        ctx.plan, ctx.drop, ctx.shapes = plan, (mu, keep, drop_p), (x.shape, h1.shape)
        ctx.has_b = (b_off is not None, b_diag is not None)
        return y.view(*x.shape[:-1], D)

    @staticmethod
    def backward(ctx, d_out):
        xb, xd, x2b, h1n, h2, v, att, kvg, phik, proj, W2, Wv, Wo = ctx.saved_tensors
        plan = ctx.plan
        bf = torch.bfloat16
        dev = h2.device
        R, Nn = plan.R, plan.Nn
        D, C = Wo.size(0), W2.size(1)
        padded = plan.Ev != plan.R
        d_out = d_out.contiguous().view(R, D)
        dy = mask_rows(d_out, plan.kind, KEEP_VALID, 0.0) if padded else d_out
        dyb = dy.to(bf)
        datt = thin_gemm_bf16(dy, Wo) if thin_wide_ok(R, D, 320) else dyb @ Wo.to(bf)
        dWo = wgrad(dyb, att)
        dbo = dy.sum(0)
        mu, keep, drop_p = ctx.drop
        AttnCore22._drop_apply(datt, plan, mu, keep, drop_p)
        dh1 = torch.empty(Nn, 384, dtype=torch.float32, device=dev)
        dh2 = torch.empty(R, C, dtype=bf, device=dev)
        dv = (torch.zeros if padded else torch.empty)(R, 320, dtype=bf, device=dev)
        ck = torch.empty(R, 192, dtype=bf, device=dev)
        dkvg = torch.empty_like(kvg)
        a = AttnCore22._args(plan, h1n, h2, v, proj, D)
        a.storage = _lib.STORAGE_BF16
        a.phik, a.ld_phik, a.kvg = phik.data_ptr(), 512, kvg.data_ptr()
        a.datt, a.ld_datt = datt.data_ptr(), 320
        a.dout, a.ld_dout = dy.data_ptr(), D
        a.dh1, a.ld_dh1 = dh1.data_ptr(), 384
        a.dh2, a.ld_dh2 = dh2.data_ptr(), C
        a.dv, a.ld_dv = dv.data_ptr(), 320
        a.ck, a.dkvg = ck.data_ptr(), dkvg.data_ptr()
        ws, nws = AttnCore22._workspace(plan, dev)
        call('hotk_attn22_backward', plan.Ev * (2 * (8 * 32 * 2 + 10 * 32 * 3) + 25) + 4 * Nn * 384 * 2, byref(a), p(ws), nws)
        seg_sum_rows_bf16(ck, plan.row_seg(), None, KEEP_VALID, out=dh1[:, 128:320])
        # fc_v backward (bf16 GEMMs, fp32 sums)
        if linear_bwd_ok(R, D, 320):
            # dx, dW and db from ONE pass over dv (bf16 in, fp32 out) instead of two library GEMMs and a column sum
            dxv32, dWv_t, dbv = linear_bwd_fused(dv, xb, Wv.t())
            dWv = dWv_t.t()
        else:
            dxv32 = (dv @ Wv.to(bf)).float()                                 # [R, D]
            dWv = wgrad(dv, xb)                                              # [320, D]
            dbv = dv.sum(0, dtype=torch.float32)
        # fc_2 backward: g = dh2; padding rows of g were never written
        g = dh2
        if padded:
            g = torch.where((plan.kind != 0).unsqueeze(1), g, torch.zeros((), dtype=bf, device=dev))
        dyn = torch.empty(Nn, 3 * C, dtype=torch.float32, device=dev)
        dW2e = None
        if (_L22_REDUCE and R >= THIN_MIN_ROWS and type(plan).__name__ == 'Plan' and R < (1 << 31) and x2b.is_contiguous()
                and lib().hotk_linear22_bwd_reduce_supported(2 * D, C, int(plan.Nmax)) == 1):
            # diagonal gather, row / column sums and the entry-level weight gradient in one pass over dh2 (bf16 in, fp32 sums)
            n = ctypes_long()
            check(lib().hotk_linear22_bwd_reduce_partials(plan.B, byref(n)), 'hotk_linear22_bwd_reduce_partials')
            part = torch.empty(n.value, 2 * D, C, dtype=torch.float32, device=dev)
            call('hotk_linear22_bwd_reduce_bf16', 2 * plan.Ev * (C + 2 * D) + 12 * Nn * C, p(dh2), C, p(x2b), 2 * D, 2 * D, C,
                 p(plan.kind), p(plan.rown), p(plan.coln), p(plan.g_ebeg), p(plan.g_eend), p(plan.n_nodes_dev), plan.B,
                 plan.Nmax, R, p(dyn), 3 * C, p(part))
            dW2e = part[0] if n.value == 1 else part.sum(0)
        else:
            node_ok = (plan.nkind == 1).unsqueeze(1)
            dyn[:, :C] = torch.where(node_ok, g.index_select(0, plan.diag_pos.clamp_min(0).long()).float(),
                                     torch.zeros((), dtype=torch.float32, device=dev))
            seg_sum_rows_bf16(g, plan.row_seg(), plan.kind, KEEP_OFFDIAG, out=dyn[:, C:2 * C])
            seg_sum_rows_bf16(g, plan.col_seg(), plan.kind, KEEP_OFFDIAG, out=dyn[:, 2 * C:])
        Wn = torch.cat([W2[0:D], W2[3 * D:4 * D], W2[4 * D:5 * D]], dim=1)
        if linear_bwd_ok(Nn, D, 3 * C):
            # node level (fp32 in this mode too): data gradient, weight gradient and both bias gradients from one pass
            dxd, dWn, dbn = linear_bwd_fused(dyn, xd, Wn)
            db_diag = dbn[0:C] if ctx.has_b[1] else None
            db_off = dbn[C:2 * C] if ctx.has_b[0] else None
        else:
            db_diag = dyn[:, 0:C].sum(0) if ctx.has_b[1] else None
            db_off = dyn[:, C:2 * C].sum(0) if ctx.has_b[0] else None
            dxd = dyn @ Wn.t()
            dWn = wgrad(xd, dyn)
        dx2 = thin_gemm_bf16in(g, W2[D:3 * D].t()) if thin_in_ok(R, C, 2 * D) else (g @ W2[D:3 * D].t().to(bf)).float()
        dx = torch.empty(R, D, dtype=torch.float32, device=dev)
        call('hotk_basis22_pair_bwd', 4 * R * 4 * D, p(dx2), 2 * D, p(plan.kind), p(plan.tpos), p(plan.rown), p(dxd), D,
             p(dxv32), D, p(dx), D, R, D)
        del dxv32
        dW2 = torch.cat([dWn[:, 0:C], dW2e if dW2e is not None else wgrad(x2b, g), dWn[:, C:2 * C], dWn[:, 2 * C:]], dim=0)
        xshape, h1shape = ctx.shapes
        return (dx.view(xshape), dh1.view(h1shape), dW2, db_off, db_diag, dWv, dbv, dWo, dbo, None, None, None, None)


def c_void_p_of(t, col):
    from ctypes import c_void_p
    return c_void_p(t.data_ptr() + 4 * col)
