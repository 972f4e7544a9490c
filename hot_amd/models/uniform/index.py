"""Device index plan for the 1 -> k-uniform extension: a set of N nodes (one segment) and B hyperedges given as
`indices [B, k] int64`.  Built once per `indices` tensor and cached on it, like the order-2 plan of a Batch.

  idx32[j]     int32 [B]   column j of `indices` (row gather index)
  seg[j]       CSR of the hyperedges that have node n in slot j: (begin [N], end [N], perm [B], max length), from
               the library's stable LSD radix sort (hotk_sort_pairs_u32) -- the backward of the k gathers is a
               deterministic segment sum instead of the reference's `gather(sparse_grad=True)` scatter-add
               (models/uniform/linear.py:33-37)
  node_seg     the single segment [0, N) (global key-value sum, pooled term);  edge_seg  the single segment [0, B)
"""
from ctypes import byref, c_size_t

import torch

from ... import _lib
from ..._lib import p, call


class UniformPlan:
    def __init__(self, n_nodes: int, indices, device):
        self.device = torch.device(device)
        self.N = int(n_nodes)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=self.device)
        self.nkind = torch.ones(self.N, dtype=torch.uint8, device=self.device)       # every node row is valid (kind 1)
        self.g_nbeg, self.g_nend = i32([0]), i32([self.N])
        self.Nmax, self.Nn = max(self.N, 1), self.N
        self.node_row_seg = torch.zeros(self.N, dtype=torch.int32, device=self.device)   # segment id of every node row
        self.iota = torch.arange(max(self.N, 0) + 1, dtype=torch.int32, device=self.device)
        self.B_rows, self.k = 0, 0
        self.Emax, self.R = 1, 0
        self.kind = torch.zeros(0, dtype=torch.uint8, device=self.device)
        self.idx32, self.seg = [], []
        if indices is not None:
            assert indices.dim() == 2 and indices.dtype == torch.int64 and indices.device == self.device
            self.B_rows, self.k = int(indices.size(0)), int(indices.size(1))
            self.Emax, self.R = max(self.B_rows, 1), self.B_rows
            self.kind = torch.full((self.B_rows,), 2, dtype=torch.uint8, device=self.device)   # query rows: all kept
            self.g_ebeg, self.g_eend = i32([0]), i32([self.B_rows])
            if self.B_rows:
                lo, hi = int(indices.min()), int(indices.max())
                if lo < 0 or hi >= self.N:
                    raise IndexError(f'hyperedge indices out of range [0, {self.N}): min {lo}, max {hi}')
            for j in range(self.k):
                col = indices[:, j].to(torch.int32).contiguous()
                self.idx32.append(col)
                self.seg.append(self._csr(col))

    def _csr(self, col):
        n = col.numel()
        keys = col.clone().view(torch.int32)
        vals = torch.arange(n, dtype=torch.int32, device=self.device)
        if n > 1:
            ws_bytes = c_size_t(0)
            _lib.check(_lib.lib().hotk_sort_workspace_bytes(n, byref(ws_bytes)), 'hotk_sort_workspace_bytes')
            ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=self.device)
            bits = max(1, (self.N - 1).bit_length())
            call('hotk_sort_pairs_u32', 16 * n, p(keys), p(vals), n, bits, p(ws), ws_bytes.value)
        # segment boundaries of the sorted keys (tiny: N + 1 binary searches)
        bounds = torch.searchsorted(keys, self.iota[:self.N + 1]).to(torch.int32)
        beg, end = bounds[:-1].contiguous(), bounds[1:].contiguous()
        maxlen = int((end - beg).max()) if self.N else 0
        return beg, end, vals, max(maxlen, 1)

    # interface used by ops.KernelAttn (same as hot_amd.plan.Plan)
    def graph_seg(self, order):
        if order == 1:
            return self.g_nbeg, self.g_nend, None, self.Nmax
        return self.g_ebeg, self.g_eend, None, self.Emax

    def unit_seg(self, n):
        return self.iota[:n], self.iota[1:n + 1], None, 1


def plan_of(x: torch.Tensor, indices) -> UniformPlan:
    """plan for node features x [N, D] and hyperedge indices [B, k] (or None), cached on the indices tensor"""
    N = x.size(0)
    if indices is None:
        return UniformPlan(N, None, x.device)
    # the plan holds copies of the index columns and their sorted CSR: it is valid only for the tensor CONTENT it was
    # built from, so the cache is keyed on the tensor's version counter and storage as well (an in-place refill of the
    # same buffer for the next minibatch rebuilds it)
    stamp = (indices._version, indices.data_ptr(), tuple(indices.shape))
    cached = getattr(indices, '_hotk_uniform_plan', None)
    if cached is None or cached[0] != stamp or cached[1].N != N or cached[1].device != x.device:
        cached = (stamp, UniformPlan(N, indices, x.device))
        try:
            indices._hotk_uniform_plan = cached
        except AttributeError:
            pass
    return cached[1]
