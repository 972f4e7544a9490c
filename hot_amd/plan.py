"""Segment plan of a padded order-2 batch: everything index-shaped that the kernels need, built once per
minibatch on the device by hotk_plan_build (K1) and shared by every Batch that re-wraps the same indices.

Replaces get_transpose_info (hot_pytorch/utils/sparse.py:102-127) and the per-call index arithmetic of
models/sparse/kernelattncoef.py:105-122,145-157 (node offsets, boolean-mask compaction, coalesce sort).
"""
from ctypes import byref, c_size_t

import numpy as np
import torch

from . import _lib
from ._lib import lib, p, check, stream_ptr


class Seg(tuple):
    """A segment list (beg, end, perm, maxlen) that remembers the plan it is a view of and its name there, so that
    derived index lists (hot_amd.ops.refine_seg) are built once per plan and kept ON the plan -- no pointer-keyed cache,
    nothing that outlives the plan or goes stale when a buffer is reused."""

    def __new__(cls, beg, end, perm, maxlen, plan=None, name=None):
        self = super().__new__(cls, (beg, end, perm, maxlen))
        self.plan, self.name = plan, name
        return self


class Plan:
    """int32 device arrays over padded rows: R = B*Emax entry rows, Nn = B*Nmax node rows."""

    def __init__(self, indices: torch.Tensor, n_nodes, n_entries):
        _lib.require_device(indices, 'indices')
        assert indices.dtype == torch.long and indices.dim() == 3 and indices.size(-1) == 2
        dev = indices.device
        indices = indices.contiguous()
        B, Emax = indices.size(0), indices.size(1)
        n_nodes = [int(x) for x in n_nodes]
        n_entries = [int(x) for x in n_entries]
        assert len(n_nodes) == B and len(n_entries) == B
        Nmax = max(n_nodes) if B else 0
        assert max(n_entries, default=0) <= Emax
        self.B, self.Emax, self.Nmax = B, Emax, Nmax
        self.R, self.Nn = B * Emax, B * Nmax
        self.Ev = int(sum(n_entries))
        nn = np.asarray(n_nodes, dtype=np.int32)
        ne = np.asarray(n_entries, dtype=np.int32)
        eofs = np.concatenate([[0], np.cumsum(ne[:-1], dtype=np.int64)]).astype(np.int64) if B else np.zeros(0, np.int64)
        i32 = dict(dtype=torch.int32, device=dev)
        self.n_nodes_dev = torch.from_numpy(nn).to(dev)
        self.n_entries_dev = torch.from_numpy(ne).to(dev)
        entry_ofs = torch.from_numpy(eofs).to(dev)
        # graph segments in padded row numbering (padding excluded)
        eb = np.arange(B, dtype=np.int64) * Emax
        nb = np.arange(B, dtype=np.int64) * Nmax
        self.g_ebeg = torch.from_numpy(eb.astype(np.int32)).to(dev)
        self.g_eend = torch.from_numpy((eb + ne).astype(np.int32)).to(dev)
        self.g_nbeg = torch.from_numpy(nb.astype(np.int32)).to(dev)
        self.g_nend = torch.from_numpy((nb + nn).astype(np.int32)).to(dev)
        self.iota = torch.arange(max(self.R, self.Nn) + 1, **i32)
        R, Nn, Ev = self.R, self.Nn, self.Ev
        self.t_indices = torch.empty(B, Emax, dtype=torch.long, device=dev)
        self.t_mask = torch.empty(B, Emax, dtype=torch.bool, device=dev)
        self.rown = torch.empty(R, **i32)
        self.coln = torch.empty(R, **i32)
        self.kind = torch.empty(R, dtype=torch.uint8, device=dev)
        self.tpos = torch.empty(R, **i32)
        self.diag_pos = torch.empty(Nn, **i32)
        self.nkind = torch.empty(Nn, dtype=torch.uint8, device=dev)
        self.col_ptr = torch.empty(Nn + 1, **i32)
        self.row_ptr = torch.empty(Nn + 1, **i32)
        self.col_perm = torch.empty(max(Ev, 1), **i32)
        self.row_perm = torch.empty(max(Ev, 1), **i32)
        status = torch.empty(4, **i32)
        need = c_size_t(0)
        check(lib().hotk_plan_workspace_bytes(B, Emax, Nmax, Ev, byref(need)), 'hotk_plan_workspace_bytes')
        ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=dev)
        check(lib().hotk_plan_build(p(indices), p(self.n_nodes_dev), p(self.n_entries_dev), p(entry_ofs), B, Emax, Nmax,
                                    Ev, p(self.t_indices), p(self.t_mask), p(self.rown), p(self.coln), p(self.kind),
                                    p(self.tpos), p(self.diag_pos), p(self.nkind), p(self.col_ptr), p(self.col_perm),
                                    p(self.row_ptr), p(self.row_perm), p(status), p(ws), need.value, stream_ptr()),
              'hotk_plan_build')
        st = status.cpu().tolist()   # one host sync per minibatch: sizes of the long-segment split + validation
        del ws
        if st[1]:
            raise ValueError('sparse Batch: an index of a valid entry is outside [0, n_nodes) of its graph')
        if st[0]:
            raise ValueError('sparse Batch: layout invariant broken - each graph must list its n diagonal entries '
                             '(v,v) first and in order, followed by off-diagonal entries only '
                             '(hot_pytorch/batch/sparse.py:38)')
        self.max_col_deg, self.max_row_deg = int(st[2]), int(st[3])
        # node row of a diagonal entry (-1 elsewhere): inverse of diag_pos, derived on the device
        self.diag_node = torch.where(self.kind == 1, self.rown, torch.full_like(self.rown, -1))
        self.device = dev

    def records(self):
        """Packed per-position records for the fused kernels (one 16-B load per column-segment entry instead of
        three dependent ones): col_rec [Ev,4] = {entry, row node, transposed entry, kind}; row_rec [Ev] = entry
        (off-diagonal) or ~entry (diagonal).  Index bookkeeping, built once per plan."""
        if not hasattr(self, '_records'):
            cp = self.col_perm[:self.Ev].long()
            col_rec = torch.stack([self.col_perm[:self.Ev], self.rown[cp], self.tpos[cp], self.kind[cp].to(torch.int32)],
                                  dim=1).contiguous()
            rp = self.row_perm[:self.Ev]
            row_rec = torch.where(self.kind[rp.long()] == 2, rp, ~rp).contiguous()
            if self.Ev == 0:
                col_rec = torch.zeros(1, 4, dtype=torch.int32, device=self.device)
                row_rec = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._records = (col_rec, row_rec)
        return self._records

    # CSR views ---------------------------------------------------------------------------------
    def col_seg(self):
        return Seg(self.col_ptr[:-1], self.col_ptr[1:], self.col_perm, self.max_col_deg, self, 'col')

    def row_seg(self):
        return Seg(self.row_ptr[:-1], self.row_ptr[1:], self.row_perm, self.max_row_deg, self, 'row')

    def graph_seg(self, order):
        if order == 1:
            return Seg(self.g_nbeg, self.g_nend, None, self.Nmax, self, 'graph1')
        return Seg(self.g_ebeg, self.g_eend, None, self.Emax, self, 'graph2')

    def unit_seg(self, n):
        return self.iota[:n], self.iota[1:n + 1], None, 1


def node_plan(n_nodes, device, Nmax=None):
    """Order-1 batches built without an order-2 parent only need node kinds and graph ranges."""
    return NodePlan(n_nodes, device, Nmax)


class NodePlan:
    def __init__(self, n_nodes, device, Nmax=None):
        n_nodes = [int(x) for x in n_nodes]
        B = len(n_nodes)
        Nmax = (max(n_nodes) if B else 0) if Nmax is None else Nmax
        self.B, self.Nmax, self.Nn = B, Nmax, B * Nmax
        self.Emax, self.R = 0, 0
        nn = np.asarray(n_nodes, dtype=np.int64)
        nb = np.arange(B, dtype=np.int64) * Nmax
        self.g_nbeg = torch.from_numpy(nb.astype(np.int32)).to(device)
        self.g_nend = torch.from_numpy((nb + nn).astype(np.int32)).to(device)
        valid = (np.arange(Nmax)[None, :] < nn[:, None]).reshape(-1).astype(np.uint8)
        self.nkind = torch.from_numpy(valid).to(device)
        self.iota = torch.arange(self.Nn + 1, dtype=torch.int32, device=device)
        self.device = device

    def graph_seg(self, order):
        assert order == 1
        return Seg(self.g_nbeg, self.g_nend, None, self.Nmax, self, 'graph1')

    def unit_seg(self, n):
        return self.iota[:n], self.iota[1:n + 1], None, 1


class DensePlan:
    """Segment plan of a DENSE order-2 batch [B, N, N, D] (hot_pytorch/batch/dense.py:9-58) read as rows
    r = (b*N + i)*N + j: the same fields as Plan (kinds, node rows, transposed row, CSR views by row and by column
    node, graph ranges), so the plan-driven kernels run on the dense layout unchanged.  Every (i, j) with
    i, j < n_b is a valid entry (diagonal where i == j); the rest of the N x N block is padding.  Pure index
    bookkeeping (torch ops on `device`), built once per (n_nodes, N) and shared by every re-wrapped batch.  The
    kernels that rely on the sparse layout invariant (the fused 2->2 core) do not take this plan."""

    def __init__(self, n_nodes, N, device):
        n_nodes = [int(x) for x in n_nodes]
        B = len(n_nodes)
        assert B == 0 or max(n_nodes) <= N
        self.B, self.Nmax, self.Emax = B, N, N * N
        self.R, self.Nn = B * N * N, B * N
        assert self.R < 2 ** 31
        dev = torch.device(device)
        i32 = dict(dtype=torch.int32, device=dev)
        nn = torch.as_tensor(n_nodes, dtype=torch.long, device=dev)
        ar = torch.arange(N, device=dev)
        vn = ar[None, :] < nn[:, None]                                   # [B, N] valid nodes
        valid = vn[:, :, None] & vn[:, None, :]                          # [B, N, N]
        eye = torch.eye(N, dtype=torch.bool, device=dev)[None]
        kind = torch.where(valid, torch.where(eye, 1, 2), 0)
        self.kind = kind.reshape(-1).to(torch.uint8)
        self.nkind = vn.reshape(-1).to(torch.uint8)
        base_n = (torch.arange(B, device=dev) * N)[:, None, None]
        base_e = (torch.arange(B, device=dev) * N * N)[:, None, None]
        I, J = ar[None, :, None].expand(B, N, N), ar[None, None, :].expand(B, N, N)
        neg = torch.full((B, N, N), -1, device=dev)
        self.rown = torch.where(valid, base_n + I, neg).reshape(-1).to(torch.int32)
        self.coln = torch.where(valid, base_n + J, neg).reshape(-1).to(torch.int32)
        self.tpos = torch.where(valid, base_e + J * N + I, neg).reshape(-1).to(torch.int32)
        dpos = (torch.arange(B, device=dev) * N * N)[:, None] + ar[None, :] * (N + 1)
        self.diag_pos = torch.where(vn, dpos, torch.full_like(dpos, -1)).reshape(-1).to(torch.int32)
        self.diag_node = torch.where(self.kind == 1, self.rown, torch.full_like(self.rown, -1))
        # CSR views: entries of row node (b, i) are the rows (b, i, j < n_b); of column node (b, j) the rows (b, i < n_b, j)
        cnt = torch.where(vn, nn[:, None].expand(B, N), torch.zeros_like(vn, dtype=torch.long)).reshape(-1)
        ptr = torch.zeros(self.Nn + 1, dtype=torch.long, device=dev)
        ptr[1:] = torch.cumsum(cnt, 0)
        self.row_ptr = ptr.to(torch.int32)
        self.col_ptr = self.row_ptr.clone()
        rows = torch.arange(self.R, device=dev).view(B, N, N)
        self.row_perm = rows[valid].to(torch.int32)                                       # row-major: grouped by i
        self.col_perm = rows.transpose(1, 2)[valid.transpose(1, 2)].to(torch.int32)       # grouped by j
        self.Ev = int(self.row_perm.numel())
        if self.Ev == 0:
            self.row_perm = torch.zeros(1, **i32)
            self.col_perm = torch.zeros(1, **i32)
        self.max_row_deg = self.max_col_deg = max(n_nodes) if B else 0
        eb = torch.arange(B, device=dev) * N * N
        nb = torch.arange(B, device=dev) * N
        self.g_ebeg, self.g_eend = eb.to(torch.int32), (eb + N * N).to(torch.int32)      # padding inside: masked by kind
        self.g_nbeg, self.g_nend = nb.to(torch.int32), (nb + nn).to(torch.int32)
        self.n_nodes_dev = nn.to(torch.int32)
        self.n_entries_dev = (nn * nn).to(torch.int32)
        self.iota = torch.arange(max(self.R, self.Nn) + 1, **i32)
        self.device = dev

    col_seg = Plan.col_seg
    row_seg = Plan.row_seg
    graph_seg = Plan.graph_seg
    unit_seg = Plan.unit_seg
