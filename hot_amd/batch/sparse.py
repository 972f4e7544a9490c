"""Sparse Batch container: same fields, constructor and helper functions as
hot_pytorch/batch/sparse.py:10-190, backed by the device segment plan (hot_amd/plan.py)."""
from typing import Union, Callable

import numpy as np
import torch

from ..utils.sparse import to_diag, get_diag, get_nondiag, do_transpose
from ..utils.set import get_mask, masked_fill
from ..plan import Plan, NodePlan


def _size_mask(sizes, device):
    return get_mask(torch.as_tensor(sizes, dtype=torch.long, device=device))


class Batch(object):
    """A minibatch of graphs as padded entry rows.  Fields (names fixed by the reference interface):
    indices [B,|E|,2] int64 or None (order 1), values [B,|E|,D], n_nodes / n_edges python lists (n_edges counts diagonal
    AND off-diagonal entries), mask [B,|E|], node_mask [B,N], order, t_indices / t_mask (transpose map), node_ofs, device."""

    def __init__(self, indices: Union[None, torch.LongTensor], values: torch.Tensor,
                 n_nodes: list, n_edges: Union[None, list], mask: torch.BoolTensor = None, skip_masking: bool = False,
                 t_indices=None, t_mask=None, node_mask: torch.BoolTensor = None, node_ofs: torch.LongTensor = None):
        dev = values.device
        order = 1 if indices is None else indices.size(-1)
        if order not in (1, 2):
            raise AssertionError(f'order-{order} batches are not supported')
        self.indices, self.values, self.n_nodes, self.n_edges = indices, values, n_nodes, n_edges
        self.device, self.order = dev, order
        self.node_mask = node_mask if node_mask is not None else _size_mask(n_nodes, dev)
        if order == 1:
            self.mask = self.node_mask
        else:
            self.mask = mask if mask is not None else _size_mask(n_edges, dev)
        self._plan = None
        # a batch whose graphs all fill the padded width has no invalid rows: the zero-filling pass (a full copy of the
        # values) is skipped for it and for every batch derived from it (batch_like copies the flag)
        # (only when the mask was derived from the sizes here: a mask handed in by the caller is taken as it is)
        sizes = n_nodes if order == 1 else n_edges
        own_mask = (node_mask is None) if order == 1 else (mask is None)
        self._padded = (not own_mask) or sizes is None or any(int(s_) != self.mask.size(1) for s_ in sizes)
        # zero-filling of invalid rows happens on the device (HIP kernel); a host-side batch remembers that it is due
        self._pending_mask = False
        if not skip_masking:
            self._mask_now_or_later()
        have_map = not (t_indices is None and t_mask is None)
        self.t_indices, self.t_mask = (t_indices, t_mask) if (have_map or order == 1) else self._transpose_map()
        if node_ofs is None:
            starts = np.cumsum([0] + list(n_nodes[:-1]))
            node_ofs = torch.tensor(starts, dtype=torch.long, device=dev)
        self.node_ofs = node_ofs

    def _mask_now_or_later(self):
        if self.values.is_cuda:
            if getattr(self, '_padded', True):
                self.apply_mask(0)
        else:
            self._pending_mask = True

    def _transpose_map(self):
        """(t_indices, t_mask) from kernel K1 once the batch lives on the device; (None, None) until then"""
        if not self.values.is_cuda:
            return None, None
        return self.plan.t_indices, self.plan.t_mask

    def __repr__(self):
        i = None if self.indices is None else list(self.indices.size())
        return f"Batch(indices {i}, values {list(self.values.size())}"

    @property
    def plan(self):
        """Device segment plan (built on first use; shared through batch_like / d / v2d)."""
        if self._plan is None:
            if self.order == 2:
                self._plan = Plan(self.indices, self.n_nodes, self.n_edges)
            else:
                self._plan = NodePlan(self.n_nodes, self.values.device, self.node_mask.size(1))
        return self._plan

    def to(self, device: Union[str, torch.device]) -> 'Batch':
        """move in place and return self (the reference's behaviour); the plan is rebuilt on the new device"""
        for name in ('indices', 't_indices', 't_mask', 'values', 'mask', 'node_mask', 'node_ofs'):
            held = getattr(self, name)
            if held is not None:
                setattr(self, name, held.to(device))
        self.device = self.values.device
        if self._plan is not None and self._plan.device != self.device:
            self._plan = None
        if self.values.is_cuda:
            if self._pending_mask:
                self._pending_mask = False
                self.apply_mask(0)
            if self.order == 2 and self.t_indices is None:
                self.t_indices, self.t_mask = self._transpose_map()
        return self

    def apply_mask(self, value=0.) -> None:
        """overwrite the rows outside `mask` (the index tensors are taken to be clean already)"""
        self.values = masked_fill(self.values, self.mask, value)


def _carry(G: Batch, H: Batch) -> Batch:
    H._plan = G._plan
    return H


def batch_like(G: Batch, values: torch.Tensor, skip_masking=False) -> Batch:
    """G with other values: shares every index / mask tensor and the plan, builds nothing"""
    H = Batch.__new__(Batch)
    H.__dict__.update(G.__dict__)
    H.values, H.device, H._pending_mask = values, values.device, False
    if not skip_masking:
        H._mask_now_or_later()
    return H


def _order2(G: Batch) -> Batch:
    if G.order != 2:
        raise AssertionError('an order-2 batch is expected')
    return G


def t(G: Batch) -> Batch:
    """values of the transposed entries: row (i,j) receives the row of (j,i), zero where (j,i) is absent"""
    _order2(G)
    return batch_like(G, do_transpose(G.values, G.t_indices, G.t_mask, G.plan), skip_masking=True)


def nd(G: Batch) -> Batch:
    """off-diagonal part (diagonal rows zeroed)"""
    _order2(G)
    return batch_like(G, get_nondiag(G.plan, G.values), skip_masking=True)


def d(G: Batch) -> Batch:
    """diagonal rows as an order-1 batch [B, N, D]"""
    _order2(G)
    H = Batch(None, get_diag(G.plan, G.values), G.n_nodes, None, G.node_mask, True, G.t_indices, G.t_mask, G.node_mask,
              G.node_ofs)
    return _carry(G, H)


def v2d(G: Batch, values: Batch) -> Batch:
    """order-1 values placed on the diagonal rows of G's index set, zero elsewhere"""
    return batch_like(G, to_diag(G.plan, values.values), skip_masking=True)


def apply(G: Union[torch.Tensor, Batch], f: Callable[[torch.Tensor], torch.Tensor], skip_masking=False) -> Union[torch.Tensor, Batch]:
    """f on the values of a batch (or on a plain tensor)"""
    return batch_like(G, f(G.values), skip_masking) if isinstance(G, Batch) else f(G)


def add_batch(G1: Union[torch.Tensor, Batch], G2: Union[torch.Tensor, Batch]) -> Union[torch.Tensor, Batch]:
    """sum of two batches over the SAME index structure (or of two equally shaped tensors)"""
    pair = (isinstance(G1, Batch), isinstance(G2, Batch))
    if pair == (True, True):
        same = G1.order == G2.order and G1.n_nodes == G2.n_nodes and G1.n_edges == G2.n_edges
        assert same, 'add_batch: the two batches do not share their structure'
        return batch_like(G1, G1.values + G2.values, skip_masking=True)
    assert pair == (False, False) and G1.size() == G2.size(), 'add_batch: two batches or two equally shaped tensors'
    return G1 + G2


def make_batch_concatenated(node_feature: torch.Tensor, edge_index: torch.LongTensor, edge_feature: torch.Tensor,
                            n_nodes: list, n_edges: list) -> Batch:
    """PyG-style concatenated graphs -> padded Batch (batch/sparse.py:134-176): per graph the n diagonal rows
    (v,v) holding [node_feat | 0], then the e edge rows holding [0 | edge_feat], then zero padding.
    Pure data placement (index bookkeeping + row copies); runs on whatever device the inputs live on."""
    assert len(node_feature.size()) == len(edge_index.size()) == len(edge_feature.size()) == 2
    B = len(n_nodes)
    dev, dt = node_feature.device, node_feature.dtype
    dn, de = node_feature.size(-1), edge_feature.size(-1)
    nn = torch.as_tensor(n_nodes, dtype=torch.long, device=dev)
    ne = torch.as_tensor(n_edges, dtype=torch.long, device=dev)
    sizes = [int(a) + int(b) for a, b in zip(n_nodes, n_edges)]
    Emax = max(sizes)
    row0 = torch.arange(B, device=dev) * Emax                                 # first padded row of each graph
    g_of_node = torch.repeat_interleave(torch.arange(B, device=dev), nn)
    g_of_edge = torch.repeat_interleave(torch.arange(B, device=dev), ne)
    local_node = torch.arange(int(nn.sum()), device=dev) - (torch.cumsum(nn, 0) - nn)[g_of_node]
    local_edge = torch.arange(int(ne.sum()), device=dev) - (torch.cumsum(ne, 0) - ne)[g_of_edge]
    node_rows = row0[g_of_node] + local_node
    edge_rows = row0[g_of_edge] + nn[g_of_edge] + local_edge
    indices = torch.zeros(B * Emax, 2, dtype=torch.long, device=dev)
    values = torch.zeros(B * Emax, dn + de, dtype=dt, device=dev)
    indices[node_rows] = local_node[:, None].expand(-1, 2)
    indices[edge_rows] = edge_index.t()
    values[node_rows, :dn] = node_feature
    values[edge_rows, dn:] = edge_feature
    return Batch(indices.view(B, Emax, 2), values.view(B, Emax, dn + de), n_nodes, sizes)


def make_batch(node_features: list, edge_indices: list, edge_features: list) -> Batch:
    """per-graph lists -> Batch (batch/sparse.py:179-190): concatenate and defer to make_batch_concatenated"""
    counts = lambda xs: [x.size(0) for x in xs]
    return make_batch_concatenated(torch.cat(node_features), torch.cat(edge_indices, dim=1), torch.cat(edge_features),
                                   counts(node_features), counts(edge_features))
