"""Sparse Batch container: same fields, constructor and helper functions as
hot_pytorch/batch/sparse.py:10-190, backed by the device segment plan (hot_amd/plan.py)."""
from typing import Union, Callable

import numpy as np
import torch

from ..utils.sparse import to_diag, get_diag, get_nondiag, do_transpose
from ..utils.set import get_mask, masked_fill
from ..plan import Plan, NodePlan


class Batch(object):
    indices: Union[None, torch.LongTensor]
    values: torch.Tensor
    n_nodes: list
    n_edges: Union[None, list]
    device: torch.device
    mask: torch.BoolTensor
    node_mask: torch.BoolTensor
    order: int
    t_indices: Union[None, torch.LongTensor]
    t_mask: Union[None, torch.BoolTensor]
    node_ofs: torch.LongTensor

    def __init__(self, indices: Union[None, torch.LongTensor], values: torch.Tensor,
                 n_nodes: list, n_edges: Union[None, list], mask: torch.BoolTensor = None, skip_masking: bool = False,
                 t_indices=None, t_mask=None, node_mask: torch.BoolTensor = None, node_ofs: torch.LongTensor = None):
        """a mini-batch of sparse graphs: indices [B,|E|,2] long (or None for order 1), values [B,|E|,D],
        n_nodes / n_edges python lists (n_edges counts diagonal + off-diagonal entries)."""
        self.indices = indices
        self.values = values
        self.n_nodes = n_nodes
        self.n_edges = n_edges
        self.device = values.device
        self.order = 1 if indices is None else indices.size(-1)
        assert self.order in (1, 2)
        self.node_mask = get_mask(torch.tensor(n_nodes, dtype=torch.long, device=self.device)) if node_mask is None else node_mask
        if self.order == 1:
            self.mask = self.node_mask
        else:
            self.mask = get_mask(torch.tensor(n_edges, dtype=torch.long, device=self.device)) if mask is None else mask
        self._plan = None
        self._pending_mask = False
        if not skip_masking:
            if self.values.is_cuda:
                self.apply_mask(0)
            else:
                self._pending_mask = True   # applied by the HIP kernel when the batch reaches the device
        if self.order == 2 and t_indices is None and t_mask is None:
            if self.values.is_cuda:
                plan = self.plan
                self.t_indices, self.t_mask = plan.t_indices, plan.t_mask
            else:
                self.t_indices, self.t_mask = None, None    # built by K1 in .to(device)
        else:
            self.t_indices, self.t_mask = t_indices, t_mask
        self.node_ofs = torch.tensor(np.cumsum([0] + self.n_nodes[:-1]), dtype=torch.long, device=self.device) if node_ofs is None else node_ofs

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
        if self.indices is not None:
            self.indices = self.indices.to(device)
            if self.t_indices is not None:
                self.t_indices = self.t_indices.to(device)
                self.t_mask = self.t_mask.to(device)
        self.values = self.values.to(device)
        self.mask = self.mask.to(device)
        self.node_mask = self.node_mask.to(device)
        self.node_ofs = self.node_ofs.to(device)
        if self._plan is not None and self._plan.device != self.values.device:
            self._plan = None
        self.device = self.values.device
        if self.values.is_cuda:
            if self._pending_mask:
                self.apply_mask(0)
                self._pending_mask = False
            if self.order == 2 and self.t_indices is None:
                self.t_indices, self.t_mask = self.plan.t_indices, self.plan.t_mask
        return self

    def apply_mask(self, value=0.) -> None:
        # mask out invalid tensor elements (indices are assumed well-masked already, as in the reference)
        self.values = masked_fill(self.values, self.mask, value)


def _carry(G: Batch, H: Batch) -> Batch:
    H._plan = G._plan
    return H


def batch_like(G: Batch, values: torch.Tensor, skip_masking=False) -> Batch:
    # only substitute values
    H = Batch.__new__(Batch)
    H.indices, H.values, H.n_nodes, H.n_edges = G.indices, values, G.n_nodes, G.n_edges
    H.device, H.order = values.device, G.order
    H.mask, H.node_mask = G.mask, G.node_mask
    H.t_indices, H.t_mask, H.node_ofs = G.t_indices, G.t_mask, G.node_ofs
    H._plan, H._pending_mask = G._plan, False
    if not skip_masking:
        if values.is_cuda:
            H.apply_mask(0)
        else:
            H._pending_mask = True
    return H


def t(G: Batch) -> Batch:
    # transpose
    assert G.order == 2
    v_t = do_transpose(G.values, G.t_indices, G.t_mask, G.plan)
    return batch_like(G, v_t, skip_masking=True)


def nd(G: Batch) -> Batch:
    # get non-diagonals
    assert G.order == 2
    v_nd = get_nondiag(G.plan, G.values)
    return batch_like(G, v_nd, skip_masking=True)


def d(G: Batch) -> Batch:
    # get diagonals
    assert G.order == 2
    v_d = get_diag(G.plan, G.values)
    H = Batch(None, v_d, G.n_nodes, None, G.node_mask, True, G.t_indices, G.t_mask, G.node_mask, G.node_ofs)
    return _carry(G, H)


def v2d(G: Batch, values: Batch) -> Batch:
    # vectors to diagonal matrices
    v_d = to_diag(G.plan, values.values)
    return batch_like(G, v_d, skip_masking=True)


def apply(G: Union[torch.Tensor, Batch], f: Callable[[torch.Tensor], torch.Tensor], skip_masking=False) -> Union[torch.Tensor, Batch]:
    if isinstance(G, torch.Tensor):
        return f(G)
    return batch_like(G, f(G.values), skip_masking)


def add_batch(G1: Union[torch.Tensor, Batch], G2: Union[torch.Tensor, Batch]) -> Union[torch.Tensor, Batch]:
    # add features of two batched graphs with identical edge structures
    if isinstance(G1, Batch) and isinstance(G2, Batch):
        assert G1.order == G2.order
        assert G1.n_nodes == G2.n_nodes
        assert G1.n_edges == G2.n_edges
        return batch_like(G1, G1.values + G2.values, skip_masking=True)
    else:
        assert isinstance(G1, torch.Tensor) and isinstance(G2, torch.Tensor)
        assert G1.size() == G2.size()
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
    """interface for sparse batch construction (batch/sparse.py:179-190)"""
    node_feature = torch.cat(node_features)
    edge_index = torch.cat(edge_indices, dim=1)
    edge_feature = torch.cat(edge_features)
    n_nodes = [x.size(0) for x in node_features]
    n_edges = [e.size(0) for e in edge_features]
    return make_batch_concatenated(node_feature, edge_index, edge_feature, n_nodes, n_edges)
