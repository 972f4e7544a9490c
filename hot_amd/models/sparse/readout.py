"""Graph-level pools over a Batch (hot_pytorch/models/sparse/readout.py:34-75), via the segment-sum kernel."""
import torch
import torch.nn as nn

from ... import ops, _lib
from ..._lib import KEEP_VALID, KEEP_DIAG, KEEP_OFFDIAG
from ...batch.sparse import Batch as B
from .basis import plan_entry_graph, plan_node_graph


def _graph_sum(G, keep):
    plan = G.plan
    if G.order == 2:
        _lib.require_unsharded(plan, 'a graph-level pool over entries')
    kind = plan.nkind if G.order == 1 else plan.kind
    rg = plan_node_graph(plan) if G.order == 1 else plan_entry_graph(plan)
    return ops.SegSumRows.apply(G.values, plan.graph_seg(G.order), kind, keep, rg)


class SumPool(nn.Module):
    """sum over the valid rows of each graph -> [B, D] (readout.py:34-45)"""

    def __init__(self, order):
        super().__init__()
        assert order in (1, 2), 'unsupported input order'
        self.order = order

    def forward(self, G: B) -> torch.Tensor:
        assert G.order == self.order
        return _graph_sum(G, KEEP_VALID)


class AvgPool(nn.Module):
    """order 1: sum / (n + 1e-5); order 2: diagonal sum / (n + 1e-5) + off-diagonal sum / (e + 1e-5)
    (readout.py:48-75)"""

    def __init__(self, order):
        super().__init__()
        assert order in (1, 2), 'unsupported input order'
        self.order = order

    def forward(self, G: B) -> torch.Tensor:
        assert G.order == self.order
        dev = G.values.device
        from .basis import _counts
        n = _counts(G.n_nodes, torch.long, dev)
        n_inv = (n + 1e-5).pow(-1).unsqueeze(1)
        if self.order == 1:
            return n_inv * _graph_sum(G, KEEP_VALID)
        e_inv = (_counts(G.n_edges, torch.long, dev) - n + 1e-5).pow(-1).unsqueeze(1)
        return _graph_sum(G, KEEP_DIAG) * n_inv + _graph_sum(G, KEEP_OFFDIAG) * e_inv
