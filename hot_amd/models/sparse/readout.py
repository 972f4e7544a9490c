"""Graph-level pools over a Batch (hot_pytorch/models/sparse/readout.py:34-75), via the segment-sum kernel."""
import torch
import torch.nn as nn

from ... import ops, _lib
from ..._lib import KEEP_VALID, KEEP_DIAG, KEEP_OFFDIAG
from ...batch.sparse import Batch as B
from .basis import plan_entry_graph, plan_node_graph


def _graph_sum(G, keep):
    """sum over the kept rows of each graph.  Edge-shard mode: nodes and diagonal entries are replicated on every rank
    (complete locally), off-diagonal entries are spread over the ranks (their sum is all-reduced)."""
    plan = G.plan
    kind = plan.nkind if G.order == 1 else plan.kind
    rg = plan_node_graph(plan) if G.order == 1 else plan_entry_graph(plan)
    seg = plan.graph_seg(G.order)
    if G.order == 1 or keep != KEEP_VALID or ops.shard_of(plan) is None:
        x = ops.SegSumRows.apply(G.values, seg, kind, keep, rg)
        return ops.entry_sum(x, plan) if (G.order == 2 and keep == KEEP_OFFDIAG) else x
    return (ops.SegSumRows.apply(G.values, seg, kind, KEEP_DIAG, rg) +
            ops.entry_sum(ops.SegSumRows.apply(G.values, seg, kind, KEEP_OFFDIAG, rg), plan))


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
        from .basis import offdiag_count
        e_inv = (offdiag_count(G.plan, G.n_nodes, G.n_edges, torch.float32, dev) + 1e-5).pow(-1)
        return _graph_sum(G, KEEP_DIAG) * n_inv + _graph_sum(G, KEEP_OFFDIAG) * e_inv
