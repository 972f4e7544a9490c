"""Pooling / broadcast basis tensors of the 'default' equivariant layers, O(E) through segment sums.
Closed forms of the reference's masked bmm poolings (models/sparse/linear.py:54-97,99-141,162-238); a pool
over an empty set is exactly 0 (the reference divides an all-zero masked sum by 1e-5)."""
import torch

from ... import ops
from ..._lib import KEEP_DIAG, KEEP_OFFDIAG


def _counts(x, dtype, device):
    return torch.as_tensor(x, dtype=dtype, device=device)


def _safe_div(x, cnt):
    cnt = cnt.to(x.dtype)
    return torch.where(cnt == 0, torch.zeros_like(x), x / torch.where(cnt == 0, torch.ones_like(cnt), cnt))


def pool_1_to_0(G, normalize):
    """[B,N,D] -> [B,D] sum (mean) over the nodes of each graph (linear.py:54-62)."""
    plan = G.plan
    V = ops.SegSumRows.apply(G.values, plan.graph_seg(1), plan.nkind, KEEP_DIAG, plan_node_graph(plan))
    if normalize:
        V = _safe_div(V, _counts(G.n_nodes, V.dtype, V.device)[:, None])
    return V


def pool_2_to_0(G, normalize):
    """[B,E,D] -> [B,2D]: trace pool and off-diagonal pool (linear.py:85-97)."""
    plan = G.plan
    eg = plan_entry_graph(plan)
    V1 = ops.SegSumRows.apply(G.values, plan.graph_seg(2), plan.kind, KEEP_DIAG, eg)
    V2 = ops.SegSumRows.apply(G.values, plan.graph_seg(2), plan.kind, KEEP_OFFDIAG, eg)
    if normalize:
        n = _counts(G.n_nodes, V1.dtype, V1.device)[:, None]
        e = _counts(G.n_edges, V1.dtype, V1.device)[:, None] - n
        V1, V2 = _safe_div(V1, n), _safe_div(V2, e)
    return torch.cat([V1, V2], -1)


def plan_entry_graph(plan):
    if not hasattr(plan, '_entry_graph'):
        plan._entry_graph = (torch.arange(plan.R, device=plan.device) // max(plan.Emax, 1)).to(torch.int32)
    return plan._entry_graph


def plan_node_graph(plan):
    if not hasattr(plan, '_node_graph'):
        plan._node_graph = (torch.arange(plan.Nn, device=plan.device) // max(plan.Nmax, 1)).to(torch.int32)
    return plan._node_graph


def basis_1_to_1(G, normalize):
    raise NotImplementedError("Linear(1,1,cfg='default') is not built yet (light cfg is)")


def basis_2_to_1(G, dG, normalize):
    raise NotImplementedError("Linear(2,1,cfg='default') is not built yet (light cfg is)")


def basis_2_to_2(G, normalize):
    raise NotImplementedError("Linear(2,2,cfg='default') is not built yet (light cfg is)")
