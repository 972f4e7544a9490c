"""Order-2 index utilities (mirror of hot_pytorch/utils/sparse.py:34-134) on top of the device plan."""
import torch

from .. import ops
from .._lib import KEEP_OFFDIAG
from ..plan import Plan


def get_transpose_info(indices: torch.LongTensor, mask: torch.BoolTensor, chunk_size=1000, n_nodes=None):
    """(t_indices [B,E] long, t_mask [B,E] bool), bit-exact with utils/sparse.py:102-127, computed by the
    sort + binary-search kernel hotk_plan_build in O(E log E) instead of the O(E^2) all-pairs match.
    `chunk_size` is accepted for signature compatibility and unused."""
    sizes = mask.sum(1).tolist()
    if n_nodes is None:
        # node count per graph = number of leading diagonal rows (layout invariant)
        diag = ((indices[..., 0] == indices[..., 1]) & mask).sum(1).tolist()
        n_nodes = diag
    plan = Plan(indices, n_nodes, sizes)
    return plan.t_indices, plan.t_mask


def do_transpose(values, transpose_indices, exist_mask, plan=None):
    """T[b,e] = values[b, t_indices[b,e]] * t_mask[b,e] (utils/sparse.py:130-134)."""
    if plan is not None:
        B, E = plan.B, plan.Emax
        tpos = plan.tpos
    else:
        B, E = exist_mask.shape
        base = (torch.arange(B, device=values.device) * E)[:, None]
        tpos = torch.where(exist_mask, transpose_indices + base, torch.full_like(transpose_indices, -1))
        tpos = tpos.reshape(-1).to(torch.int32)
    return ops.GatherRows.apply(values, tpos, tpos, B * E).view(B, E, values.size(-1))


def get_diag(plan: Plan, values):
    """[B,E,C] -> [B,Nmax,C] (utils/sparse.py:59-87)."""
    return ops.GatherRows.apply(values, plan.diag_pos, plan.diag_node, plan.Nn).view(plan.B, plan.Nmax, values.size(-1))


def to_diag(plan: Plan, node_values):
    """[B,Nmax,C] -> [B,E,C], zero off the diagonal (utils/sparse.py:34-56)."""
    return ops.GatherRows.apply(node_values, plan.diag_node, plan.diag_pos, plan.R).view(plan.B, plan.Emax,
                                                                                          node_values.size(-1))


def get_nondiag(plan: Plan, values):
    """Zero diagonal and padding rows (utils/sparse.py:90-99)."""
    return ops.MaskRows.apply(values, plan.kind, KEEP_OFFDIAG, 0.0)
