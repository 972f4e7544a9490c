"""Dense Batch container: same fields, constructor and helper functions as hot_pytorch/batch/dense.py:9-107
([B, N, D] sets and [B, N, N, D] graphs with a validity mask).

The dense layout is read as a sparse batch over the COMPLETE index set {(i, j)}: `sparse_view` wraps the same
memory ([B, N*N, D], no copy) as a hot_amd.batch.sparse.Batch whose segment plan is a DensePlan, so every
plan-driven HIP kernel of the sparse path (gathers, segment sums, feature maps, key-value sums, the factored
Linear(2,2,'light')) runs on it unchanged.  Nothing here computes on the CPU."""
from typing import Union, Callable

import torch

from ..utils.set import get_mask, masked_fill
from ..plan import DensePlan
from . import sparse as S


class Batch(object):
    """dense minibatch: A [B, N, D] (sets, order 1) or [B, N, N, D] (graphs, order 2), n_nodes python list, mask [B, N(, N)]
    marking the valid positions"""

    def __init__(self, A: torch.Tensor, n_nodes: list, mask: torch.BoolTensor = None, skip_masking: bool = False):
        order = A.dim() - 2
        if order == 0:
            raise AssertionError('zeroth-order data (plain [B, D] features) is handled as a tensor, not a Batch')
        if order not in (1, 2):
            raise AssertionError('dense batching is supported up to second order')
        self.A, self.n_nodes, self.device, self.order = A, n_nodes, A.device, order
        self.mask = mask if mask is not None else self.get_mask()
        if self.mask.size(1) != max(n_nodes) or self.mask.shape != A.shape[:-1]:
            raise AssertionError(f'mask {tuple(self.mask.shape)} does not describe A {tuple(A.shape)} / n_nodes')
        self._plan = None
        if not skip_masking:
            self.apply_mask(0)

    def __repr__(self):
        return f"Batch(A {list(self.A.size())})"

    def to(self, device: Union[str, torch.device]) -> 'Batch':
        self.A, self.mask = self.A.to(device), self.mask.to(device)
        self.device = self.A.device
        if self._plan is not None and self._plan.device != self.device:
            self._plan = None
        return self

    @torch.no_grad()
    def get_mask(self) -> torch.BoolTensor:
        """valid nodes [B, N]; for order 2 the outer product of that with itself [B, N, N]"""
        valid = get_mask(torch.as_tensor(self.n_nodes, dtype=torch.long, device=self.device))
        return valid if self.order == 1 else valid[:, :, None] & valid[:, None, :]

    def apply_mask(self, value=0.) -> None:
        self.A = masked_fill(self.A, self.mask, value)

    @property
    def plan(self) -> DensePlan:
        """complete-index-set segment plan (built on first use, shared through batch_like / d / v2d)"""
        if self._plan is None:
            self._plan = DensePlan(self.n_nodes, self.A.size(1), self.A.device)
        return self._plan


def _carry(G: Batch, H: Batch) -> Batch:
    if G.A.size(1) == H.A.size(1) and G.n_nodes == H.n_nodes:
        H._plan = G._plan
    return H


def sparse_view(G: Batch) -> S.Batch:
    """the same memory as a sparse-path Batch driven by the DensePlan (order 1: [B, N, D]; order 2: [B, N*N, D])"""
    H = S.Batch.__new__(S.Batch)
    plan = G.plan
    B = G.A.size(0)
    H.indices = None
    H.values = G.A if G.order == 1 else G.A.reshape(B, plan.Emax, G.A.size(-1))
    H.n_nodes = G.n_nodes
    H.n_edges = None if G.order == 1 else [n * n for n in G.n_nodes]
    H.device, H.order = G.A.device, G.order
    nm = (plan.nkind != 0).view(B, plan.Nmax)
    H.node_mask = nm
    H.mask = nm if G.order == 1 else (plan.kind != 0).view(B, plan.Emax)
    H.t_indices = H.t_mask = H.node_ofs = None
    H._plan, H._pending_mask = plan, False
    return H


def from_sparse_view(G: Batch, H: S.Batch, order: int) -> Batch:
    """wrap the values of a sparse-path result (already masked by the kernels) as a dense Batch sharing G's plan"""
    plan = G.plan
    B, N = plan.B, plan.Nmax
    A = H.values if order == 1 else H.values.reshape(B, N, N, H.values.size(-1))
    mask = (plan.nkind != 0).view(B, N) if order == 1 else (plan.kind != 0).view(B, N, N)
    if G.order == order:
        mask = G.mask
    out = Batch(A, G.n_nodes, mask, skip_masking=True)
    out._plan = plan
    return out


def batch_like(G: Batch, A: torch.Tensor, skip_masking=False) -> Batch:
    """G's node counts and mask around other values"""
    return _carry(G, Batch(A, G.n_nodes, G.mask, skip_masking))


def t(G: Batch) -> Batch:
    """swap the two node axes (a view)"""
    assert G.order == 2
    return _carry(G, Batch(G.A.transpose(2, 1), G.n_nodes, G.mask, skip_masking=True))


def nd(G: Batch) -> Batch:
    """off-diagonal part"""
    assert G.order == 2
    H = S.nd(sparse_view(G))
    return from_sparse_view(G, H, 2)


def d(G: Batch, mask: torch.Tensor = None) -> Batch:
    """diagonal as an order-1 batch"""
    assert G.order == 2 and (mask is None or mask.dim() == 2)
    H = S.d(sparse_view(G))
    out = Batch(H.values, G.n_nodes, mask, skip_masking=True)
    out._plan = G._plan
    return out


def v2d(G: Batch, mask: torch.Tensor = None) -> Batch:
    """order-1 values on the diagonal of an order-2 batch, zero elsewhere"""
    assert G.order == 1 and (mask is None or mask.dim() == 3)
    plan = G.plan
    Sv = sparse_view(G)
    carrier = S.Batch.__new__(S.Batch)
    carrier.__dict__.update(Sv.__dict__)
    carrier.order = 2
    H = S.v2d(carrier, Sv)
    A = H.values.reshape(plan.B, plan.Nmax, plan.Nmax, G.A.size(-1))
    out = Batch(A, G.n_nodes, mask, skip_masking=True)
    out._plan = plan
    return out


def apply(G: Union[torch.Tensor, Batch], f: Callable[[torch.Tensor], torch.Tensor], skip_masking=False) -> Union[torch.Tensor, Batch]:
    """f on the values of a batch (or on a plain tensor)"""
    return _carry(G, Batch(f(G.A), G.n_nodes, G.mask, skip_masking)) if isinstance(G, Batch) else f(G)


def add_batch(G1: Union[torch.Tensor, Batch], G2: Union[torch.Tensor, Batch]) -> Union[torch.Tensor, Batch]:
    """sum of two batches of the same shape and node counts (or of two equally shaped tensors)"""
    pair = (isinstance(G1, Batch), isinstance(G2, Batch))
    if pair == (True, True):
        same = G1.order == G2.order and G1.n_nodes == G2.n_nodes and G1.A.size() == G2.A.size()
        assert same, 'add_batch: the two batches do not share their structure'
        return _carry(G1, Batch(G1.A + G2.A, G1.n_nodes, G1.mask, skip_masking=True))
    assert pair == (False, False) and G1.size() == G2.size(), 'add_batch: two batches or two equally shaped tensors'
    return G1 + G2
