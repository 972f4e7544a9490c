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
    A: torch.Tensor
    n_nodes: list
    device: torch.device
    mask: torch.BoolTensor
    order: int

    def __init__(self, A: torch.Tensor, n_nodes: list, mask: torch.BoolTensor = None, skip_masking: bool = False):
        """a mini-batch of dense (hyper)graphs: A [B, N^k, D], n_nodes python list, mask [B, N^k]"""
        self.A = A
        self.n_nodes = n_nodes
        self.device = self.A.device
        self.order = len(A.size()) - 2
        assert self.order != 0, "zeroth-order data (N x D batched features) is expected to be handled separately"
        assert self.order in (1, 2), "we currently support dense batching up to second order"
        self.mask = self.get_mask() if mask is None else mask
        assert self.mask.size(1) == max(n_nodes)
        assert self.mask.size() == A.size()[:-1]
        self._plan = None
        if not skip_masking:
            self.apply_mask(0)

    def __repr__(self):
        return f"Batch(A {list(self.A.size())})"

    def to(self, device: Union[str, torch.device]) -> 'Batch':
        self.A = self.A.to(device)
        self.mask = self.mask.to(device)
        if self._plan is not None and self._plan.device != self.A.device:
            self._plan = None
        self.device = self.A.device
        return self

    @torch.no_grad()
    def get_mask(self) -> torch.BoolTensor:
        node_mask = get_mask(torch.tensor(self.n_nodes, dtype=torch.long, device=self.device))  # [B, N]
        if self.order == 1:
            return node_mask
        return node_mask.unsqueeze(1) & node_mask.unsqueeze(2)  # [B, N, N]

    def apply_mask(self, value=0.) -> None:
        # mask out invalid tensor elements
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
    # only substitute A
    return _carry(G, Batch(A, G.n_nodes, G.mask, skip_masking))


def t(G: Batch) -> Batch:
    # transpose
    assert G.order == 2
    return _carry(G, Batch(G.A.transpose(2, 1), G.n_nodes, G.mask, skip_masking=True))


def nd(G: Batch) -> Batch:
    # get non-diagonals
    assert G.order == 2
    H = S.nd(sparse_view(G))
    return from_sparse_view(G, H, 2)


def d(G: Batch, mask: torch.Tensor = None) -> Batch:
    # get diagonals
    assert G.order == 2
    if mask is not None:
        assert len(mask.size()) == 2
    H = S.d(sparse_view(G))
    out = Batch(H.values, G.n_nodes, mask, skip_masking=True)
    out._plan = G._plan
    return out


def v2d(G: Batch, mask: torch.Tensor = None) -> Batch:
    # vectors to diagonal matrices
    assert G.order == 1
    if mask is not None:
        assert len(mask.size()) == 3
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
    if isinstance(G, torch.Tensor):
        return f(G)
    return _carry(G, Batch(f(G.A), G.n_nodes, G.mask, skip_masking))


def add_batch(G1: Union[torch.Tensor, Batch], G2: Union[torch.Tensor, Batch]) -> Union[torch.Tensor, Batch]:
    # add features of two batched graphs with identical edge structures
    if isinstance(G1, Batch) and isinstance(G2, Batch):
        assert G1.order == G2.order
        assert G1.n_nodes == G2.n_nodes
        assert G1.A.size() == G2.A.size()
        return _carry(G1, Batch(G1.A + G2.A, G1.n_nodes, G1.mask, skip_masking=True))
    else:
        assert isinstance(G1, torch.Tensor) and isinstance(G2, torch.Tensor)
        assert G1.size() == G2.size()
        return G1 + G2
