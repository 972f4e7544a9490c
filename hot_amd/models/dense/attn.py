"""Softmax self-attention on the DENSE Batch -- the order-0 readouts (set -> vector, graph -> vector): same constructor,
attribute names (state_dict keys fc_0 / fc_1 / fc_2 / fc_v / fc_o) and forward signature as
hot_pytorch/models/dense/attn.py:10-98,138-157.

One query vector per graph attends over the graph's nodes (1 -> 0) or over its nodes and its off-diagonal entries (2 -> 0;
the reference masks the self loops of the order-2 keys, dense/attncoef.py:40-45): linear in the size of the batch.  A dense
batch is the complete index set, so the layer is the sparse-path readout (models/sparse/attn.py) run on the same memory
through the DensePlan, with the dense Linear layers underneath (their order-0 pools divide by the dense layout's counts).
Orders with an output of order >= 1 are quadratic in N^2 there as here and are not built (KernelSelfAttn covers them)."""
import torch

from ...batch import dense as DB
from ..sparse.attn import SelfAttn as _SparseSelfAttn
from .linear import Linear


class SelfAttn(_SparseSelfAttn):
    linear_cls = Linear

    def forward(self, G: DB.Batch) -> torch.Tensor:
        assert isinstance(G, DB.Batch) and G.order == self.ord_in
        return _SparseSelfAttn.forward(self, DB.sparse_view(G))
