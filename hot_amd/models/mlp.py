"""MLP of order-changing equivariant linear layers on the sparse Batch.

Same constructor, attribute names (`layers.<k>.weight.weight`, `layers.<k>.bias.bias.<j>`) and forward signature as
hot_pytorch/models/mlp.py:15-40: every layer is `Linear(cfg='default', normalize=True)`, i.e. the full 15 / 5 / 2-operation
basis, with a pointwise nonlinearity and dropout between layers.  sparse=False takes the dense [B,N,N,D] layout
(hot_amd/models/dense, the same kernels through the DensePlan)."""
from typing import List, Union

import torch.nn as nn

from ..batch.sparse import Batch as S
from ..batch.dense import Batch as D
from .sparse.linear import Linear as S_Linear
from .sparse.f import Nonlinear as S_Nonlinear, Apply as S_Apply
from .dense.linear import Linear as D_Linear
from .dense.f import Nonlinear as D_Nonlinear, Apply as D_Apply


class MLP(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden: Union[List, int], f='relu', dropout=0.,
                 sparse=True):
        super().__init__()
        Linear = S_Linear if sparse else D_Linear
        Nonlinear = S_Nonlinear if sparse else D_Nonlinear
        Apply = S_Apply if sparse else D_Apply
        self.sparse = sparse
        if not isinstance(dim_hidden, list):
            dim_hidden = [dim_hidden] * len(ord_hidden)
        ords = [ord_in] + list(ord_hidden) + [ord_out]
        dims = [dim_in] + list(dim_hidden) + [dim_out]
        layers = []
        n_lin = len(ords) - 1
        for idx in range(n_lin):
            layers.append(Linear(ords[idx], ords[idx + 1], dims[idx], dims[idx + 1], bias=True, cfg='default', normalize=True))
            if idx < n_lin - 1:
                layers.append(Nonlinear(f))
                layers.append(Apply(nn.Dropout(p=dropout)))
        self.layers = nn.Sequential(*layers)

    def forward(self, G: Union[S, D]):
        assert isinstance(G, S if self.sparse else D)
        return self.layers(G)
