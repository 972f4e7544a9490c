"""MLP of order-changing equivariant linear layers on the sparse Batch.

Same constructor, attribute names (`layers.<k>.weight.weight`, `layers.<k>.bias.bias.<j>`) and forward signature as
hot_pytorch/models/mlp.py:15-40: every layer is `Linear(cfg='default', normalize=True)`, i.e. the full 15 / 5 / 2-operation
basis, with a pointwise nonlinearity and dropout between layers.  Only the sparse layout is built here (the dense
[B,N,N,D] layout is out of scope): `sparse=False` raises."""
from typing import List, Union

import torch.nn as nn

from ..batch.sparse import Batch as S
from .sparse.linear import Linear
from .sparse.f import Nonlinear, Apply


class MLP(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden: Union[List, int], f='relu', dropout=0.,
                 sparse=True):
        super().__init__()
        if not sparse:
            raise NotImplementedError('hot_amd builds the sparse layout only (dense [B,N,N,D] layers are out of scope)')
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

    def forward(self, G: S):
        assert isinstance(G, S)
        return self.layers(G)
