"""MLP of order-changing equivariant linear layers, sparse or dense layout.

Interface contract of hot_pytorch/models/mlp.py:15-40: `layers` is one nn.Sequential holding, per layer,
Linear(cfg='default', normalize=True) -- the full 15 / 5 / 2-operation basis -- followed (except after the last layer) by a
pointwise nonlinearity and a dropout, so the parameter keys are `layers.<3k>.weight.weight`, `layers.<3k>.bias.bias.<j>`."""
import torch.nn as nn

from . import _stack


class MLP(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden, f='relu', dropout=0., sparse=True):
        super().__init__()
        lay = _stack.layout_of(sparse)
        self.sparse = sparse
        self._batch_type = lay.batch_type
        plan = _stack.orders_and_widths(ord_in, ord_hidden, ord_out, dim_in, dim_hidden, dim_out)
        members = []
        for k, ((o_src, w_src), (o_dst, w_dst)) in enumerate(zip(plan[:-1], plan[1:])):
            members.append(lay.linear(o_src, o_dst, w_src, w_dst, bias=True, cfg='default', normalize=True))
            if k + 2 < len(plan):     # not the last layer
                members += [lay.nonlinear(f), lay.apply(nn.Dropout(p=dropout))]
        self.layers = nn.Sequential(*members)

    def forward(self, G):
        if not isinstance(G, self._batch_type):
            raise AssertionError(f'a {"sparse" if self.sparse else "dense"} Batch is expected, got {type(G).__name__}')
        return self.layers(G)
