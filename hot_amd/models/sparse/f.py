"""Elementwise wrappers over Batch values (hot_pytorch/models/sparse/f.py:10-36)."""
from typing import Union, Callable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...batch.sparse import Batch as B, apply, add_batch, batch_like
from ... import ops


class Apply(nn.Module):
    def __init__(self, f: Callable[[torch.Tensor], torch.Tensor], skip_masking=False):
        super().__init__()
        self.f = f
        self.skip_masking = skip_masking

    def forward(self, G: Union[torch.Tensor, B]) -> Union[torch.Tensor, B]:
        if isinstance(self.f, nn.LayerNorm) and isinstance(G, B) and ops.layernorm_ok(G.values, self.f):
            # LayerNorm and the zeroing of the padding rows in one HIP kernel (encoder.py:36,40-41: Apply(LayerNorm))
            kind = None
            if not self.skip_masking and G._padded:
                kind = G.plan.kind if G.order == 2 else G.plan.nkind
            y = ops.LayerNormRows.apply(G.values, self.f.weight, self.f.bias, kind, self.f.eps)
            return batch_like(G, y, skip_masking=True)
        return apply(G, self.f, self.skip_masking)


class Nonlinear(Apply):
    def __init__(self, f: str, skip_masking=False):
        super().__init__({'relu': F.relu, 'leakyrelu': F.leaky_relu, 'gelu': F.gelu, 'tanh': torch.tanh}[f], skip_masking)


class Add(nn.Module):
    @staticmethod
    def forward(G1: Union[torch.Tensor, B], G2: Union[torch.Tensor, B]) -> Union[torch.Tensor, B]:
        return add_batch(G1, G2)
