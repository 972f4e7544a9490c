"""Order-{0,1,2} equivariant linear layers on the sparse Batch: same constructor, parameter shapes and
state_dict keys (weight.weight, bias.bias.k) as hot_pytorch/models/sparse/linear.py:16-347.

Forward = [basis tensors] @ weight + bias, padding zeroed.  The basis tensors are produced by HIP kernels in
O(E) (gathers and segment sums over the device plan) instead of the reference's O(E^2) boolean-mask bmm's;
the dense projection is a plain library GEMM (torch.matmul -> rocBLAS/hipBLASLt)."""
from typing import Union

import torch
import torch.nn as nn

from ...batch.sparse import Batch as B, batch_like, d
from ... import ops
from ..._lib import KEEP_DIAG, KEEP_OFFDIAG
from . import basis


N_WEIGHTS = {(0, 0): {'default': 1, 'light': 1}, (1, 0): {'default': 1, 'light': 1}, (1, 1): {'default': 2, 'light': 1},
             (1, 2): {'default': 5, 'light': 3}, (2, 0): {'default': 2, 'light': 2}, (2, 1): {'default': 5, 'light': 1},
             (2, 2): {'default': 15, 'light': 5}}


class _Weight(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_out, cfg='default', normalize=False):
        super().__init__()
        assert cfg in ('default', 'light')
        if (ord_in, ord_out) not in N_WEIGHTS:
            raise NotImplementedError('Currently supports up to second-order only')
        self.ord_in, self.ord_out, self.cfg, self.normalize = ord_in, ord_out, cfg, normalize
        self.weight = nn.Parameter(torch.Tensor(dim_in * N_WEIGHTS[(ord_in, ord_out)][cfg], dim_out))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.xavier_normal_(self.weight)

    def forward(self, G: Union[B, torch.Tensor]):
        """returns (carrier batch or None, projected values)"""
        key = (self.ord_in, self.ord_out)
        if isinstance(G, B):
            assert G.order == self.ord_in
        if key == (0, 0):
            return None, ops.RowsMatmul.apply(G, self.weight, None)
        if key == (1, 2):
            raise NotImplementedError('Sparse set-to-graph is inefficient; use a dense layer')
        if key == (1, 0):
            return None, ops.RowsMatmul.apply(basis.pool_1_to_0(G, self.normalize), self.weight, None)
        if key == (2, 0):
            return None, ops.RowsMatmul.apply(basis.pool_2_to_0(G, self.normalize), self.weight, None)
        if key == (1, 1):
            V = G.values if self.cfg == 'light' else basis.basis_1_to_1(G, self.normalize)
            return G, ops.RowsMatmul.apply(V, self.weight, None)
        if key == (2, 1):
            dG = d(G)
            V = dG.values if self.cfg == 'light' else basis.basis_2_to_1(G, dG, self.normalize)
            return dG, ops.RowsMatmul.apply(V, self.weight, None)
        V = ops.Basis22Light.apply(G.values, G.plan) if self.cfg == 'light' else basis.basis_2_to_2(G, self.normalize)
        return G, ops.RowsMatmul.apply(V, self.weight, None)


class _Bias(nn.Module):
    def __init__(self, ord_out, dim_out):
        super().__init__()
        if ord_out not in (0, 1, 2):
            raise NotImplementedError('Currently supports up to second-order only')
        self.ord_out = ord_out
        n_b = 2 if ord_out == 2 else 1
        self.bias = nn.ParameterList([nn.Parameter(torch.Tensor(dim_out)) for _ in range(n_b)])
        self.reset_parameters()

    def reset_parameters(self) -> None:
        [nn.init.constant_(b, 0.) for b in self.bias]


class Linear(nn.Module):
    def __init__(self, ord_in, ord_out, in_dim, out_dim, bias=True, cfg='default', normalize=True):
        super().__init__()
        self.weight = _Weight(ord_in, ord_out, in_dim, out_dim, cfg, normalize)
        if bias:
            self.bias = _Bias(ord_out, out_dim)
        self.factored = True   # 2->2 'light': node-level blocks projected per node (ops.Linear22Light)

    def forward(self, G: Union[B, torch.Tensor]) -> Union[B, torch.Tensor]:
        """A' = L A with L a learned combination of the equivariant basis; returns Batch (order 1/2) or
        Tensor [B, D'] (order 0)."""
        bias = self.bias.bias if hasattr(self, 'bias') else None
        w = self.weight
        if self.factored and (w.ord_in, w.ord_out, w.cfg) == (2, 2, 'light'):
            assert isinstance(G, B) and G.order == 2
            y = ops.Linear22Light.apply(G.values, w.weight, None if bias is None else bias[0],
                                        None if bias is None else bias[1], G.plan)
            return batch_like(G, y, skip_masking=True)
        H, y = self.weight(G)
        if H is None:
            return y if bias is None else y + bias[0][None, :]
        plan = H.plan
        if H.order == 2:
            # bias[0] on off-diagonal rows, bias[1] on diagonal rows, padding zeroed (linear.py:297-306,345-346)
            y = ops.BiasMask.apply(y, plan.kind, None if bias is None else bias[0], None if bias is None else bias[1],
                                   plan.Ev == plan.R)
        else:
            y = ops.BiasMask.apply(y, plan.nkind, None, None if bias is None else bias[0], sum(H.n_nodes) == plan.Nn)
        return batch_like(H, y.view(*H.values.shape[:-1], y.size(-1)), skip_masking=True)
