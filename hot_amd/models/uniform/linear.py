"""Equivariant linear layers of the 1 -> k-uniform extension on plain tensors: x [N, D] (one set), hyperedges
`indices [B, k]`.  Same constructor, parameter shapes and state_dict keys (`weight.weight`, `bias`) as
hot_pytorch/models/uniform/linear.py:5-81.

1 -> k:  out[b] = cat_j x[indices[b, j]] @ W[:kD] (+ pool(x) @ W[kD:]) + bias.  The concatenation is linear in its
blocks, so the k products are taken once per NODE (one [N, D] x [D, k D'] library GEMM, N << B) and the hyperedge
rows are k coalesced row gathers summed in place (hotk_gather_rows with accumulate); the backward of each gather is
a segment sum over the hyperedges that reference the node (UniformPlan.seg), not an atomic scatter."""
import torch
import torch.nn as nn

from ... import ops
from .index import plan_of


class _Weight(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_out, cfg='default', normalize=False):
        super().__init__()
        assert cfg in ('default', 'light')
        self.ord_in, self.ord_out, self.dim_in, self.cfg, self.normalize = ord_in, ord_out, dim_in, cfg, normalize
        if ord_in == 1:
            n_w = {'default': ord_out + 1, 'light': ord_out}[cfg]
        else:
            raise NotImplementedError('This extension is only for 1->k-uniform')
        self.weight = nn.Parameter(torch.Tensor(dim_in * n_w, dim_out))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.xavier_normal_(self.weight)

    def _pool(self, x, plan):
        """[1, D] mean or sum over the set (uniform/linear.py:29,41)"""
        s = ops.SegSumRows.apply(x, plan.graph_seg(1), plan.nkind, ops.KEEP_VALID, plan.node_row_seg)
        return s / max(x.size(0), 1) if self.normalize else s

    def _1_to_1(self, x, plan):
        if self.cfg == 'light':
            return ops.RowsMatmul.apply(x, self.weight, None)
        D = self.dim_in
        y = ops.RowsMatmul.apply(x, self.weight[:-D], None)
        return y + ops.RowsMatmul.apply(self._pool(x, plan), self.weight[-D:], None)

    def _1_to_k(self, x, plan):
        D, k = self.dim_in, self.ord_out
        Dout = self.weight.size(1)
        # per-node products of all k blocks at once: P[:, j*Dout:(j+1)*Dout] = x @ W_j
        Wn = self.weight[:k * D].view(k, D, Dout).permute(1, 0, 2).reshape(D, k * Dout)
        P = ops.RowsMatmul.apply(x, Wn, None)
        y = ops.GatherSum.apply(P, plan, Dout)
        if self.cfg == 'default':
            y = y + ops.RowsMatmul.apply(self._pool(x, plan), self.weight[-D:], None)
        return y

    def forward(self, x: torch.Tensor, indices: torch.LongTensor = None) -> torch.Tensor:
        assert len(x.size()) == 2
        if self.ord_out == 1:
            assert indices is None
            return self._1_to_1(x, plan_of(x, None))
        assert len(indices.size()) == 2 and indices.size(1) == self.ord_out
        return self._1_to_k(x, plan_of(x, indices))


class Linear(nn.Module):
    def __init__(self, ord_in, ord_out, in_dim, out_dim, bias=True, cfg='default', normalize=True):
        super().__init__()
        self.weight = _Weight(ord_in, ord_out, in_dim, out_dim, cfg, normalize)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_dim))
            self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.constant_(self.bias, 0.)

    def forward(self, x: torch.Tensor, indices: torch.LongTensor = None) -> torch.Tensor:
        """x [N, D], indices [B, k] or None -> [B, D'] or [N, D']"""
        x = self.weight(x, indices)
        if hasattr(self, 'bias'):
            x = x + self.bias[None, :]
        return x
