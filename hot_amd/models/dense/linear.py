"""Order-{0,1,2} equivariant linear layers on the DENSE Batch: same constructor, parameter shapes and state_dict
keys (weight.weight, bias.bias.k) as hot_pytorch/models/dense/linear.py:9-278.

A dense batch is the complete index set, so every case the sparse path has (1->1, 2->2, 2->1, 2->0, 1->0) runs the
sparse-path implementation on the same memory through the DensePlan (batch/dense.py `sparse_view`): O(N^2) gathers
and segment sums instead of the reference's [N^k, N^l] mask tensors and einsums (masksum.py:33-82).  The one case the
sparse path lacks, set -> graph (1->2), is here in factored form: node-level blocks projected per node, then one
broadcast kernel (ops.Assemble12)."""
from typing import Union

import torch

from ...batch import dense as DB
from ...batch import sparse as SB
from ... import ops
from ..._lib import KEEP_DIAG
from ..sparse.linear import Linear as _SparseLinear
from ..sparse import basis


def _order2_carrier(Sv: SB.Batch, values: torch.Tensor) -> SB.Batch:
    """order-2 sparse-path view [B, N*N, C] on the plan of an order-1 view"""
    plan = Sv.plan
    H = SB.Batch.__new__(SB.Batch)
    H.__dict__.update(Sv.__dict__)
    H.order = 2
    H.values = values.view(plan.B, plan.Emax, values.size(-1))
    H.n_edges = [n * n for n in Sv.n_nodes]
    H.mask = (plan.kind != 0).view(plan.B, plan.Emax)
    return H


class Linear(_SparseLinear):
    def __init__(self, ord_in, ord_out, dim_in, dim_out, bias=True, cfg='default', normalize=True):
        super().__init__(ord_in, ord_out, dim_in, dim_out, bias, cfg, normalize)

    def _1_to_2(self, Sv: SB.Batch) -> SB.Batch:
        """basis of models/dense/linear.py:92-111: diagonal x_i [, (tot - x_i)/(n-1)]; off-diagonal x_j, x_i
        [, (tot - x_i - x_j)/(n-2)] -- the last one split evenly between the row and the column node table"""
        w = self.weight
        plan = Sv.plan
        x = Sv.values.reshape(plan.Nn, -1)
        D = x.size(-1)
        W = w.weight
        C = W.size(1)
        if w.cfg == 'light':
            yn = ops.RowsMatmul.apply(x, torch.cat([W[0:D], W[2 * D:3 * D], W[D:2 * D]], dim=1), None)
        else:
            ng = basis.plan_node_graph(plan)
            n = torch.as_tensor(Sv.n_nodes, dtype=x.dtype, device=x.device)[ng.long()].unsqueeze(1)
            tot = ops.SegSumRows.apply(x, plan.graph_seg(1), plan.nkind, KEEP_DIAG, ng)
            tot = ops.GatherSeg.apply(tot, ng, plan.graph_seg(1), plan.Nn)
            # an empty pool is an exact zero (the reference divides a zero sum by its 1e-5 guard, masksum.py:58-60)
            inf = torch.full_like(n, float('inf'))
            c1 = torch.where(n - 1 <= 0, inf, n - 1 if w.normalize else torch.ones_like(n))
            c2 = torch.where(n - 2 <= 0, inf, n - 2 if w.normalize else torch.ones_like(n))
            a4 = (tot - x) / c1
            half = (0.5 * tot - x) / c2
            Fd = torch.cat([x, a4], -1)
            Fr = torch.cat([x, half], -1)
            yn = torch.cat([ops.RowsMatmul.apply(Fd, torch.cat([W[0:D], W[3 * D:4 * D]], 0), None),
                            ops.RowsMatmul.apply(Fr, torch.cat([W[2 * D:3 * D], W[4 * D:5 * D]], 0), None),
                            ops.RowsMatmul.apply(Fr, torch.cat([W[D:2 * D], W[4 * D:5 * D]], 0), None)], dim=1)
        bias = self.bias.bias if hasattr(self, 'bias') else None
        y = ops.Assemble12.apply(yn, None if bias is None else bias[0], None if bias is None else bias[1], plan)
        return _order2_carrier(Sv, y)

    def forward_view(self, Sv: Union[SB.Batch, torch.Tensor]):
        """sparse-path view in, sparse-path view (or [B, D'] tensor) out"""
        w = self.weight
        if (w.ord_in, w.ord_out) == (1, 2):
            assert isinstance(Sv, SB.Batch) and Sv.order == 1
            return self._1_to_2(Sv)
        if (w.ord_in, w.ord_out, w.cfg) == (2, 2, 'default'):
            # the dense class lists the pooled column sums before the row sums (dense/linear.py:146-148, A6 / A7), the
            # sparse one the other way round (sparse/linear.py:187-199): blocks 5 and 6 of the weight trade places
            assert isinstance(Sv, SB.Batch) and Sv.order == 2
            D = w.weight.size(0) // 15
            W = torch.cat([w.weight[:5 * D], w.weight[6 * D:7 * D], w.weight[5 * D:6 * D], w.weight[7 * D:]], 0)
            plan = Sv.plan
            y = ops.RowsMatmul.apply(basis.basis_2_to_2(Sv, w.normalize), W, None)
            bias = self.bias.bias if hasattr(self, 'bias') else None
            y = ops.BiasMask.apply(y, plan.kind, None if bias is None else bias[0], None if bias is None else bias[1],
                                   plan.Ev == plan.R)
            return SB.batch_like(Sv, y.view(*Sv.values.shape[:-1], y.size(-1)), skip_masking=True)
        return _SparseLinear.forward(self, Sv)

    def forward(self, G: Union[DB.Batch, torch.Tensor]) -> Union[DB.Batch, torch.Tensor]:
        if isinstance(G, torch.Tensor):
            return _SparseLinear.forward(self, G)
        if isinstance(G, SB.Batch):
            return self.forward_view(G)
        assert G.order == self.weight.ord_in
        out = self.forward_view(DB.sparse_view(G))
        if isinstance(out, torch.Tensor):
            return out
        return DB.from_sparse_view(G, out, self.weight.ord_out)
