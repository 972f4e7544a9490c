"""Kernelised attention coefficients on the sparse Batch: same classes, constructor arguments and forward
signatures as hot_pytorch/models/sparse/kernelattncoef.py:13-234, computed by the segmented HIP kernels."""
from typing import Union, List

import torch
import torch.nn as nn
from torch import Tensor as T

from ...batch.sparse import Batch as B, batch_like
from ... import ops, _lib
from ..._lib import KEEP_DIAG, KEEP_OFFDIAG
from ..common.kernel import KernelFeatureMap


class KernelFeatureMapWrapper(nn.Module):
    def __init__(self, feature_map: KernelFeatureMap, dim_qk, n_heads):
        super().__init__()
        self.feature_map = feature_map
        self.dim_qk_head = dim_qk // n_heads if dim_qk >= n_heads else 1

    def forward(self, G: B, is_query=False) -> B:
        """[B,|E|,H*d] -> [B,|E|,H*m]: every width-d head chunk goes through the feature map; rows on the
        diagonal of an order-2 batch and padding rows come out as exact zeros (phi(0) != 0, so the masks
        matter).  One fused kernel instead of split/stack/einsum/masked_fill/cat (kernelattncoef.py:19-28)."""
        assert self.feature_map.dim_features == self.dim_qk_head
        plan = G.plan
        kind, keep = (plan.kind, KEEP_OFFDIAG) if G.order > 1 else (plan.nkind, KEEP_DIAG)
        # edge-shard mode: the softmax key map's single maximum (common/kernel.py:112) is MAX-all-reduced inside ops.Phi
        v = self.feature_map(G.values, is_query, kind, keep, getattr(plan, 'shard', None))
        return batch_like(G, v, skip_masking=True)


def _cat_values(items):
    vals = [x.values if isinstance(x, B) else x for x in items]
    return vals[0] if len(vals) == 1 else torch.cat(vals, dim=-1)


class KernelAttnCoef(nn.Module):
    def __init__(self, ord_q, ord_k, dim_qk_head, dim_v, n_heads):
        super().__init__()
        assert ord_q in (1, 2) and ord_k in (1, 2)
        self.ord_q = ord_q
        self.ord_k = ord_k
        self.dim_qk_head = dim_qk_head  # caution: feature map dimension
        self.dim_v = dim_v
        self.dim_v_head = dim_v // n_heads if dim_v >= n_heads else 1
        self.n_heads = n_heads

    def get_attn_coef(self, query, key):
        raise NotImplementedError('explicit attention coefficients are quadratic in |E|; the linear-cost path '
                                  'never materialises them (use forward)')

    def forward(self, query: Union[B, List[B]], key: Union[B, List[B]], value: Union[B, List[B]],
                diagonal: tuple = None) -> Union[B, List[B]]:
        """Higher-order kernelised attention, key-value first (kernelattncoef.py:198-234).
        query [B,|Eq|,m*H], key [B,|Ek|,m*H], value [B,|Ev|,Dv].  diagonal=None: per-graph key-value sums;
        diagonal in {(1,2),(2,3),(1,3)} (order-1 keys, order-2 values): KV_j = sum_i K_i (x) V_ij and
        O_j = Q_j.KV_j, O_ij = Q_ij.KV_j (dim1 = 2) or O_ij = Q_ij.KV_i (dim1 = 1).
        Lists of S tuples are run as S*H heads in one pass and split back."""
        as_list = isinstance(value, list)
        vs = value if as_list else [value]
        # queries / keys of a list call may also arrive as ONE batch with the S slots already side by side
        qs = query if isinstance(query, list) else [query]
        ks = key if isinstance(key, list) else [key]
        q0, k0, v0 = qs[0], ks[0], vs[0]
        assert isinstance(q0, B) and isinstance(k0, B) and isinstance(v0, B)
        assert q0.order == self.ord_q and k0.order == self.ord_k
        if diagonal is None:
            assert v0.order == self.ord_k
        else:
            assert self.ord_k == 1 and v0.order == 2
            assert (self.ord_q, diagonal[0], diagonal[1]) in [(1, 1, 2), (2, 2, 3), (2, 1, 3)]
        # the plan of an order-2 participant describes nodes and entries alike
        plan = next((x.plan for x in (v0, q0, k0) if x.order == 2), q0.plan)
        att = ops.KernelAttn.apply(_cat_values(qs), _cat_values(ks), _cat_values(vs), plan, self.ord_q, self.ord_k,
                                   self.dim_qk_head, self.dim_v_head, diagonal)
        if not as_list:
            return batch_like(q0, att, skip_masking=True)
        return [batch_like(q0, a, skip_masking=True) for a in att.split(self.dim_v, -1)]
