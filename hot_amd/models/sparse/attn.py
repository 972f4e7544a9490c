"""Softmax self-attention on the sparse Batch - the order-0 readouts only.

Same constructor, attribute names (state_dict keys fc_0 / fc_1 / fc_2 / fc_v / fc_o) and forward signature as
hot_pytorch/models/sparse/attn.py:10-131.  The set->vector (1,0) and graph->vector (2,0) layers are linear in the number
of entries (one query vector per graph) and close the Encoder(…, ord_out=0) used by the PCQM4M configuration; they run
on the segmented softmax-pooling kernel hotk_seg_softmax_pool_fwd/bwd.  Every other (ord_in, ord_out) of the softmax
variant is quadratic in |E| (attncoef.py:61-163) and outside the linear-cost path this package implements."""
import torch
import torch.nn as nn

from ...batch.sparse import Batch as B, batch_like, d
from ..common.mudrop import MuDropout
from ... import ops, _lib
from ..._lib import KEEP_DIAG, KEEP_OFFDIAG
from .linear import Linear


class SelfAttn(nn.Module):
    linear_cls = Linear      # (the dense layout's readout, models/dense/attn.py, builds on its own Linear)

    def __init__(self, ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads, cfg='default', dropout=0., drop_mu=0.):
        super().__init__()
        Linear = self.linear_cls
        assert cfg in ('default', 'local')
        self.is_local = cfg == 'local'
        self.ord_in, self.ord_out = ord_in, ord_out
        self.dim_in, self.dim_v, self.dim_qk, self.n_heads = dim_in, dim_v, dim_qk, n_heads
        if (ord_in, ord_out) == (1, 0):
            n_v = 1
            self.fc_0 = Linear(1, 0, dim_in, dim_qk + dim_in, cfg='light')
            self.fc_1 = Linear(1, 1, dim_in, dim_qk, cfg='light')
        elif (ord_in, ord_out) == (2, 0):
            n_v = 2
            self.fc_0 = Linear(2, 0, dim_in, dim_qk * 2 + dim_in, cfg='light')
            self.fc_1 = Linear(2, 1, dim_in, dim_qk, cfg='light')
            self.fc_2 = Linear(2, 2, dim_in, dim_qk, cfg='light')
        elif (ord_in, ord_out) == (1, 2):
            raise NotImplementedError('Sparse set-to-graph is inefficient; use a dense layer or specialized layers')
        else:
            raise NotImplementedError('softmax attention with an order >= 1 output is quadratic in |E|; hot_amd '
                                      'implements the linear-cost kernel variant (KernelSelfAttn) for those orders')
        self.fc_v = nn.Linear(dim_in, dim_v * n_v)
        self.fc_o = nn.Linear(dim_v * n_v, dim_in)
        self.reset_vo_parameters()
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.mu_dropout = MuDropout(p=drop_mu)

    def reset_vo_parameters(self):
        nn.init.xavier_normal_(self.fc_v.weight)
        nn.init.xavier_normal_(self.fc_o.weight)
        nn.init.constant_(self.fc_v.bias, 0.)
        nn.init.constant_(self.fc_o.bias, 0.)

    def _values(self, G: B):
        v = batch_like(G, ops.linear_rows(G.values, self.fc_v), skip_masking=False)
        return [batch_like(G, x, skip_masking=True) for x in v.values.split(self.dim_v, -1)]

    def _1_to_0(self, G: B):
        h_0 = self.fc_0(G)                                   # [B, dqk + D]
        non_att, q_0 = h_0[..., -self.dim_in:], h_0[..., :self.dim_qk]
        k_1 = self.fc_1(G)                                   # [B, N, dqk]
        v_1 = self._values(G)[0]
        plan = G.plan
        att_1 = ops.SoftmaxPool.apply(q_0, k_1.values, v_1.values, plan.graph_seg(1), plan.nkind, KEEP_DIAG, self.n_heads)
        return non_att + self.fc_o(self.dropout(att_1))

    def _2_to_0(self, G: B):
        h_0 = self.fc_0(G)                                   # [B, 2 dqk + D]
        non_att = h_0[..., -self.dim_in:]
        q_a, q_b = h_0[..., :self.dim_qk], h_0[..., self.dim_qk:2 * self.dim_qk]
        k_1 = self.fc_1(G)                                   # [B, N, dqk]
        k_2 = self.fc_2(G)                                   # [B, |E|, dqk]
        v = self._values(G)
        plan = G.plan
        # set -> vector over the nodes, graph -> vector over the off-diagonal entries (loop mask, attncoef.py:45-48)
        att_1 = ops.SoftmaxPool.apply(q_a, k_1.values, d(v[0]).values, plan.graph_seg(1), plan.nkind, KEEP_DIAG,
                                      self.n_heads)
        # (edge-shard mode: the nodes are replicated on every rank, the off-diagonal entries are spread over the ranks)
        att_2 = ops.SoftmaxPool.apply(q_b, k_2.values, v[1].values, plan.graph_seg(2), plan.kind, KEEP_OFFDIAG,
                                      self.n_heads, ops.shard_of(plan))
        return non_att + self.fc_o(self.dropout(torch.cat([att_1, att_2], -1)))

    def forward(self, G: B):
        assert G.order == self.ord_in
        return self._1_to_0(G) if self.ord_in == 1 else self._2_to_0(G)
