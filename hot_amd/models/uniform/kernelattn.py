"""Kernel (FAVOR+) self-attention of the 1 -> k-uniform extension: one set of N nodes, queries either the nodes
themselves (1 -> 1) or B hyperedges (1 -> k).  Same constructor, attribute names and forward signature as
hot_pytorch/models/uniform/kernelattn.py:7-101.

One global key-value sum  KV = sum_n phi(K_n) (x) [V_n | 1]  (hotk_seg_outer_sum over the single segment), applied
to every query row (hotk_row_contract) -- the same general kernels as the per-graph terms of the sparse layer."""
import torch
import torch.nn as nn

from ... import ops
from .linear import Linear
from .index import plan_of


class KernelSelfAttn(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads, cfg='default', dropout=0., drop_mu=0., feature_map=None):
        super().__init__()
        assert cfg == 'default'
        self.ord_in, self.ord_out = ord_in, ord_out
        self.dim_in, self.dim_v, self.dim_qk, self.n_heads = dim_in, dim_v, dim_qk, n_heads
        self.feature_map = feature_map
        self.dim_qk_head = dim_qk // n_heads if dim_qk >= n_heads else 1
        self.dim_v_head = dim_v // n_heads if dim_v >= n_heads else 1
        self.feat_dim = feature_map.num_features
        if (ord_in, ord_out) == (1, 1):
            self.fc_1 = Linear(1, 1, dim_in, dim_qk * 2 + dim_in, bias=True, cfg='light')
        elif ord_in == 1 and ord_out > 1:
            self.fc_1 = Linear(1, 1, dim_in, dim_qk, bias=True, cfg='light')
            self.fc_k = Linear(1, ord_out, dim_in, dim_qk + dim_in, bias=True, cfg='light')
        else:
            raise NotImplementedError('This extension is only for 1->k-uniform')
        self.fc_v = nn.Linear(dim_in, dim_v)
        self.fc_o = nn.Linear(dim_v, dim_in)
        self.reset_vo_parameters()
        self.dropout = nn.Dropout(p=dropout, inplace=True)

    def reset_vo_parameters(self):
        nn.init.xavier_normal_(self.fc_v.weight)
        nn.init.xavier_normal_(self.fc_o.weight)
        nn.init.constant_(self.fc_v.bias, 0.)
        nn.init.constant_(self.fc_o.bias, 0.)

    def _attend(self, q_raw, k_raw, x, plan, ord_q):
        """phi(q) . KV / guard(phi(q) . ks) with KV, ks summed over the N nodes (uniform/kernelattn.py:52-60,76-84)"""
        q = self.feature_map(q_raw, is_query=True)      # [rows, H m]; the map splits heads itself
        k = self.feature_map(k_raw, is_query=False)
        v = ops.RowsMatmul.apply(x, self.fc_v.weight.t(), self.fc_v.bias)
        att = ops.KernelAttn.apply(q, k, v, plan, ord_q, 1, self.feat_dim, self.dim_v_head, None)
        return ops.RowsMatmul.apply(self.dropout(att), self.fc_o.weight.t(), self.fc_o.bias)

    def _1_to_1(self, x):
        h_1 = self.fc_1(x)
        q_1, k_1 = h_1[:, :self.dim_qk], h_1[:, self.dim_qk:2 * self.dim_qk]
        non_att = h_1[:, -self.dim_in:]
        return non_att + self._attend(q_1, k_1, x, plan_of(x, None), 1)

    def _1_to_k(self, x, indices):
        k_1 = self.fc_1(x)
        h_k = self.fc_k(x, indices)
        q_k = h_k[:, :self.dim_qk]
        non_att = h_k[:, -self.dim_in:]
        return non_att + self._attend(q_k, k_1, x, plan_of(x, indices), 2)

    def forward(self, x: torch.Tensor, indices: torch.LongTensor = None) -> torch.Tensor:
        """x [N, D], indices [B, k] or None -> [B, D'] or [N, D']"""
        assert len(x.size()) == 2
        if self.ord_out == 1:
            assert indices is None
            return self._1_to_1(x)
        assert len(indices.size()) == 2 and indices.size(1) == self.ord_out
        return self._1_to_k(x, indices)
