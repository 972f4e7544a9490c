"""Kernel (FAVOR+) self-attention on the DENSE Batch: same constructor, attribute names (state_dict keys fc_1 / fc_2 /
fc_v / fc_o / feature_map.*) and forward signature as hot_pytorch/models/dense/kernelattn.py:10-206.

Orders (1,1), (2,1) and (2,2) run the sparse-path layer on the same memory through the DensePlan (the complete index
set; general kernels, not the fused core, which relies on the sparse row layout).  Set -> graph (1,2), the layer of
config C3 (set-to-graph/main/models/model.py:9-30), exists only in the dense layout in the reference
(dense/kernelattn.py:95-116; the sparse class raises NotImplementedError): two per-graph key-value sums over the
nodes, applied to every off-diagonal entry query (term 1) and to the node queries, scattered to the diagonal (term 2)."""
import torch.nn as nn

from ...batch import dense as DB
from ...batch import sparse as SB
from ..common.mudrop import MuDropout
from ..sparse.kernelattn import KernelSelfAttn as _SparseKernelSelfAttn
from ..sparse.kernelattncoef import KernelFeatureMapWrapper, KernelAttnCoef
from .linear import Linear, _order2_carrier


class KernelSelfAttn(_SparseKernelSelfAttn):
    def __init__(self, ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads, cfg='default', dropout=0., drop_mu=0., feature_map=None):
        if (ord_in, ord_out) != (1, 2):
            super().__init__(ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads, cfg, dropout, drop_mu, feature_map)
            self.use_fused = False
            return
        nn.Module.__init__(self)
        assert cfg in ('default', 'local')
        self.is_local = cfg == 'local'
        self.ord_in, self.ord_out = ord_in, ord_out
        self.dim_in, self.dim_v, self.dim_qk, self.n_heads = dim_in, dim_v, dim_qk, n_heads
        self.feature_map = KernelFeatureMapWrapper(feature_map, dim_qk, n_heads)
        self.feat_dim = feature_map.num_features
        n_qk1, n_qk2, n_v = 3, 1, 2
        self.fc_1 = Linear(1, 1, dim_in, dim_qk * n_qk1, cfg='light')
        self.fc_2 = Linear(1, 2, dim_in, dim_qk * n_qk2 + dim_in, cfg='light')
        self.att_1_1 = KernelAttnCoef(1, 1, self.feat_dim, dim_v, n_heads)
        self.att_2_1 = KernelAttnCoef(2, 1, self.feat_dim, dim_v, n_heads)
        self.fc_v = nn.Linear(dim_in, dim_v * n_v)
        self.fc_o = nn.Linear(dim_v * n_v, dim_in)
        self.reset_vo_parameters()
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.mu_dropout = MuDropout(p=drop_mu)
        self.use_fused = False

    def _1_to_2(self, G: SB.Batch) -> SB.Batch:
        h_1 = self.fc_1.forward_view(G)                      # [B, N, 3 dqk]: one query slot, two key slots
        q_1 = self._phi_slots(h_1, 0, 1, True)[0]
        k_1 = self._phi_slots(h_1, 1, 3, False)
        h_2 = self.fc_2.forward_view(G)                      # [B, N*N, dqk + D]: one query slot, skip path
        non_att = SB.batch_like(h_2, h_2.values[..., -self.dim_in:], skip_masking=True)
        q_2 = self._phi_slots(h_2, 0, 1, True)[0]            # zero on the diagonal and on padding
        v = self._values(G)
        att_1 = self.att_2_1(q_2, k_1[0], v[0])              # set -> graph
        att_2 = SB.v2d(h_2, self.att_1_1(q_1, k_1[1], v[1]))  # set -> set, on the diagonal
        return SB.add_batch(non_att, self._project_out(q_2, [att_1, att_2]))

    def forward(self, G: DB.Batch) -> DB.Batch:
        assert isinstance(G, DB.Batch) and G.order == self.ord_in
        Sv = DB.sparse_view(G)
        if (self.ord_in, self.ord_out) == (1, 2):
            out = self._1_to_2(Sv)
        else:
            out = _SparseKernelSelfAttn.forward(self, Sv)
        assert out.order == self.ord_out
        return DB.from_sparse_view(G, out, self.ord_out)
