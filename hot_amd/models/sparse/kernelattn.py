"""Second-order kernel (FAVOR+) self-attention layer on the sparse Batch.

Same constructor, attribute names (state_dict keys fc_1 / fc_2 / fc_v / fc_o / feature_map.*) and forward
signature as hot_pytorch/models/sparse/kernelattn.py:10-177.  NOTE the positional order (dim_v before dim_qk),
kept as in the reference even though EncLayer passes (dim_qk, dim_v) into these slots."""
import torch
import torch.nn as nn

from ...batch.sparse import Batch as B, batch_like, t, v2d, d, add_batch
from ..common.mudrop import MuDropout
from ... import ops
from .linear import Linear
from .kernelattncoef import KernelFeatureMapWrapper, KernelAttnCoef

# (n_qk1, n_qk2, n_v) per (ord_in, ord_out): number of dim_qk-wide query/key slots produced by fc_1 (node level)
# and fc_2 (entry level), and of dim_v-wide value slots (kernelattn.py:25-46)
_SLOTS = {(1, 1): (2, 0, 1), (2, 1): (7, 1, 4), (2, 2): (12, 8, 10)}


class KernelSelfAttn(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads, cfg='default', dropout=0., drop_mu=0., feature_map=None):
        super().__init__()
        assert cfg in ('default', 'local')
        self.is_local = cfg == 'local'
        self.ord_in, self.ord_out = ord_in, ord_out
        self.dim_in, self.dim_v, self.dim_qk, self.n_heads = dim_in, dim_v, dim_qk, n_heads
        self.feature_map = KernelFeatureMapWrapper(feature_map, dim_qk, n_heads)
        self.feat_dim = feature_map.num_features
        key = (ord_in, ord_out)
        if key in ((1, 0), (2, 0)):
            raise ValueError('Kernel gives no asymptotic improvement. Use softmax instead')
        if key == (1, 2):
            raise NotImplementedError('Sparse set-to-graph is inefficient; use a dense layer')
        if key not in _SLOTS:
            raise NotImplementedError
        n_qk1, n_qk2, n_v = _SLOTS[key]
        coef = lambda oq, ok: KernelAttnCoef(oq, ok, self.feat_dim, dim_v, n_heads)
        if key == (1, 1):
            self.fc_1 = Linear(1, 1, dim_in, dim_qk * n_qk1 + dim_in, cfg='light')
            self.att_1_1 = coef(1, 1)
        elif key == (2, 1):
            self.fc_1 = Linear(2, 1, dim_in, dim_qk * n_qk1 + dim_in, cfg='light')
            self.fc_2 = Linear(2, 2, dim_in, dim_qk * n_qk2, cfg='light')
            self.att_1_1, self.att_1_2 = coef(1, 1), coef(1, 2)
        else:
            self.fc_1 = Linear(2, 1, dim_in, dim_qk * n_qk1, cfg='light')
            self.fc_2 = Linear(2, 2, dim_in, dim_qk * n_qk2 + dim_in, cfg='light')
            self.att_1_1, self.att_2_1 = coef(1, 1), coef(2, 1)
            self.att_1_2, self.att_2_2 = coef(1, 2), coef(2, 2)
        self.fc_v = nn.Linear(dim_in, dim_v * n_v)
        self.fc_o = nn.Linear(dim_v * n_v, dim_in)
        self.reset_vo_parameters()
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.mu_dropout = MuDropout(p=drop_mu)
        self.use_fused = True   # fused HIP attention core when the shape allows it (see _fused_ok)
        self.storage = 'f32'    # 'bf16': wide intermediates of the fused 2->2 path in bf16 (set_storage)

    def set_storage(self, storage):
        """'f32' (the reference's precision) or 'bf16': h2 = fc_2(G), v = fc_v(values), the attention terms and their
        gradients are stored in bf16 with fp32 accumulation (ops.AttnLayer22BF16; fused 2->2 shape only -- other layers
        ignore the switch).  Parameters, layer inputs / outputs and node-level tensors stay fp32."""
        assert storage in ('f32', 'bf16')
        self.storage = storage
        return self

    def reset_vo_parameters(self):
        nn.init.xavier_normal_(self.fc_v.weight)
        nn.init.xavier_normal_(self.fc_o.weight)
        nn.init.constant_(self.fc_v.bias, 0.)
        nn.init.constant_(self.fc_o.bias, 0.)

    # -- helpers ------------------------------------------------------------------------------------
    def _phi_slots(self, H: B, lo, hi, is_query, groups=None):
        """feature-map the dim_qk-wide slots [lo, hi) of H.values and return them as a list of batches, one per slot --
        or one per GROUP of adjacent slots (groups = slot counts): slots that one attention call takes as a list arrive
        already side by side, as a view, instead of being concatenated again (a full copy of both)"""
        x = batch_like(H, H.values[..., lo * self.dim_qk:hi * self.dim_qk], skip_masking=True)
        phi = self.feature_map(x, is_query=is_query)
        F = self.feat_dim * self.n_heads
        if groups is None:
            return [batch_like(H, v, skip_masking=True) for v in phi.values.split(F, -1)]
        assert sum(groups) == hi - lo
        return [batch_like(H, v, skip_masking=True) for v in phi.values.split([g * F for g in groups], -1)]

    def _values(self, G: B):
        v = batch_like(G, ops.linear_rows(G.values, self.fc_v), skip_masking=False)
        return [batch_like(G, x, skip_masking=True) for x in v.values.split(self.dim_v, -1)]

    def _project_out(self, H: B, att_list):
        n_v = self.fc_o.in_features // self.dim_v
        if len(att_list) != n_v:
            # the reference sizes fc_o for every term and then concatenates only the local ones: it fails with
            # the same shape error for cfg='local' (kernelattn.py:49-50,109-116,146-158)
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied: {len(att_list)} attention terms "
                               f"for an fc_o built for {n_v} (cfg='local' is unusable in the reference as well)")
        x = torch.cat(self.mu_dropout([a.values for a in att_list]), -1)
        return batch_like(H, ops.linear_rows(self.dropout(x), self.fc_o), skip_masking=False)

    # -- order-specific forwards ------------------------------------------------------------------------
    def _1_to_1(self, G: B):
        h_1 = self.fc_1(G)
        non_att = batch_like(h_1, h_1.values[..., -self.dim_in:], skip_masking=True)
        q_1 = self._phi_slots(h_1, 0, 1, True)[0]
        k_1 = self._phi_slots(h_1, 1, 2, False)[0]
        att_1 = self.att_1_1(q_1, k_1, self._values(G)[0])
        att = batch_like(G, ops.linear_rows(self.dropout(att_1.values), self.fc_o), skip_masking=False)
        return add_batch(non_att, att)

    def _2_to_1(self, G: B):
        h_1 = self.fc_1(G)
        non_att = batch_like(h_1, h_1.values[..., -self.dim_in:], skip_masking=True)
        q_1 = self._phi_slots(h_1, 0, 4, True, groups=(2, 1, 1))       # [slots 0-1 | 2 | 3]
        k_1 = self._phi_slots(h_1, 4, 7, False, groups=(2, 1))         # [slots 0-1 | 2]
        h_2 = self.fc_2(G)
        k_2 = self._phi_slots(h_2, 0, 1, False)[0]
        v = self._values(G)
        # graph -> set:  O_j = Q_j . sum_i K_i (x) V_ij   for V and V^T
        att = list(self.att_1_1(q_1[0], k_1[0], [v[0], t(v[1])], diagonal=(1, 2)))
        if not self.is_local:
            att.append(self.att_1_1(q_1[1], k_1[1], d(v[2])))      # set -> set (global)
            att.append(self.att_1_2(q_1[2], k_2, v[3]))            # graph -> set (global)
        return add_batch(non_att, self._project_out(h_1, att))

    def _fused_ok(self):
        """the fused attention core covers the headline shape: 4 heads x 16 features (one wavefront per slot),
        head width 8, value width 8, ReLU features, every term present.  Dropout and MuDropout sit between the core and
        fc_o (kernelattn.py:70-72) and are applied there by ops.AttnCore22, so training configs stay on this path."""
        fm = self.feature_map.feature_map
        return (self.use_fused and not self.is_local and fm.generalized_attention and self.n_heads == 4
                and self.dim_qk == 32 and self.dim_v == 32 and self.feat_dim == 16)

    def _2_to_2(self, G: B):
        sharded = getattr(G.plan, 'shard', None) is not None     # (edge-shard mode runs the fp32 core in phases)
        if self._fused_ok() and (self.storage == 'f32' or sharded):
            # the three input projections as one autograd node (the layer input's gradient is written once), then the
            # fused core with fc_o; padding rows are ignored by the kernels: no masking pass
            b1 = self.fc_1.bias.bias[0] if hasattr(self.fc_1, 'bias') else None
            b2 = self.fc_2.bias.bias if hasattr(self.fc_2, 'bias') else (None, None)
            h_1, h_2, v = ops.Proj22.apply(G.values, self.fc_1.weight.weight, b1, self.fc_2.weight.weight, b2[0], b2[1],
                                           self.fc_v.weight, self.fc_v.bias, G.plan, sum(G.n_nodes) != G.plan.Nn)
            drop_p = self.dropout.p if self.training else 0.
            mu_p = self.mu_dropout.p if self.training else 0.
            out = ops.AttnCore22.apply(h_1, h_2, v, self.feature_map.feature_map.projection_matrix,
                                       self.fc_o.weight, self.fc_o.bias, G.plan, drop_p, mu_p)
            return batch_like(G, out, skip_masking=True)
        h_1 = self.fc_1(G)                      # [B, N, 12 dqk]: 4 query slots, 8 key slots
        if self.storage == 'bf16' and self._fused_ok():
            drop_p = self.dropout.p if self.training else 0.
            mu_p = self.mu_dropout.p if self.training else 0.
            fb = self.fc_2.bias.bias if hasattr(self.fc_2, 'bias') else (None, None)
            out = ops.AttnLayer22BF16.apply(G.values, h_1.values, self.fc_2.weight.weight, fb[0], fb[1],
                                            self.fc_v.weight, self.fc_v.bias, self.fc_o.weight, self.fc_o.bias,
                                            self.feature_map.feature_map.projection_matrix, G.plan, drop_p, mu_p)
            return batch_like(G, out, skip_masking=True)
        h_2 = self.fc_2(G)                      # [B, |E|, 8 dqk + D]: 6 query slots, 2 key slots, skip path
        non_att = batch_like(h_2, h_2.values[..., -self.dim_in:], skip_masking=True)
        # slots that one attention call takes as a pair stay side by side (views of the feature-map output)
        q_1 = self._phi_slots(h_1, 0, 4, True, groups=(2, 1, 1))             # [0-1 | 2 | 3]
        k_1 = self._phi_slots(h_1, 4, 12, False, groups=(2, 2, 2, 1, 1))     # [0-1 | 2-3 | 4-5 | 6 | 7]
        q_2 = self._phi_slots(h_2, 0, 6, True, groups=(2, 2, 1, 1))          # [0-1 | 2-3 | 4 | 5]
        k_2 = self._phi_slots(h_2, 6, 8, False)
        v = self._values(G)
        # node outputs from column-segmented sums, placed on the diagonal
        a1, a2 = self.att_1_1(q_1[0], k_1[0], [v[0], t(v[1])], diagonal=(1, 2))
        # entry outputs reading the column sum (a3, a5) or the row-node's sum (a4, a6)
        a3, a5 = self.att_2_1(q_2[0], k_1[1], [v[2], t(v[3])], diagonal=(2, 3))
        a4, a6 = self.att_2_1(q_2[1], k_1[2], [t(v[4]), v[5]], diagonal=(1, 3))
        att = [v2d(G, a1), v2d(G, a2), a3, a4, a5, a6]
        if not self.is_local:
            att.append(v2d(G, self.att_1_1(q_1[1], k_1[3], d(v[6]))))    # set -> set
            att.append(v2d(G, self.att_1_2(q_1[2], k_2[0], v[7])))       # graph -> set
            att.append(self.att_2_1(q_2[2], k_1[4], d(v[8])))            # set -> graph
            att.append(self.att_2_2(q_2[3], k_2[1], v[9]))               # graph -> graph
        return add_batch(non_att, self._project_out(G, att))

    def forward(self, G: B):
        assert G.order == self.ord_in
        key = (self.ord_in, self.ord_out)
        if key == (1, 1):
            G_att = self._1_to_1(G)
        elif key == (2, 1):
            G_att = self._2_to_1(G)
        elif key == (2, 2):
            G_att = self._2_to_2(G)
        else:
            raise NotImplementedError('Currently supports up to second-order invariance only')
        assert G_att.order == self.ord_out
        return G_att
