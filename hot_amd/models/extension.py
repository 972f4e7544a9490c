"""1 -> k-uniform hyperedge extension: N node features in, one output row per k-node hyperedge.

Interface contract of hot_pytorch/models/extension.py:14-124 (class names, constructor arguments, attribute names ->
state_dict keys, forward(x, indices)); layers from hot_amd.models.uniform (HIP kernels), assembly shared with the other
stacks (models/_stack.py).  Tensors are plain [N, D] / [B, D] here, so the pointwise members are torch modules."""
import torch
import torch.nn as nn

from . import _stack
from .uniform.kernelattn import KernelSelfAttn
from .uniform.f import Nonlinear, Add
from .uniform.linear import Linear


def _check_indices(indices):
    """a hyperedge names k DISTINCT nodes (extension.py:14-17): after sorting each row no two neighbours are equal"""
    if indices is None or indices.size(1) < 2:
        return
    ordered, _ = torch.sort(indices, dim=1)
    repeated = (ordered[:, 1:] == ordered[:, :-1]).any()
    assert not bool(repeated), 'hyperedges with a repeated node are outside the 1 -> k-uniform extension'


def _only(value, allowed, what):
    assert value == allowed, f'the k-uniform extension supports {what}={allowed!r} only (got {value!r})'


class EncLayer(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_qk, dim_v, dim_ff, n_heads, cfg='default', att_cfg='default',
                 dropout=0., drop_mu=0., feature_map=None, sparse=False):
        super().__init__()
        _only(cfg, 'default', 'cfg')
        _only(att_cfg, 'generalized_kernel', 'att_cfg')
        _only(ord_in, 1, 'ord_in')
        self.ord_out = ord_out
        self.residual = False
        self.add = Add()
        self.ln = nn.LayerNorm(dim_in)
        # (dim_qk, dim_v) go positionally into the attention's (dim_v, dim_qk) parameters, as in extension.py:29
        self.attn = KernelSelfAttn(1, ord_out, dim_in, dim_qk, dim_v, n_heads, cfg, dropout, drop_mu, feature_map)
        self.ffn = _stack.pre_norm_block(nn.LayerNorm(dim_in), nn.Linear(dim_in, dim_ff), nn.GELU(),
                                         nn.Linear(dim_ff, dim_in), nn.Dropout(dropout, inplace=True))

    def forward(self, x, indices=None):
        attended = self.attn(self.ln(x), indices)
        if self.residual:
            attended = self.add(x, attended)
        return self.add(attended, self.ffn(attended))


class Encoder(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden, dim_qk, dim_v, dim_ff, n_heads,
                 readout_dim_qk, readout_dim_v, readout_n_heads, enc_cfg='default', att_cfg='generalized_kernel',
                 drop_input=0., dropout=0., drop_mu=0., sparse=False):
        super().__init__()
        _only(enc_cfg, 'default', 'enc_cfg')
        _only(att_cfg, 'generalized_kernel', 'att_cfg')
        assert ord_out > 1, 'the output of this extension is a k-uniform hyperedge (k >= 2)'
        self.input = nn.Sequential(Linear(1, 1, dim_in, dim_hidden, cfg='light'), nn.Dropout(drop_input, inplace=True))
        self.feature_map, self.skip_redraw_projections = _stack.shared_feature_map(att_cfg, dim_qk, n_heads)
        orders = [ord_in] + list(ord_hidden) + [ord_out]
        assert all(o == 1 for o in orders[:-1]), 'every layer input of this extension is a set (order 1)'
        self.layers = nn.ModuleList(
            EncLayer(1, dst, dim_hidden, dim_qk, dim_v, dim_ff, n_heads, enc_cfg, att_cfg, dropout, drop_mu,
                     self.feature_map, sparse) for dst in orders[1:])
        self.output = nn.Sequential(nn.LayerNorm(dim_hidden), nn.Linear(dim_hidden, dim_out))

    def forward(self, x, indices):
        _check_indices(indices)
        _stack.maybe_redraw(self)
        x = self.input(x)
        for layer in self.layers:
            x = layer(x, indices if layer.ord_out != 1 else None)
        return self.output(x)


class MLP(nn.Module):
    """`layers` is a ModuleList of per-layer ModuleLists [Linear(cfg='default'), (Nonlinear, Dropout)]: keys `layers.k.0.*`"""

    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden, f='relu', dropout=0., sparse=False):
        super().__init__()
        assert ord_out > 1, 'the output of this extension is a k-uniform hyperedge (k >= 2)'
        plan = _stack.orders_and_widths(ord_in, ord_hidden, ord_out, dim_in, dim_hidden, dim_out)
        assert all(o == 1 for o, _ in plan[:-1]), 'every layer input of this extension is a set (order 1)'
        self.layers = nn.ModuleList()
        for k, ((_, w_src), (o_dst, w_dst)) in enumerate(zip(plan[:-1], plan[1:])):
            members = [Linear(1, o_dst, w_src, w_dst, bias=True, cfg='default', normalize=True)]
            if k + 2 < len(plan):
                members += [Nonlinear(f), nn.Dropout(p=dropout)]
            self.layers.append(nn.ModuleList(members))

    def forward(self, x, indices):
        _check_indices(indices)
        for members in self.layers:
            lin = members[0]
            x = lin(x) if lin.weight.ord_out == 1 else lin(x, indices)
            for post in members[1:]:
                x = post(x)
        return x
