"""The extension for k-uniform hyperedge prediction: a set of N nodes in, B k-node
hyperedges out.  Same class names, constructors, attribute names (state_dict keys) and forward signatures as
hot_pytorch/models/extension.py:14-124; the layers are hot_amd.models.uniform (HIP kernels)."""
from typing import Union, List

import torch
import torch.nn as nn

from .uniform.kernelattn import KernelSelfAttn
from .uniform.f import Nonlinear, Add
from .uniform.linear import Linear
from .common.kernel import KernelFeatureMap


def _check_indices(indices: torch.LongTensor):
    """every hyperedge must name k distinct nodes (extension.py:14-17, there via masksum.count_unique)"""
    if indices is not None:
        k = indices.size(1)
        s = indices.sort(dim=1).values
        assert k < 2 or bool((s[:, 1:] != s[:, :-1]).all()), \
            'This extension is only for 1->k-uniform, but given indices contain loops'


class EncLayer(nn.Module):
    def __init__(self, ord_in, ord_out, dim_in, dim_qk, dim_v, dim_ff, n_heads, cfg='default', att_cfg='default',
                 dropout=0., drop_mu=0., feature_map=None, sparse=False):
        super().__init__()
        assert cfg == 'default' and att_cfg == 'generalized_kernel'
        assert ord_in == 1
        self.ord_out = ord_out
        self.add = Add()
        self.ln = nn.LayerNorm(dim_in)
        # positional (dim_qk, dim_v) into (dim_v, dim_qk) slots, as in the reference (extension.py:29)
        self.attn = KernelSelfAttn(1, ord_out, dim_in, dim_qk, dim_v, n_heads, cfg, dropout, drop_mu, feature_map)
        self.residual = False
        self.ffn = nn.Sequential(
            nn.LayerNorm(dim_in),
            nn.Linear(dim_in, dim_ff),
            nn.GELU(),
            nn.Linear(dim_ff, dim_in),
            nn.Dropout(dropout, inplace=True)
        )

    def forward(self, x: torch.Tensor, indices: torch.LongTensor = None):
        h = self.ln(x)
        h = self.attn(h, indices)
        x = self.add(x, h) if self.residual else h
        h = self.ffn(x)
        return self.add(x, h)


class Encoder(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden, dim_qk, dim_v, dim_ff, n_heads,
                 readout_dim_qk, readout_dim_v, readout_n_heads, enc_cfg='default', att_cfg='generalized_kernel',
                 drop_input=0., dropout=0., drop_mu=0., sparse=False):
        super().__init__()
        assert enc_cfg == 'default' and att_cfg == 'generalized_kernel'
        assert ord_out > 1
        self.input = nn.Sequential(
            Linear(1, 1, dim_in, dim_hidden, cfg='light'),
            nn.Dropout(drop_input, inplace=True)
        )
        feat_dim = dim_qk // n_heads if dim_qk >= n_heads else 1
        self.feature_map = KernelFeatureMap(feat_dim, generalized_attention=(att_cfg == 'generalized_kernel'))
        self.skip_redraw_projections = False
        layers = []
        for ord1, ord2 in zip([ord_in] + ord_hidden, ord_hidden + [ord_out]):
            assert ord1 == 1
            layers.append(EncLayer(1, ord2, dim_hidden, dim_qk, dim_v, dim_ff, n_heads, enc_cfg, att_cfg, dropout, drop_mu,
                                   self.feature_map, sparse))
        self.layers = nn.ModuleList(layers)
        self.output = nn.Sequential(
            nn.LayerNorm(dim_hidden),
            nn.Linear(dim_hidden, dim_out)
        )

    def forward(self, x: torch.Tensor, indices: torch.LongTensor):
        _check_indices(indices)
        if (self.feature_map is not None) and (not self.skip_redraw_projections):
            self.feature_map.redraw_projections()
        x = self.input(x)
        for layer in self.layers:
            x = layer(x) if layer.ord_out == 1 else layer(x, indices)
        return self.output(x)


class MLP(nn.Module):
    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden: Union[List, int], f='relu', dropout=0.,
                 sparse=False):
        super().__init__()
        assert ord_out > 1
        if not isinstance(dim_hidden, list):
            dim_hidden = [dim_hidden] * len(ord_hidden)
        ords = [ord_in] + list(ord_hidden) + [ord_out]
        dims = [dim_in] + list(dim_hidden) + [dim_out]
        layers = []
        n_lin = len(ords) - 1
        for idx in range(n_lin):
            assert ords[idx] == 1
            layer = nn.ModuleList()
            layer.append(Linear(1, ords[idx + 1], dims[idx], dims[idx + 1], bias=True, cfg='default', normalize=True))
            if idx < n_lin - 1:
                layer.append(Nonlinear(f))
                layer.append(nn.Dropout(p=dropout))
            layers.append(layer)
        self.layers = nn.ModuleList(layers)

    def forward(self, x: torch.Tensor, indices: torch.LongTensor):
        _check_indices(indices)
        for layer in self.layers:
            x = layer[0](x) if layer[0].weight.ord_out == 1 else layer[0](x, indices)
            if len(layer) > 1:
                x = layer[1](x)
                x = layer[2](x)
        return x
