"""Shared assembly helpers for the model stacks (Encoder / EncLayer / MLP in the sparse, dense and k-uniform layouts).

What is fixed by checkpoint compatibility with the reference (hot_pytorch/models/encoder.py:20-96, mlp.py:15-40,
extension.py:20-124) is the set of attribute names and the position of parameterised modules inside each nn.Sequential --
nothing else.  This module holds the pieces the three stacks share, written once:

  * a `Layout` record naming the building blocks of a data layout (which Linear, which pointwise wrapper, which attention);
  * `orders_and_widths`, the (order, width) schedule of a stack from its constructor arguments;
  * `pre_norm_block`, the LayerNorm -> Linear -> GELU -> Linear -> Dropout feed-forward block whose parameterised members
    sit at Sequential positions 0, 1 and 3 (keys `ffn.0.*`, `ffn.1.*`, `ffn.3.*`);
  * `shared_feature_map`, the one FAVOR+ map an encoder hands to all its kernel-attention layers, and `maybe_redraw`.
"""
from collections import namedtuple

import torch.nn as nn

from .common.kernel import KernelFeatureMap

Layout = namedtuple('Layout', 'name linear apply nonlinear add kernel_attn readout_attn batch_type')


def sparse_layout():
    from .sparse.linear import Linear
    from .sparse.f import Apply, Nonlinear, Add
    from .sparse.kernelattn import KernelSelfAttn
    from .sparse.attn import SelfAttn
    from ..batch.sparse import Batch
    return Layout('sparse', Linear, Apply, Nonlinear, Add, KernelSelfAttn, SelfAttn, Batch)


def dense_layout():
    from .dense.linear import Linear
    from .dense.f import Apply, Nonlinear, Add
    from .dense.kernelattn import KernelSelfAttn
    from .dense.attn import SelfAttn
    from ..batch.dense import Batch
    return Layout('dense', Linear, Apply, Nonlinear, Add, KernelSelfAttn, SelfAttn, Batch)


def layout_of(sparse: bool) -> Layout:
    return sparse_layout() if sparse else dense_layout()


def orders_and_widths(ord_in, ord_hidden, ord_out, dim_in, dim_hidden, dim_out):
    """[(order_k, width_k)] for k = 0 .. n_layers: the tensors between the layers of a stack.  dim_hidden may be one int
    (all hidden widths equal) or a list with one width per hidden tensor."""
    hidden = list(ord_hidden)
    widths = list(dim_hidden) if isinstance(dim_hidden, (list, tuple)) else [dim_hidden] * len(hidden)
    if len(widths) != len(hidden):
        raise ValueError(f'{len(hidden)} hidden orders but {len(widths)} hidden widths')
    return list(zip([ord_in] + hidden + [ord_out], [dim_in] + widths + [dim_out]))


def pre_norm_block(norm, first, act, second, drop):
    """feed-forward block; parameterised members at positions 0, 1, 3"""
    return nn.Sequential(norm, first, act, second, drop)


def shared_feature_map(att_cfg, dim_qk, n_heads):
    """(feature map or None, skip_redraw flag): kernel attention variants share ONE projection per encoder, redrawn on
    every training forward unless the flag is set by the user (encoder.py:71-75)"""
    if att_cfg not in ('kernel', 'generalized_kernel'):
        return None, True
    width = dim_qk // n_heads if dim_qk >= n_heads else 1
    return KernelFeatureMap(width, generalized_attention=(att_cfg == 'generalized_kernel')), False


def maybe_redraw(module):
    if module.feature_map is not None and not module.skip_redraw_projections:
        module.feature_map.redraw_projections()
