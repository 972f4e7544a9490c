"""Encoder / EncLayer for the sparse and dense layouts.

Interface contract (constructor arguments, attribute names -> state_dict keys, forward signature) follows
hot_pytorch/models/encoder.py:20-96; the assembly is table-driven (models/_stack.py).  Built here: kernel attention
(att_cfg 'kernel' / 'generalized_kernel') for output orders >= 1 and, as in the reference (encoder.py:34-35), the softmax
readout SelfAttn(., 0) for an order-0 output, which is linear in |E|.  Quadratic softmax attention for output orders >= 1
(att_cfg='default') is outside the linear-cost path: NotImplementedError.  The dense layout has the same readout
(models/dense/attn.py)."""
import torch.nn as nn

from . import _stack


class EncLayer(nn.Module):
    """pre-norm attention block followed by a pre-norm feed-forward block with a skip connection around the latter only
    (`residual` False, as shipped)"""

    def __init__(self, ord_in, ord_out, dim_in, dim_qk, dim_v, dim_ff, n_heads, cfg='default', att_cfg='default',
                 dropout=0., drop_mu=0., feature_map=None, sparse=True):
        super().__init__()
        if cfg not in ('default', 'local') or att_cfg not in ('default', 'kernel', 'generalized_kernel'):
            raise AssertionError(f'unknown configuration cfg={cfg!r}, att_cfg={att_cfg!r}')
        lay = _stack.layout_of(sparse)
        self.sparse = sparse
        self.residual = False
        self.add = lay.add()
        self.ln = lay.apply(nn.LayerNorm(dim_in))
        self.attn = self._attention(lay, ord_in, ord_out, dim_in, dim_qk, dim_v, n_heads, cfg, att_cfg, dropout, drop_mu,
                                    feature_map)
        self.ffn = _stack.pre_norm_block(
            lay.apply(nn.LayerNorm(dim_in)),
            lay.linear(ord_out, ord_out, dim_in, dim_ff, cfg='light'),
            lay.apply(nn.GELU(), skip_masking=True),
            lay.linear(ord_out, ord_out, dim_ff, dim_in, cfg='light'),
            lay.apply(nn.Dropout(dropout, inplace=True), skip_masking=True))

    @staticmethod
    def _attention(lay, ord_in, ord_out, dim_in, dim_qk, dim_v, n_heads, cfg, att_cfg, dropout, drop_mu, feature_map):
        # NOTE the attention classes take (dim_v, dim_qk) in that order while this constructor receives (dim_qk, dim_v)
        # and passes them through positionally -- the reference does the same (encoder.py:35,37), so the two widths are
        # swapped inside; checkpoints depend on it.
        if ord_out == 0:
            if lay.readout_attn is None:
                raise NotImplementedError(f'no order-0 softmax readout in the {lay.name} layout')
            return lay.readout_attn(ord_in, 0, dim_in, dim_qk, dim_v, n_heads, cfg, dropout, drop_mu)
        if att_cfg == 'default':
            raise NotImplementedError('softmax SelfAttn with an order >= 1 output is quadratic in |E| and not part of '
                                      'the linear-cost path (use att_cfg="kernel" / "generalized_kernel")')
        return lay.kernel_attn(ord_in, ord_out, dim_in, dim_qk, dim_v, n_heads, cfg, dropout, drop_mu, feature_map)

    def forward(self, G):
        attended = self.attn(self.ln(G))
        if self.residual:
            attended = self.add(G, attended)
        return self.add(attended, self.ffn(attended))


class Encoder(nn.Module):
    """input projection -> EncLayer per consecutive pair of orders -> LayerNorm + nn.Linear head"""

    def __init__(self, ord_in, ord_out, ord_hidden: list, dim_in, dim_out, dim_hidden, dim_qk, dim_v, dim_ff, n_heads,
                 readout_dim_qk, readout_dim_v, readout_n_heads, enc_cfg='default', att_cfg='default',
                 drop_input=0., dropout=0., drop_mu=0., sparse=True):
        super().__init__()
        lay = _stack.layout_of(sparse)
        self.sparse = sparse
        self._batch_type = lay.batch_type
        self.input = nn.Sequential(lay.linear(ord_in, ord_in, dim_in, dim_hidden, cfg='light'),
                                   lay.apply(nn.Dropout(drop_input, inplace=True), skip_masking=True))
        self.feature_map, self.skip_redraw_projections = _stack.shared_feature_map(att_cfg, dim_qk, n_heads)
        orders = [ord_in] + list(ord_hidden) + [ord_out]
        body = []
        for src, dst in zip(orders[:-1], orders[1:]):
            # an order-0 layer is the readout and has its own attention widths
            qk, v, heads = (readout_dim_qk, readout_dim_v, readout_n_heads) if dst == 0 else (dim_qk, dim_v, n_heads)
            body.append(EncLayer(src, dst, dim_hidden, qk, v, dim_ff, heads, enc_cfg, att_cfg, dropout, drop_mu,
                                 self.feature_map, sparse))
        self.layers = nn.Sequential(*body)
        self.output = lay.apply(nn.Sequential(nn.LayerNorm(dim_hidden), nn.Linear(dim_hidden, dim_out)))

    def forward(self, G):
        if not isinstance(G, self._batch_type):
            raise AssertionError(f'a {"sparse" if self.sparse else "dense"} Batch is expected, got {type(G).__name__}')
        _stack.maybe_redraw(self)
        return self.output(self.layers(self.input(G)))
