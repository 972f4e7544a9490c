"""FAVOR+ random-feature map (Performer) - constructor, buffers and redraw logic as in
hot_pytorch/models/common/kernel.py:12-171; the map itself is the fused HIP kernel hotk_phi_fwd/bwd."""
import math

import torch
import torch.nn as nn

from ... import ops, _lib


@torch.no_grad()
def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0, device=None):
    """[nb_rows, nb_columns] block-orthogonal Gaussian matrix (kernel.py:12-44): stacked transposed Q factors
    of square Gaussian draws, rows rescaled by chi-distributed norms (scaling 0) or sqrt(D) (scaling 1).
    Initialisation-time host/device plumbing (torch.linalg.qr), not part of the per-step path."""
    blocks = []
    rows_left = nb_rows
    while rows_left > 0:
        q, _ = torch.linalg.qr(torch.randn((nb_columns, nb_columns), device=device), mode='reduced')
        blocks.append(q.t()[:min(rows_left, nb_columns)])
        rows_left -= nb_columns
    mat = torch.cat(blocks)
    if scaling == 0:
        mult = torch.randn((nb_rows, nb_columns), device=device).norm(dim=1)
    elif scaling == 1:
        mult = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,), device=device)
    else:
        raise ValueError(f'Invalid scaling {scaling}')
    return mult[:, None] * mat


class KernelFeatureMap(nn.Module):
    """phi: [..., D] -> [..., m], phi(x) > 0.  generalized_attention=True: ReLU features + 1e-3;
    otherwise the positive softmax features exp(w.x - |x|^2/2 - stabiliser) + 1e-4, scaled by m^-1/2."""

    def __init__(self, dim_features, num_features=None, ortho_scaling=0, generalized_attention=False, kernel_fn=nn.ReLU(),
                 feature_redraw_interval=0):
        super().__init__()
        if num_features is None:
            num_features = int(dim_features * math.log(dim_features))   # m = Theta(d log d)
        self.dim_features = dim_features
        self.num_features = num_features
        self.ortho_scaling = ortho_scaling
        self.register_buffer('projection_matrix', self.create_projection())
        self.generalized_attention = generalized_attention
        if generalized_attention and not isinstance(kernel_fn, nn.ReLU):
            raise NotImplementedError('the HIP feature-map kernel implements kernel_fn = ReLU only')
        self.kernel_fn = kernel_fn
        self.feature_redraw_interval = feature_redraw_interval
        self.register_buffer('calls_since_last_redraw', torch.tensor(0))

    def create_projection(self, device=None):
        return gaussian_orthogonal_random_matrix(self.num_features, self.dim_features, self.ortho_scaling, device)

    def mode(self, is_query):
        if self.generalized_attention:
            return _lib.PHI_RELU
        return _lib.PHI_SOFTMAX_Q if is_query else _lib.PHI_SOFTMAX_K

    @torch.no_grad()
    def redraw_projection_matrix(self, device):
        self.projection_matrix.copy_(self.create_projection(device=device))

    def fix_projections_(self):
        self.feature_redraw_interval = None

    def redraw_projections(self):
        if not self.training:
            return
        if (self.feature_redraw_interval is not None) and (self.calls_since_last_redraw >= self.feature_redraw_interval):
            self.redraw_projection_matrix(self.projection_matrix.device)
            self.calls_since_last_redraw.zero_()
            return
        self.calls_since_last_redraw += 1

    def forward(self, x, is_query=False, kind=None, keep=_lib.KEEP_VALID, shard=None):
        """x [..., k*D] -> [..., k*m] (k head chunks of width D each; k = 1 reproduces the reference call).
        shard: the edge-shard context of the batch's plan (hot_amd.dist.ShardInfo) or None."""
        return ops.Phi.apply(x, self.projection_matrix, kind, keep, self.dim_features, self.mode(is_query), shard)
