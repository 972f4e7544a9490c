import torch
import torch.nn as nn


class MuDropout(nn.Module):
    """Per-term ("mu") dropout over a list of equally shaped tensors: each term of each batch element is
    kept with probability 1-p and rescaled (hot_pytorch/models/common/mudrop.py:5-20).  Identity in eval
    mode or when p == 0, which is the configuration every shipped experiment and the benchmark use."""

    def __init__(self, p=0.):
        super().__init__()
        self.p = p

    def forward(self, x_list):
        if not self.training or self.p == 0.:
            return x_list
        x0 = x_list[0]
        shape = [len(x_list), len(x0)] + [1] * (x0.dim() - 1)
        keep = (torch.rand(shape, device=x0.device) > self.p).to(x0.dtype) / (1 - self.p)
        return [x * k for x, k in zip(x_list, keep)]
