"""Varying-size set helpers; masking convention: data == True, padding == False
(mirror of hot_pytorch/utils/set.py:24-48)."""
from typing import Tuple

import torch

from .. import ops
from .._lib import KEEP_DIAG

MASK = 0


def get_mask(sizes: torch.LongTensor) -> torch.BoolTensor:
    """[B] sizes -> [B, max] bool (utils/set.py:36-38).  Index bookkeeping only."""
    idx = torch.arange(int(sizes.max()) if sizes.numel() else 0, device=sizes.device)
    return idx[None, :] < sizes[:, None]


def masked_fill(x: torch.Tensor, mask: torch.BoolTensor, value: float) -> torch.Tensor:
    """Copy of x with rows outside mask set to value (utils/set.py:47-48); HIP kernel hotk_mask_rows."""
    kind = mask.reshape(-1).view(torch.uint8)   # 0 = padding, 1 = data
    return ops.MaskRows.apply(x, kind, KEEP_DIAG, float(value))


def to_batch(x: torch.Tensor, sizes: list, mask: torch.BoolTensor, value: float = MASK) -> torch.Tensor:
    """Concatenated rows [sum(sizes), D] -> padded [B, max, D] (utils/set.py:41-44)."""
    B, N = mask.shape
    dev = x.device
    sz = torch.as_tensor(sizes, dtype=torch.long, device=dev)
    ofs = torch.cumsum(sz, 0) - sz
    idx = (torch.arange(N, device=dev)[None, :] + ofs[:, None]).to(torch.int32)
    idx = torch.where(mask, idx, torch.full_like(idx, -1)).reshape(-1)
    inv = torch.nonzero(mask.reshape(-1)).squeeze(-1).to(torch.int32)
    out = ops.GatherRows.apply(x, idx, inv, B * N).view(B, N, x.size(-1))
    if value != 0:
        out = out + (~mask).unsqueeze(-1).to(x.dtype) * value
    return out


def to_masked_batch(x: torch.Tensor, sizes: list, value: float = MASK) -> Tuple[torch.Tensor, torch.BoolTensor]:
    """utils/set.py:24-33."""
    mask = get_mask(torch.tensor(sizes, dtype=torch.long, device=x.device))
    return to_batch(x, sizes, mask, value), mask
