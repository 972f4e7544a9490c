"""Multi-GPU helpers for the graph-minibatch mode: graphs are independent units, so the layer
needs no exchange in forward or backward; only parameter gradients cross GPUs (one flat all-reduce over RCCL,
`backend="nccl"` on ROCm).  Stands in for the Lightning DDPPlugin of regression/main/entry.py:110-113.

One process per GPU; every function takes the process group explicitly or uses the default one."""
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def partition_graphs(sizes: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous graph ranges [(lo, hi)) per rank, balanced by total entry count.

    Rank r gets the graphs whose cumulative entry count (exclusive of the graph itself) falls into
    [r*T/world, (r+1)*T/world).  Contiguity keeps the padded Batch of a rank a slice of the global one."""
    assert world >= 1
    sizes = np.asarray([int(s) for s in sizes], dtype=np.int64)
    prefix = np.concatenate([[0], np.cumsum(sizes)])            # entries before graph g
    total = int(prefix[-1])
    cuts = [0]
    for r in range(1, world):
        # first graph whose start is at or beyond rank r's share of the entries
        g = int(np.searchsorted(prefix[:-1], -(-r * total // world), side='left'))
        cuts.append(max(g, cuts[-1]))
    cuts.append(len(sizes))
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def _all_reduce(t: torch.Tensor, op, group=None) -> torch.Tensor:
    """all_reduce in place; a gloo group (CPU rehearsal of the multi-GPU path) gets device tensors staged through
    the host, since gloo collectives on HIP tensors are not usable on this stack."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> None:
    """Sum (or average) the gradients of `params` over the group with ONE collective on a flat bucket
    (the C1 layer has 80 k parameters = 320 KB: latency-bound, so a single call beats per-tensor calls).
    Parameters without a gradient contribute zeros, so every rank issues an identical collective."""
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    _all_reduce(flat, dist.ReduceOp.SUM, group)
    if average:
        flat /= dist.get_world_size(group)
    ofs = 0
    for p in params:
        n = p.numel()
        g = flat[ofs:ofs + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        ofs += n


def max_over_ranks(value: float, device, group=None) -> float:
    """MAX of a host scalar over the group (bench timing: the slowest rank defines the step)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX, group)
    return float(t.item())
