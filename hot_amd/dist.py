"""Multi-GPU helpers.  One process per GPU; `backend="nccl"` is RCCL on ROCm.

Graph-minibatch mode: graphs are independent units, so the layer needs no exchange in forward or backward; only
parameter gradients cross GPUs (one flat all-reduce).  Stands in for the Lightning DDPPlugin of
regression/main/entry.py:110-113.

Edge-shard mode (one giant graph): every rank holds ALL nodes (diagonal entries, N << E) and a share of the
off-diagonal entries in which (i,j) and (j,i) are co-located (`shard_graph`).  Node-level work is replicated; the
only exchanges are all-reduces of the key-value summaries built from entries (per-node and per-graph
sum phi(K) (x) [V|1] in forward, their gradients in backward) and the final parameter-gradient all-reduce.
Replicated diagonal entries enter entry-level sums with weight 1/world so that the reduced sums count them once
(bit-exact for power-of-two worlds; any other world size adds one fp32 rounding of 1/world per diagonal term, ~6e-8
relative: tests/test_gpu_edge_shard.py runs 3 ranks against the oracle at the same 1e-5 bar); the caller weights replicated output rows by `ShardInfo.entry_weight` in its loss.

Every function takes the process group explicitly or uses the default one."""
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


class _Done:
    """handle of a collective that has already completed"""

    @staticmethod
    def wait():
        return True


class ShardInfo:
    """Edge-shard context attached to a Plan (`plan.shard`): ops that sum over entries all-reduce their summaries."""

    def __init__(self, rank: int, world: int, entry_weight: torch.Tensor, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.entry_weight = entry_weight      # [R] fp32: 1/world on diagonal rows, 1 on off-diagonal rows, 0 on padding

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            _all_reduce(t, dist.ReduceOp.SUM, self.group)
        return t

    def all_reduce_start_(self, t: torch.Tensor):
        """SUM all-reduce of t in place, not waited for: returns a handle whose .wait() orders the current stream behind
        it.  With RCCL the collective runs on the backend's own stream, so kernels launched between start and wait --
        which must not touch t -- overlap it (the fused core's per-graph half runs under the per-node sums' exchange:
        ops.AttnCore22).  A gloo group (host-staged rehearsal) or a world of one completes at once."""
        if self.world > 1 and t.is_cuda and dist.get_backend(self.group) != 'gloo':
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.all_reduce_(t)
        return _Done()

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            _all_reduce(t, dist.ReduceOp.MAX, self.group)
        return t


def shard_owner(i: torch.Tensor, j: torch.Tensor, world: int) -> torch.Tensor:
    """owner rank of off-diagonal entry (i,j): a function of the unordered pair, so transposes stay together"""
    lo, hi = torch.minimum(i, j), torch.maximum(i, j)
    return (lo * 1000003 + hi * 7919) % world


def shard_graph(node_feature: torch.Tensor, edge_index: torch.Tensor, edge_feature: torch.Tensor, world: int, rank: int,
                group=None):
    """Local Batch of `rank` for ONE graph given PyG-style (node_feature [N,Dv], edge_index [2,E], edge_feature [E,De]):
    all N nodes + the owned edges (original order kept).  Returns (Batch, owned edge positions [E_local])."""
    from .batch.sparse import make_batch_concatenated
    own = (shard_owner(edge_index[0], edge_index[1], world) == rank).nonzero(as_tuple=True)[0]
    G = make_batch_concatenated(node_feature, edge_index[:, own], edge_feature[own], [node_feature.size(0)], [own.numel()])
    plan = G.plan
    w = torch.where(plan.kind == 1, torch.full((plan.R,), 1.0 / world, device=plan.device),
                    (plan.kind == 2).to(torch.float32))
    plan.shard = ShardInfo(rank, world, w, group)
    return G, own
