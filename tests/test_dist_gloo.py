"""Graph-minibatch data parallelism on CPU: world size 2 over gloo.

The N>1 path of the product is partition_graphs -> per-rank layer forward/backward (no exchange) ->
allreduce_gradients.  The HIP kernels cannot run in this container, so the per-rank compute here is the oracle
(tests may use it); what is under test is the host logic of hot_amd.dist: the partition covers every graph once
and is balanced, and the reduced parameter gradients equal the single-process gradients of the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hot_amd.dist import partition_graphs, allreduce_gradients, max_over_ranks  # noqa: E402
from oracle import hot_oracle as O  # noqa: E402

N_GRAPHS = 6


def _graphs():
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    out = []
    for n in (12, 20, 9, 16, 14, 11):
        ei = torch.from_numpy(O.ba_edges(n, 3, rng))
        out.append((torch.randn(n, 16, generator=g), ei, torch.randn(ei.size(1), 16, generator=g)))
    return out


def _state():
    from hot_amd.models.common import KernelFeatureMap
    from hot_amd.models.sparse import KernelSelfAttn
    torch.manual_seed(0)
    layer = KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=True))
    return {k: v.detach().clone() for k, v in layer.state_dict().items()}


def _loss_grads(graphs, sd, names):
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
    full = dict(sd)
    full.update(params)
    G = O.build_batch_from_lists([a for a, _, _ in graphs], [b for _, b, _ in graphs], [c for _, _, c in graphs])
    y = O.kernel_self_attn(G, full, 2, 2, 32, 32, 32, 4, 'default', True)
    y.values.sum().backward()
    return params


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        graphs, sd = _graphs(), _state()
        names = [k for k, v in sd.items() if v.dtype == torch.float32 and 'projection' not in k]
        sizes = [a.size(0) + c.size(0) for a, _, c in graphs]
        lo, hi = partition_graphs(sizes, world)[rank]
        params = _loss_grads(graphs[lo:hi], sd, names)
        allreduce_gradients(params.values())
        slowest = max_over_ranks(float(rank + 1), 'cpu')
        q.put((rank, {k: p.grad.clone().numpy() for k, p in params.items()}, slowest, (lo, hi)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_partition_covers_and_balances():
    for sizes, world in (([654] * 15291, 8), ([5, 900, 3, 3, 40, 41, 7], 3), ([10], 4), ([], 2)):
        parts = partition_graphs(sizes, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(sizes)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        assert all(lo <= hi for lo, hi in parts)
    parts = partition_graphs([654] * 15291, 8)
    loads = [hi - lo for lo, hi in parts]
    assert max(loads) - min(loads) <= 1


def test_two_rank_gradient_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    graphs, sd = _graphs(), _state()
    names = [k for k, v in sd.items() if v.dtype == torch.float32 and 'projection' not in k]
    ref = _loss_grads(graphs, sd, names)
    ranges = sorted(r[3] for r in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == N_GRAPHS and ranges[0][1] == ranges[1][0]
    for rank, grads, slowest, _ in results:
        assert slowest == float(world)
        for k in names:
            a, b = grads[k], ref[k].grad.numpy()
            scale = max(np.abs(b).max(), 1e-12)
            assert np.abs(a - b).max() / scale < 2e-5, (rank, k)
