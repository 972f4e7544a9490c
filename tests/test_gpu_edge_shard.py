"""GPU (-m gpu): edge-shard mode of ONE graph over two / four ranks.  All ranks share cuda:0 and talk over
gloo (host-staged collectives in hot_amd.dist stand in for RCCL, which needs one GPU per rank); the arithmetic on
every rank is the HIP path.  Each rank checks its local outputs, its local input gradients and the all-reduced
parameter gradients against the CPU ORACLE run on the whole graph.  Two sharded paths: the fused core
in phases (hotk_attn22_phase, all-reduces of kvg / kvn and their gradients between the phases) and the general
kernels (all-reduce of the materialised key-value sums in ops.KernelAttn)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _graph(seed=3, n=300, n_pairs=1200):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    und = set()
    while len(und) < n_pairs:
        i, j = rng.integers(0, n, 2)
        if i != j:
            und.add((int(min(i, j)), int(max(i, j))))
    pairs = set()
    for k, (i, j) in enumerate(sorted(und)):
        pairs.add((i, j))
        if k % 5:            # every fifth pair stays one-directional: entries without a transposed partner
            pairs.add((j, i))
    ei = torch.tensor(sorted(pairs), dtype=torch.long).t().contiguous()
    return torch.randn(n, 16, generator=g), ei, torch.randn(ei.size(1), 16, generator=g), g


def _layer(generalized=True):
    from hot_amd.models.common import KernelFeatureMap
    from hot_amd.models.sparse import KernelSelfAttn
    torch.manual_seed(0)
    return KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=generalized)).to(DEV)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _worker(rank, world, port, q, fused, generalized=True):
    import datetime
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    ok = False
    try:
        torch.cuda.set_device(0)
        from oracle import hot_oracle as O
        from hot_amd.batch import sparse as bs
        from hot_amd.dist import shard_graph, allreduce_gradients
        nf, ei, ef, g = _graph()
        n, E = nf.size(0), ei.size(1)
        Rw_cpu = torch.randn(n + E, 32, generator=g)              # loss weights per global row (nodes, then edges)
        Rw = Rw_cpu.to(DEV)
        # the WHOLE graph on the CPU oracle (pinned restatement of the reference): the reference every rank is held to
        layer = _layer(generalized)
        sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32 and 'projection' not in k)
              for k, v in layer.state_dict().items()}
        GO = O.build_batch(nf, ei, ef, [n], [E])
        xo = GO.values.clone().requires_grad_(True)
        yo = O.kernel_self_attn(GO.like(xo), sd, 2, 2, 32, 32, 32, 4, 'default', generalized)
        (yo.values[0] * Rw_cpu).sum().backward()
        ref_y, ref_gx = yo.values[0].detach().to(DEV), xo.grad[0].to(DEV)
        ref_gp = {k: sd[k].grad.to(DEV) for k, _ in layer.named_parameters()}
        # this rank's shard on the HIP path
        layer2 = _layer(generalized)
        layer2.use_fused = fused
        Gs, own = shard_graph(nf.to(DEV), ei.to(DEV), ef.to(DEV), world, rank)
        rows = torch.cat([torch.arange(n, device=DEV), n + own])         # global row of every local row
        xs = Gs.values.clone().requires_grad_(True)
        ys = layer2(bs.batch_like(Gs, xs, skip_masking=True))
        w = Gs.plan.shard.entry_weight.view(-1, 1)
        (ys.values[0] * Rw[rows] * w).sum().backward()
        allreduce_gradients(layer2.parameters())
        gx = xs.grad[0].clone()
        gx_nodes = gx[:n].clone()
        Gs.plan.shard.all_reduce_(gx_nodes)                        # replicated inputs: their gradient is the sum over ranks
        # ReLU features: rows whose pre-activation sits within rounding of 0 may flip their gate against the CPU
        # (tests/test_gpu_parity.py::test_layer_vs_oracle_ba_graphs): gradients are held row-wise
        scale = ref_gx.abs().max()
        bad_e = float((((gx[n:] - ref_gx[n + own]).abs().amax(-1) / scale) > 2e-5).float().mean())
        bad_n = float((((gx_nodes - ref_gx[:n]).abs().amax(-1) / scale) > 2e-5).float().mean())
        res = {'n_local_edges': int(own.numel()),
               'y': _rel(ys.values[0].detach(), ref_y[rows]),
               'gx_edges_bad_rows': bad_e, 'gx_nodes_bad_rows': bad_n,
               'gp': max(_rel(p.grad, ref_gp[k]) for k, p in layer2.named_parameters())}
        q.put((rank, res))
        ok = True
    except Exception as exc:                                       # report instead of leaving the parent to time out
        import traceback
        q.put((rank, {'error': repr(exc), 'trace': traceback.format_exc()}))
    finally:
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def _worker_models(rank, world, port, q):
    """Linear(cfg='default') of every order, the graph-level pools and a whole Encoder(2, 0, ...) with its softmax readout
    on a sharded graph, each against the CPU oracle on the whole graph.  Replicated outputs (order 1 / order 0) enter the
    local loss with weight 1 / world, local order-2 rows with ShardInfo.entry_weight."""
    import datetime
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    ok = False
    try:
        torch.cuda.set_device(0)
        from oracle import hot_oracle as O
        from hot_amd.batch import sparse as bs
        from hot_amd.dist import shard_graph, allreduce_gradients
        from hot_amd.models import sparse as ms
        from hot_amd.models.encoder import Encoder
        nf, ei, ef, g = _graph(seed=11, n=120, n_pairs=420)
        n, E = nf.size(0), ei.size(1)
        GO = O.build_batch(nf, ei, ef, [n], [E])
        Gs, own = shard_graph(nf.to(DEV), ei.to(DEV), ef.to(DEV), world, rank)
        rows = torch.cat([torch.arange(n, device=DEV), n + own])
        w_rows = Gs.plan.shard.entry_weight.view(-1, 1)
        res = {'n_local_edges': int(own.numel())}

        def check(tag, module, oracle_fn, out_order):
            module = module.to(DEV)
            sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32 and 'projection' not in k)
                  for k, v in module.state_dict().items()}
            xo = GO.values.clone().requires_grad_(True)
            yo = oracle_fn(GO.like(xo), sd)
            yo = yo.values[0] if out_order > 0 else yo[0]
            Rw = torch.randn(yo.shape, generator=g)
            (yo * Rw).sum().backward()
            xs = Gs.values.clone().requires_grad_(True)
            ys = module(bs.batch_like(Gs, xs, skip_masking=True))
            ys = ys.values[0] if out_order > 0 else ys[0]
            if out_order == 2:
                ref_y, wl = yo.detach().to(DEV)[rows], Rw.to(DEV)[rows] * w_rows
            else:
                ref_y, wl = yo.detach().to(DEV), Rw.to(DEV) / world
            (ys * wl).sum().backward()
            allreduce_gradients(module.parameters())
            gx = xs.grad[0].clone()
            gx_nodes = gx[:n].clone()
            Gs.plan.shard.all_reduce_(gx_nodes)
            ref_gx = xo.grad[0].to(DEV)
            scale = ref_gx.abs().max()
            gscale = max(float(sd[k].grad.abs().max()) for k, _ in module.named_parameters())
            res[tag] = {
                'y': _rel(ys.detach(), ref_y),
                'gx_edges_bad_rows': float((((gx[n:] - ref_gx[n + own]).abs().amax(-1) / scale) > 2e-5).float().mean()),
                'gx_nodes_bad_rows': float((((gx_nodes - ref_gx[:n]).abs().amax(-1) / scale) > 2e-5).float().mean()),
                'gp': max(float((p.grad.cpu() - sd[k].grad).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3 * gscale)
                          for k, p in module.named_parameters())}

        torch.manual_seed(0)
        for normalize in (True, False):
            t = 'n' if normalize else 'u'
            check(f'lin22_default_{t}', ms.Linear(2, 2, 32, 24, cfg='default', normalize=normalize),
                  lambda G, sd, nz=normalize: O.linear_forward(G, 2, 2, sd['weight.weight'],
                                                               [sd['bias.bias.0'], sd['bias.bias.1']], 'default', nz), 2)
            check(f'lin21_default_{t}', ms.Linear(2, 1, 32, 24, cfg='default', normalize=normalize),
                  lambda G, sd, nz=normalize: O.linear_forward(G, 2, 1, sd['weight.weight'], [sd['bias.bias.0']], 'default', nz), 1)
            check(f'lin20_{t}', ms.Linear(2, 0, 32, 24, cfg='light', normalize=normalize),
                  lambda G, sd, nz=normalize: O.linear_forward(G, 2, 0, sd['weight.weight'], [sd['bias.bias.0']], 'light', nz), 0)
        enc = Encoder(2, 0, [2], 32, 8, 32, 32, 32, 32, 4, 32, 32, 4, 'default', 'generalized_kernel', 0., 0., 0., sparse=True)
        enc.skip_redraw_projections = True
        enc.eval()
        check('encoder_2_0', enc, lambda G, sd: O.encoder_2_0(G, sd, 1, 32, 32, 32, 32, 4, 32, 32, 4, True), 0)
        q.put((rank, res))
        ok = True
    except Exception as exc:
        import traceback
        q.put((rank, {'error': repr(exc), 'trace': traceback.format_exc()}))
    finally:
        if ok:
            dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('world,fused,generalized', [(2, True, True), (2, False, True), (4, True, True), (2, False, False),
                                                    (3, True, True), (3, False, False)])
def test_edge_shard_matches_oracle_on_whole_graph(world, fused, generalized):
    """generalized = False: the softmax feature map, whose key branch subtracts ONE maximum over the whole batch
    (common/kernel.py:112) -- MAX-all-reduced across the shards, its gradient SUM-all-reduced (ops.Phi).
    world = 3: replicated diagonal entries enter the entry sums with weight 1/3, which fp32 does not represent exactly --
    the reduced sums then carry one more rounding per term (3 x fl(1/3) x = x (1 +- 2^-24)), far inside the 1e-5 bar; only
    power-of-two worlds reproduce the unsharded sums bit for bit (hot_amd/dist.py)"""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, fused, generalized), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=200) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=10)
    errors = {r: v for r, v in results.items() if 'error' in v}
    assert not errors, errors
    assert all(p.exitcode == 0 for p in procs)
    assert sum(r['n_local_edges'] for r in results.values()) == _graph()[1].size(1)
    assert all(r['n_local_edges'] > 0 for r in results.values())
    print('edge-shard results', results)
    for rank, r in results.items():
        assert r['y'] < 1e-5, (rank, r)
        assert r['gx_edges_bad_rows'] <= 5e-3 and r['gx_nodes_bad_rows'] <= 5e-3 and r['gp'] < 1e-3, (rank, r)


def _run_ranks(target, world, extra=()):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=200) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=10)
    errors = {r: v for r, v in results.items() if 'error' in v}
    assert not errors, errors
    assert all(p.exitcode == 0 for p in procs)
    return results


@pytest.mark.parametrize('world', [2, 4])
def test_edge_shard_default_basis_pools_and_readout_match_oracle(world):
    """the sums over entries outside the attention core in edge-shard mode: row / column / graph pools of
    Linear(cfg='default') (models/sparse/linear.py:85-238) and their element counts, the order-2 -> order-0 pool, and the
    softmax readout of Encoder(2, 0, ...) (sparse/attn.py:112-131, attncoef.py:25-59) -- all-reduced sums, MAX-reduced
    softmax stabiliser -- against the oracle on the whole graph"""
    results = _run_ranks(_worker_models, world)
    print('edge-shard model results', results)
    for rank, r in results.items():
        assert r.pop('n_local_edges') > 0
        for tag, v in r.items():
            assert v['y'] < 1e-5, (rank, tag, v)
            assert v['gx_edges_bad_rows'] <= 5e-3 and v['gx_nodes_bad_rows'] <= 5e-3 and v['gp'] < 1e-3, (rank, tag, v)


def test_fused_phases_single_rank_equal_one_pass():
    """hotk_attn22_phase with a one-rank shard context (no exchange, diagonal weight 1) against hotk_attn22_forward /
    _backward on the same batch: the phases share the arithmetic of the VALU/DPP kernels, the one-pass forward runs
    the MFMA tiles, so outputs agree to rounding (1e-5) and not bitwise."""
    from hot_amd import _lib
    from hot_amd.batch import sparse as bs
    from hot_amd.dist import ShardInfo
    nf, ei, ef, g = _graph(seed=5, n=200)
    n, E = nf.size(0), ei.size(1)
    res = []
    for sharded in (False, True):
        layer = _layer()
        G = bs.make_batch_concatenated(nf.to(DEV), ei.to(DEV), ef.to(DEV), [n], [E])
        if sharded:
            G.plan.shard = ShardInfo(0, 1, (G.plan.kind != 0).to(torch.float32))
        x = G.values.clone().requires_grad_(True)
        _lib.timing_begin(None)
        y = layer(bs.batch_like(G, x, skip_masking=True))
        y.values.square().sum().backward()
        census = _lib.timing_end()
        assert ('hotk_attn22_phase' in census) == sharded and ('hotk_attn22_forward' in census) != sharded, sorted(census)
        res.append((y.values.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    (y0, gx0, gp0), (y1, gx1, gp1) = res
    assert _rel(y1, y0) < 1e-5 and _rel(gx1, gx0) < 2e-5
    assert max(_rel(gp1[k], gp0[k]) for k in gp0) < 2e-5


@pytest.mark.parametrize('sharded', [False, True])
def test_fused_one_large_graph_many_chunks(sharded):
    """ONE graph with 3 000 nodes and ~21 000 entries: the per-graph sums are reduced from 83 row chunks
    (k_graph_reduce_wide, taken above 16 chunks per graph) and column / row segments span the whole range of node ids.
    Fused core (one pass, and in phases with a one-rank shard context) against the general kernels on the same batch.
    ReLU features: a pre-activation within rounding of 0 may flip its gate between the two kernels, so gradients are
    held row-wise as in test_gpu_parity.test_layer_vs_oracle_ba_graphs."""
    from hot_amd.batch import sparse as bs
    from hot_amd.dist import ShardInfo
    nf, ei, ef, g = _graph(seed=9, n=3000, n_pairs=10000)
    n, E = nf.size(0), ei.size(1)
    res = []
    for fused in (False, True):
        layer = _layer()
        layer.use_fused = fused
        G = bs.make_batch_concatenated(nf.to(DEV), ei.to(DEV), ef.to(DEV), [n], [E])
        assert G.plan.Emax > 16 * 256
        if fused and sharded:
            G.plan.shard = ShardInfo(0, 1, (G.plan.kind != 0).to(torch.float32))
        x = G.values.clone().requires_grad_(True)
        y = layer(bs.batch_like(G, x, skip_masking=True))
        y.values.square().sum().backward()
        res.append((y.values.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    (y0, gx0, gp0), (y1, gx1, gp1) = res
    assert _rel(y1, y0) < 1e-5
    row_err = (gx1 - gx0).abs().amax(-1) / gx0.abs().max()
    assert float((row_err > 2e-5).float().mean()) <= 2e-3
    assert max(_rel(gp1[k], gp0[k]) for k in gp0) < 1e-3


def test_fused_phases_padded_batch():
    """the phased core on a batch of two graphs of different sizes (padding node and entry rows: their kvn / nd0 / dd0
    rows are never written or read) with a one-rank shard context, against the one-pass core"""
    from hot_amd.batch import sparse as bs
    from hot_amd.dist import ShardInfo
    (nf1, ei1, ef1, _), (nf2, ei2, ef2, _) = _graph(seed=21, n=120, n_pairs=400), _graph(seed=22, n=75, n_pairs=150)
    nf, ef = torch.cat([nf1, nf2]), torch.cat([ef1, ef2])
    ei = torch.cat([ei1, ei2], dim=1)
    res = []
    for sharded in (False, True):
        layer = _layer()
        G = bs.make_batch_concatenated(nf.to(DEV), ei.to(DEV), ef.to(DEV), [120, 75], [ei1.size(1), ei2.size(1)])
        assert G.plan.Ev != G.plan.R and G.plan.B == 2
        if sharded:
            G.plan.shard = ShardInfo(0, 1, (G.plan.kind != 0).to(torch.float32))
        x = G.values.clone().requires_grad_(True)
        y = layer(bs.batch_like(G, x, skip_masking=True))
        y.values.square().sum().backward()
        res.append((y.values.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    (y0, gx0, gp0), (y1, gx1, gp1) = res
    assert _rel(y1, y0) < 1e-5 and _rel(gx1, gx0) < 2e-5
    assert max(_rel(gp1[k], gp0[k]) for k in gp0) < 2e-5
    pad = (G.plan.kind == 0).view(-1)
    assert float(y1.view(-1, y1.size(-1))[pad].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------------------
# C5 at ONE RANK'S REAL SIZE (bench.py --mode edge-shard at 1 GPU: 125 000 nodes, 3.125 M undirected pairs = 6.25 M
# directed entries + the diagonal, B = 1, ~50 entries per node).  Reached only here: the branch R >= 24 Nn that switches
# the work tickets off (csrc/fused22.hip ws_tickets), k_graph_reduce_wide over thousands of row chunks, refinements of
# refinements in the segment sums (one graph of 6.4 M rows).  No oracle runs at this size and the per-graph terms make
# every output depend on the whole graph, so the checks are size-independent properties.
# ------------------------------------------------------------------------------------------------------------------
def _c5_shard_graph(n_nodes=125000, n_pairs=3125000, seed=4321, perm=None):
    """the graph of bench.py::edge_shard_mode (same seed and construction); perm: relabel node v as perm[v]"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    i = torch.randint(0, n_nodes, (int(n_pairs * 1.02),), device=DEV, generator=g)
    j = torch.randint(0, n_nodes, (int(n_pairs * 1.02),), device=DEV, generator=g)
    lo, hi = torch.minimum(i, j), torch.maximum(i, j)
    key = torch.unique((lo * n_nodes + hi)[lo != hi])[:n_pairs]
    lo, hi = key // n_nodes, key % n_nodes
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    nf = torch.randn(n_nodes, 16, device=DEV, generator=g)
    ef = torch.randn(ei.size(1), 16, device=DEV, generator=g)
    if perm is not None:
        ei = perm[ei]
        nf = torch.empty_like(nf).index_copy_(0, perm, nf)        # node perm[v] carries v's features
    order = torch.argsort(ei[0] * n_nodes + ei[1])
    return nf, ei[:, order].contiguous(), ef[order].contiguous()


class _LoneRankOf(object):
    """edge-shard context of rank 0 of a world of `world` ranks with no peers attached: all-reduces return the rank's own
    partial sums.  What the kernels compute under it -- diagonal entries weighted 1 / world in every entry sum -- is the
    same quantity on the phased fused core and on the general kernels, which is what the test compares."""

    def __init__(self, world, entry_weight):
        self.rank, self.world, self.group, self.entry_weight = 0, world, None, entry_weight

    def all_reduce_(self, t):
        return t

    def all_reduce_max_(self, t):
        return t

    def all_reduce_start_(self, t):
        from hot_amd.dist import _Done
        return _Done()


def _c5_step(G, fused, shard=None, square=False):
    from hot_amd import _lib
    from hot_amd.batch import sparse as bs
    layer = _layer()
    layer.use_fused = fused
    plan = G.plan
    if shard is None:
        plan.__dict__.pop('shard', None)
    else:
        plan.shard = shard
    x = G.values.clone().requires_grad_(True)
    _lib.timing_begin(None)
    y = layer(bs.batch_like(G, x, skip_masking=True))
    (y.values.square() if square else y.values).sum().backward()
    census = set(_lib.timing_end())
    plan.__dict__.pop('shard', None)
    return y.values.detach()[0], x.grad[0].clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}, census


def test_c5_one_rank_real_shard_size():
    from hot_amd.batch import sparse as bs
    from hot_amd.dist import ShardInfo
    n = 125000
    nf, ei, ef = _c5_shard_graph(n)
    E = ei.size(1)
    G = bs.make_batch_concatenated(nf, ei, ef, [n], [E])
    plan = G.plan
    assert plan.B == 1 and plan.R == n + E and plan.R >= 24 * plan.Nn and plan.R > 6_000_000
    # transpose map at this size: an involution on the valid rows, every entry of a symmetric graph has its partner
    tI, tM = G.t_indices[0], G.t_mask[0]
    assert bool(tM.all()) and bool((tI[tI] == torch.arange(plan.R, device=DEV)).all())
    assert bool((G.indices[0][tI] == G.indices[0].flip(-1)).all())
    # one-pass fused core, twice: every sum runs in a fixed order -> bit-identical
    y0, gx0, gp0, c0 = _c5_step(G, True)
    y0b, gx0b, gp0b, _ = _c5_step(G, True)
    assert {'hotk_attn22_forward', 'hotk_attn22_backward'} <= c0
    assert torch.equal(y0, y0b) and torch.equal(gx0, gx0b) and all(torch.equal(gp0[k], gp0b[k]) for k in gp0)
    assert bool(torch.isfinite(y0).all()) and float(y0.abs().max()) > 0
    # the core in phases with a one-rank shard context == the one-pass core (to rounding: the one-pass forward runs the
    # MFMA tiles, the phases the VALU kernels; the backward arithmetic and its ReLU gates are the same code)
    y1, gx1, gp1, c1 = _c5_step(G, True, ShardInfo(0, 1, (plan.kind != 0).to(torch.float32)))
    assert 'hotk_attn22_phase' in c1 and 'hotk_attn22_forward' not in c1
    assert _rel(y1, y0) < 1e-5 and _rel(gx1, gx0) < 2e-5, (_rel(y1, y0), _rel(gx1, gx0))
    assert max(_rel(gp1[k], gp0[k]) for k in gp0) < 2e-5
    del y1, gx1, gp1
    # rank 0 of a world of 8 and of 3 (diagonal entries weighted 1/8, 1/3) without peers: phased fused core against the
    # general kernels under the same context.  Both forward paths and both backward paths differ in code here, so ReLU
    # gates within rounding of 0 may differ: gradients row-wise, as in test_fused_one_large_graph_many_chunks
    for world in (8, 3):
        w = torch.where(plan.kind == 1, torch.full((plan.R,), 1.0 / world, device=DEV), (plan.kind == 2).float())
        ya, gxa, gpa, ca = _c5_step(G, True, _LoneRankOf(world, w), square=True)
        yb, gxb, gpb, cb = _c5_step(G, False, _LoneRankOf(world, w), square=True)
        assert 'hotk_attn22_phase' in ca and 'hotk_attn22_phase' not in cb and 'hotk_seg_outer_sum' in cb
        assert _rel(ya, yb) < 1e-5, (world, _rel(ya, yb))
        assert float((ya - y0).abs().max()) > 1e-3 * float(y0.abs().max())        # the weights do change the sums
        row_err = (gxa - gxb).abs().amax(-1) / gxb.abs().max()
        assert float((row_err > 2e-5).float().mean()) <= 2e-3, world
        assert max(_rel(gpa[k], gpb[k]) for k in gpa) < 1e-3, world
        del ya, gxa, gpa, yb, gxb, gpb
    # permutation equivariance -- the defining property of the layer (README.md:9-13) -- at full size: relabel the nodes,
    # rebuild the batch (other row order, other segments, other chunking of every sum) and compare row by row
    perm = torch.randperm(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    nf2, ei2, ef2 = _c5_shard_graph(n, perm=perm)
    G2 = bs.make_batch_concatenated(nf2, ei2, ef2, [n], [E])
    y2, gx2, gp2, _ = _c5_step(G2, True)
    key = lambda idx: idx[:, 0] * n + idx[:, 1]
    k1 = key(perm[G.indices[0]])                  # where row r of G lives in G2, by its relabelled (i, j)
    k2 = key(G2.indices[0])
    o2 = torch.argsort(k2)
    pos = o2[torch.searchsorted(k2[o2], k1)]
    assert bool((k2[pos] == k1).all())
    assert _rel(y2[pos], y0) < 1e-5, _rel(y2[pos], y0)
    row_err = (gx2[pos] - gx0).abs().amax(-1) / gx0.abs().max()
    assert float((row_err > 2e-5).float().mean()) <= 2e-3
    assert max(_rel(gp2[k], gp0[k]) for k in gp0) < 1e-3
