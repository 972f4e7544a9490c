"""C5-shaped measurement (SURVEY 8d S5, 8e edge-shard): ONE giant random graph, N nodes, M undirected unique pairs
-> 2M directed entries + N diagonal entries, KernelSelfAttn(2,2,32,32,32,4) generalized FAVOR+, fwd + backward.

Two runs on the one GPU of the box:
  whole : rank 0's share (1/world of the pairs, all N nodes) treated as a whole graph  -> fused core, no exchange
  shard : the same entries with the edge-shard context of `world` ranks attached -> fused core in phases, then the
          general kernels; every
          all-reduce point is counted (bytes per step) and executed as a no-op, so the time is one rank's COMPUTE
          and the byte count prices the exchange (7 xGMI links x ~153 GB/s per GPU, ring all-reduce moves
          2 (w-1)/w of the buffer per rank).
    python scratch/bench_c5.py [--nodes 1000000] [--pairs 25000000] [--world 8] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def giant_graph(N, M, world, rank, device, seed=7):
    """owned directed edges [2, E_local] of a graph with M unique undirected random pairs (both directions kept)"""
    from hot_amd.dist import shard_owner
    g = torch.Generator(device=device).manual_seed(seed)
    keys = []
    chunk = 1 << 24
    got = 0
    while got < M:
        i = torch.randint(0, N, (chunk,), device=device, generator=g)
        j = torch.randint(0, N, (chunk,), device=device, generator=g)
        ok = i != j
        lo, hi = torch.minimum(i, j)[ok], torch.maximum(i, j)[ok]
        keys.append(lo * N + hi)
        got += int(ok.sum())
    key = torch.unique(torch.cat(keys))[:M]
    lo, hi = key // N, key % N
    own = shard_owner(lo, hi, world) == rank
    lo, hi = lo[own], hi[own]
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    return ei, int(key.numel())


class CountingShard:
    """ShardInfo stand-in: counts the bytes every all-reduce point would move; the reduction itself is the identity"""

    def __init__(self, rank, world, entry_weight):
        self.rank, self.world, self.group, self.entry_weight = rank, world, None, entry_weight
        self.bytes, self.calls = 0, 0

    def all_reduce_(self, t):
        self.bytes += t.numel() * t.element_size()
        self.calls += 1
        return t


def timed(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=1_000_000)
    ap.add_argument('--pairs', type=int, default=25_000_000)
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--skip-shard', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    from bench import make_layer, D
    from hot_amd import _lib
    from hot_amd.batch.sparse import make_batch_concatenated, batch_like
    from hot_amd.plan import Plan

    N = args.nodes
    ei, M = giant_graph(N, args.pairs, args.world, 0, dev)
    E = ei.size(1)
    g = torch.Generator(device=dev).manual_seed(11)
    nf = torch.randn(N, D // 2, device=dev, generator=g)
    ef = torch.randn(E, D // 2, device=dev, generator=g)
    print(f'graph: N={N} pairs={M} world={args.world}: rank 0 owns {E} directed entries + {N} diagonal', flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    G = make_batch_concatenated(nf, ei, ef, [N], [E])
    torch.cuda.synchronize()
    t_batch = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    Plan(G.indices, G.n_nodes, G.n_edges)
    torch.cuda.synchronize()
    t_plan = (time.perf_counter() - t0) * 1e3
    n_entries = N + E
    layer = make_layer(dev)
    x0 = G.values
    out = {'N': N, 'pairs': M, 'world': args.world, 'entries_rank0': n_entries, 'make_batch_ms': round(t_batch, 2),
           'k1_plan_build_ms': round(t_plan, 2)}

    def step():
        x = x0.detach().requires_grad_(True)
        y = layer(batch_like(G, x, skip_masking=True))
        y.values.sum().backward()
        for p in layer.parameters():
            p.grad = None

    # whole: fused core
    layer.use_fused = True
    ms = timed(step, args.steps)
    _lib.timing_begin(None)
    step()
    torch.cuda.synchronize()
    census = _lib.timing_end()
    out['whole_fused'] = {'ms_per_step': round(ms, 2), 'entries_per_s': n_entries / ms * 1e3,
                          'census_ms': {k: round(v[0], 2) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])[:8]}}
    print(json.dumps(out['whole_fused']), flush=True)

    if not args.skip_shard:
        plan = G.plan
        w = torch.where(plan.kind == 1, torch.full((plan.R,), 1.0 / args.world, device=dev), (plan.kind == 2).to(torch.float32))
        plan.shard = CountingShard(0, args.world, w)
        wd = args.world
        for name, fused in (('shard_fused_phases', True), ('shard_general', False)):
            layer.use_fused = fused
            ms = timed(step, args.steps)
            plan.shard.bytes, plan.shard.calls = 0, 0
            _lib.timing_begin(None)
            step()
            torch.cuda.synchronize()
            census = _lib.timing_end()
            xb, calls = plan.shard.bytes, plan.shard.calls
            # ring all-reduce: every rank sends and receives 2 (w-1)/w of the buffer; 7 links x 153 GB/s per GPU
            xgmi_ms = xb * 2 * (wd - 1) / wd / (7 * 153e9) * 1e3
            out[name] = {'ms_per_step_compute': round(ms, 2), 'entries_per_s_rank': n_entries / ms * 1e3,
                         'allreduce_bytes_per_step': xb, 'allreduce_calls_per_step': calls,
                         'allreduce_ms_at_xgmi_peak': round(xgmi_ms, 2),
                         'census_ms': {k: round(v[0], 2) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])[:8]}}
            print(json.dumps(out[name]), flush=True)
        plan.shard = None
    out['peak_mem_GB'] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
