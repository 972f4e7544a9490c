"""Headline benchmark: entries/s forward+backward of the 2->2 sparse kernel-attention layer (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1: when no launcher has set WORLD_SIZE, this process (which never touches the GPU) starts the N ranks itself as
children of `python -m torch.distributed.run` (one per GPU, RCCL), forwards rank 0's JSON line and exits with the
launcher's return code -- the stand-in for Lightning's DDP launcher of regression/main/entry.py:110-113.  Launched
from outside by torch.distributed.run it runs as one rank, as before.

Workload "S1/C1" (SURVEY 8d): B Barabasi-Albert(64, 5) graphs -> 654 entries each (64 diagonal + 590 directed
edges), ~10 M entries per GPU; layer KernelSelfAttn(2,2, dim_in=32, dim_v=32, dim_qk=32, n_heads=4) with the
generalized (ReLU) FAVOR+ map (d=8, m=16), fp32, dropout 0, fixed projection.  One step = layer(G) followed by
out.values.sum().backward() (run_perf_tests.py:62-88), inputs resident in HBM.  Graph structures come from an own
seeded BA generator (512 distinct graphs, tiled), features are seeded Gaussians generated on the device.
Multi-GPU: graphs are independent, so every rank runs its own B graphs (weak scaling) and the parameter
gradients are all-reduced over RCCL each step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table
N_NODES, BA_M, ENTRIES_PER_GRAPH = 64, 5, 654
D, DQK, DV, HEADS = 32, 32, 32, 4
# algorithmic bytes of the attention core per entry, forward + backward: 96*dqk + 224*dv + 50, plus the
# per-node term 2*4*12*dqk + 3*2*2*4*S*H*m*(dvh+1) (S=2, H=4, m=16, dvh=8) amortised over entries/node
BYTES_PER_ENTRY = 96 * DQK + 224 * DV + 50
BYTES_PER_NODE = 2 * 4 * 12 * DQK + 3 * 2 * 2 * 4 * 2 * HEADS * 16 * (DV // HEADS + 1)
# HBM traffic of the fused core per valid entry on THIS workload, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes (FETCH_SIZE x2 as calibrated on a known 1 GiB copy, MI355X_MICROARCH.md HBM section):
# profiles/r02_pmc_traffic_attn22.txt, collected with scratch/pmc_traffic.sh at the benchmarked size (10 M entries; the
# round-1 pass at 2 M entries gave 4903 / 8584, the mid-round-2 pass 4886 / 8592; re-collected at the end of round 2: the
# ticketed kernels hold more waves resident and re-read more of a node's rows past L2 -- they fetch more and run faster)
PMC_TRAFFIC_BYTES_PER_ENTRY = {'hotk_attn22_forward': 5261, 'hotk_attn22_backward': 10058}
PMC_TRAFFIC_SOURCE = 'profiles/r03_pmc_traffic_attn22.txt'


def build_structure(n_graphs, distinct=512, seed=1234):
    """edge_index [2, n_graphs*590] (local ids) from `distinct` seeded BA graphs, tiled"""
    from hot_amd.utils.graphs import ba_edges
    rng = np.random.default_rng(seed)
    base = [ba_edges(N_NODES, BA_M, rng) for _ in range(min(distinct, n_graphs))]
    reps = (n_graphs + len(base) - 1) // len(base)
    ei = np.concatenate((base * reps)[:n_graphs], axis=1)
    return ei


def make_inputs(n_graphs, device, seed):
    from hot_amd.batch.sparse import make_batch_concatenated
    ei = torch.from_numpy(build_structure(n_graphs)).to(device)
    g = torch.Generator(device=device).manual_seed(seed)
    e_per = ei.size(1) // n_graphs
    nf = torch.randn(n_graphs * N_NODES, D // 2, device=device, generator=g)
    ef = torch.randn(ei.size(1), D // 2, device=device, generator=g)
    return make_batch_concatenated(nf, ei, ef, [N_NODES] * n_graphs, [e_per] * n_graphs)


def make_layer(device):
    from hot_amd.models.common import KernelFeatureMap
    from hot_amd.models.sparse import KernelSelfAttn
    torch.manual_seed(0)
    fmap = KernelFeatureMap(DQK // HEADS, generalized_attention=True)
    return KernelSelfAttn(2, 2, D, DV, DQK, HEADS, 'default', 0., 0., fmap).to(device)


def cpu_baseline(n_graphs=192, threads=None):
    """The oracle (CPU restatement of the reference algorithm, kind 'port') on the host cores, same layer and
    graph family, bounded sample."""
    from oracle import hot_oracle as O
    threads = threads or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(99)
    g = torch.Generator().manual_seed(99)
    nfs, eis, efs = [], [], []
    for _ in range(n_graphs):
        ei = torch.from_numpy(O.ba_edges(N_NODES, BA_M, rng))
        nfs.append(torch.randn(N_NODES, D // 2, generator=g)); eis.append(ei)
        efs.append(torch.randn(ei.size(1), D // 2, generator=g))
    G = O.build_batch_from_lists(nfs, eis, efs)
    layer = make_layer('cpu')
    sd = {k: v.detach().clone().requires_grad_(v.dtype == torch.float32 and 'projection' not in k)
          for k, v in layer.state_dict().items()}
    n_entries = sum(G.n_edges)
    best = None
    for it in range(3):
        x = G.values.clone().requires_grad_(True)
        t0 = time.perf_counter()
        y = O.kernel_self_attn(G.like(x), sd, 2, 2, D, DV, DQK, HEADS, 'default', True)
        y.values.sum().backward()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {'value': n_entries / best, 'unit': 'entries/s', 'cores': threads, 'host_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': f'{n_graphs} BA(64,5) graphs = {n_entries} entries, fwd+bwd of the same layer, best of 3, '
                      f'oracle/hot_oracle.py on torch-CPU fp32'}


def sweep(device, repeat=3, n_max=None):
    """The reference's only published artefact restated (run_perf_tests.py:91-243, docs/linear-scaling.png): forward /
    backward time and peak memory of a 4-layer sparse model on ONE Barabasi-Albert(n, 5) graph, n swept over
    2^(5..18)/5, dim = dim_qk = dim_v = dim_ff = 32, 4 heads.  Curves: SK = Encoder(2,0,[2]*4,...,'generalized_kernel')
    (fixed projections), SL = MLP(2,0,[2]*4) of Linear(cfg='default').  HIP events on the current stream, as the
    reference's CUDA events; an own seeded BA generator stands in for networkx."""
    from hot_amd.batch.sparse import make_batch
    from hot_amd.models import Encoder, MLP
    from hot_amd.utils.graphs import ba_edges
    n_list = sorted(set(int(v) // 5 for v in 2 ** np.linspace(5, 18, 27, endpoint=True)))
    n_list = [n for n in n_list if n > 5 and (n_max is None or n <= n_max)]
    torch.manual_seed(0)
    models = {
        'SK': Encoder(2, 0, [2] * 4, 32, 32, 32, 32, 32, 32, 4, 32, 32, 4, 'default', 'generalized_kernel', 0., 0., 0.,
                      sparse=True).to(device),
        'SL': MLP(2, 0, [2] * 4, 32, 32, 32, sparse=True).to(device),
    }
    models['SK'].skip_redraw_projections = True
    out = {k: [] for k in models}
    for n in n_list:
        for name, model in models.items():
            ft, bt, pm, ne = [], [], [], 0
            for i in range(repeat + 1):                      # first repetition = warm-up, dropped
                rng = np.random.default_rng(i)
                ei = torch.from_numpy(ba_edges(n, 5, rng)).to(device)
                g = torch.Generator(device=device).manual_seed(i)
                nf = torch.randn(n, 16, device=device, generator=g)
                ef = torch.randn(ei.size(1), 16, device=device, generator=g)
                G = make_batch([nf], [ei], [ef])
                ne = sum(G.n_edges)
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                y = model(G)
                e1.record()
                y.sum().backward()
                e2.record()
                torch.cuda.synchronize()
                if i:
                    ft.append(e0.elapsed_time(e1)); bt.append(e1.elapsed_time(e2))
                    pm.append(torch.cuda.max_memory_allocated() / 2 ** 30)
                model.zero_grad(set_to_none=True)
                del G, y
            out[name].append({'n': n, 'entries': ne, 'forward_ms': round(float(np.mean(ft)), 3),
                              'backward_ms': round(float(np.mean(bt)), 3), 'peak_mem_GiB': round(float(np.mean(pm)), 4)})
    return {'harness': 'run_perf_tests.py:221-243 restated: bsize 1, 4 layers, dim 32, 4 heads, BA(n,5), repeat %d' % repeat,
            'curves': out}


def edge_shard_mode(args, device, world, rank, dist):
    """--mode edge-shard: ONE graph sharded by entries over the ranks.  Weak scaling: 125 000
    nodes and 3.125 M undirected pairs per GPU, so that 8 GPUs hold exactly C5 (1 M nodes, 25 M pairs = 50 M directed
    entries + 1 M diagonal).  Every rank generates the same graph from one seed and keeps its shard
    (hot_amd.dist.shard_graph); the per-node / per-graph key-value sums are all-reduced over RCCL between the phases of
    the fused core, parameter gradients at the end of the step."""
    from hot_amd import _lib
    from hot_amd.batch.sparse import batch_like
    from hot_amd.dist import shard_graph, allreduce_gradients, max_over_ranks
    n_nodes = args.shard_nodes * world
    n_pairs = args.shard_pairs * world
    g = torch.Generator(device=device).manual_seed(4321)
    # unique unordered pairs (i < j) from random draws, duplicates dropped (a few per million)
    i = torch.randint(0, n_nodes, (int(n_pairs * 1.02),), device=device, generator=g)
    j = torch.randint(0, n_nodes, (int(n_pairs * 1.02),), device=device, generator=g)
    lo, hi = torch.minimum(i, j), torch.maximum(i, j)
    key = torch.unique((lo * n_nodes + hi)[lo != hi])[:n_pairs]
    lo, hi = key // n_nodes, key % n_nodes
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    order = torch.argsort(ei[0] * n_nodes + ei[1])
    ei = ei[:, order].contiguous()
    nf = torch.randn(n_nodes, D // 2, device=device, generator=g)
    ef = torch.randn(ei.size(1), D // 2, device=device, generator=g)
    group = None
    G, own = shard_graph(nf, ei, ef, world, rank, group)
    del nf, ef, i, j, lo, hi, key, order
    layer = make_layer(device)
    params = list(layer.parameters())
    x0 = G.values
    w = G.plan.shard.entry_weight.view(1, -1, 1)

    def step():
        x = x0.detach().requires_grad_(True)
        out = layer(batch_like(G, x, skip_masking=True))
        (out.values * w).sum().backward()          # replicated diagonal rows count once over the ranks
        if dist is not None:
            allreduce_gradients(params)
        for p_ in params:
            p_.grad = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.timing_begin(None)
    step()
    census = _lib.timing_end()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    n_total = n_nodes + ei.size(1)
    local = int(G.plan.Ev)
    if rank == 0:
        print(json.dumps({
            'metric': 'edges/sec fwd+bwd, 2->2 sparse kernel attn layer', 'value': n_total * args.steps / elapsed,
            'unit': 'entries/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'S5/C5 edge-shard: ONE graph, {n_nodes} nodes, {ei.size(1)} directed entries + diagonal, '
                                   f'sharded by unordered pair over {world} GPU(s) ({local} local rows on rank 0); '
                                   f'KernelSelfAttn(2,2,32,32,32,4) generalized FAVOR+, phased fused core + all-reduces',
                       'entries_total': n_total, 'parallelism': f'edge-shard x{world}'},
            'exchange_bytes_per_step_per_rank': 2 * 4 * (2 * G.plan.Nn * 1152 + G.plan.Nn * 128) if world > 1 else 0,
            'kernel_census_ms_per_step': {k: round(v[0], 3) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])},
            'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'preconditioned': args.preconditioned,      # (see precondition(): device memory written once by an earlier process)
        }), flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Parent of an N-rank run: start `python -m torch.distributed.run --nproc-per-node N bench.py <argv>` as a CHILD
    process (never an exec: this parent may live on, the children initialise the GPUs), pass rank 0's JSON line through
    on stdout, everything else to stderr; returns the launcher's exit code (non-zero when any rank failed).  The parent
    makes no device call and loads no HIP library."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(16, (os.cpu_count() or 1) // n))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    lines = 0
    for ln in proc.stdout:
        if ln.startswith('{') and ln.rstrip().endswith('}'):
            sys.stdout.write(ln)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and lines == 0:
        sys.stderr.write('bench.py: the ranks exited cleanly but rank 0 printed no result line\n')
        rc = 1
    return rc


def dry_launch(args, world, rank, dist):
    """HOTK_BENCH_DRY_LAUNCH=1: the N > 1 start-up path up to (not including) device selection -- rendezvous, one MAX
    all-reduce (the timing reduction of the real run), rank 0's line -- so that the launcher is testable without a GPU."""
    from hot_amd.dist import max_over_ranks
    slowest = max_over_ranks(float(rank + 1), 'cpu')
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'dry-launch', 'n_gpus': world, 'gpus_arg': args.gpus, 'slowest_rank_plus_1': slowest,
                          'mode': args.mode}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def c2_mode(args, device, world, rank, dist):
    """--config c2: BASELINE config 2, the PCQM4M-LSC enc-small training step (regression/examples/pcqm4m-lsc/enc-small.sh:
    Encoder(2, 0, [2]*8, 512, 256, 256, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel'), 14.06 M parameters)
    on --molecules synthetic PCQM-like graphs per GPU (n ~ U{8..20} atoms, ~1.05 n bonds, symmetrised; the
    shipped per-GPU batch is 128), forward + L1 loss backward, fixed projections.  A second line, never the headline: the
    headline metric is the 2->2 layer on 10 M entries.  One GPU: the whole step is captured once into a HIP graph and
    replayed (every launch goes to torch's current stream, nothing allocates outside the caching allocator); several GPUs:
    eager steps with the parameter gradients all-reduced (graph-minibatch data parallelism)."""
    from hot_amd import _lib
    from hot_amd.batch.sparse import make_batch_concatenated
    from hot_amd.dist import allreduce_gradients, max_over_ranks
    from hot_amd.models.encoder import Encoder
    rng = np.random.default_rng(1000 + rank)
    nn_, ne_, eis = [], [], []
    for _ in range(args.molecules):
        n = int(rng.integers(8, 21))
        und = {(i, i + 1) for i in range(n - 1)}
        while len(und) < int(round(1.05 * n)):
            i, j = sorted(int(v) for v in rng.integers(0, n, 2))
            if i != j:
                und.add((i, j))
        e = sorted([(i, j) for i, j in und] + [(j, i) for i, j in und])
        eis.append(np.array(e, dtype=np.int64).T); nn_.append(n); ne_.append(len(e))
    ei = torch.from_numpy(np.concatenate(eis, axis=1)).to(device)
    g = torch.Generator(device=device).manual_seed(77 + rank)
    nf = torch.randn(sum(nn_), 256, device=device, generator=g)
    ef = torch.randn(sum(ne_), 256, device=device, generator=g)
    G = make_batch_concatenated(nf, ei, ef, nn_, ne_)
    torch.manual_seed(0)
    enc = Encoder(2, 0, [2] * 8, 512, 256, 256, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel', 0., 0., 0.,
                  sparse=True).to(device)
    enc.skip_redraw_projections = True
    head = torch.nn.Linear(256, 1).to(device)
    params = list(enc.parameters()) + list(head.parameters())
    n_entries = sum(G.n_edges)

    def fwd_bwd():
        y = head(enc(G)).squeeze(-1)
        torch.nn.functional.l1_loss(y, torch.zeros_like(y)).backward()

    def step():
        fwd_bwd()
        if dist is not None:
            allreduce_gradients(params)
        for p_ in params:
            p_.grad = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # small batches: the projections are library GEMMs of 10-40 us; pick the library's best kernel per shape during warm-up
    from hot_amd import ops
    from hot_amd.utils import tuning
    tuned = None
    if args.gemm_tuning == 'on' or (args.gemm_tuning == 'auto' and G.values.size(0) * G.values.size(1) < ops.G6_MIN_ROWS):
        tuning.start()
    for _ in range(max(args.warmup, 2)):
        step()
    if args.gemm_tuning != 'off':
        tuned = tuning.finish()
    fence()
    _lib.timing_begin(None)
    step()
    census = _lib.timing_end()
    n_kernels = None
    try:                                    # device kernels of one eager step (library GEMMs and torch glue included)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        n_kernels = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    except Exception:
        pass
    replay = dist is None and not args.no_graph
    if replay:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fwd_bwd()
        run = graph.replay
        for _ in range(3):
            run()
    else:
        run = step
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        print(json.dumps({
            'metric': 'entries/sec, PCQM4M-LSC enc-small training step (fwd + L1 loss bwd)', 'value': n_entries * world * args.steps / elapsed,
            'unit': 'entries/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'C2: enc-small Encoder (8 x 2->2 kernel-attention layers + softmax readout, D = 256, 14.06 M '
                                   f'parameters), {args.molecules} PCQM-like molecules per GPU = {n_entries} entries '
                                   f'({G.values.size(0) * G.values.size(1)} padded rows), '
                                   f'{"HIP-graph replay" if replay else "eager"}',
                       'entries_per_gpu': n_entries, 'parallelism': f'graph-minibatch dp{world}'},
            'hotk_launches_per_step': int(sum(v[1] for v in census.values())),
            'device_kernels_per_step': n_kernels,
            'library_gemm_tuning': (f'TunableOp, {tuned} shapes chosen in warm-up (hot_amd/utils/tuning.py)' if tuned else 'off'),
            'kernel_census_ms_per_step': {k: round(v[0], 3) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])[:12]},
            'cpu_reference_entries_per_s': 1.19e3,
            'cpu_reference_note': 'the real reference on 8 CPU cores, same model, 128 graphs (SURVEY section 6 / BASELINE.md)',
            'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'preconditioned': args.preconditioned,      # (see precondition(): device memory written once by an earlier process)
        }), flush=True)


_PRECONDITION = """
import sys, torch
dev = torch.device('cuda', int(sys.argv[1]))
free, _ = torch.cuda.mem_get_info(dev)
n = min(int(float(sys.argv[2]) * 2 ** 30), int(0.8 * free))
x = torch.empty(n, dtype=torch.uint8, device=dev)
x.zero_()
torch.cuda.synchronize(dev)
"""


def precondition(local_rank, gib=100.0):
    """Device memory that no process has used since the box came up is slower: the FIRST process to use the GPU of a fresh
    box runs every kernel of the step about 2 % slower than any later process (68.0-68.3 against 66.1-66.9 ms per step on
    five boxes; not a matter of warm-up length -- 25 or 100 warm-up steps: 67.6 / 68.1 -- nor of creating a context first:
    68.0; writing 200-270 GiB and freeing them inside the measuring process: 66.9 / 68.1).  After a short-lived process has
    written ~100 GiB of it and exited, the next process runs at the later-process speed (66.7).  So, before this process
    creates its own context, a child does exactly that.  Nothing of the measured job runs there; --no-precondition
    skips it, and the result line says which was done."""
    import subprocess
    t0 = time.perf_counter()
    try:
        rc = subprocess.run([sys.executable, '-c', _PRECONDITION, str(local_rank), str(gib)], timeout=300,
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
    except Exception:
        rc = -1
    return {'child_wrote_GiB': gib, 'child_rc': rc, 'seconds': round(time.perf_counter() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--entries', type=float, default=10e6, help='entries per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-kernel', default='auto', help='C-ABI entry point timed live with HIP events')
    ap.add_argument('--mode', default='graph-minibatch', choices=['graph-minibatch', 'edge-shard'])
    ap.add_argument('--gemm-tuning', default='auto', choices=['auto', 'on', 'off'],
                    help='--config c2: TunableOp kernel selection for the library GEMMs (auto: batches below the tiled GEMM threshold)')
    ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'],
                    help="bf16: the layer's bf16 STORAGE mode (wide intermediates in bf16, fp32 accumulation; tolerance "
                         "2e-2). A second line, never the headline: the reference trains in fp32")
    ap.add_argument('--shard-nodes', type=int, default=125000, help='edge-shard mode: nodes per GPU')
    ap.add_argument('--shard-pairs', type=int, default=3125000, help='edge-shard mode: undirected pairs per GPU')
    ap.add_argument('--config', default='c1', choices=['c1', 'c2'],
                    help='c1 (default, the headline): the 2->2 layer on ~10 M entries of BA(64,5) graphs; c2: the PCQM4M-LSC '
                         'enc-small training step (a second line)')
    ap.add_argument('--molecules', type=int, default=128, help='--config c2: molecules per GPU (enc-small.sh ships 128)')
    ap.add_argument('--no-graph', action='store_true', help='--config c2: eager steps instead of HIP-graph replay')
    ap.add_argument('--no-precondition', action='store_true',
                    help='do not let a short-lived child process write device memory before this process uses the GPU '
                         '(see precondition(): the first process on a fresh box is ~2 %% slower)')
    ap.add_argument('--sweep', action='store_true', help='run_perf_tests.py-style size sweep (SK / SL curves) instead')
    ap.add_argument('--sweep-max-n', type=int, default=None)
    args = ap.parse_args()

    if os.environ.get('HOTK_BENCH_DEBUG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['HOTK_BENCH_DEBUG']), exit=True)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus in the result '
                         f'line must equal --gpus')
    dry = os.environ.get('HOTK_BENCH_DRY_LAUNCH') == '1'
    # (before anything in this process touches the GPU)
    # (not under rocprofv3: its preloaded tool library has initialised the GPU in this process already, and a process that
    # has must not start another program)
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    args.preconditioned = None if (dry or args.no_precondition or args.sweep or profiled) else precondition(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # one rank per GPU over RCCL; HOTK_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs
        backend = 'gloo' if dry else os.environ.get('HOTK_BENCH_BACKEND', 'nccl')
        if dry:
            dist.init_process_group(backend)
            return dry_launch(args, world, rank, dist)
        if backend == 'nccl' and torch.cuda.device_count() < world:
            raise SystemExit(f'bench.py: {world} RCCL ranks need {world} GPUs, this node shows {torch.cuda.device_count()}')
        local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    if dry:
        return dry_launch(args, world, rank, None)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    from hot_amd import _lib
    from hot_amd.batch.sparse import batch_like
    from hot_amd.dist import allreduce_gradients, max_over_ranks

    if args.sweep:
        if rank == 0:
            print(json.dumps(sweep(device, n_max=args.sweep_max_n)), flush=True)
        return
    if args.config == 'c2':
        c2_mode(args, device, world, rank, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == 'edge-shard':
        edge_shard_mode(args, device, world, rank, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    n_graphs = max(1, int(round(args.entries / ENTRIES_PER_GRAPH)))
    G = make_inputs(n_graphs, device, seed=1234 + rank)
    n_entries = sum(G.n_edges)
    n_offdiag = n_entries - sum(G.n_nodes)
    # K1 (plan: radix sorts, CSR views, transpose map) is per minibatch, not per layer: timed apart from the step
    from hot_amd.plan import Plan
    Plan(G.indices, G.n_nodes, G.n_edges)
    torch.cuda.synchronize()
    t_k1 = time.perf_counter()
    Plan(G.indices, G.n_nodes, G.n_edges)
    torch.cuda.synchronize()
    k1_ms = (time.perf_counter() - t_k1) * 1e3
    layer = make_layer(device).set_storage(args.storage)
    params = [p for p in layer.parameters()]
    x0 = G.values

    def step():
        x = x0.detach().requires_grad_(True)
        out = layer(batch_like(G, x, skip_masking=True))
        out.values.sum().backward()
        if dist is not None:
            allreduce_gradients(params)
        for p in params:
            p.grad = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # which C-ABI entry point to time live: the one with the largest share in a one-step census
    if args.profile_kernel == 'auto':
        _lib.timing_begin(None)
        step()
        torch.cuda.synchronize()
        census = _lib.timing_end()
        prof_name = max(census, key=lambda k: census[k][0]) if census else None
    else:
        census, prof_name = {}, args.profile_kernel
    fence()
    _lib.timing_begin(prof_name)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timed = _lib.timing_end()
    elapsed = max_over_ranks(elapsed, device)
    total_entries = n_entries * world
    value = total_entries * args.steps / elapsed

    if rank == 0:
        bpe_entry = BYTES_PER_ENTRY if args.storage == 'f32' else (96 * DQK + 224 * DV) // 2 + 50
        bpe = bpe_entry + BYTES_PER_NODE * sum(G.n_nodes) / n_entries
        layer_frac = bpe * (value / world) / HBM_PEAK
        roof = None
        if prof_name and prof_name in timed:
            tot_ms, calls, alg_bytes = timed[prof_name]
            ach = alg_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
            traffic = PMC_TRAFFIC_BYTES_PER_ENTRY.get(prof_name)
            roof = {'bound': 'hbm', 'kernel': prof_name, 'achieved': ach, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'frac': ach * 1e9 / HBM_PEAK, 'traffic': None if traffic is None else traffic * n_entries,
                    'traffic_source': 'NOT measured in this run (a bench cannot profile itself): constant from separate rocprofv3 '
                                      '--pmc FETCH_SIZE / WRITE_SIZE passes at the benchmarked size (10 M entries, '
                                      'traffic_profile), B/entry x entries of this launch; fp32 storage',
                    'traffic_profile': PMC_TRAFFIC_SOURCE,
                    'launches': calls,
                    'avg_launch_ms': tot_ms / max(calls, 1),
                    'note': 'achieved = algorithmic bytes (operands read once + results written once) of the '
                            'dominant C-ABI kernel / its HIP-event time inside the timed region; the kernel is '
                            'instruction-issue bound (VALU 86 % busy; L2 -> fabric traffic 1.4-1.6x algorithmic), see DESIGN.md 4'}
        line = {
            'metric': 'edges/sec fwd+bwd, 2->2 sparse kernel attn layer', 'value': value, 'unit': 'entries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.storage == 'f32' else 'bf16 storage / f32 accumulate', 'data': 'synthetic',
            'config': {'workload': f'S1/C1: {n_graphs} BA(64,5) graphs per GPU = {n_entries} entries '
                                   f'({n_offdiag} off-diagonal), KernelSelfAttn(2,2,32,32,32,4) generalized FAVOR+ '
                                   f'(d=8,m=16), fwd + sum().backward()',
                       'entries_per_gpu': n_entries, 'parallelism': f'graph-minibatch dp{world}'},
            'layer_roofline': {'bytes_per_entry': bpe, 'frac_of_hbm_peak_per_gpu': layer_frac,
                               'note': 'SURVEY 8d whole-layer figure: attention-core algorithmic bytes x entries/s '
                                       '/ 8 TB/s, GEMM time included in the denominator'},
            'roofline': roof,
            'k1_plan_build_ms': round(k1_ms, 3),
            'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'preconditioned': args.preconditioned,      # (see precondition(): device memory written once by an earlier process)
            'kernel_census_ms_per_step': {k: round(v[0], 3) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])},
        }
        if args.no_cpu_baseline or world > 1:
            line['cpu_baseline'] = None
        else:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
