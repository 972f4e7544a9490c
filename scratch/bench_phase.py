"""scratch: the fused 2->2 layer in PHASES (one-rank shard context) on the C1 batch: step time and census; run under
rocprofv3 --kernel-trace --stats to see the split-phase kernels.  usage: bench_phase.py <entries>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib
from hot_amd.batch.sparse import batch_like
from hot_amd.dist import ShardInfo
from hot_amd.models.common import KernelFeatureMap
from hot_amd.models.sparse import KernelSelfAttn

entries = float(sys.argv[1]) if len(sys.argv) > 1 else 2e6
dev = torch.device('cuda', 0)
G = bench.make_inputs(int(round(entries / 654)), dev, 1234)
G.plan.shard = ShardInfo(0, 1, (G.plan.kind != 0).to(torch.float32))
torch.manual_seed(0)
layer = KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=True)).to(dev)
params = list(layer.parameters())
def step():
    x = G.values.detach().requires_grad_(True)
    layer(batch_like(G, x, skip_masking=True)).values.sum().backward()
    g = float(x.grad.abs().sum())
    for p in params: p.grad = None
    return g
for _ in range(2): gsum = step()
torch.cuda.synchronize()
_lib.timing_begin(None); step(); census = _lib.timing_end()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 3
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
n = sum(G.n_edges)
print(f'entries {n}: {dt*1e3:.1f} ms/step phased; |dx|_1 = {gsum:.6e}')
for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])[:6]:
    print(f'  {k:32s} {v[0]:9.2f} ms  {v[1]:4d} calls')
