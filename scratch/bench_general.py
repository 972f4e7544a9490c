"""scratch: the 2->2 layer on the GENERAL kernels (csrc/attn.hip, csrc/phi.hip) at a given head shape: step time and the
per-entry-point census.  usage: bench_general.py <entries> <dqk> [generalized=1]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib
from hot_amd.batch.sparse import batch_like
from hot_amd.models.common import KernelFeatureMap
from hot_amd.models.sparse import KernelSelfAttn

entries = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
dqk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
gen = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
dev = torch.device('cuda', 0)
G = bench.make_inputs(int(round(entries / 654)), dev, 1234)
torch.manual_seed(0)
layer = KernelSelfAttn(2, 2, 32, dqk, dqk, 4, 'default', 0., 0., KernelFeatureMap(dqk // 4, generalized_attention=gen)).to(dev)
layer.use_fused = False
params = list(layer.parameters())
def step():
    x = G.values.detach().requires_grad_(True)
    layer(batch_like(G, x, skip_masking=True)).values.sum().backward()
    for p in params: p.grad = None
for _ in range(2): step()
torch.cuda.synchronize()
_lib.timing_begin(None); step(); census = _lib.timing_end()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 3
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
n = sum(G.n_edges)
print(f'entries {n} dqk {dqk} m {layer.feat_dim} generalized {gen}: {dt*1e3:.1f} ms/step = {n/dt/1e6:.2f} M entries/s')
for k, v in sorted(census.items(), key=lambda kv: -kv[1][0]):
    print(f'  {k:32s} {v[0]:9.2f} ms  {v[1]:4d} calls')
