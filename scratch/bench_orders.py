"""scratch: step time + census of the layers that run on the GENERAL kernels at the C1 head shape (d = 8, m = 16):
KernelSelfAttn(2,1), (1,1) on the node batch, and (2,2) with the softmax feature map.  usage: bench_orders.py <entries>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib
from hot_amd.batch.sparse import batch_like
from hot_amd.models.common import KernelFeatureMap
from hot_amd.models.sparse import KernelSelfAttn

entries = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
dev = torch.device('cuda', 0)
G = bench.make_inputs(int(round(entries / 654)), dev, 1234)
n = sum(G.n_edges)

def run(tag, layer, Gin):
    params = list(layer.parameters())
    def step():
        x = Gin.values.detach().requires_grad_(True)
        layer(batch_like(Gin, x, skip_masking=True)).values.sum().backward()
        for p in params: p.grad = None
    for _ in range(2): step()
    torch.cuda.synchronize()
    _lib.timing_begin(None); step(); census = _lib.timing_end()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'{tag}: {dt*1e3:.1f} ms/step = {n/dt/1e6:.2f} M entries/s')
    for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f'    {k:30s} {v[0]:8.2f} ms {v[1]:4d} calls')

torch.manual_seed(0)
run('(2,2) softmax map, general kernels', KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=False)).to(dev), G)
l21 = KernelSelfAttn(2, 1, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=True)).to(dev)
run('(2,1) ReLU map', l21, G)
with torch.no_grad():
    G1 = l21(G)
run('(1,1) ReLU map (node batch)', KernelSelfAttn(1, 1, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=True)).to(dev), G1)
