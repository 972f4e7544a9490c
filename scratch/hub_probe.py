"""one BA(n,5) graph, single KernelSelfAttn(2,2) layer fwd+bwd: census per C-ABI entry point and degree statistics"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hot_amd import _lib
from hot_amd.batch.sparse import make_batch, batch_like
from hot_amd.utils.graphs import ba_edges
from bench import make_layer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 52428
dev = 'cuda'
ei = torch.from_numpy(ba_edges(n, 5, np.random.default_rng(1))).to(dev)
nf = torch.randn(n, 16, device=dev); ef = torch.randn(ei.size(1), 16, device=dev)
G = make_batch([nf], [ei], [ef])
p = G.plan
print('entries', p.Ev, 'max col deg', p.max_col_deg, 'max row deg', p.max_row_deg)
deg = (p.col_ptr[1:] - p.col_ptr[:-1]).float()
for t in (64, 128, 256, 512, 1024):
    print('nodes with col degree >', t, int((deg > t).sum()), 'entries in them', int(deg[deg > t].sum()))
layer = make_layer(dev)
for it in range(3):
    x = G.values.detach().requires_grad_(True)
    if it == 2:
        _lib.timing_begin(None)
    y = layer(batch_like(G, x, skip_masking=True))
    y.values.sum().backward()
census = _lib.timing_end()
for k, v in sorted(census.items(), key=lambda kv: -kv[1][0]):
    print(f'{k:28s} {v[0]:8.3f} ms  {v[1]} calls')
