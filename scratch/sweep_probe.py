"""the SK model of bench.py --sweep at one size: per-kernel profile target"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hot_amd.batch.sparse import make_batch
from hot_amd.models import Encoder, MLP
from hot_amd.utils.graphs import ba_edges
n = int(sys.argv[1]) if len(sys.argv) > 1 else 52428
which = sys.argv[2] if len(sys.argv) > 2 else 'SK'
dev = 'cuda'
torch.manual_seed(0)
if which == 'SK':
    model = Encoder(2, 0, [2] * 4, 32, 32, 32, 32, 32, 32, 4, 32, 32, 4, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(dev)
    model.skip_redraw_projections = True
else:
    model = MLP(2, 0, [2] * 4, 32, 32, 32, sparse=True).to(dev)
for it in range(3):
    ei = torch.from_numpy(ba_edges(n, 5, np.random.default_rng(it))).to(dev)
    nf = torch.randn(n, 16, device=dev); ef = torch.randn(ei.size(1), 16, device=dev)
    G = make_batch([nf], [ei], [ef])
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); y = model(G); e1.record(); y.sum().backward(); e2.record(); torch.cuda.synchronize()
    print(it, 'fwd', e0.elapsed_time(e1), 'bwd', e1.elapsed_time(e2))
    model.zero_grad(set_to_none=True)
