"""time hotk_linear22_bwd_reduce alone at the C1 size: python scratch/bench_l22_reduce.py [n_graphs]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 15291
G = bench.make_inputs(n_graphs, 'cuda', 1)
plan = G.plan
C = 288
g = torch.randn(plan.R, C, device='cuda')
x2 = torch.randn(plan.R, 64, device='cuda')
L = _lib.lib(); p = _lib.p
n = ctypes.c_int64(); L.hotk_linear22_bwd_reduce_partials(plan.B, ctypes.byref(n))
dyn = torch.empty(plan.Nn, 3 * C, device='cuda'); part = torch.empty(n.value, 64, C, device='cuda')
def run():
    rc = L.hotk_linear22_bwd_reduce(p(g), C, p(x2), 64, 64, C, p(plan.kind), p(plan.rown), p(plan.coln), p(plan.g_ebeg),
                                    p(plan.g_eend), p(plan.n_nodes_dev), plan.B, plan.Nmax, plan.R, p(dyn), 3 * C, p(part),
                                    _lib.stream_ptr())
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'{os.environ.get("HOTK_LIB", "default")}: {ms:.3f} ms  {plan.R * 4 * (C + 64) / ms / 1e9:.2f} TB/s  rows {plan.R}', flush=True)
