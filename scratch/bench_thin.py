"""Micro-benchmark of the thin GEMM entry points at the C1 shapes (scratch tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib, ops
from hot_amd._lib import p, call

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
torch.manual_seed(0)

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for K, C in [(32, 320), (320, 32), (288, 64), (64, 288)]:
    A = torch.randn(R, K, device=dev); W = torch.randn(K, C, device=dev); b = torch.randn(C, device=dev)
    t_thin = timeit(lambda: ops.thin_gemm(A, W, b))
    t_lib = timeit(lambda: torch.addmm(b, A, W))
    gb = 4.0 * R * (K + C) / 1e9
    print(f'K={K:3d} C={C:3d}: thin {t_thin:6.3f} ms ({gb / t_thin:5.2f} TB/s... {2e-9 * R * K * C / t_thin:6.1f} TFLOP/s)   library {t_lib:6.3f} ms')
    del A, W
# entry GEMM with the assembly epilogue on the bench batch
nG = int(round(R / 654))
G = bench.make_inputs(nG, dev, 1234)
plan = G.plan
n, D, Cc = plan.R, 32, 288
x2 = torch.randn(n, 2 * D, device=dev); W = torch.randn(2 * D, Cc, device=dev); yn = torch.randn(plan.Nn, 3 * Cc, device=dev)
bo = torch.randn(Cc, device=dev); bd = torch.randn(Cc, device=dev); y = torch.empty(n, Cc, device=dev)
t = timeit(lambda: call('hotk_linear22_entry_fwd', 0, p(x2), 2 * D, p(W), 2 * D, Cc, p(yn), 3 * Cc, p(plan.kind), p(plan.rown),
                        p(plan.coln), p(bo), p(bd), p(y), Cc, n, plan.Nn))
print(f'entry GEMM + assembly ({n} rows): {t:6.3f} ms ({4.0 * n * (2 * D + Cc) / 1e9 / t:5.2f} TB/s)')
