"""scratch: the tiled GEMMs over the valid rows of a padded batch (70 % valid, per-graph prefixes) against all rows"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M, per = 253952, 62
g = torch.Generator().manual_seed(0)
nv = torch.randint(25, per + 1, (M // per + 1,), generator=g)
pos = torch.arange(M)
mask = (pos % per) < nv[pos // per]
rows = torch.nonzero(mask).view(-1).to(torch.int32).cuda()
print(f'{rows.numel()} of {M} rows valid ({rows.numel() / M:.2f})')
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K, N in [(512, 512), (256, 320), (256, 864), (320, 256), (1280, 512), (512, 256)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') / K ** 0.5; G = torch.randn(M, N, device='cuda')
    a, b = t(lambda: ops.mm(A, W)), t(lambda: ops.mm(A, W, rows=rows))
    line = f'[{M},{K}]x[{K},{N}] nn all {a:.3f} ms, listed {b:.3f} ms ({b / a:.2f})'
    if ops.wgrad6_ok(A, G):
        c, d = t(lambda: ops.wgrad6(A, G)), t(lambda: ops.wgrad6(A, G, rows))
        line += f'; tn all {c:.3f} ms, listed {d:.3f} ms ({d / c:.2f})'
    print(line)
