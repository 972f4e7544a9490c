import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M = 254016
for K, N in [(512, 512), (512, 512), (256, 256), (256, 768), (512, 256), (1280, 512), (256, 1536)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') / K ** 0.5
    y = ops.mm(A, W); err = float((y - A @ W).abs().max())
    for _ in range(3): ops.mm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.mm(A, W)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f'[{M},{K}]x[{K},{N}] nn {t:.3f} ms ({2.0 * M * K * N / t / 1e9:.0f} TF) max diff vs library {err:.1e}')
