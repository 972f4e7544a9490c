import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M = 254016
for K, N in [(512, 512), (256, 320), (256, 864), (128, 576), (320, 256), (1280, 512), (256, 128)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') / K ** 0.5
    for _ in range(3): ops.mm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.mm(A, W)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    for _ in range(3): A @ W
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): A @ W
    e1.record(); torch.cuda.synchronize(); tl = e0.elapsed_time(e1) / 10
    print(f'[{M},{K}]x[{K},{N}] nn {t:.3f} ms ({2.0 * M * K * N / t / 1e9:.0f} TF; library {tl:.3f} ms)')
