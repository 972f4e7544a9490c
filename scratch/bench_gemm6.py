"""scratch: split-bf16 tiled GEMM (csrc/gemm6.hip) against the library's fp32 GEMM at C2-shaped sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 254016
dev = 'cuda'


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for K, N in [(512, 512), (256, 320), (320, 256), (512, 256), (256, 864), (128, 576), (640, 64), (1280, 512)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5; G = torch.randn(M, N, device=dev)
    fl = 2.0 * M * K * N
    t6 = timeit(lambda: ops.mm(A, W)); tl = timeit(lambda: A @ W)
    ops._G6 = False; twl = timeit(lambda: ops.wgrad(A, G)); ops._G6 = True
    tw6 = timeit(lambda: ops.wgrad(A, G)) if ops.wgrad6_ok(A, G) else float('nan')
    print(f'[{M},{K}]x[{K},{N}]  nn: gemm6 {t6:.3f} ms ({fl / t6 / 1e9:.0f} TF)  lib {tl:.3f} ms ({fl / tl / 1e9:.0f} TF) | '
          f'tn: gemm6 {tw6:.3f} ms ({fl / tw6 / 1e9:.0f} TF)  lib split-K {twl:.3f} ms ({fl / twl / 1e9:.0f} TF)')
