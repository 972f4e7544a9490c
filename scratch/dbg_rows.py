import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops, _lib
M, K, N = 16477, 256, 320
A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda')
rows = torch.arange(5, M - 9, dtype=torch.int32, device='cuda')
for name, r in (('all', None), ('rows', rows)):
    try:
        y = ops.mm(A, W, rows=r); torch.cuda.synchronize(); print(name, 'ok', float(y.abs().max()))
    except Exception as e:
        print(name, 'FAILED', e)
