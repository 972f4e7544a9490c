import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M, K, N = 254016, 512, 512
A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') / K ** 0.5; G = torch.randn(M, N, device='cuda')
for _ in range(3):
    ops.mm(A, W); ops.wgrad(A, G)
torch.cuda.synchronize()
