import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
M, per = 253952, 62
K, N = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(0)
nv = torch.randint(25, per + 1, (M // per + 1,), generator=g)
pos = torch.arange(M)
rows = torch.nonzero((pos % per) < nv[pos // per]).view(-1).to(torch.int32).cuda()
A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') / K ** 0.5
for _ in range(5): ops.mm(A, W)
for _ in range(5): ops.mm(A, W, rows=rows)
torch.cuda.synchronize()
