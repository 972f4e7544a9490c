"""scratch: one-rank RCCL process group on the box's GPU: the collectives hot_amd.dist issues (flat gradient bucket,
MAX of a scalar, all-reduce of a large fp32 buffer) run through backend 'nccl' end to end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from hot_amd.dist import allreduce_gradients, max_over_ranks, ShardInfo
lin = torch.nn.Linear(8, 8).cuda(); lin(torch.randn(4, 8, device='cuda')).sum().backward()
g0 = lin.weight.grad.clone()
# world 1: allreduce_gradients returns early; call the collective directly as well
allreduce_gradients(lin.parameters()); assert torch.equal(g0, lin.weight.grad)
t = torch.ones(1 << 28, device='cuda')      # 1 GiB fp32
dist.all_reduce(t); torch.cuda.synchronize(); t0 = time.perf_counter(); dist.all_reduce(t); torch.cuda.synchronize()
print('rccl all_reduce 1 GiB (1 rank): %.2f ms, sum check %s' % ((time.perf_counter() - t0) * 1e3, float(t[0]) == 1.0))
print('max_over_ranks', max_over_ranks(3.5, torch.device('cuda', 0)))
dist.barrier(); dist.destroy_process_group(); print('rccl probe ok')
