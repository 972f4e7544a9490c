import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import rel_err
from oracle import hot_oracle as O
from hot_amd.batch import sparse as bs
from hot_amd.models.common import KernelFeatureMap
from hot_amd.models import sparse as ms
DEV='cuda'
rng = np.random.default_rng(5)
nfs, eis, efs = [], [], []
g = torch.Generator().manual_seed(7)
for _ in range(12):
    ei = torch.from_numpy(O.ba_edges(64, 5, rng))
    nfs.append(torch.randn(64, 16, generator=g)); eis.append(ei); efs.append(torch.randn(ei.size(1), 16, generator=g))
G0 = bs.make_batch([x.to(DEV) for x in nfs], [x.to(DEV) for x in eis], [x.to(DEV) for x in efs])
torch.manual_seed(0)
layer = ms.KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0., 0., KernelFeatureMap(8, generalized_attention=True)).to(DEV)
x = G0.values.clone().requires_grad_(True)
y = layer(bs.batch_like(G0, x, skip_masking=True))
R = torch.randn(y.values.shape, generator=g)
(y.values * R.to(DEV)).sum().backward()
res = {}
for dt in (torch.float32, torch.float64):
    GO = O.build_batch_from_lists([a.to(dt) for a in nfs], eis, [a.to(dt) for a in efs])
    sd = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()).clone().requires_grad_(v.dtype == torch.float32 and 'projection' not in k) for k, v in layer.state_dict().items()}
    xo = GO.values.clone().requires_grad_(True)
    yo = O.kernel_self_attn(GO.like(xo), sd, 2, 2, 32, 32, 32, 4, 'default', True)
    (yo.values * R.to(dt)).sum().backward()
    res[dt] = (yo.values.detach(), xo.grad, {k: sd[k].grad for k, _ in layer.named_parameters()})
y64, gx64, gp64 = res[torch.float64]
y32, gx32, gp32 = res[torch.float32]
print('y   hip-vs-64', rel_err(y.values.detach().cpu(), y64), ' o32-vs-64', rel_err(y32, y64))
print('gx  hip-vs-64', rel_err(x.grad.cpu(), gx64), ' o32-vs-64', rel_err(gx32, gx64))
for k, p_ in layer.named_parameters():
    print(k, 'hip-vs-64', rel_err(p_.grad.cpu(), gp64[k]), ' o32-vs-64', rel_err(gp32[k], gp64[k]))
d = (x.grad.cpu().double() - gx64).abs()
idx = torch.nonzero(d > 1e-4)
print('n bad', idx.shape[0], idx[:10].tolist(), d.max().item())
rows = torch.unique(idx[:, :2], dim=0)
print('bad rows', rows[:10].tolist(), 'n_nodes 64')
# which pre-activation of node (7,39) is near zero?
with torch.no_grad():
    h1 = layer.fc_1(bs.batch_like(G0, G0.values, skip_masking=True)).values.cpu().double()   # [B,N,12*32]
    P = layer.feature_map.feature_map.projection_matrix.cpu().double()
    ch = h1.view(12, 64, 48, 8)   # [B,N,slot*head,d]
    pre = (ch * 8 ** -0.25) @ P.t()
    a = pre.abs()
    print('min |pre| overall', a.min().item(), 'at', np.unravel_index(a.argmin().item(), a.shape))
    print('min |pre| node(7,39)', a[7, 39].min().item())
    print('count |pre|<1e-6', (a < 1e-6).sum().item(), ' <1e-5', (a < 1e-5).sum().item())
