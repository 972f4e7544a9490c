"""HIP layer vs the oracle in fp64, with the oracle's own fp32-vs-fp64 gap beside it (tests/test_gpu_parity.py bars)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_gpu_parity as T
import test_gpu_c2 as C2
from oracle import hot_oracle as O
DEV = 'cuda'
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def oracle(run, sd_src, x_src, dt):
    sd = {k: (v.detach().cpu().clone().to(dt) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in sd_src.items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'projection' not in k:
            v.requires_grad_(True)
    xo = x_src.clone().to(dt).requires_grad_(True)
    y = run(xo, sd, dt)
    return y.detach(), xo.grad, {k: v.grad for k, v in sd.items() if v.requires_grad}


def report(tag, y, gx, pg, o32, o64):
    print(f'== {tag}')
    print(f'  out : hip-o64 {rel(y, o64[0]):.2e}   o32-o64 {rel(o32[0], o64[0]):.2e}')
    s = o64[1].abs().max()
    rh = (gx.double() - o64[1]).abs().amax(-1) / s
    ro = (o32[1].double() - o64[1]).abs().amax(-1) / s
    print(f'  gx  : hip-o64 max {float(rh.max()):.2e} rows>2e-5 {float((rh > 2e-5).double().mean()):.2e}   o32-o64 max {float(ro.max()):.2e} rows>2e-5 {float((ro > 2e-5).double().mean()):.2e}')
    scale = max(float(v.abs().max()) for v in o64[2].values())
    for k in pg:
        ref = o64[2][k]
        den = max(float(ref.abs().max()), 1e-3 * scale)
        print(f'  {k:40s} hip-o64 {float((pg[k].double() - ref).abs().max()) / den:.2e}   o32-o64 {float((o32[2][k].double() - ref).abs().max()) / den:.2e}')


def layer_case(tag, nfs, eis, efs, g, generalized, fused, dqk):
    _, _, _, bs, KernelFeatureMap, ms, _ = T._product()
    GO = O.build_batch_from_lists(nfs, eis, efs)
    G0 = bs.make_batch([x.to(DEV) for x in nfs], [x.to(DEV) for x in eis], [x.to(DEV) for x in efs])
    torch.manual_seed(0)
    layer = ms.KernelSelfAttn(2, 2, 32, dqk, dqk, 4, 'default', 0., 0., KernelFeatureMap(dqk // 4, generalized_attention=generalized)).to(DEV)
    layer.use_fused = fused
    x = G0.values.clone().requires_grad_(True)
    y = layer(bs.batch_like(G0, x, skip_masking=True))
    R = torch.randn(y.values.shape, generator=g)
    (y.values * R.to(DEV)).sum().backward()

    def run(xo, sd, dt):
        yo = O.kernel_self_attn(GO.like(xo), sd, 2, 2, 32, dqk, dqk, 4, 'default', generalized)
        (yo.values * R.to(dt)).sum().backward()
        return yo.values
    o32 = oracle(run, layer.state_dict(), GO.values, torch.float32)
    o64 = oracle(run, layer.state_dict(), GO.values, torch.float64)
    report(tag, y.values.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in layer.named_parameters()}, o32, o64)


for gen, fused, dqk in [(False, False, 32), (True, False, 32), (True, True, 32), (True, False, 64), (False, False, 64)]:
    layer_case(f'BA generalized={gen} fused={fused} dqk={dqk}', *T._ba_case(12, 5), gen, fused, dqk)
layer_case('ragged fused', *T._ragged_case(), True, True, 32)

# C2 real dims
from hot_amd.batch import sparse as bs
from hot_amd.models.encoder import Encoder
L, DH = 3, 256
nfs, eis, efs, g = C2._molecules(128, 21, 2 * DH)
GO = O.build_batch_from_lists(nfs, eis, efs)
G0 = C2._to_dev(nfs, eis, efs)
torch.manual_seed(0)
enc = Encoder(2, 0, [2] * L, 2 * DH, DH, DH, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(DEV)
enc.skip_redraw_projections = True
enc.eval()
x = G0.values.clone().requires_grad_(True)
y = enc(bs.batch_like(G0, x, skip_masking=True))
R = torch.randn(y.shape, generator=g)
(y * R.to(DEV)).sum().backward()


def run(xo, sd, dt):
    yo = O.encoder_2_0(GO.like(xo), sd, L, DH, 32, 32, 256, 4, 128, 128, 8, True)
    (yo * R.to(dt)).sum().backward()
    return yo
o32 = oracle(run, enc.state_dict(), GO.values, torch.float32)
o64 = oracle(run, enc.state_dict(), GO.values, torch.float64)
pg = {k: p.grad.cpu() for k, p in enc.named_parameters()}
scale = max(float(v.abs().max()) for v in o64[2].values())
worst = sorted(((float((pg[k].double() - o64[2][k]).abs().max()) / max(float(o64[2][k].abs().max()), 1e-3 * scale),
                 float((o32[2][k].double() - o64[2][k]).abs().max()) / max(float(o64[2][k].abs().max()), 1e-3 * scale), k) for k in pg), reverse=True)
print('== C2 real dims')
print(f'  out : hip-o64 {rel(y.detach().cpu(), o64[0]):.2e}   o32-o64 {rel(o32[0], o64[0]):.2e}')
s = o64[1].abs().max()
rh = (x.grad.cpu().double() - o64[1]).abs().amax(-1) / s
ro = (o32[1].double() - o64[1]).abs().amax(-1) / s
print(f'  gx  : hip-o64 max {float(rh.max()):.2e} rows>2e-5 {float((rh > 2e-5).double().mean()):.2e}   o32-o64 max {float(ro.max()):.2e} rows>2e-5 {float((ro > 2e-5).double().mean()):.2e}')
for e, e32, k in worst[:8]:
    print(f'  {k:50s} hip-o64 {e:.2e}   o32-o64 {e32:.2e}')
