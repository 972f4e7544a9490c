"""scratch: error of the entry GEMM (hotk_linear22_entry_fwd through Linear(2,2,32,288,'light')) and of the thin K = 32
product against fp64, for the current HOTK_GEMM_F32 / HOTK_GEMM_B6_NARROW setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
torch.manual_seed(0)
R = 1 << 20
A = torch.randn(R, 32, device='cuda'); W = torch.randn(32, 320, device='cuda') * 32 ** -0.5
y = ops.thin_gemm(A, W); ref = A.double() @ W.double()
print('env F32=%s NARROW=%s' % (os.environ.get('HOTK_GEMM_F32'), os.environ.get('HOTK_GEMM_B6_NARROW')))
print('thin K=32 -> 320   max|err|/max|ref| = %.3e' % float((y.double() - ref).abs().max() / ref.abs().max()))
from bench import make_inputs
from hot_amd.models.sparse import Linear
G = make_inputs(1600, torch.device('cuda', 0), 1)
lin = Linear(2, 2, 32, 288, cfg='light').cuda()
y = lin(G).values
lin64 = Linear(2, 2, 32, 288, cfg='light').cuda()
lin64.load_state_dict(lin.state_dict())
ops.THIN_MIN_ROWS = 1 << 40          # library path (fp32) as a second reference
y_lib = lin64(G).values
# fp64 reference of the factored form
D, C = 32, 288
Wt = lin.weight.weight.double(); x = G.values.double().view(-1, D); plan = G.plan
kind = plan.kind.long(); rn = plan.rown.long().clamp_min(0); cn = plan.coln.long().clamp_min(0); tp = plan.tpos.long()
xd = x[plan.diag_pos.long().clamp_min(0)]
xt = torch.where((tp >= 0).unsqueeze(1), x[tp.clamp_min(0)], torch.zeros_like(x))
b_off, b_diag = [b.double() for b in lin.bias.bias]
off = x @ Wt[D:2 * D] + xt @ Wt[2 * D:3 * D] + xd[rn] @ Wt[3 * D:4 * D] + xd[cn] @ Wt[4 * D:5 * D] + b_off
dg = x @ Wt[0:D] + b_diag
ref = torch.where((kind == 2).unsqueeze(1), off, torch.where((kind == 1).unsqueeze(1), dg, torch.zeros_like(off)))
e = lambda t: float((t.double().view(-1, C) - ref).abs().max() / ref.abs().max())
print('entry GEMM (hand-written) vs fp64: %.3e   library path vs fp64: %.3e' % (e(y), e(y_lib)))
