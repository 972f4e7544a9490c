"""scratch: per-wave in-kernel timing of k_seg_fwd<1> (instrumented library build only)"""
import sys, os
from ctypes import byref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from hot_amd import _lib, ops
from hot_amd._lib import p
entries = float(sys.argv[1]) if len(sys.argv) > 1 else 3e6
dev = torch.device('cuda', 0)
G = bench.make_inputs(int(round(entries / 654)), dev, 1234)
plan = G.plan
h1 = torch.randn(plan.Nn, 384, device=dev); h2 = torch.randn(plan.R, 288, device=dev); v = torch.randn(plan.R, 320, device=dev)
proj = torch.randn(16, 8, device=dev); att = torch.empty(plan.R, 320, device=dev); kvg = torch.empty(plan.B, 4, 25, 64, device=dev)
ck = torch.zeros(plan.R, 192, device=dev)
a = ops.AttnCore22._args(plan, h1, h2, v, proj, 32)
phik = ops.AttnCore22._phik(plan, h1, proj); a.phik, a.ld_phik = phik.data_ptr(), 512
a.att, a.ld_att, a.kvg, a.ck = att.data_ptr(), 320, kvg.data_ptr(), ck.data_ptr()
ws, nws = ops.AttnCore22._workspace(plan, dev)
L = _lib.lib()
for _ in range(3):
    rc = L.hotk_attn22_forward(byref(a), p(ws), nws, _lib.stream_ptr()); assert rc == 0
torch.cuda.synchronize()
nw = 8192 * 4
d = ck.view(-1)[:nw * 8].view(torch.int32).view(nw, 8).cpu().numpy().astype(np.int64) & 0xffffffff
life, tacc, ent, nodes = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
t0 = d[:, 4] + (d[:, 5] << 32)
t1 = t0 + life
xcc = d[:, 7] & 15
hw = d[:, 7] >> 4
print('waves', nw, 'entries/wave %.1f nodes/wave %.1f' % (ent.mean(), nodes.mean()))
print('lifetime cycles mean %.0f max %.0f min %.0f; acc-phase share %.2f; cycles/entry %.0f' % (life.mean(), life.max(), life.min(), tacc.sum() / life.sum(), life.sum() / ent.sum()))
for x in range(8):
    m = xcc == x
    if m.sum() == 0: continue
    span = t1[m].max() - t0[m].min()
    print('xcc', x, 'waves', m.sum(), 'span', span, 'avg resident waves %.0f' % (life[m].sum() / span), 'per CU %.1f' % (life[m].sum() / span / 32))
    # concurrency profile over 10 time slices
    edges = np.linspace(t0[m].min(), t1[m].max(), 11)
    prof = [int(((t0[m] < e1) & (t1[m] > e0)).sum()) for e0, e1 in zip(edges[:-1], edges[1:])]
    print('   active-in-slice', prof)
    break
print('hw_id sample', [hex(int(h)) for h in hw[:8]])
