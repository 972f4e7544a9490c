"""Micro-benchmark of the fused attention-core entry points alone (no GEMMs): scratch tool."""
import sys, os, time
from ctypes import byref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib, ops
from hot_amd._lib import p

entries = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
nG = int(round(entries / 654))
G = bench.make_inputs(nG, dev, 1234)
plan = G.plan
torch.manual_seed(0)
h1 = torch.randn(plan.Nn, 384, device=dev)
h2 = torch.randn(plan.R, 288, device=dev)
v = torch.randn(plan.R, 320, device=dev)
proj = torch.randn(16, 8, device=dev)
att = torch.empty(plan.R, 320, device=dev)
kvg = torch.empty(plan.B, 4, 25, 64, device=dev)
datt = torch.randn(plan.R, 320, device=dev)
dout = torch.randn(plan.R, 32, device=dev)
dh1 = torch.empty(plan.Nn, 384, device=dev); dh2 = torch.empty(plan.R, 288, device=dev)
dv = torch.empty(plan.R, 320, device=dev); ck = torch.empty(plan.R, 192, device=dev); dkvg = torch.empty_like(kvg)
a = ops.AttnCore22._args(plan, h1, h2, v, proj, 32)
phik = ops.AttnCore22._phik(plan, h1, proj); a.phik, a.ld_phik = phik.data_ptr(), 512
a.att, a.ld_att, a.kvg = att.data_ptr(), 320, kvg.data_ptr()
a.datt, a.ld_datt = datt.data_ptr(), 320
a.dout, a.ld_dout = dout.data_ptr(), 32
a.dh1, a.ld_dh1 = dh1.data_ptr(), 384
a.dh2, a.ld_dh2 = dh2.data_ptr(), 288
a.dv, a.ld_dv = dv.data_ptr(), 320
a.ck, a.dkvg = ck.data_ptr(), dkvg.data_ptr()
ws, nws = ops.AttnCore22._workspace(plan, dev)
L = _lib.lib()
def run(fn):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    rc = fn(byref(a), p(ws), nws, _lib.stream_ptr()); assert rc == 0, rc
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn(byref(a), p(ws), nws, _lib.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
f = run(L.hotk_attn22_forward); b = run(L.hotk_attn22_backward)
print(f'entries {plan.Ev} nodes {plan.Nn}: fwd {f:.3f} ms  bwd {b:.3f} ms')
