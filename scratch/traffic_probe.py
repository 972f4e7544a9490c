"""scratch: one launch of the fused core (fwd, bwd) at bench size plus calibration copies of known byte counts, for
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads)."""
import sys, os
from ctypes import byref, c_float
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib, ops
from hot_amd._lib import p

entries = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
dev = torch.device('cuda', 0)
L = _lib.lib()
# calibration: hotk_mask_rows on 2^28 floats (1 GiB in, 1 GiB out): float4 path (C=256) and scalar path (C=257 -> ld 257)
n = 1 << 20
kind = torch.full((n,), 2, dtype=torch.uint8, device=dev)
a = torch.randn(n, 256, device=dev); b = torch.empty_like(a)
L.hotk_mask_rows(p(a), 256, p(kind), 6, c_float(0.), p(b), 256, n, 256, _lib.stream_ptr())
a1 = torch.randn(n, 257, device=dev); b1 = torch.empty_like(a1)
L.hotk_mask_rows(p(a1), 257, p(kind), 6, c_float(0.), p(b1), 257, n, 257, _lib.stream_ptr())
torch.cuda.synchronize()
del a, b, a1, b1
G = bench.make_inputs(int(round(entries / 654)), dev, 1234)
plan = G.plan
h1 = torch.randn(plan.Nn, 384, device=dev); h2 = torch.randn(plan.R, 288, device=dev); v = torch.randn(plan.R, 320, device=dev)
proj = torch.randn(16, 8, device=dev); att = torch.empty(plan.R, 320, device=dev); kvg = torch.empty(plan.B, 4, 25, 64, device=dev)
datt = torch.randn(plan.R, 320, device=dev); dout = torch.randn(plan.R, 32, device=dev)
dh1 = torch.empty(plan.Nn, 384, device=dev); dh2 = torch.empty(plan.R, 288, device=dev)
dv = torch.empty(plan.R, 320, device=dev); ck = torch.empty(plan.R, 192, device=dev); dkvg = torch.empty_like(kvg)
A = ops.AttnCore22._args(plan, h1, h2, v, proj, 32)
phik = ops.AttnCore22._phik(plan, h1, proj); A.phik, A.ld_phik = phik.data_ptr(), 512
A.att, A.ld_att, A.kvg = att.data_ptr(), 320, kvg.data_ptr()
A.datt, A.ld_datt, A.dout, A.ld_dout = datt.data_ptr(), 320, dout.data_ptr(), 32
A.dh1, A.ld_dh1, A.dh2, A.ld_dh2, A.dv, A.ld_dv = dh1.data_ptr(), 384, dh2.data_ptr(), 288, dv.data_ptr(), 320
A.ck, A.dkvg = ck.data_ptr(), dkvg.data_ptr()
ws, nws = ops.AttnCore22._workspace(plan, dev)
assert L.hotk_attn22_forward(byref(A), p(ws), nws, _lib.stream_ptr()) == 0
assert L.hotk_attn22_backward(byref(A), p(ws), nws, _lib.stream_ptr()) == 0
torch.cuda.synchronize()
print('entries', plan.Ev, 'nodes', plan.Nn, 'calib_bytes_f4', n * 256 * 4, 'calib_bytes_f1', n * 257 * 4)
