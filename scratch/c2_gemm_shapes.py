"""scratch: per-shape census of ops.mm / ops.wgrad in one C2-shaped step (4 096 molecules)"""
import sys, os, runpy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops
log = []
def wrap(name, f):
    def g(A, W, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(A, W, *a, **k); e1.record()
        rows = k.get('rows', a[-1] if (a and torch.is_tensor(a[-1]) and a[-1].dtype == torch.int32) else None)
        log.append((name, tuple(A.shape), tuple(W.shape), k.get('trans_w', False), None if rows is None else rows.numel(), e0, e1))
        return out
    return g
ops.mm = wrap('mm', ops.mm); ops.wgrad = wrap('wgrad', ops.wgrad); ops.thin_gemm = wrap('thin', ops.thin_gemm)
os.environ['HOTK_C2_GRAPH'] = '0'
sys.argv = ['bench_c2.py', '4096']
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_c2.py'), run_name='__main__')
torch.cuda.synchronize()
steps = 13
agg = collections.OrderedDict()
for name, a, w, tw, nr, e0, e1 in log:
    k = (name, a, w, tw, nr)
    ms, n = agg.get(k, (0., 0))
    agg[k] = (ms + e0.elapsed_time(e1), n + 1)
tot = 0.
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f'{k[0]:6s} A{k[1]} W{k[2]} trans_w={k[3]} rows={k[4]}: {n // steps} calls/step, {ms / n * 1e3:.0f} us each, {ms / steps:.2f} ms/step')
    tot += ms / steps
print(f'total {tot:.1f} ms/step')
