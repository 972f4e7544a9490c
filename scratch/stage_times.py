"""scratch: per-stage cycles of k_seg_bwd per (part, node) (probe build -DHOTK_X_STAGE_TIMING,
HOTK_LIB=scratch/ub/libhotk_stage.so): mean cycles a wave spends in the accumulation walk, the query side, the layout
conversion and the key / value walk"""
import sys, os, ctypes, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hot_amd import _lib
L = _lib.lib()
L.hotk_debug_stage_buffer.argtypes = [ctypes.c_void_p]
entries = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
Nn = int(round(entries / 654)) * 64
buf = torch.zeros(3 * Nn * 4, dtype=torch.int32, device='cuda')
assert L.hotk_debug_stage_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
sys.argv = ['bench_fused.py', str(entries), '1']
g = runpy.run_path(os.path.join(os.path.dirname(__file__), 'bench_fused.py'), run_name='__main__')
assert g['plan'].Nn == Nn, (g['plan'].Nn, Nn)
torch.cuda.synchronize()
L.hotk_debug_stage_buffer(ctypes.c_void_p(0))
v = buf.view(3, Nn, 4).double()
for P in range(3):
    m = v[P].mean(0).tolist()
    print('part %d: accumulate %.0f  query %.0f  convert %.0f  keyvalue %.0f  cycles per node (sum %.0f)' % (P, *m, sum(m)))
