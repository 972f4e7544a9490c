"""scratch: who calls mask_rows / bias_mask in one C2-shaped step (4 096 molecules)"""
import sys, os, runpy, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hot_amd import ops, _lib
sites = collections.Counter()
orig_call = _lib.call
def call(name, nbytes, *a):
    if name in ('hotk_mask_rows', 'hotk_bias_mask_rows', 'hotk_gather_rows'):
        fr = [f for f in traceback.extract_stack()[:-1] if 'hot_amd' in f.filename][-3:]
        sites[(name, nbytes, ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(fr)))] += 1
    return orig_call(name, nbytes, *a)
ops.call = call
os.environ['HOTK_C2_GRAPH'] = '0'
sys.argv = ['bench_c2.py', '4096']
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_c2.py'), run_name='__main__')
for (name, nb, where), n in sorted(sites.items(), key=lambda kv: -kv[1] * kv[0][1]):
    print(f'{name:22s} {n // 13:3d}/step  {nb / 1e6:8.1f} MB  {where}')
