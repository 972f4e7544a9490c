"""debug: per-phase cycle totals of k_seg_fwd<2> (library built with -DHOTK_STAMPS, HOTK_LIB pointing at it)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], os.environ.get('STAMP_ENTRIES', '2e6'), '1']
import torch
exec(open(os.path.join(os.path.dirname(__file__), 'bench_fused.py')).read().split("f = run(")[0])
buf = (ctypes.c_ulonglong * 16)()
L.hotk_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.hotk_debug_stamps(buf, 1)
rc = L.hotk_attn22_forward(byref(a), p(ws), nws, _lib.stream_ptr()); assert rc == 0
torch.cuda.synchronize()
L.hotk_debug_stamps(buf, 1)
v = list(buf)
names = {9: 'drain stores', 10: 'col: wait rows', 11: 'col: arithmetic', 1: 'startup+nkind', 2: 'col_ptr', 3: 'col records', 4: 'column walk', 5: 'F->C', 6: 'row_ptr', 7: 'row records', 8: 'row walk'}
waves = v[15]
tot = sum(v[1:12])
print('waves(all parts)', waves, 'nodes', plan.Nn)
for i in range(1, 12):
    print(f'{names[i]:16s} {v[i] / plan.Nn:10.0f} ticks/node  {100 * v[i] / tot:5.1f}%')
print('total ticks/node', tot / plan.Nn, '(s_memtime ticks: 100 MHz constant clock or shader clock, see ratio to wave cycles)')
