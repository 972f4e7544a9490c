"""fuzz hotk_linear22_bwd_reduce / _bf16 on random ragged batches against float64 (python scratch/fuzz_l22_reduce.py [n])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_gpu_l22_reduce as T
from hot_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(123)
L = _lib.lib(); p = _lib.p
worst = 0.
for it in range(N):
    nmax = int(rng.integers(1, 65))
    B = int(rng.integers(1, 400)) if it % 5 == 0 else int(rng.integers(1, 40))
    graphs = []
    for _ in range(B):
        n = int(rng.integers(1, nmax + 1))
        e = int(rng.integers(0, min(n * (n - 1), 6 * n) + 1))
        pairs = set()
        while len(pairs) < e:
            i, j = rng.integers(0, n, 2)
            if i != j: pairs.add((int(i), int(j)))
        ei = np.array(sorted(pairs), dtype=np.int64).T.reshape(2, -1)
        ei = ei[:, rng.permutation(ei.shape[1])]
        graphs.append((n, ei))
    G = T._batch(graphs, it)
    plan = G.plan
    if L.hotk_linear22_bwd_reduce_supported(64, T.C, plan.Nmax) != 1 or plan.R == 0:
        continue
    g, x2 = T._inputs(plan, 1000 + it)
    for bf in (False, True):
        gg, xx = (g.to(torch.bfloat16), x2.to(torch.bfloat16)) if bf else (g, x2)
        n = ctypes.c_int64(); L.hotk_linear22_bwd_reduce_partials(plan.B, ctypes.byref(n))
        dyn = torch.full((plan.Nn, 3 * T.C), float('nan'), device='cuda'); part = torch.full((n.value, 64, T.C), float('nan'), device='cuda')
        fn = L.hotk_linear22_bwd_reduce_bf16 if bf else L.hotk_linear22_bwd_reduce
        rc = fn(p(gg), T.C, p(xx), 64, 64, T.C, None, p(plan.rown), p(plan.coln), p(plan.g_ebeg), p(plan.g_eend), p(plan.n_nodes_dev),
                plan.B, plan.Nmax, plan.R, p(dyn), 3 * T.C, p(part), _lib.stream_ptr())
        assert rc == 0, rc
        torch.cuda.synchronize()
        dW = part.sum(0)
        g0 = torch.where(plan.kind.view(-1, 1) == 0, torch.zeros_like(gg), gg).float()
        d64, w64 = T._reference64(plan, g0, xx.float())
        assert not torch.isnan(dyn).any() and not torch.isnan(dW).any(), (it, bf)
        e1 = float((dyn.double() - d64).abs().max()) / max(float(d64.abs().max()), 1e-30)
        e2 = float((dW.double() - w64).abs().max()) / max(float(w64.abs().max()), 1.)
        worst = max(worst, e1, e2)
        assert e1 <= 3e-6 and e2 <= 3e-6, (it, bf, e1, e2, plan.B, plan.Nmax, plan.R)
    if it % 10 == 0: print('case', it, 'B', plan.B, 'Nmax', plan.Nmax, 'R', plan.R, 'worst', worst, flush=True)
print('fuzz ok, worst relative error', worst)
