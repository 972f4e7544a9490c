"""GPU (-m gpu), collected last: hotk_attn22_forward_static -- the forward core for callers that cannot allocate
(include/hotk.h) -- against hotk_attn22_forward on the same inputs, through the C ABI.  The two entry points share
their kernels, so the results must agree bit for bit (the bar for this path: att and the saved per-graph sums kvg).
Two calls in a row on one stream, the second with a smaller batch (the library's scratch array keeps the first call's
partial sums); then the size limit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _core_inputs(n_graphs, n_lo, n_hi, seed):
    """a batch of random symmetric graphs and random raw projections h1 / h2 / v of the fused shape"""
    from hot_amd.batch import sparse as bs
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    nfs, eis, efs = [], [], []
    for _ in range(n_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        pairs = set()
        while len(pairs) < min(2 * n, n * (n - 1) // 2):
            i, j = (int(x) for x in rng.integers(0, n, 2))
            if i != j:
                pairs.add((min(i, j), max(i, j)))
        und = sorted(pairs)
        if rng.random() < 0.5:     # also one-directional edges: entries without a transposed partner
            src = [a for a, b in und] + [b for a, b in und[::2]]
            dst = [b for a, b in und] + [a for a, b in und[::2]]
        else:
            src = [a for a, b in und] + [b for a, b in und]
            dst = [b for a, b in und] + [a for a, b in und]
        order = np.lexsort((dst, src))
        ei = torch.tensor([[src[k] for k in order], [dst[k] for k in order]], dtype=torch.int64)
        nfs.append(torch.randn(n, 16, generator=g)); eis.append(ei); efs.append(torch.randn(ei.size(1), 16, generator=g))
    G = bs.make_batch([x.to(DEV) for x in nfs], [x.to(DEV) for x in eis], [x.to(DEV) for x in efs])
    plan = G.plan
    h1 = torch.randn(plan.Nn, 384, generator=g).to(DEV)
    h2 = torch.randn(plan.R, 256 + 32, generator=g).to(DEV)
    v = torch.randn(plan.R, 320, generator=g).to(DEV)
    proj = torch.randn(16, 8, generator=g).to(DEV)
    return plan, h1, h2, v, proj


def _run(plan, h1, h2, v, proj, static):
    from hot_amd import _lib, ops
    L = _lib.lib()
    att = torch.zeros(plan.R, 320, device=DEV)
    kvg = torch.zeros(plan.B, 4, 25, 64, device=DEV)
    a = ops.AttnCore22._args(plan, h1, h2, v, proj, 32)
    phik = ops.AttnCore22._phik(plan, h1, proj)
    a.phik, a.ld_phik = phik.data_ptr(), 512
    a.att, a.ld_att, a.kvg = att.data_ptr(), 320, kvg.data_ptr()
    if static:
        rc = L.hotk_attn22_forward_static(ctypes.byref(a), _lib.stream_ptr())
    else:
        ws, nws = ops.AttnCore22._workspace(plan, h1.device)
        rc = L.hotk_attn22_forward(ctypes.byref(a), _lib.p(ws), nws, _lib.stream_ptr())
    assert rc == 0, L.hotk_strerror(rc)
    torch.cuda.synchronize()
    return att, kvg


def test_forward_static_matches_workspace_form_twice():
    from hot_amd import _lib
    assert _lib.lib().hotk_attn22_static_init() == 0      # set-up call: resolves the array's address on this device
    big = _core_inputs(96, 10, 40, seed=11)      # 96 graphs, up to ~200 rows each
    small = _core_inputs(7, 3, 9, seed=12)       # second call: fewer, smaller graphs
    for inputs in (big, small):
        att_ref, kvg_ref = _run(*inputs, static=False)
        att, kvg = _run(*inputs, static=True)
        assert att_ref.abs().max() > 0
        assert torch.equal(att, att_ref)
        assert torch.equal(kvg, kvg_ref)
    # and once more the large one after the small one: nothing of the small call's partial sums survives
    att_ref, kvg_ref = _run(*big, static=False)
    att, kvg = _run(*big, static=True)
    assert torch.equal(att, att_ref) and torch.equal(kvg, kvg_ref)
    # on a side stream: the entry point takes the device (and with it the array's address) from the stream it is handed
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        att, kvg = _run(*small, static=True)
    att_ref, kvg_ref = _run(*small, static=False)
    assert torch.equal(att, att_ref) and torch.equal(kvg, kvg_ref)


def test_forward_static_rejects_oversize_without_launching():
    from hot_amd import _lib, ops
    L = _lib.lib()
    plan, h1, h2, v, proj = _core_inputs(4, 3, 6, seed=13)
    a = ops.AttnCore22._args(plan, h1, h2, v, proj, 32)
    att = torch.full((plan.R, 320), 7.0, device=DEV)
    kvg = torch.zeros(plan.B, 4, 25, 64, device=DEV)
    phik = ops.AttnCore22._phik(plan, h1, proj)
    a.phik, a.ld_phik = phik.data_ptr(), 512
    a.att, a.ld_att, a.kvg = att.data_ptr(), 320, kvg.data_ptr()
    # claim a batch whose partial sums need more than the library's scratch array: B * chunks * 9 216 B > 32 MiB
    a.B, a.Emax = 1 << 14, 64
    a.R, a.Nn = a.B * a.Emax, a.B * a.Nmax
    need = ctypes.c_size_t(0)
    assert L.hotk_attn22_workspace_bytes(a.B, a.Emax, ctypes.byref(need)) == 0 and need.value > (32 << 20)
    rc = L.hotk_attn22_forward_static(ctypes.byref(a), _lib.stream_ptr())
    assert rc == -5 and b'scratch capacity' in L.hotk_strerror(rc)
    torch.cuda.synchronize()
    assert bool((att == 7.0).all())
