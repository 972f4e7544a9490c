"""hotk_linear22_bwd_reduce (csrc/l22bwd.hip): the entry-level reductions of the backward pass of the factored
Linear(2, 2, 'light') -- diagonal gather, row / column sums over off-diagonal entries, weight gradient of the entry blocks
(models/sparse/linear.py:143-161, 293-306 under autograd) -- in one pass over dy, through the C ABI against float64 and
against the three-kernel form it replaces."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
C = 288


def _product():
    from hot_amd import _lib, ops
    from hot_amd.batch import sparse as bs
    return _lib, ops, bs


def _batch(graphs, seed):
    """graphs: list of (n nodes, edge index [2, e]) -> sparse Batch with 4-wide features (only the plan is used)"""
    _, _, bs = _product()
    g = torch.Generator().manual_seed(seed)
    nfs, eis, efs = [], [], []
    for n, ei in graphs:
        nfs.append(torch.randn(n, 4, generator=g).to(DEV))
        eis.append(torch.from_numpy(np.ascontiguousarray(ei)).to(DEV))
        efs.append(torch.randn(ei.shape[1], 4, generator=g).to(DEV))
    return bs.make_batch(nfs, eis, efs)


def _ba_graphs(n_graphs, seed, n=64, m=5):
    from hot_amd.utils.graphs import ba_edges
    rng = np.random.default_rng(seed)
    return [(n, ba_edges(n, m, rng)) for _ in range(n_graphs)]


def _ragged_graphs(nmax, seed):
    """directed graphs of 1 .. nmax nodes with uneven entry counts (padding rows and padding nodes), one without edges,
    one with a hub whose row and column hold every other node"""
    rng = np.random.default_rng(seed)
    graphs = []
    for n in (nmax, 2, 1, nmax // 2, 5, nmax - 1, 3):
        pairs = set()
        want = min(n * (n - 1), int(rng.integers(0, 4 * n + 1)))
        while len(pairs) < want:
            i, j = rng.integers(0, n, 2)
            if i != j:
                pairs.add((int(i), int(j)))
        ei = np.array(sorted(pairs), dtype=np.int64).T.reshape(2, -1)
        ei = ei[:, rng.permutation(ei.shape[1])]          # entries in no particular order
        graphs.append((n, ei))
    graphs.append((4, np.zeros((2, 0), dtype=np.int64)))
    n = nmax
    star = np.stack([np.zeros(n - 1, dtype=np.int64), np.arange(1, n)])
    graphs.append((n, np.concatenate([star, star[::-1]], axis=1)))
    return graphs


def _reduce(plan, g, x2):
    _lib, _, _ = _product()
    L = _lib.lib()
    n = ctypes.c_int64()
    assert L.hotk_linear22_bwd_reduce_partials(plan.B, ctypes.byref(n)) == 0
    # NaN-filled outputs: every element the caller reads must have been written
    dyn = torch.full((plan.Nn, 3 * C), float('nan'), device=DEV)
    part = torch.full((n.value, 64, C), float('nan'), device=DEV)
    p = _lib.p
    rc = L.hotk_linear22_bwd_reduce(p(g), C, p(x2), 64, 64, C, p(plan.kind), p(plan.rown), p(plan.coln), p(plan.g_ebeg),
                                    p(plan.g_eend), p(plan.n_nodes_dev), plan.B, plan.Nmax, plan.R, p(dyn), 3 * C, p(part),
                                    _lib.stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return dyn, part.sum(0)


def _reference64(plan, g, x2):
    kind = plan.kind.long()
    g64, x64 = g.double(), x2.double()
    dyn = torch.zeros(plan.Nn, 3 * C, dtype=torch.float64, device=DEV)
    dg = kind == 1
    dyn[plan.rown[dg].long(), 0:C] = g64[dg]
    od = kind == 2
    dyn[:, C:2 * C].index_add_(0, plan.rown[od].long(), g64[od])
    dyn[:, 2 * C:].index_add_(0, plan.coln[od].long(), g64[od])
    valid = kind != 0
    dW = x64[valid].t() @ g64[valid]
    return dyn, dW


def _inputs(plan, seed):
    gen = torch.Generator().manual_seed(seed)
    g = torch.randn(plan.R, C, generator=gen).to(DEV)
    x2 = torch.randn(plan.R, 64, generator=gen).to(DEV)
    x2[plan.kind != 2] = 0.            # (x2 = [x_ij | x_ji] is zero on diagonal and padding rows)
    # padding rows of an upstream gradient may hold anything: the kernel must not read them
    g[plan.kind == 0] = float('nan')
    return g, x2


@pytest.mark.parametrize('case', ['ragged40', 'ragged64', 'ragged17', 'ba300', 'one_graph'])
def test_bwd_reduce_vs_fp64(case):
    """every dyn element written (NaN-prefilled), sums and weight gradient against float64, two runs bit-identical; more
    graphs than workgroups (ba300: 300 graphs over 256 persistent workgroups), one / two one-hot node tiles (Nmax 17 / 40,
    64), padding rows holding NaN"""
    if case.startswith('ragged'):
        G = _batch(_ragged_graphs(int(case[6:]), 3), 4)
    elif case == 'ba300':
        G = _batch(_ba_graphs(300, 5), 6)
    else:
        G = _batch(_ba_graphs(1, 7), 8)
    plan = G.plan
    _lib, _, _ = _product()
    assert _lib.lib().hotk_linear22_bwd_reduce_supported(64, C, plan.Nmax) == 1
    g, x2 = _inputs(plan, 9)
    dyn, dW = _reduce(plan, g, x2)
    assert not torch.isnan(dyn).any() and not torch.isnan(dW).any()
    g0 = torch.where(plan.kind.view(-1, 1) == 0, torch.zeros_like(g), g)
    dyn64, dW64 = _reference64(plan, g0, x2)
    scale = float(dyn64.abs().max())
    assert float((dyn.double() - dyn64).abs().max()) <= 2e-6 * scale
    # the diagonal block is a copy, padding node rows are zero
    assert torch.equal(dyn[:, 0:C].double(), dyn64[:, 0:C])
    assert float(dyn[plan.nkind == 0].abs().max() if (plan.nkind == 0).any() else 0.) == 0.
    assert float((dW.double() - dW64).abs().max()) <= 2e-6 * max(float(dW64.abs().max()), 1.)
    dyn_b, dW_b = _reduce(plan, g, x2)
    assert torch.equal(dyn, dyn_b) and torch.equal(dW, dW_b)


@pytest.mark.parametrize('case', ['ragged40', 'ragged17', 'ba300'])
def test_bwd_reduce_bf16_inputs_vs_fp64(case):
    """hotk_linear22_bwd_reduce_bf16 (the bf16 storage mode's form): dy and x2 as bf16 tensors, every product exact, fp32 sums
    -- held to float64 sums of the stored bf16 values; NaN in the padding rows; two runs bit-identical"""
    G = _batch(_ragged_graphs(int(case[6:]), 3), 4) if case.startswith('ragged') else _batch(_ba_graphs(300, 5), 6)
    plan = G.plan
    _lib, _, _ = _product()
    L = _lib.lib()
    g32, x32 = _inputs(plan, 21)
    g, x2 = g32.to(torch.bfloat16), x32.to(torch.bfloat16)

    def run():
        n = ctypes.c_int64()
        assert L.hotk_linear22_bwd_reduce_partials(plan.B, ctypes.byref(n)) == 0
        dyn = torch.full((plan.Nn, 3 * C), float('nan'), device=DEV)
        part = torch.full((n.value, 64, C), float('nan'), device=DEV)
        p = _lib.p
        rc = L.hotk_linear22_bwd_reduce_bf16(p(g), C, p(x2), 64, 64, C, None, p(plan.rown), p(plan.coln), p(plan.g_ebeg),
                                             p(plan.g_eend), p(plan.n_nodes_dev), plan.B, plan.Nmax, plan.R, p(dyn), 3 * C,
                                             p(part), _lib.stream_ptr())
        assert rc == 0, rc
        torch.cuda.synchronize()
        return dyn, part.sum(0)
    dyn, dW = run()
    assert not torch.isnan(dyn).any() and not torch.isnan(dW).any()
    g0 = torch.where(plan.kind.view(-1, 1) == 0, torch.zeros_like(g), g)
    dyn64, dW64 = _reference64(plan, g0.float(), x2.float())
    assert float((dyn.double() - dyn64).abs().max()) <= 2e-6 * float(dyn64.abs().max())
    assert torch.equal(dyn[:, 0:C].double(), dyn64[:, 0:C])
    assert float((dW.double() - dW64).abs().max()) <= 2e-6 * max(float(dW64.abs().max()), 1.)
    dyn_b, dW_b = run()
    assert torch.equal(dyn, dyn_b) and torch.equal(dW, dW_b)
    # misaligned leading dimension (bf16 rows move in 16-byte pieces)
    assert L.hotk_linear22_bwd_reduce_bf16(_lib.p(g), C + 4, _lib.p(x2), 64, 64, C, None, _lib.p(plan.rown), _lib.p(plan.coln),
                                           _lib.p(plan.g_ebeg), _lib.p(plan.g_eend), _lib.p(plan.n_nodes_dev), plan.B, plan.Nmax,
                                           plan.R, _lib.p(dyn), 3 * C, _lib.p(dW), _lib.stream_ptr()) != 0


def test_bwd_reduce_unsupported_shapes_are_refused():
    _lib, _, _ = _product()
    L = _lib.lib()
    assert L.hotk_linear22_bwd_reduce_supported(64, 256, 64) == 0
    assert L.hotk_linear22_bwd_reduce_supported(128, C, 64) == 0
    assert L.hotk_linear22_bwd_reduce_supported(64, C, 65) == 0
    assert L.hotk_linear22_bwd_reduce_supported(64, C, 64) == 1


def test_bwd_reduce_without_entry_rows_leaves_zeros():
    """R = 0 (a batch of graphs without a single entry row): dyn and the one partial block leave as zeros, nothing is read"""
    _lib, _, _ = _product()
    L = _lib.lib()
    B, Nmax = 2, 3
    n = ctypes.c_int64()
    assert L.hotk_linear22_bwd_reduce_partials(B, ctypes.byref(n)) == 0 and n.value == 2
    dyn = torch.full((B * Nmax, 3 * C + 8), float('nan'), device=DEV)      # (padded leading dimension: the pad stays as it is)
    part = torch.full((n.value, 64, C), float('nan'), device=DEV)
    rc = L.hotk_linear22_bwd_reduce(None, C, None, 64, 64, C, None, None, None, None, None, None, B, Nmax, 0,
                                    _lib.p(dyn), 3 * C + 8, _lib.p(part), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert float(dyn[:, :3 * C].abs().max()) == 0. and torch.isnan(dyn[:, 3 * C:]).all()
    assert float(part.abs().max()) == 0.
    # invalid: no partial buffer
    assert L.hotk_linear22_bwd_reduce(None, C, None, 64, 64, C, None, None, None, None, None, None, B, Nmax, 0,
                                      _lib.p(dyn), 3 * C + 8, None, _lib.stream_ptr()) != 0


def test_light_basis_backward_uses_it_and_matches_the_three_kernel_form():
    """ops._l22_backward at the headline widths (D = 32 -> C = 288) on 60 BA(64,5) graphs: the one-pass reductions run
    (census), and every gradient equals the gather + pair-sum + library weight-gradient form to fp32 rounding"""
    _lib, ops, bs = _product()
    G = _batch(_ba_graphs(60, 11), 12)
    plan = G.plan
    assert plan.R >= ops.THIN_MIN_ROWS
    gen = torch.Generator().manual_seed(13)
    x0 = torch.randn(plan.B, plan.Emax, 32, generator=gen).to(DEV)
    W = (torch.randn(5 * 32, C, generator=gen) / 8).to(DEV).requires_grad_(True)
    b_off = torch.randn(C, generator=gen).to(DEV).requires_grad_(True)
    b_diag = torch.randn(C, generator=gen).to(DEV).requires_grad_(True)
    R = torch.randn(plan.B, plan.Emax, C, generator=gen).to(DEV)

    def run(on):
        ops._L22_REDUCE = on
        try:
            x = x0.clone().requires_grad_(True)
            _lib.timing_begin(None)
            (ops.Linear22Light.apply(x, W, b_off, b_diag, plan) * R).sum().backward()
            ran = _lib.timing_end()
            out = [x.grad.clone(), W.grad.clone(), b_off.grad.clone(), b_diag.grad.clone()]
            W.grad = b_off.grad = b_diag.grad = None
            return ran, out
        finally:
            ops._L22_REDUCE = True
    ran1, g1 = run(True)
    ran0, g0 = run(False)
    assert ran1['hotk_linear22_bwd_reduce'][1] == 1 and 'hotk_seg_sum_rows_pair_own' not in ran1
    assert 'hotk_linear22_bwd_reduce' not in ran0 and ran0['hotk_seg_sum_rows_pair_own'][1] == 1
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
