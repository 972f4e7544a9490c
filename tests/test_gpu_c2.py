"""GPU (-m gpu): configuration C2 (PCQM4M-LSC enc-small, regression/examples/pcqm4m-lsc/enc-small.sh) at its REAL layer
dimensions -- dim_hidden 256, dim_qk = dim_v = 32 with 4 heads, dim_ff 256, readout 128 x 8 heads, 128 molecule-sized
graphs per step -- against the CPU oracle (pinned by tests/test_oracle_golden.py), plus the training-mode pieces the
shipped config switches on: dropout 0.1 between the attention core and fc_o, and the per-forward projection redraw."""
import numpy as np
import pytest
import torch

from conftest import rel_err, oracle_fp32_fp64, hold_to_fp64, err64  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-5


def _molecules(n_graphs, seed, dim):
    """PCQM-like batch: n ~ U{8..20} atoms, ~1.05 n bonds, symmetrised; features [dim/2 | dim/2]"""
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    nfs, eis, efs = [], [], []
    for _ in range(n_graphs):
        n = int(rng.integers(8, 21))
        bonds = {(i, i + 1) for i in range(n - 1)}                 # a chain keeps the molecule connected
        while len(bonds) < int(1.05 * n):
            i, j = sorted(int(x) for x in rng.integers(0, n, 2))
            if i != j:
                bonds.add((i, j))
        und = np.array(sorted(bonds), dtype=np.int64)
        ei = np.concatenate([und, und[:, ::-1]], 0)
        ei = ei[np.lexsort((ei[:, 1], ei[:, 0]))].T.copy()
        nfs.append(torch.randn(n, dim // 2, generator=g)); eis.append(torch.from_numpy(ei))
        efs.append(torch.randn(ei.shape[1], dim // 2, generator=g))
    return nfs, eis, efs, g


def _to_dev(nfs, eis, efs):
    from hot_amd.batch import sparse as bs
    return bs.make_batch([x.to(DEV) for x in nfs], [x.to(DEV) for x in eis], [x.to(DEV) for x in efs])


def _find_node(fn, name, seen=None):
    """the autograd node called `name` upstream of fn"""
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return None
    seen.add(fn)
    if type(fn).__name__ == name:
        return fn
    for nxt, _ in fn.next_functions:
        r = _find_node(nxt, name, seen)
        if r is not None:
            return r
    return None


def test_c2_encoder_real_dims_vs_oracle():
    """Encoder(2, 0, [2]*3, 512, 256, 256, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel') -- enc-small with 3
    of its 8 identical layers -- forward + backward on 128 molecules against the oracle.  The fused attention core runs
    with D = 256 (ld_h2 = 512); the readout is the segmented softmax kernel with 8 heads of width 16.
    Criteria, all against the oracle run in FLOAT64 with the float32 oracle's own distance beside it: outputs <= max(1e-5,
    2 x that distance).  Gradients: this case (three stacked ReLU-feature layers at D = 256, ~1e7 pre-activations) does hit
    ReLU kinks -- pre-activations within fp32 rounding of 0 whose gate opens on one side only; each moves one gradient row
    by O(1e-4) -- and the float32 oracle hits them as well (measured: it is up to 1.9e-4 from float64, 0.11 % of the
    input-gradient rows beyond 2e-5; the HIP path 3.7e-4 / 0.11 %).  Which gates flip is a property of each side's
    rounding, so the bar is the model-wide kink level of the reference's own precision: every gradient tensor <= max(2e-5,
    4 x G), G = the float32 oracle's largest distance from float64 over all gradient tensors (round 2: flat 1e-3), and the
    fraction of input-gradient rows beyond 2e-5 <= twice the float32 oracle's + 0.1 % (round 2: flat 0.5 %)."""
    from oracle import hot_oracle as O
    from hot_amd import _lib
    from hot_amd.batch import sparse as bs
    from hot_amd.models.encoder import Encoder
    L, DH = 3, 256
    nfs, eis, efs, g = _molecules(128, 21, 2 * DH)
    GO = O.build_batch_from_lists(nfs, eis, efs)
    G0 = _to_dev(nfs, eis, efs)
    assert np.array_equal(G0.t_indices.cpu().numpy(), GO.t_indices.numpy())
    torch.manual_seed(0)
    enc = Encoder(2, 0, [2] * L, 2 * DH, DH, DH, 32, 32, 256, 4, 128, 128, 8, 'default', 'generalized_kernel', 0., 0., 0.,
                  sparse=True).to(DEV)
    enc.skip_redraw_projections = True
    enc.eval()
    for lyr in list(enc.layers)[:L]:
        assert lyr.attn._fused_ok()
    x = G0.values.clone().requires_grad_(True)
    _lib.timing_begin(None)
    y = enc(bs.batch_like(G0, x, skip_masking=True))
    R = torch.randn(y.shape, generator=g)
    (y * R.to(DEV)).sum().backward()
    census = _lib.timing_end()
    assert census['hotk_attn22_forward'][1] == L and census['hotk_attn22_backward'][1] == L
    assert census['hotk_seg_softmax_pool_fwd'][1] == 2

    def run(xo, sd, dt):
        yo = O.encoder_2_0(GO.like(xo), sd, L, DH, 32, 32, 256, 4, 128, 128, 8, True)
        (yo * R.to(dt)).sum().backward()
        return yo
    o32, o64 = oracle_fp32_fp64(run, enc.state_dict(), GO.values)
    e_y, gap_y = hold_to_fp64('out', y, o32[0], o64[0], TOL)
    # gradient scale of a parameter: its own largest gradient, or 1e-3 of the largest in the model (a softmax is invariant
    # to a shift of its logits, so the readout's key biases have an exactly zero gradient that every side computes as noise)
    scale = max(float(v.abs().max()) for v in o64[2].values())
    params = dict(enc.named_parameters())
    gaps = {k: err64(o32[2][k], o64[2][k], 1e-3 * scale) for k in params}
    gaps['d input'] = err64(o32[1], o64[1])
    G = max(gaps.values())
    errs = {k: err64(p_.grad, o64[2][k], 1e-3 * scale) for k, p_ in params.items()}
    errs['d input'] = err64(x.grad, o64[1])
    s_in = float(o64[1].abs().max())
    rows = lambda gx: float((((gx.detach().cpu().double() - o64[1]).abs().amax(-1) / s_in) > 2 * TOL).double().mean())
    bad, bad32 = rows(x.grad), rows(o32[1])
    worst_k = max(errs, key=errs.get)
    print(f'C2 real dims: out err {e_y:.2e} (oracle fp32 {gap_y:.2e}); gradients: worst {errs[worst_k]:.2e} ({worst_k}), '
          f'oracle fp32 worst {G:.2e}; input-gradient rows off {bad:.2e} (oracle fp32 {bad32:.2e})')
    for k, e in errs.items():
        assert e <= max(2 * TOL, 4 * G), f'{k}: {e:.2e} against the fp64 oracle; the fp32 oracle is within {G:.2e}'
    assert bad <= 2 * bad32 + 1e-3, (bad, bad32)


def test_training_dropout_stays_on_fused_core():
    """enc-small trains with --dropout_rate 0.1: the dropout sits between cat(att) and fc_o (kernelattn.py:70-72), so the
    fused core must stay on the path.  Checks: hotk_attn22_forward / _backward ran; what fc_o saw is the eval-mode att
    with ~10 % of its elements zeroed and the rest scaled by 1/0.9; MuDropout zeroes whole (graph, term) blocks; the
    gradients are those of exactly that masked computation (weight gradient from the masked att, value-bias gradient
    against a central finite difference with the masks replayed from the same seed)."""
    from hot_amd import _lib
    from hot_amd.batch import sparse as bs
    from hot_amd.models.common import KernelFeatureMap
    from hot_amd.models.sparse import KernelSelfAttn
    nfs, eis, efs, g = _molecules(64, 5, 32)
    G0 = _to_dev(nfs, eis, efs)
    torch.manual_seed(1)
    layer = KernelSelfAttn(2, 2, 32, 32, 32, 4, 'default', 0.1, 0.2, KernelFeatureMap(8, generalized_attention=True)).to(DEV)
    valid = G0.mask.view(-1)

    def run(train, seed=7):
        layer.train(train)
        layer.zero_grad()
        torch.manual_seed(seed)
        x = G0.values.clone().requires_grad_(True)
        y = layer(bs.batch_like(G0, x, skip_masking=True)).values
        node = _find_node(y.grad_fn, 'AttnCore22Backward')
        return x, y, node.saved_tensors[5]

    _, y_eval, att_eval = run(False)
    _lib.timing_begin(None)
    x, y, att = run(True)
    Rn = torch.randn(y.shape, generator=g).to(DEV)
    (y * Rn).sum().backward()
    census = _lib.timing_end()
    assert census['hotk_attn22_forward'][1] == 1 and census['hotk_attn22_backward'][1] == 1
    assert 'hotk_seg_outer_sum' not in census          # the general kernels did not run
    a0, a1 = att_eval[valid], att[valid]
    live = a0 != 0
    ratio = a1[live] / a0[live]
    kept = ratio != 0
    # mu-dropout (p = 0.2) and dropout (p = 0.1) multiply: kept elements carry 1 / (0.8 * 0.9)
    assert float((ratio[kept] - 1. / 0.72).abs().max()) < 1e-5
    frac0 = 1. - float(kept.float().mean())
    assert abs(frac0 - (1. - 0.8 * 0.9)) < 0.03, frac0
    blocks = att.view(G0.plan.B, G0.plan.Emax, 10, 32).abs().amax(dim=(1, 3))          # [B, 10]
    mu_dropped = float((blocks == 0).float().mean())
    assert abs(mu_dropped - 0.2) < 0.06, mu_dropped
    # gradients of the masked computation
    dy = Rn.view(-1, 32) * valid.unsqueeze(1)
    assert rel_err(layer.fc_o.weight.grad.cpu(), (dy.t() @ att).cpu()) < TOL
    gb = layer.fc_v.bias.grad.clone()
    direction = torch.randn(gb.shape, generator=g).to(DEV)
    eps = 1e-2
    vals = []
    for s in (+1, -1):
        with torch.no_grad():
            layer.fc_v.bias.add_(s * eps * direction)
        _, ys, _ = run(True)
        vals.append(float((ys.double() * Rn.double()).sum()))
        with torch.no_grad():
            layer.fc_v.bias.sub_(s * eps * direction)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((gb.double() * direction.double()).sum())
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-3), (fd, an)


def test_projection_redraw_on_device():
    """The reference redraws the FAVOR+ projection on EVERY training forward (common/kernel.py:149-158 with the default
    feature_redraw_interval = 0, encoder.py:92-93).  On the device: the matrix changes, its two 8-row blocks are
    orthogonal, calls_since_last_redraw stays 0, eval forwards do not redraw, and the fused core computes with the new
    matrix (train-mode output == eval-mode output right after)."""
    from hot_amd.batch import sparse as bs
    from hot_amd.models.encoder import Encoder
    nfs, eis, efs, _ = _molecules(16, 9, 32)
    G0 = _to_dev(nfs, eis, efs)
    torch.manual_seed(3)
    enc = Encoder(2, 2, [2], 32, 32, 32, 32, 32, 64, 4, 32, 32, 4, 'default', 'generalized_kernel', 0., 0., 0., sparse=True).to(DEV)
    assert enc.skip_redraw_projections is False
    fm = enc.feature_map
    assert fm is enc.layers[0].attn.feature_map.feature_map          # one shared map, as in the reference
    p0 = fm.projection_matrix.clone()
    enc.eval()
    y_old = enc(G0).values.clone()
    assert torch.equal(fm.projection_matrix, p0) and int(fm.calls_since_last_redraw) == 0
    enc.train()
    y_train = enc(G0).values.clone()
    p1 = fm.projection_matrix.clone()
    assert not torch.equal(p1, p0) and int(fm.calls_since_last_redraw) == 0
    for blk in (p1[:8], p1[8:]):
        gram = blk @ blk.t()
        off = gram - torch.diag(torch.diag(gram))
        assert float(off.abs().max()) < 1e-4 * float(torch.diag(gram).max())
    enc.eval()
    y_new = enc(G0).values
    assert torch.equal(fm.projection_matrix, p1)
    assert rel_err(y_train.cpu(), y_new.cpu()) < 1e-6
    assert rel_err(y_old.cpu(), y_new.cpu()) > 1e-3
    # an interval of 2: two counted calls, then a redraw
    fm.feature_redraw_interval = 2
    enc.train()
    seen = []
    for _ in range(3):
        enc(G0)
        seen.append((int(fm.calls_since_last_redraw), torch.equal(fm.projection_matrix, p1)))
    assert seen == [(1, True), (2, True), (0, False)]


def test_wide_layers_skip_the_gradient_masks_without_changing_gradients():
    """At C2's widths and >= 16 k padded rows the dense products run on the tiled GEMMs over the list of valid rows; the
    backward functions then do not zero the padding rows of the gradients they receive (nothing reads them: ops._l22_backward,
    ops.AttnCore22.backward).  One EncLayer(2, 2) at D = 256 on 320 molecules: fewer hotk_mask_rows launches, and every
    gradient equals the masked run's -- bit for bit except fc_o's bias, whose column sum changes its order of addition
    (by row kind instead of over all rows): 1e-6."""
    from hot_amd import ops, _lib
    from hot_amd.batch.sparse import batch_like
    from hot_amd.models.encoder import EncLayer
    from hot_amd.models.common import KernelFeatureMap
    DH = 256
    nfs, eis, efs, g = _molecules(320, 5, DH)
    G = _to_dev(nfs, eis, efs)
    plan = G.plan
    assert plan.R >= ops.G6_MIN_ROWS and plan.Ev < plan.R
    torch.manual_seed(0)
    layer = EncLayer(2, 2, DH, 32, 32, DH, 4, 'default', 'generalized_kernel', 0., 0., KernelFeatureMap(8, generalized_attention=True)).to(DEV)
    params = dict(layer.named_parameters())
    Wout = torch.randn(G.values.shape, generator=g).to(DEV)

    def run(skip):
        ops._L22_SKIP_MASK = skip
        try:
            x = G.values.detach().clone().requires_grad_(True)
            _lib.timing_begin(None)
            (layer(batch_like(G, x, skip_masking=True)).values * Wout).sum().backward()
            ran = _lib.timing_end()
            grads = {'x': x.grad.clone()}
            for n, p_ in params.items():
                grads[n] = p_.grad.clone()
                p_.grad = None
            return ran.get('hotk_mask_rows', (0., 0, 0))[1], grads
        finally:
            ops._L22_SKIP_MASK = True
    n_off, g_off = run(False)
    n_on, g_on = run(True)
    assert n_on <= n_off - 4, (n_on, n_off)            # fc_2 of the attention, both FFN linears, the core's output gradient
    for n in g_on:
        if n == 'attn.fc_o.bias':
            assert rel_err(g_on[n].cpu(), g_off[n].cpu()) < 1e-6
        else:
            assert torch.equal(g_on[n], g_off[n]), n
