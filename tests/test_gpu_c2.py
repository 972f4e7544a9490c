"""GPU (-m gpu): configuration C2 (PCQM4M-LSC enc-small, regression/examples/pcqm4m-lsc/enc-small.sh) at its REAL layer
dimensions -- dim_hidden 256, dim_qk = dim_v = 32 with 4 heads, dim_ff 256, readout 128 x 8 heads, 128 molecule-sized
graphs per step -- against the CPU oracle (pinned by tests/test_oracle_golden.py), plus the training-mode pieces the
shipped config switches on: dropout 0.1 between the attention core and fc_o, and the per-forward projection redraw."""
import numpy as np
import pytest
import torch

from conftest import rel_err

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
    Criteria: outputs <= 1e-5 of the output scale; input gradients row-wise <= 2e-5 of the gradient scale on all but <=
    0.5 % of the rows (ReLU kinks, see test_layer_vs_oracle_ba_graphs), parameter gradients <= 1e-3."""
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
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32 and 'projection' not in k)
          for k, v in enc.state_dict().items()}
    x = G0.values.clone().requires_grad_(True)
    _lib.timing_begin(None)
    y = enc(bs.batch_like(G0, x, skip_masking=True))
    R = torch.randn(y.shape, generator=g)
    (y * R.to(DEV)).sum().backward()
    census = _lib.timing_end()
    assert census['hotk_attn22_forward'][1] == L and census['hotk_attn22_backward'][1] == L
    assert census['hotk_seg_softmax_pool_fwd'][1] == 2
    xo = GO.values.clone().requires_grad_(True)
    yo = O.encoder_2_0(GO.like(xo), sd, L, DH, 32, 32, 256, 4, 128, 128, 8, True)
    (yo * R).sum().backward()
    e_y = rel_err(y.detach().cpu(), yo.detach())
    gx, gxo = x.grad.cpu(), xo.grad
    row_err = (gx - gxo).abs().amax(-1) / gxo.abs().max()
    bad = float((row_err > 2 * TOL).float().mean())
    # a softmax is invariant to a shift of its logits, so the readout's key biases have an exactly zero gradient that
    # both sides compute as rounding noise: errors are measured against the larger of the parameter's own gradient
    # scale and 1e-3 of the largest gradient in the model (as tests/test_gpu_parity.py::_check_param_grads)
    scale = max(float(sd[k].grad.abs().max()) for k, _ in enc.named_parameters())
    worst, worst_k = 0., None
    for k, p_ in enc.named_parameters():
        ref = sd[k].grad
        e = float((p_.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * scale)
        if e > worst:
            worst, worst_k = e, k
    print(f'C2 real dims: out err {e_y:.2e}, grad rows off {bad:.2e}, worst parameter grad err {worst:.2e} ({worst_k})')
    assert e_y < TOL, e_y
    assert bad <= 5e-3, bad
    assert worst < 1e-3, worst


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
