"""Generate the golden fixtures in this directory by IMPORTING THE REAL REFERENCE (jw9730/hot).

Run only in the build container, where the reference checkout is mounted read-only:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden.py

The reference never travels to the GPU box; only the ``.npz`` files written here (inputs, weights
and the reference's outputs / gradients) do.  Every file is plain numeric data loaded with
``numpy.load(allow_pickle=False)``.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
from hot_pytorch.batch.sparse import Batch, make_batch, t, d, nd, v2d, batch_like  # noqa: E402
from hot_pytorch.models.common.kernel import KernelFeatureMap  # noqa: E402
from hot_pytorch.models.sparse.kernelattncoef import KernelFeatureMapWrapper, KernelAttnCoef  # noqa: E402
from hot_pytorch.models.sparse.kernelattn import KernelSelfAttn  # noqa: E402
from hot_pytorch.models.sparse.linear import Linear  # noqa: E402
from hot_pytorch.models.encoder import EncLayer, Encoder  # noqa: E402
from hot_pytorch.models.sparse.attn import SelfAttn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def npy(x):
    return x.detach().cpu().numpy()


def rand_graphs(sizes, p, dn, de, seed, symmetric=False):
    g = torch.Generator().manual_seed(seed)
    nf, ei, ef = [], [], []
    for n in sizes:
        A = (torch.rand(n, n, generator=g) < p)
        if symmetric:
            A = A | A.t()
        A = A & ~torch.eye(n, dtype=torch.bool)
        idx = A.nonzero().t().contiguous()
        nf.append(torch.randn(n, dn, generator=g))
        ei.append(idx)
        ef.append(torch.randn(idx.size(1), de, generator=g))
    return nf, ei, ef


def ba_graphs(n, count, dn, de, seed):
    import networkx as nx
    g = torch.Generator().manual_seed(seed)
    nf, ei, ef = [], [], []
    for s in range(count):
        adj = nx.adjacency_matrix(nx.barabasi_albert_graph(n, 5, s)).tocoo()
        idx = torch.tensor(np.vstack((adj.row, adj.col)), dtype=torch.long)
        nf.append(torch.randn(n, dn, generator=g))
        ei.append(idx)
        ef.append(torch.randn(idx.size(1), de, generator=g))
    return nf, ei, ef


def pack_inputs(nf, ei, ef):
    return dict(node_feature=npy(torch.cat(nf)), edge_index=npy(torch.cat(ei, 1)), edge_feature=npy(torch.cat(ef)),
                n_nodes=np.array([x.size(0) for x in nf]), n_edges=np.array([e.size(0) for e in ef]))


def ref_tests_transpose_fixture():
    """The 4-graph fixture of the reference's own test_transpose (tests/test_sparse_util.py:37-57)."""
    A1_nodes = torch.tensor([1, 5, 9, 0]); A1_edges = torch.tensor([[0, 2, 0, 3], [4, 0, 6, 7], [0, 8, 0, 8], [7, 0, 6, 0]])
    A2_nodes = torch.tensor([1]); A2_edges = torch.tensor([[0]])
    A3_nodes = torch.tensor([1, 0]); A3_edges = torch.tensor([[0, 2], [3, 0]])
    A4_nodes = torch.tensor([1, 0]); A4_edges = torch.tensor([[0, 0], [2, 0]])
    nodes = [A1_nodes, A2_nodes, A3_nodes, A4_nodes]
    edges = [A1_edges, A2_edges, A3_edges, A4_edges]
    nf = [x[:, None].repeat(1, 4).float() for x in nodes]
    sp = [e.to_sparse().coalesce() for e in edges]
    ei = [e.indices() for e in sp]
    ef = [e.values()[:, None].repeat(1, 4).float() for e in sp]
    return nf, ei, ef


def dump_util(name, nf, ei, ef):
    G = make_batch(nf, ei, ef)
    out = pack_inputs(nf, ei, ef)
    out.update(indices=npy(G.indices), values=npy(G.values), mask=npy(G.mask), node_mask=npy(G.node_mask),
               t_indices=npy(G.t_indices), t_mask=npy(G.t_mask), node_ofs=npy(G.node_ofs),
               sizes=np.array(G.n_edges),
               t_values=npy(t(G).values), d_values=npy(d(G).values), nd_values=npy(nd(G).values),
               v2d_values=npy(v2d(G, d(G)).values), tt_values=npy(t(t(G)).values))
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_linear(name, nf, ei, ef, dout, seed):
    torch.manual_seed(seed)
    G = make_batch(nf, ei, ef)
    D = G.values.size(-1)
    out = pack_inputs(nf, ei, ef)
    for (oi, oo) in [(2, 2), (2, 1), (2, 0), (1, 1), (1, 0)]:
        for cfg in ('light', 'default'):
            for normalize in (True, False):
                tag = f'l{oi}{oo}_{cfg}_{int(normalize)}'
                lin = Linear(oi, oo, D, dout, bias=True, cfg=cfg, normalize=normalize)
                for b in lin.bias.bias:
                    torch.nn.init.normal_(b)
                x = G.values.clone().requires_grad_(True)
                Gin = batch_like(G, x, skip_masking=True)
                if oi == 1:
                    Gin = d(Gin)
                captured = {}
                if (oi, oo) == (2, 2):
                    orig = lin.weight.get_output
                    def hook(G_, vl, _o=orig, _c=captured):
                        _c['V'] = [v.detach().clone() for v in vl]
                        return _o(G_, vl)
                    lin.weight.get_output = hook
                y = lin(Gin)
                yv = y if oo == 0 else y.values
                R = torch.randn_like(yv)
                (yv * R).sum().backward()
                out[tag + '_w'] = npy(lin.weight.weight)
                for k, b in enumerate(lin.bias.bias):
                    out[tag + f'_b{k}'] = npy(b)
                    out[tag + f'_gb{k}'] = npy(b.grad)
                out[tag + '_y'] = npy(yv)
                out[tag + '_R'] = npy(R)
                out[tag + '_gx'] = npy(x.grad)
                out[tag + '_gw'] = npy(lin.weight.weight.grad)
                if 'V' in captured:
                    out[tag + '_V'] = npy(torch.stack(captured['V'], 0))
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_phi_coef(name, nf, ei, ef, d_head, n_heads, dv, seed):
    """KernelFeatureMapWrapper for order 1/2 x query/key x generalized/softmax, and the seven
    KernelAttnCoef cases of the reference's test_kernel_attn with random (not all-ones) inputs."""
    torch.manual_seed(seed)
    G2 = make_batch(nf, ei, ef)
    B, E, _ = G2.values.shape
    Nmax = G2.node_mask.size(1)
    dqk = d_head * n_heads
    out = pack_inputs(nf, ei, ef)
    out['dims'] = np.array([d_head, n_heads, dv])
    x2 = torch.randn(B, E, dqk)
    x1 = torch.randn(B, Nmax, dqk)
    v2 = torch.randn(B, E, dv)
    v1 = torch.randn(B, Nmax, dv)
    out.update(x1=npy(x1), x2=npy(x2), v1=npy(v1), v2=npy(v2))
    X2 = batch_like(G2, x2)
    X1 = Batch(None, x1, G2.n_nodes, None)
    V2 = batch_like(G2, v2)
    V1 = Batch(None, v1, G2.n_nodes, None)
    for gen in (True, False):
        fmap = KernelFeatureMap(d_head, generalized_attention=gen)
        m = fmap.num_features
        wrap = KernelFeatureMapWrapper(fmap, dqk, n_heads)
        g = 'g' if gen else 's'
        out[f'proj_{g}'] = npy(fmap.projection_matrix)
        q1, k1 = wrap(X1, True), wrap(X1, False)
        q2, k2 = wrap(X2, True), wrap(X2, False)
        out[f'phi_{g}_q1'] = npy(q1.values); out[f'phi_{g}_k1'] = npy(k1.values)
        out[f'phi_{g}_q2'] = npy(q2.values); out[f'phi_{g}_k2'] = npy(k2.values)
        a11 = KernelAttnCoef(1, 1, m, dv, n_heads); a12 = KernelAttnCoef(1, 2, m, dv, n_heads)
        a21 = KernelAttnCoef(2, 1, m, dv, n_heads); a22 = KernelAttnCoef(2, 2, m, dv, n_heads)
        out[f'att_{g}_111'] = npy(a11(q1, k1, V1).values)
        out[f'att_{g}_112_d12'] = npy(a11(q1, k1, V2, (1, 2)).values)
        out[f'att_{g}_122'] = npy(a12(q1, k2, V2).values)
        out[f'att_{g}_211'] = npy(a21(q2, k1, V1).values)
        out[f'att_{g}_212_d13'] = npy(a21(q2, k1, V2, (1, 3)).values)
        out[f'att_{g}_212_d23'] = npy(a21(q2, k1, V2, (2, 3)).values)
        out[f'att_{g}_222'] = npy(a22(q2, k2, V2).values)
        # list form: two tuples at once, second value transposed (as KernelSelfAttn uses it)
        xq = torch.randn(B, E, dqk); xk = torch.randn(B, Nmax, dqk); vv = torch.randn(B, E, dv)
        out[f'l_{g}_xq'] = npy(xq); out[f'l_{g}_xk'] = npy(xk); out[f'l_{g}_vv'] = npy(vv)
        qb = wrap(batch_like(G2, xq), True); kb = wrap(Batch(None, xk, G2.n_nodes, None), False)
        res = a21([q2, qb], [k1, kb], [V2, t(batch_like(G2, vv))], (2, 3))
        out[f'l_{g}_o0'] = npy(res[0].values); out[f'l_{g}_o1'] = npy(res[1].values)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_layer(name, nf, ei, ef, dim_qk, dim_v, n_heads, seed):
    torch.manual_seed(seed)
    G = make_batch(nf, ei, ef)
    D = G.values.size(-1)
    out = pack_inputs(nf, ei, ef)
    out['dims'] = np.array([D, dim_v, dim_qk, n_heads])
    d_head = dim_qk // n_heads
    for gen in (True, False):
        # cfg='local' is not captured: in the reference it fails with a shape error for (2,2) and (2,1)
        # (fc_o is sized for all n_v terms but only the local ones are concatenated, kernelattn.py:49-50,146-158)
        for (oi, oo, cfg) in [(2, 2, 'default'), (2, 1, 'default'), (1, 1, 'default')]:
            tag = f"{'g' if gen else 's'}{oi}{oo}{cfg[0]}"
            fmap = KernelFeatureMap(d_head, generalized_attention=gen)
            layer = KernelSelfAttn(oi, oo, D, dim_v, dim_qk, n_heads, cfg, 0., 0., fmap)
            for n_, p in layer.named_parameters():
                if 'bias' in n_:
                    torch.nn.init.normal_(p, std=0.1)
            x = G.values.clone().requires_grad_(True)
            Gin = batch_like(G, x, skip_masking=True)
            if oi == 1:
                Gin = d(Gin)
            y = layer(Gin)
            R = torch.randn_like(y.values)
            (y.values * R).sum().backward()
            for k, v in layer.state_dict().items():
                out[f'{tag}/sd/{k}'] = npy(v)
            for k, p in layer.named_parameters():
                out[f'{tag}/grad/{k}'] = npy(p.grad)
            out[f'{tag}/y'] = npy(y.values); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(x.grad)
    # one full EncLayer(2,2) with the generalized kernel
    fmap = KernelFeatureMap(d_head, generalized_attention=True)
    enc = EncLayer(2, 2, D, dim_qk, dim_v, 2 * D, n_heads, 'default', 'generalized_kernel', 0., 0., fmap, sparse=True)
    x = G.values.clone().requires_grad_(True)
    y = enc(batch_like(G, x, skip_masking=True))
    R = torch.randn_like(y.values)
    (y.values * R).sum().backward()
    for k, v in enc.state_dict().items():
        out[f'enc/sd/{k}'] = npy(v)
    for k, p in enc.named_parameters():
        out[f'enc/grad/{k}'] = npy(p.grad)
    out['enc/y'] = npy(y.values); out['enc/R'] = npy(R); out['enc/gx'] = npy(x.grad)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_readout(name, nf, ei, ef, dim_qk, dim_v, n_heads, seed):
    """F8: the linear-cost softmax readouts SelfAttn(2,0) / SelfAttn(1,0) (sparse/attn.py:82-131, attncoef.py:25-59)
    and a whole Encoder(2, 0, [2], ...) with the generalized kernel (8f-1), fixed projections."""
    torch.manual_seed(seed)
    G = make_batch(nf, ei, ef)
    D = G.values.size(-1)
    out = pack_inputs(nf, ei, ef)
    out['dims'] = np.array([D, dim_v, dim_qk, n_heads])
    for oi in (2, 1):
        tag = f'r{oi}0'
        layer = SelfAttn(oi, 0, D, dim_v, dim_qk, n_heads, 'default', 0., 0.)
        for n_, p in layer.named_parameters():
            if 'bias' in n_:
                torch.nn.init.normal_(p, std=0.1)
        x = G.values.clone().requires_grad_(True)
        Gin = batch_like(G, x, skip_masking=True)
        if oi == 1:
            Gin = d(Gin)
        y = layer(Gin)
        R = torch.randn_like(y)
        (y * R).sum().backward()
        for k, v in layer.state_dict().items():
            out[f'{tag}/sd/{k}'] = npy(v)
        for k, p in layer.named_parameters():
            out[f'{tag}/grad/{k}'] = npy(p.grad)
        out[f'{tag}/y'] = npy(y); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(x.grad)
    dh, dout = 16, 3
    out['enc_dims'] = np.array([D, dout, dh, dim_qk, dim_v, 2 * dh, n_heads, dim_qk, dim_v, n_heads])
    enc = Encoder(2, 0, [2], D, dout, dh, dim_qk, dim_v, 2 * dh, n_heads, dim_qk, dim_v, n_heads, 'default',
                  'generalized_kernel', 0., 0., 0., sparse=True)
    enc.skip_redraw_projections = True
    for n_, p in enc.named_parameters():
        if 'bias' in n_ and p.dim() == 1:
            torch.nn.init.normal_(p, std=0.1)
    x = G.values.clone().requires_grad_(True)
    y = enc(batch_like(G, x, skip_masking=True))
    R = torch.randn_like(y)
    (y * R).sum().backward()
    for k, v in enc.state_dict().items():
        out[f'enc0/sd/{k}'] = npy(v)
    for k, p in enc.named_parameters():
        out[f'enc0/grad/{k}'] = npy(p.grad)
    out['enc0/y'] = npy(y); out['enc0/R'] = npy(R); out['enc0/gx'] = npy(x.grad)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_mlp(name, nf, ei, ef, seed):
    """F9: MLP stacks of Linear(cfg='default', normalize=True) (models/mlp.py:15-40, 8f-3): 2->2->2 (order-2 output),
    2->2->1->0 (graph vector) and 2->1->1 (order-1 output); biases made non-zero so that they are pinned too."""
    from hot_pytorch.models.mlp import MLP
    torch.manual_seed(seed)
    G = make_batch(nf, ei, ef)
    D = G.values.size(-1)
    out = pack_inputs(nf, ei, ef)
    cases = {'m222': (2, 2, [2], D, 5, 7, 'relu'), 'm2210': (2, 0, [2, 1], D, 3, [6, 4], 'gelu'), 'm211': (2, 1, [1], D, 4, 5, 'tanh')}
    for tag, (oi, oo, oh, di, do, dh, f) in cases.items():
        mlp = MLP(oi, oo, oh, di, do, dh, f, 0., sparse=True)
        for n_, p in mlp.named_parameters():
            if 'bias' in n_:
                torch.nn.init.normal_(p, std=0.1)
        x = G.values.clone().requires_grad_(True)
        y = mlp(batch_like(G, x, skip_masking=True))
        yv = y if isinstance(y, torch.Tensor) else y.values
        R = torch.randn_like(yv)
        (yv * R).sum().backward()
        for k, v in mlp.state_dict().items():
            out[f'{tag}/sd/{k}'] = npy(v)
        for k, p in mlp.named_parameters():
            out[f'{tag}/grad/{k}'] = npy(p.grad)
        out[f'{tag}/y'] = npy(yv); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(x.grad)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_uniform(name, N, B, k, D, dqk, dv, H, seed):
    """F10: the 1 -> k-uniform extension (models/uniform/*, models/extension.py, 8f-2): Linear 1->1 / 1->k in both
    cfgs, KernelSelfAttn 1->1 / 1->k with the generalized kernel, and a whole extension Encoder / MLP.  The GPS data of
    k-uniform-hyperedge has 221 nodes and 3-node hyperedges; the fixture is a small synthetic set of that shape."""
    from hot_pytorch.models.uniform.linear import Linear as ULinear
    from hot_pytorch.models.uniform.kernelattn import KernelSelfAttn as UAttn
    from hot_pytorch.models import extension as X
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    x0 = torch.randn(N, D, generator=g)
    idx = torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(B)])       # k distinct nodes per hyperedge
    out = {'x': npy(x0), 'indices': idx.numpy(), 'dims': np.array([N, B, k, D, dqk, dv, H])}

    def run(tag, mod, use_idx):
        for n_, p in mod.named_parameters():
            if 'bias' in n_ and p.dim() == 1:
                torch.nn.init.normal_(p, std=0.1)
        x = x0.clone().requires_grad_(True)
        y = mod(x, idx) if use_idx else mod(x)
        R = torch.randn(y.shape, generator=g)
        (y * R).sum().backward()
        for k_, v in mod.state_dict().items():
            out[f'{tag}/sd/{k_}'] = npy(v)
        for k_, p in mod.named_parameters():
            out[f'{tag}/grad/{k_}'] = npy(p.grad)
        gx = x.grad.to_dense() if x.grad.is_sparse else x.grad     # gather(sparse_grad=True) yields a sparse gradient
        out[f'{tag}/y'] = npy(y); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(gx)

    for cfg in ('light', 'default'):
        run(f'lin11_{cfg}', ULinear(1, 1, D, 5, cfg=cfg), False)
        run(f'lin1k_{cfg}', ULinear(1, k, D, 6, cfg=cfg), True)
    run('att11', UAttn(1, 1, D, dv, dqk, H, 'default', 0., 0., KernelFeatureMap(dqk // H, generalized_attention=True)), False)
    run('att1k', UAttn(1, k, D, dv, dqk, H, 'default', 0., 0., KernelFeatureMap(dqk // H, generalized_attention=True)), True)
    enc = X.Encoder(1, k, [1], D, 3, 16, dqk, dv, 32, H, dqk, dv, H, 'default', 'generalized_kernel', 0., 0., 0.)
    enc.skip_redraw_projections = True
    run('enc', enc, True)
    run('mlp', X.MLP(1, k, [1], D, 4, 9, 'relu', 0.), True)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_dense(name, n_nodes, N, D, dqk, dv, H, dff, seed):
    """F11: the dense layout (batch/dense.py, models/dense/*, 8f-4 / config C3: set input -> graph output): Linear
    1->1 / 1->2 / 2->2 / 2->1 / 2->0 / 1->0 in both cfgs on [B,N,D] / [B,N,N,D] batches with ragged n_nodes,
    KernelSelfAttn 1->1 and 1->2 with the generalized kernel, d / v2d / nd / t, and a whole
    Encoder(1, 2, [1, 1], ..., sparse=False) as set-to-graph/main/models/model.py:9-30 builds it."""
    from hot_pytorch.batch import dense as DB
    from hot_pytorch.models.dense.linear import Linear as DLinear
    from hot_pytorch.models.dense.kernelattn import KernelSelfAttn as DAttn
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    B = len(n_nodes)
    x1 = torch.randn(B, N, D, generator=g)
    x2 = torch.randn(B, N, N, D, generator=g)
    out = {'x1': npy(x1), 'x2': npy(x2), 'n_nodes': np.array(n_nodes), 'dims': np.array([N, D, dqk, dv, H, dff])}

    def run(tag, mod, order):
        for n_, p in mod.named_parameters():
            if 'bias' in n_ and p.dim() == 1:
                torch.nn.init.normal_(p, std=0.1)
        x = (x1 if order == 1 else x2).clone().requires_grad_(True)
        y = mod(DB.Batch(x, n_nodes))
        ya = y if isinstance(y, torch.Tensor) else y.A
        R = torch.randn(ya.shape, generator=g)
        (ya * R).sum().backward()
        for k_, v in mod.state_dict().items():
            out[f'{tag}/sd/{k_}'] = npy(v)
        for k_, p in mod.named_parameters():
            out[f'{tag}/grad/{k_}'] = npy(p.grad)
        out[f'{tag}/y'] = npy(ya); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(x.grad)
        if not isinstance(y, torch.Tensor):
            out[f'{tag}/mask'] = npy(y.mask)

    G2 = DB.Batch(x2.clone(), n_nodes)
    out['util/d'] = npy(DB.d(G2).A); out['util/nd'] = npy(DB.nd(G2).A); out['util/t'] = npy(DB.t(G2).A)
    out['util/v2d'] = npy(DB.v2d(DB.Batch(x1.clone(), n_nodes)).A); out['util/masked2'] = npy(G2.A)
    for cfg in ('light', 'default'):
        run(f'lin11_{cfg}', DLinear(1, 1, D, 5, cfg=cfg), 1)
        run(f'lin12_{cfg}', DLinear(1, 2, D, 6, cfg=cfg), 1)
        run(f'lin22_{cfg}', DLinear(2, 2, D, 7, cfg=cfg), 2)
        run(f'lin21_{cfg}', DLinear(2, 1, D, 5, cfg=cfg), 2)
    run('lin20', DLinear(2, 0, D, 4), 2)
    run('lin10', DLinear(1, 0, D, 4), 1)
    fm = lambda: KernelFeatureMap(dqk // H, generalized_attention=True)
    run('att11', DAttn(1, 1, D, dv, dqk, H, 'default', 0., 0., fm()), 1)
    run('att12', DAttn(1, 2, D, dv, dqk, H, 'default', 0., 0., fm()), 1)
    run('att21', DAttn(2, 1, D, dv, dqk, H, 'default', 0., 0., fm()), 2)
    run('att22', DAttn(2, 2, D, dv, dqk, H, 'default', 0., 0., fm()), 2)
    enc22 = Encoder(2, 2, [2], D, 3, 16, dqk, dv, dff, H, 0, 0, 0, 'default', 'generalized_kernel', 0., 0., 0., sparse=False)
    enc22.skip_redraw_projections = True
    run('enc22', enc22, 2)
    # the same model and inputs in float64: how far the reference's own fp32 result is from the exact one
    enc22.zero_grad(set_to_none=True)
    e64 = enc22.double()
    x64 = x2.double().clone().requires_grad_(True)
    y64 = e64(DB.Batch(x64, n_nodes)).A
    (y64 * torch.from_numpy(out['enc22/R']).double()).sum().backward()
    out['enc22/f64/y'] = npy(y64); out['enc22/f64/gx'] = npy(x64.grad)
    for k_, p in e64.named_parameters():
        out[f'enc22/f64/grad/{k_}'] = npy(p.grad)
    enc = Encoder(1, 2, [1, 1], D, 3, 16, dqk, dv, dff, H, 0, 0, 0, 'default', 'generalized_kernel', 0., 0., 0., sparse=False)
    enc.skip_redraw_projections = True
    run('enc', enc, 1)
    np.savez_compressed(os.path.join(HERE, name), **out)


def dump_dense_readout(name, n_nodes, N, D, dqk, dv, H, dff, seed):
    """the order-0 softmax readout in the DENSE layout (models/dense/attn.py:84-98,138-157 with attncoef.py:22-59): SelfAttn
    1->0 and 2->0 on ragged [B,N,D] / [B,N,N,D] batches, and a whole Encoder(2, 0, [2], ..., sparse=False) whose last layer
    is that readout (encoder.py:34-35)."""
    from hot_pytorch.batch import dense as DB
    from hot_pytorch.models.dense.attn import SelfAttn as DSelfAttn
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    B = len(n_nodes)
    x1 = torch.randn(B, N, D, generator=g)
    x2 = torch.randn(B, N, N, D, generator=g)
    out = {'x1': npy(x1), 'x2': npy(x2), 'n_nodes': np.array(n_nodes), 'dims': np.array([N, D, dqk, dv, H, dff])}

    def run(tag, mod, order):
        for n_, p in mod.named_parameters():
            if 'bias' in n_ and p.dim() == 1:
                torch.nn.init.normal_(p, std=0.1)
        x = (x1 if order == 1 else x2).clone().requires_grad_(True)
        y = mod(DB.Batch(x, n_nodes))
        R = torch.randn(y.shape, generator=g)
        (y * R).sum().backward()
        for k_, v in mod.state_dict().items():
            out[f'{tag}/sd/{k_}'] = npy(v)
        for k_, p in mod.named_parameters():
            out[f'{tag}/grad/{k_}'] = npy(p.grad)
        out[f'{tag}/y'] = npy(y); out[f'{tag}/R'] = npy(R); out[f'{tag}/gx'] = npy(x.grad)

    run('att10', DSelfAttn(1, 0, D, dv, dqk, H), 1)
    run('att20', DSelfAttn(2, 0, D, dv, dqk, H), 2)
    enc = Encoder(2, 0, [2], D, 3, 16, dqk, dv, dff, H, dqk, dv, H, 'default', 'generalized_kernel', 0., 0., 0., sparse=False)
    enc.skip_redraw_projections = True
    run('enc20', enc, 2)
    np.savez_compressed(os.path.join(HERE, name), **out)


def main():
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == 'dense_readout':
        dump_dense_readout('dense_readout.npz', [5, 3, 4, 1], N=5, D=4, dqk=8, dv=8, H=2, dff=12, seed=91)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'dense':
        dump_dense('dense_small.npz', [5, 3, 4], N=5, D=4, dqk=8, dv=8, H=2, dff=12, seed=81)
        dump_dense('dense_c3.npz', [20, 20], N=20, D=2, dqk=32, dv=32, H=4, dff=32, seed=82)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'uniform':
        dump_uniform('uniform_small.npz', N=13, B=21, k=3, D=6, dqk=8, dv=8, H=2, seed=71)
        dump_uniform('uniform_gps.npz', N=221, B=400, k=3, D=16, dqk=32, dv=32, H=4, seed=72)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mlp':
        dump_mlp('mlp_small.npz', *rand_graphs([5, 3, 6, 1, 4], 0.5, 3, 2, seed=61), seed=62)
        dump_mlp('mlp_sym.npz', *rand_graphs([7, 4, 9], 0.35, 4, 4, seed=63, symmetric=True), seed=64)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'readout':     # later additions are generated alone: earlier files stay as committed
        dump_readout('readout_small.npz', *rand_graphs([5, 3, 6, 1, 4], 0.5, 4, 4, seed=51), dim_qk=8, dim_v=8, n_heads=2, seed=52)
        dump_readout('readout_c2.npz', *rand_graphs([14, 9, 20], 0.15, 16, 16, seed=53, symmetric=True), dim_qk=32, dim_v=32, n_heads=4, seed=54)
        return
    # F1/F2 index utilities
    dump_util('util_reftest.npz', *ref_tests_transpose_fixture())
    dump_util('util_random.npz', *rand_graphs([5, 1, 7, 3, 4, 6], 0.4, 3, 2, seed=11))
    dump_util('util_ba64.npz', *ba_graphs(64, 2, 2, 2, seed=12))
    # F3 linear basis
    dump_linear('linear_random.npz', *rand_graphs([5, 1, 7, 2, 6], 0.45, 3, 3, seed=21), dout=5, seed=22)
    dump_linear('linear_sym.npz', *rand_graphs([6, 4, 3], 0.5, 2, 2, seed=23, symmetric=True), dout=4, seed=24)
    # F4/F5 feature maps and attention coefficients: odd sizes (d=4 -> m=5, dvh=3) and C1 sizes (d=8 -> m=16)
    dump_phi_coef('coef_small.npz', *rand_graphs([2, 3, 4, 5], 0.5, 2, 2, seed=31), d_head=4, n_heads=3, dv=9, seed=32)
    dump_phi_coef('coef_c1.npz', *rand_graphs([9, 6], 0.4, 2, 2, seed=33, symmetric=True), d_head=8, n_heads=4, dv=32, seed=34)
    # F6/F7 layers
    dump_layer('layer_small.npz', *rand_graphs([5, 3, 6, 1], 0.5, 4, 4, seed=41), dim_qk=8, dim_v=8, n_heads=2, seed=42)
    dump_layer('layer_c1.npz', *ba_graphs(16, 2, 16, 16, seed=43), dim_qk=32, dim_v=32, n_heads=4, seed=44)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
