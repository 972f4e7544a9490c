"""CPU oracle for the second-order sparse kernel-attention path of jw9730/hot.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch O(E) CPU restatement (numpy for the
integer/index work, torch-CPU for the float work so that autograd supplies reference gradients)
of the algorithms on the hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product (``hot_amd``) never does.

Parity is PINNED: every function below is checked against golden vectors produced by importing
the real reference in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``)
and against the hand-written expected values held by the reference's own tests
(``hot_pytorch/tests/test_sparse_util.py``, ``test_sparse_linear.py``); see ``tests/test_oracle_*.py``.

Reference locations restated here (paths relative to the reference checkout):
  batch/sparse.py:23-60,134-176   -> OBatch, build_batch
  utils/sparse.py:102-127         -> transpose_info      (sort + binary search instead of O(E^2) match)
  utils/sparse.py:34-99,130-134   -> to_diag / get_diag / nondiag / transpose_values
  utils/set.py:36-48              -> size_mask
  models/common/kernel.py:81-138  -> softmax_features / generalized_features
  models/sparse/kernelattncoef.py:13-28    -> feature_map_batch
  models/sparse/kernelattncoef.py:74-234   -> kernel_attention (segment sums instead of coalesce)
  models/sparse/linear.py:16-347  -> linear_forward (closed-form O(E) pooling instead of O(E^2) masks)
  models/sparse/kernelattn.py:74-159 -> kernel_self_attn
  models/encoder.py:20-53         -> enc_layer
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# container
# --------------------------------------------------------------------------------------------
def size_mask(sizes: Sequence[int]) -> torch.Tensor:
    """[B, max(sizes)] bool, True on data (utils/set.py:36-38)."""
    s = torch.as_tensor(list(sizes), dtype=torch.long)
    return torch.arange(int(s.max()) if len(s) else 0)[None, :] < s[:, None]


class OBatch:
    """Plain record mirroring the fields of the reference Batch (batch/sparse.py:10-60)."""

    def __init__(self, indices, values, n_nodes, n_edges, mask=None, node_mask=None,
                 t_indices=None, t_mask=None, node_ofs=None, zero_invalid=True):
        self.indices = indices
        self.values = values
        self.n_nodes = list(n_nodes)
        self.n_edges = None if n_edges is None else list(n_edges)
        self.order = 1 if indices is None else 2
        self.node_mask = size_mask(self.n_nodes) if node_mask is None else node_mask
        if self.order == 1:
            self.mask = self.node_mask
        else:
            self.mask = size_mask(self.n_edges) if mask is None else mask
        if zero_invalid:
            self.values = self.values * self.mask.unsqueeze(-1).to(self.values.dtype)
        if self.order == 2 and t_indices is None:
            t_indices, t_mask = transpose_info(indices, self.mask)
        self.t_indices, self.t_mask = t_indices, t_mask
        if node_ofs is None:
            node_ofs = torch.as_tensor(np.cumsum([0] + self.n_nodes[:-1]), dtype=torch.long)
        self.node_ofs = node_ofs

    def like(self, values, zero_invalid=False) -> "OBatch":
        return OBatch(self.indices, values, self.n_nodes, self.n_edges, self.mask, self.node_mask,
                      self.t_indices, self.t_mask, self.node_ofs, zero_invalid)

    def as_order1(self, values) -> "OBatch":
        return OBatch(None, values, self.n_nodes, None, None, self.node_mask,
                      self.t_indices, self.t_mask, self.node_ofs, False)


def transpose_info(indices: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """t_indices[b,e] = first row e' of graph b holding the reversed pair, 0 if none;
    t_mask = found & mask (utils/sparse.py:102-127).  Padded rows take part in the search
    exactly as in the reference (they hold (0,0))."""
    idx = indices.numpy().astype(np.int64)
    B, E, _ = idx.shape
    t_idx = np.zeros((B, E), dtype=np.int64)
    found = np.zeros((B, E), dtype=bool)
    if E == 0:
        return torch.from_numpy(t_idx), torch.from_numpy(found)
    K = int(idx.max()) + 1
    fwd = idx[..., 0] * K + idx[..., 1]      # [B,E]
    rev = idx[..., 1] * K + idx[..., 0]
    for b in range(B):
        order = np.argsort(fwd[b], kind="stable")          # first occurrence first
        keys = fwd[b][order]
        pos = np.searchsorted(keys, rev[b], side="left")
        pos_c = np.minimum(pos, E - 1)
        hit = (pos < E) & (keys[pos_c] == rev[b])
        t_idx[b] = np.where(hit, order[pos_c], 0)
        found[b] = hit
    return torch.from_numpy(t_idx), torch.from_numpy(found) & mask


def build_batch(node_feature, edge_index, edge_feature, n_nodes, n_edges) -> OBatch:
    """Concatenated PyG-style input -> padded order-2 batch (batch/sparse.py:134-176):
    per graph the n diagonal rows (v,v) carrying [node_feat | 0] come first, then the e edge rows
    carrying [0 | edge_feat], then zero padding."""
    B = len(n_nodes)
    dn, de = node_feature.size(-1), edge_feature.size(-1)
    sizes = [n + e for n, e in zip(n_nodes, n_edges)]
    Emax = max(sizes)
    indices = torch.zeros(B, Emax, 2, dtype=torch.long)
    values = torch.zeros(B, Emax, dn + de, dtype=node_feature.dtype)
    no = eo = 0
    for b, (n, e) in enumerate(zip(n_nodes, n_edges)):
        indices[b, :n, 0] = torch.arange(n)
        indices[b, :n, 1] = torch.arange(n)
        indices[b, n:n + e] = edge_index[:, eo:eo + e].t()
        values[b, :n, :dn] = node_feature[no:no + n]
        values[b, n:n + e, dn:] = edge_feature[eo:eo + e]
        no += n
        eo += e
    return OBatch(indices, values, n_nodes, sizes)


def build_batch_from_lists(node_features, edge_indices, edge_features) -> OBatch:
    """batch/sparse.py:179-190."""
    return build_batch(torch.cat(node_features), torch.cat(edge_indices, 1), torch.cat(edge_features),
                       [x.size(0) for x in node_features], [e.size(0) for e in edge_features])


# --------------------------------------------------------------------------------------------
# order-2 index utilities
# --------------------------------------------------------------------------------------------
def _diag_mask(G: OBatch):
    return (G.indices[..., 0] == G.indices[..., 1]) & G.mask


def transpose_values(values, t_indices, t_mask):
    """utils/sparse.py:130-134."""
    T = torch.gather(values, 1, t_indices.unsqueeze(-1).expand(-1, -1, values.size(-1)))
    return T * t_mask.unsqueeze(-1).to(values.dtype)


def get_diag(G: OBatch, values=None):
    """[B,E,C] -> [B,Nmax,C] (utils/sparse.py:59-87): k-th diagonal row -> k-th valid node slot."""
    values = G.values if values is None else values
    rows = values[_diag_mask(G)]
    out = torch.zeros(len(G.n_nodes), G.node_mask.size(1), values.size(-1), dtype=values.dtype)
    sel = G.node_mask.reshape(-1).nonzero().squeeze(-1)
    return out.reshape(-1, values.size(-1)).index_copy(0, sel, rows).reshape(out.shape)


def to_diag(G: OBatch, node_values):
    """[B,Nmax,C] -> [B,E,C], zeros off the diagonal (utils/sparse.py:34-56)."""
    C = node_values.size(-1)
    B, E = G.mask.shape
    sel_dst = _diag_mask(G).reshape(-1).nonzero().squeeze(-1)
    sel_src = G.node_mask.reshape(-1).nonzero().squeeze(-1)
    out = torch.zeros(B * E, C, dtype=node_values.dtype)
    out = out.index_copy(0, sel_dst, node_values.reshape(-1, C).index_select(0, sel_src))
    return out.reshape(B, E, C)


def nondiag(G: OBatch, values=None):
    """utils/sparse.py:90-99."""
    values = G.values if values is None else values
    keep = (G.indices[..., 0] != G.indices[..., 1]) & G.mask
    return values * keep.unsqueeze(-1).to(values.dtype)


def op_t(G):
    return G.like(transpose_values(G.values, G.t_indices, G.t_mask))


def op_d(G):
    return G.as_order1(get_diag(G))


def op_nd(G):
    return G.like(nondiag(G))


def op_v2d(G, V: OBatch):
    return G.like(to_diag(G, V.values))


# --------------------------------------------------------------------------------------------
# FAVOR+ feature maps
# --------------------------------------------------------------------------------------------
def generalized_features(x, proj, eps=1e-3):
    """relu((x d^-1/4) P^T) + eps  (models/common/kernel.py:116-138)."""
    d = x.size(-1)
    return torch.relu((x * d ** -0.25) @ proj.t()) + eps


def softmax_features(x, proj, is_query, eps=1e-4):
    """models/common/kernel.py:81-114; the key branch subtracts the max over the WHOLE tensor."""
    d = x.size(-1)
    m = proj.size(0)
    nrm = d ** -0.25
    dd = (x * nrm) @ proj.t()
    diag = (x ** 2).sum(-1, keepdim=True) / 2.0 * nrm ** 2
    if is_query:
        stab = dd.max(dim=-1, keepdim=True)[0]
    else:
        stab = dd.max()
    return m ** -0.5 * (torch.exp(dd - diag - stab) + eps)


def feature_map_batch(G: OBatch, proj, d_head, generalized, is_query) -> OBatch:
    """models/sparse/kernelattncoef.py:19-28: per-head map, then zero loops (order 2) and padding."""
    chunks = torch.stack(G.values.split(d_head, -1), 0)          # [HH,B,E,d]
    phi = generalized_features(chunks, proj) if generalized else softmax_features(chunks, proj, is_query)
    if G.order == 2:
        keep = (G.indices[..., 0] != G.indices[..., 1])[None, :, :, None].to(phi.dtype)
        phi = phi * keep
    phi = torch.cat(phi.unbind(0), -1)
    return G.like(phi, zero_invalid=True)


# --------------------------------------------------------------------------------------------
# kernel attention (key-value first)
# --------------------------------------------------------------------------------------------
def _guard(x):
    return torch.where(x == 0, torch.full_like(x, 1e-5), x)


def _heads(x, width):
    """[..., HH*width] -> [..., HH, width]"""
    return x.reshape(*x.shape[:-1], x.size(-1) // width, width)


def kernel_attention(ord_q, query: OBatch, key: OBatch, value: OBatch, m, dvh, diagonal=None) -> OBatch:
    """models/sparse/kernelattncoef.py:198-234 for single (already channel-concatenated) inputs.
    query/key hold feature-mapped values [.., HH*m]; value holds [.., HH*dvh]."""
    q = _heads(query.values, m)
    k = _heads(key.values, m)
    v = _heads(value.values, dvh)
    if diagonal is None:
        kv = torch.einsum('blhi,blhj->bhij', k, v)          # per graph, all rows (padding is zero)
        ks = k.sum(1)                                        # [B,HH,m]
        num = torch.einsum('bqhi,bhij->bqhj', q, kv)
        den = _guard(torch.einsum('bqhi,bhi->bqh', q, ks)).unsqueeze(-1)
        att = (num / den).flatten(-2)
    else:
        dim1, _ = diagonal
        B, E = value.mask.shape
        Nmax = key.mask.size(1)
        ntot = B * Nmax
        gidx = value.indices + (torch.arange(B) * Nmax).view(B, 1, 1)       # padded-node ids
        sel = value.mask.reshape(-1).nonzero().squeeze(-1)
        gi = gidx.reshape(-1, 2).index_select(0, sel)
        kf = k.reshape(ntot, *k.shape[2:])
        vf = v.reshape(B * E, *v.shape[2:]).index_select(0, sel)
        kg = kf.index_select(0, gi[:, 0])                                    # key of the ROW node
        outer = kg.unsqueeze(-1) * vf.unsqueeze(-2)                          # [Ev,HH,m,dvh]
        kv = torch.zeros(ntot, *outer.shape[1:], dtype=outer.dtype).index_add(0, gi[:, 1], outer)
        ks = torch.zeros(ntot, *kg.shape[1:], dtype=kg.dtype).index_add(0, gi[:, 1], kg)
        if ord_q == 1:
            qf = q.reshape(ntot, *q.shape[2:])
            num = torch.einsum('nhi,nhij->nhj', qf, kv)
            den = _guard(torch.einsum('nhi,nhi->nh', qf, ks)).unsqueeze(-1)
            att = (num / den).flatten(-2).reshape(B, Nmax, -1)
        else:
            qsel = query.mask.reshape(-1).nonzero().squeeze(-1)
            qg = gidx.reshape(-1, 2).index_select(0, qsel)[:, 1 if dim1 == 2 else 0]
            qf = q.reshape(B * E, *q.shape[2:]).index_select(0, qsel)
            num = torch.einsum('ehi,ehij->ehj', qf, kv.index_select(0, qg))
            den = _guard(torch.einsum('ehi,ehi->eh', qf, ks.index_select(0, qg))).unsqueeze(-1)
            o = (num / den).flatten(-2)
            att = torch.zeros(B * E, o.size(-1), dtype=o.dtype).index_copy(0, qsel, o).reshape(B, E, -1)
    if ord_q == 2:
        keep = (query.indices[..., 0] != query.indices[..., 1]).unsqueeze(-1).to(att.dtype)
        att = att * keep
    return query.like(att, zero_invalid=True)


def kernel_attention_list(ord_q, qs: List[OBatch], ks: List[OBatch], vs: List[OBatch], m, dvh, dv, diagonal):
    """models/sparse/kernelattncoef.py:177-196: S tuples run as S*H heads."""
    q = qs[0].like(torch.cat([x.values for x in qs], -1))
    k = ks[0].like(torch.cat([x.values for x in ks], -1))
    v = vs[0].like(torch.cat([x.values for x in vs], -1))
    att = kernel_attention(ord_q, q, k, v, m, dvh, diagonal)
    return [q.like(a) for a in att.values.split(dv, -1)]


# --------------------------------------------------------------------------------------------
# equivariant linear layers
# --------------------------------------------------------------------------------------------
def _safe(x):
    return torch.where(x == 0, torch.full_like(x, 1e-5), x)


def _pool_norm(x, c):
    """sum / count.  The reference divides a MASKED sum by max(count, 1e-5) (linear.py:11-13); with an
    empty mask that sum is exactly 0.  The closed forms below reach the same pools by subtraction, so an
    empty pool carries rounding residue: force it to the exact 0 the reference produces."""
    return torch.where(c == 0, torch.zeros_like(x), x / _safe(c))


def basis_2_to_2(G: OBatch, cfg='default', normalize=True) -> List[torch.Tensor]:
    """The 5 ('light') or 15 ('default') order-2 -> order-2 basis tensors [B,E,D]
    (models/sparse/linear.py:143-238) in closed form from row/column/graph sums."""
    A = G.values
    B, E, D = A.shape
    Nmax = G.node_mask.size(1)
    dt = A.dtype
    i, j = G.indices[..., 0], G.indices[..., 1]
    valid = G.mask
    off = (i != j) & valid
    offf = off.unsqueeze(-1).to(dt)
    diag_nodes = get_diag(G)                                  # [B,Nmax,D]
    V1 = to_diag(G, diag_nodes)
    tau = G.t_mask.unsqueeze(-1).to(dt)
    At = transpose_values(A, G.t_indices, G.t_mask)
    gat = lambda X, idx: torch.gather(X, 1, idx.clamp(min=0).unsqueeze(-1).expand(-1, -1, X.size(-1)))
    Aii, Ajj = gat(diag_nodes, i), gat(diag_nodes, j)
    out = [V1, A * offf, At * offf, Aii * offf, Ajj * offf]
    if cfg == 'light':
        return out
    ex = lambda idx: idx.unsqueeze(-1).expand(-1, -1, D)
    Aoff = A * offf
    R = torch.zeros(B, Nmax, D, dtype=dt).scatter_add(1, ex(i), Aoff)         # off-diagonal row sums
    C = torch.zeros(B, Nmax, D, dtype=dt).scatter_add(1, ex(j), Aoff)         # off-diagonal column sums
    od = torch.zeros(B, Nmax, dtype=dt).scatter_add(1, i, off.to(dt)).unsqueeze(-1)
    idg = torch.zeros(B, Nmax, dtype=dt).scatter_add(1, j, off.to(dt)).unsqueeze(-1)
    T = diag_nodes.sum(1, keepdim=True)                                       # [B,1,D]
    S = Aoff.sum(1, keepdim=True)
    Eo = off.sum(1).to(dt).view(B, 1, 1)
    n = torch.as_tensor(G.n_nodes, dtype=dt).view(B, 1, 1)
    nrm = _pool_norm if normalize else (lambda x, c: x)
    nm = G.node_mask.unsqueeze(-1).to(dt)
    # node-level quantities placed on the diagonal
    V6 = to_diag(G, nrm(R, od) * nm)
    V7 = to_diag(G, nrm(C, idg) * nm)
    V12 = to_diag(G, nrm(T - diag_nodes, n - 1) * nm)
    V13 = to_diag(G, nrm(S - R - C, Eo - od - idg) * nm)
    # edge-level quantities (off-diagonal rows only)
    Ri, Rj, Ci, Cj = gat(R, i), gat(R, j), gat(C, i), gat(C, j)
    odi, odj, idi, idj = gat(od, i), gat(od, j), gat(idg, i), gat(idg, j)
    V8 = nrm(Cj - A, idj - 1) * offf
    V9 = nrm(Ri - A, odi - 1) * offf
    V10 = nrm(Rj - At, odj - tau) * offf
    V11 = nrm(Ci - At, idi - tau) * offf
    V14 = nrm(T - Aii - Ajj, n - 2) * offf
    V15 = nrm(S - Ri - Ci - Rj - Cj + A + At, Eo - odi - idi - odj - idj + 1 + tau) * offf
    return out + [V6, V7, V8, V9, V10, V11, V12, V13, V14, V15]


def basis_2_to_1(G: OBatch, cfg='default', normalize=True) -> List[torch.Tensor]:
    """models/sparse/linear.py:99-141 -> list of [B,Nmax,D].  As in the reference, the
    'sum of the other diagonals' term is NOT masked on padded node rows (the layer masks later)."""
    A = G.values
    B, E, D = A.shape
    Nmax = G.node_mask.size(1)
    dt = A.dtype
    diag_nodes = get_diag(G)
    if cfg == 'light':
        return [diag_nodes]
    i, j = G.indices[..., 0], G.indices[..., 1]
    off = (i != j) & G.mask
    Aoff = A * off.unsqueeze(-1).to(dt)
    ex = lambda idx: idx.unsqueeze(-1).expand(-1, -1, D)
    R = torch.zeros(B, Nmax, D, dtype=dt).scatter_add(1, ex(i), Aoff)
    C = torch.zeros(B, Nmax, D, dtype=dt).scatter_add(1, ex(j), Aoff)
    od = torch.zeros(B, Nmax, dtype=dt).scatter_add(1, i, off.to(dt)).unsqueeze(-1)
    idg = torch.zeros(B, Nmax, dtype=dt).scatter_add(1, j, off.to(dt)).unsqueeze(-1)
    T = diag_nodes.sum(1, keepdim=True)
    S = Aoff.sum(1, keepdim=True)
    Eo = off.sum(1).to(dt).view(B, 1, 1)
    n = torch.as_tensor(G.n_nodes, dtype=dt).view(B, 1, 1)
    nrm = _pool_norm if normalize else (lambda x, c: x)
    nm = G.node_mask.unsqueeze(-1).to(dt)
    # the three masked bmm terms are zero on padded node rows (their masks contain node_mask)
    return [diag_nodes, nrm(R, od) * nm, nrm(C, idg) * nm, nrm(T - diag_nodes, n - 1),
            nrm(S - R - C, Eo - od - idg) * nm]


def basis_2_to_0(G: OBatch, normalize=True) -> List[torch.Tensor]:
    """models/sparse/linear.py:85-97 -> [trace pool, off-diagonal pool], each [B,D]."""
    dt = G.values.dtype
    V1 = get_diag(G).sum(1)
    V2 = nondiag(G).sum(1)
    if normalize:
        n = torch.as_tensor(G.n_nodes, dtype=dt).unsqueeze(-1)
        e = torch.as_tensor(G.n_edges, dtype=dt).unsqueeze(-1) - n
        V1, V2 = V1 / _safe(n), V2 / _safe(e)
    return [V1, V2]


def basis_1_to_1(G: OBatch, cfg='default', normalize=True) -> List[torch.Tensor]:
    """models/sparse/linear.py:64-80."""
    x = G.values
    if cfg == 'light':
        return [x]
    n = torch.as_tensor(G.n_nodes, dtype=x.dtype).view(-1, 1, 1) - 1
    xn = x / _safe(n) if normalize else x
    return [x, xn.sum(1, keepdim=True) - xn]


def basis_1_to_0(G: OBatch, normalize=True) -> List[torch.Tensor]:
    """models/sparse/linear.py:54-62."""
    V = G.values.sum(1)
    if normalize:
        V = V / _safe(torch.as_tensor(G.n_nodes, dtype=V.dtype).unsqueeze(-1))
    return [V]


def linear_forward(G, ord_in, ord_out, weight, biases: Optional[Sequence[torch.Tensor]],
                   cfg='default', normalize=True):
    """Linear = basis @ weight + bias, then zero the padding (models/sparse/linear.py:330-347).
    weight: [n_w*D, Dout]; biases: list of 1 (order 0/1) or 2 (order 2: [off-diagonal, diagonal])."""
    if (ord_in, ord_out) == (2, 2):
        Vs = basis_2_to_2(G, cfg, normalize)
    elif (ord_in, ord_out) == (2, 1):
        Vs = basis_2_to_1(G, cfg, normalize)
    elif (ord_in, ord_out) == (2, 0):
        Vs = basis_2_to_0(G, normalize)
    elif (ord_in, ord_out) == (1, 1):
        Vs = basis_1_to_1(G, cfg, normalize)
    elif (ord_in, ord_out) == (1, 0):
        Vs = basis_1_to_0(G, normalize)
    else:
        raise NotImplementedError((ord_in, ord_out))
    y = torch.cat(Vs, -1) @ weight
    if ord_out == 0:
        return y if biases is None else y + biases[0]
    if ord_out == 1:
        H = G.as_order1(y) if G.order == 2 else G.like(y)
        if biases is not None:
            y = y + biases[0]
        return H.like(y, zero_invalid=True)
    if biases is not None:
        dt = y.dtype
        i, j = G.indices[..., 0], G.indices[..., 1]
        off = ((i != j) & G.mask).unsqueeze(-1).to(dt)
        dia = ((i == j) & G.mask).unsqueeze(-1).to(dt)
        y = y + off * biases[0] + dia * biases[1]
    return G.like(y, zero_invalid=True)


def _lin(G, sd, prefix, ord_in, ord_out, cfg='light', normalize=True):
    w = sd[prefix + '.weight.weight']
    nb = 2 if ord_out == 2 else 1
    b = [sd[f'{prefix}.bias.bias.{k}'] for k in range(nb)] if f'{prefix}.bias.bias.0' in sd else None
    return linear_forward(G, ord_in, ord_out, w, b, cfg, normalize)


# --------------------------------------------------------------------------------------------
# KernelSelfAttn (models/sparse/kernelattn.py)
# --------------------------------------------------------------------------------------------
def kernel_self_attn(G: OBatch, sd: dict, ord_in, ord_out, dim_in, dim_v, dim_qk, n_heads,
                     cfg='default', generalized=True) -> OBatch:
    """Functional restatement of KernelSelfAttn.forward with dropout = 0.
    ``sd`` uses the reference's state_dict key names."""
    proj = sd['feature_map.feature_map.projection_matrix']
    m = proj.size(0)
    d_head = dim_qk // n_heads if dim_qk >= n_heads else 1
    dvh = dim_v // n_heads if dim_v >= n_heads else 1
    fm = lambda X, is_q: feature_map_batch(X, proj, d_head, generalized, is_q)
    split_qk = lambda X: [X.like(v) for v in X.values.split(m * n_heads, -1)]

    def v_list():
        v = G.like(F.linear(G.values, sd['fc_v.weight'], sd['fc_v.bias']), zero_invalid=True)
        return [G.like(x) for x in v.values.split(dim_v, -1)]

    def combine(H, atts):
        y = F.linear(torch.cat([a.values for a in atts], -1), sd['fc_o.weight'], sd['fc_o.bias'])
        return H.like(y, zero_invalid=True)

    att1 = lambda q, k, v, dg=None: kernel_attention(1, q, k, v, m, dvh, dg)
    att2 = lambda q, k, v, dg=None: kernel_attention(2, q, k, v, m, dvh, dg)
    attl = lambda o, q, k, v, dg: kernel_attention_list(o, q, k, v, m, dvh, dim_v, dg)

    if (ord_in, ord_out) == (1, 1):
        h1 = _lin(G, sd, 'fc_1', 1, 1)
        non_att = h1.like(h1.values[..., -dim_in:])
        q1 = fm(h1.like(h1.values[..., :dim_qk]), True)
        k1 = fm(h1.like(h1.values[..., dim_qk:2 * dim_qk]), False)
        a = att1(q1, k1, v_list()[0])
        y = F.linear(a.values, sd['fc_o.weight'], sd['fc_o.bias'])
        att = G.like(y, zero_invalid=True)
        return G.like(non_att.values + att.values)
    if (ord_in, ord_out) == (2, 1):
        h1 = _lin(G, sd, 'fc_1', 2, 1)
        non_att = h1.like(h1.values[..., -dim_in:])
        q1 = split_qk(fm(h1.like(h1.values[..., :4 * dim_qk]), True))
        k1 = split_qk(fm(h1.like(h1.values[..., 4 * dim_qk:7 * dim_qk]), False))
        k2 = fm(_lin(G, sd, 'fc_2', 2, 2), False)
        v = v_list()
        atts = attl(1, q1[0:2], k1[0:2], [v[0], op_t(v[1])], (1, 2))
        if cfg != 'local':
            atts += [att1(q1[2], k1[2], op_d(v[2])), att1(q1[3], k2, v[3])]
        att = combine(q1[0], atts)
        return h1.like(non_att.values + att.values)
    assert (ord_in, ord_out) == (2, 2)
    h1 = _lin(G, sd, 'fc_1', 2, 1)
    q1 = split_qk(fm(h1.like(h1.values[..., :4 * dim_qk]), True))
    k1 = split_qk(fm(h1.like(h1.values[..., 4 * dim_qk:]), False))
    h2 = _lin(G, sd, 'fc_2', 2, 2)
    non_att = h2.values[..., -dim_in:]
    q2 = split_qk(fm(h2.like(h2.values[..., :6 * dim_qk]), True))
    k2 = split_qk(fm(h2.like(h2.values[..., 6 * dim_qk:8 * dim_qk]), False))
    v = v_list()
    a1, a2 = attl(1, q1[0:2], k1[0:2], [v[0], op_t(v[1])], (1, 2))
    a3, a5 = attl(2, q2[0:2], k1[2:4], [v[2], op_t(v[3])], (2, 3))
    a4, a6 = attl(2, q2[2:4], k1[4:6], [op_t(v[4]), v[5]], (1, 3))
    atts = [op_v2d(G, a1), op_v2d(G, a2), a3, a4, a5, a6]
    if cfg != 'local':
        atts += [op_v2d(G, att1(q1[2], k1[6], op_d(v[6]))),
                 op_v2d(G, att1(q1[3], k2[0], v[7])),
                 att2(q2[4], k1[7], op_d(v[8])),
                 att2(q2[5], k2[1], v[9])]
    att = combine(G, atts)
    return G.like(non_att + att.values)


def enc_layer_2_2(G: OBatch, sd: dict, dim_in, dim_qk, dim_v, dim_ff, n_heads, cfg='default', generalized=True):
    """EncLayer(2,2,...,att_cfg kernel) with dropout 0 (models/encoder.py:20-53).  Note the reference
    passes (dim_qk, dim_v) into KernelSelfAttn's (dim_v, dim_qk) slots; reproduced here."""
    ln = lambda X, p: X.like(F.layer_norm(X.values, (dim_in,), sd[p + '.weight'], sd[p + '.bias']), zero_invalid=True)
    h = ln(G, 'ln.f')
    asd = {k[len('attn.'):]: v for k, v in sd.items() if k.startswith('attn.')}
    h = kernel_self_attn(h, asd, 2, 2, dim_in, dim_qk, dim_v, n_heads, cfg, generalized)
    G = h
    h = ln(G, 'ffn.0.f')
    h = _lin(h, sd, 'ffn.1', 2, 2)
    h = h.like(F.gelu(h.values))
    h = _lin(h, sd, 'ffn.3', 2, 2)
    return G.like(G.values + h.values)


# --------------------------------------------------------------------------------------------
# order-0 softmax readout SelfAttn(1|2, 0) and the whole Encoder(2, 0, [2]*L, ...) (models/sparse/attn.py:84-131,
# models/sparse/attncoef.py:25-59,165-173, models/encoder.py:20-96); dropout = 0
# --------------------------------------------------------------------------------------------
def _vector_query_attend(query, key: OBatch, value: OBatch, n_heads):
    """alpha = softmax over the kept rows of q_h . k_h / sqrt(d_head) (attncoef.py:25-59: keys of an order-2 batch
    keep valid OFF-diagonal rows only), out = sum_rows alpha * v per head, heads concatenated (attncoef.py:165-173).
    A graph without a kept row gives softmax over an all -inf row = NaN in torch; the reference fills alpha with 0
    AFTER the softmax, so NaN * ... -- restated as the reference computes it: masked_fill(~mask, 0) on alpha."""
    dqk, dv = query.size(-1), value.values.size(-1)
    dh = dqk // n_heads if dqk >= n_heads else 1
    dvh = dv // n_heads if dv >= n_heads else 1
    q_ = torch.stack(query.split(dh, -1), 0)                    # [H, B, dh]
    k_ = torch.stack(key.values.split(dh, -1), 0)               # [H, B, E, dh]
    if key.order == 1:
        keep = key.mask
    else:
        keep = (key.indices[..., 0] != key.indices[..., 1]) & key.mask
    sdp = torch.einsum('hbd,hbed->hbe', q_, k_) / math.sqrt(dh)
    sdp = sdp.masked_fill(~keep.unsqueeze(0), -float('inf'))
    alpha = torch.softmax(sdp, 2).masked_fill(~keep.unsqueeze(0), 0)
    v_ = torch.stack((value.values * value.mask.unsqueeze(-1).to(value.values.dtype)).split(dvh, -1), 0)
    att = torch.einsum('hbe,hbed->hbd', alpha, v_)
    return torch.cat(att.unbind(0), -1)


def self_attn_readout(G: OBatch, sd: dict, ord_in, dim_in, dim_v, dim_qk, n_heads):
    """SelfAttn(ord_in, 0).forward (attn.py:84-96 `_1_to_0`, :112-131 `_2_to_0`) -> [B, dim_in]"""
    h0 = _lin(G, sd, 'fc_0', ord_in, 0)
    non_att = h0[..., -dim_in:]
    k1 = _lin(G, sd, 'fc_1', ord_in, 1)
    v = G.like(F.linear(G.values, sd['fc_v.weight'], sd['fc_v.bias']), zero_invalid=True)
    if ord_in == 1:
        att = _vector_query_attend(h0[..., :dim_qk], k1, v, n_heads)
    else:
        k2 = _lin(G, sd, 'fc_2', 2, 2)
        v0, v1 = [G.like(x) for x in v.values.split(dim_v, -1)]
        att = torch.cat([_vector_query_attend(h0[..., :dim_qk], k1, op_d(v0), n_heads),
                         _vector_query_attend(h0[..., dim_qk:2 * dim_qk], k2, v1, n_heads)], -1)
    return non_att + F.linear(att, sd['fc_o.weight'], sd['fc_o.bias'])


def enc_layer_2_0(G: OBatch, sd: dict, dim_in, dim_qk, dim_v, n_heads):
    """EncLayer(2, 0, ...) (models/encoder.py:20-53): LayerNorm -> SelfAttn(2,0) -> FFN on vectors (Linear(0,0) = plain
    matmul + bias, linear.py:50-52,288-291) -> add.  (dim_qk, dim_v) land in SelfAttn's (dim_v, dim_qk) slots
    (encoder.py:35), reproduced."""
    h = G.like(F.layer_norm(G.values, (dim_in,), sd['ln.f.weight'], sd['ln.f.bias']), zero_invalid=True)
    asd = {k[len('attn.'):]: v for k, v in sd.items() if k.startswith('attn.')}
    g = self_attn_readout(h, asd, 2, dim_in, dim_qk, dim_v, n_heads)        # SelfAttn(ord_in, 0, dim_in, dim_v=dim_qk, dim_qk=dim_v)
    h = F.layer_norm(g, (dim_in,), sd['ffn.0.f.weight'], sd['ffn.0.f.bias'])
    h = h @ sd['ffn.1.weight.weight'] + sd['ffn.1.bias.bias.0']
    h = F.gelu(h)
    h = h @ sd['ffn.3.weight.weight'] + sd['ffn.3.bias.bias.0']
    return g + h


def encoder_2_0(G: OBatch, sd: dict, n_layers, dim_hidden, dim_qk, dim_v, dim_ff, n_heads, readout_dim_qk,
                readout_dim_v, readout_n_heads, generalized=True):
    """Encoder(2, 0, [2]*n_layers, ..., 'default', '(generalized_)kernel', sparse=True).forward with dropout 0 and
    fixed projections (models/encoder.py:56-96): input Linear(2,2,light) -> n_layers kernel EncLayer(2,2) -> softmax
    readout EncLayer(2,0) -> LayerNorm + nn.Linear head.  -> [B, dim_out]"""
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    G = _lin(G, sd, 'input.0', 2, 2)
    for l in range(n_layers):
        G = enc_layer_2_2(G, sub(f'layers.{l}.'), dim_hidden, dim_qk, dim_v, dim_ff, n_heads, 'default', generalized)
    g = enc_layer_2_0(G, sub(f'layers.{n_layers}.'), dim_hidden, readout_dim_qk, readout_dim_v, readout_n_heads)
    g = F.layer_norm(g, (dim_hidden,), sd['output.f.0.weight'], sd['output.f.0.bias'])
    return F.linear(g, sd['output.f.1.weight'], sd['output.f.1.bias'])


# --------------------------------------------------------------------------------------------
# synthetic workloads shared by tests and bench (own seeded generator; SURVEY 8d "S1")
# --------------------------------------------------------------------------------------------
def ba_edges(n: int, m: int, rng: np.random.Generator) -> np.ndarray:
    """Barabasi-Albert preferential attachment (same process as run_perf_tests.py:16 uses through
    networkx; own implementation, not bit-identical).  Returns a symmetric directed edge list
    [2, 2*(n-m)*m] in row-major COO order."""
    targets = list(range(m))
    repeated: List[int] = []
    und = []
    for src in range(m, n):
        und.extend((src, t) for t in targets)
        repeated.extend(targets)
        repeated.extend([src] * m)
        chosen = set()
        while len(chosen) < m:
            chosen.add(repeated[int(rng.integers(len(repeated)))])
        targets = list(chosen)
    und = np.asarray(und, dtype=np.int64)
    ei = np.concatenate([und, und[:, ::-1]], 0)
    ei = ei[np.lexsort((ei[:, 1], ei[:, 0]))]
    return np.ascontiguousarray(ei.T)
