"""Pooling / broadcast basis tensors of the 'default' equivariant layers, O(E) through segment sums.
Closed forms of the reference's masked bmm poolings (models/sparse/linear.py:54-97,99-141,162-238); a pool
over an empty set is exactly 0 (the reference divides an all-zero masked sum by 1e-5)."""
import torch

from ... import ops, _lib
from ..._lib import KEEP_DIAG, KEEP_OFFDIAG


_COUNT_CACHE = {}


def _counts(x, dtype, device):
    """python list of per-graph sizes -> device tensor, uploaded once per list object (a host-to-device copy per
    forward would also forbid capturing the step into a HIP graph)"""
    key = (id(x), len(x), dtype, str(device))
    hit = _COUNT_CACHE.get(key)
    if hit is not None and hit[0] is x:
        return hit[1]
    if len(_COUNT_CACHE) > 64:
        _COUNT_CACHE.clear()
    t = torch.as_tensor(x, dtype=dtype, device=device)
    _COUNT_CACHE[key] = (x, t)
    return t


def _safe_div(x, cnt):
    cnt = cnt.to(x.dtype)
    return torch.where(cnt == 0, torch.zeros_like(x), x / torch.where(cnt == 0, torch.ones_like(cnt), cnt))


def pool_1_to_0(G, normalize):
    """[B,N,D] -> [B,D] sum (mean) over the nodes of each graph (linear.py:54-62)."""
    plan = G.plan
    V = ops.SegSumRows.apply(G.values, plan.graph_seg(1), plan.nkind, KEEP_DIAG, plan_node_graph(plan))
    if normalize:
        V = _safe_div(V, _counts(G.n_nodes, V.dtype, V.device)[:, None])
    return V


def offdiag_count(plan, n_nodes, n_edges, dtype, device):
    """[B,1] number of off-diagonal entries per graph; in edge-shard mode the count over every rank's entries"""
    if not hasattr(plan, '_eo_graph'):
        e = (_counts(n_edges, torch.float32, device) - _counts(n_nodes, torch.float32, device))[:, None].clone()
        shard = ops.shard_of(plan)
        plan._eo_graph = e if shard is None else shard.all_reduce_(e)
    return plan._eo_graph.to(dtype)


def pool_2_to_0(G, normalize):
    """[B,E,D] -> [B,2D]: trace pool and off-diagonal pool (linear.py:85-97).  Edge-shard mode: the diagonal is replicated
    on every rank (its pool is complete locally), the off-diagonal pool is added over the ranks."""
    plan = G.plan
    eg = plan_entry_graph(plan)
    V1 = ops.SegSumRows.apply(G.values, plan.graph_seg(2), plan.kind, KEEP_DIAG, eg)
    V2 = ops.entry_sum(ops.SegSumRows.apply(G.values, plan.graph_seg(2), plan.kind, KEEP_OFFDIAG, eg), plan)
    if normalize:
        n = _counts(G.n_nodes, V1.dtype, V1.device)[:, None]
        V1, V2 = _safe_div(V1, n), _safe_div(V2, offdiag_count(plan, G.n_nodes, G.n_edges, V1.dtype, V1.device))
    return torch.cat([V1, V2], -1)


def plan_entry_graph(plan):
    if not hasattr(plan, '_entry_graph'):
        plan._entry_graph = (torch.arange(plan.R, device=plan.device) // max(plan.Emax, 1)).to(torch.int32)
    return plan._entry_graph


def plan_node_graph(plan):
    if not hasattr(plan, '_node_graph'):
        plan._node_graph = (torch.arange(plan.Nn, device=plan.device) // max(plan.Nmax, 1)).to(torch.int32)
    return plan._node_graph


def _plan_counts(plan, n_nodes, n_edges):
    """Element counts of every pool (integer bookkeeping from the CSR pointers, cached on the plan): od / idg =
    off-diagonal out / in degree per node row, n / Eo per graph and their gathers to node rows (fp32)."""
    if not hasattr(plan, '_counts'):
        dev = plan.device
        f = dict(dtype=torch.float32, device=dev)
        valid = (plan.nkind == 1)
        od = ((plan.row_ptr[1:] - plan.row_ptr[:-1] - 1).clamp(min=0) * valid).to(torch.float32).unsqueeze(1)
        idg = ((plan.col_ptr[1:] - plan.col_ptr[:-1] - 1).clamp(min=0) * valid).to(torch.float32).unsqueeze(1)
        shard = ops.shard_of(plan)
        if shard is not None:      # degrees over every rank's entries (each CSR row holds the replicated diagonal once)
            shard.all_reduce_(od)
            shard.all_reduce_(idg)
        n_g = torch.as_tensor(n_nodes, **f).unsqueeze(1)                            # [B,1]
        eo_g = offdiag_count(plan, n_nodes, n_edges, torch.float32, dev)
        ng = plan_node_graph(plan).long()
        plan._counts = dict(od=od.contiguous(), idg=idg.contiguous(), n_graph=n_g.contiguous(), eo_graph=eo_g.contiguous(),
                            n_node=n_g[ng], eo_node=eo_g[ng])
    return plan._counts


def _nrm(x, cnt, normalize):
    return _safe_div(x, cnt) if normalize else x


def _order2_sums(G):
    """diag [Nn,D], off-diagonal row sums R and column sums C [Nn,D], trace T and off-diagonal total S [B,D].
    Edge-shard mode (hot_amd.dist): R, C and S are sums over off-diagonal entries, which are spread over the ranks --
    the local sums are all-reduced in ONE flat buffer (ops.AllReduceSum: forward and gradient); diag and T come from the
    diagonal entries, which every rank holds."""
    plan = G.plan
    A = G.values.reshape(plan.R, -1)
    D = A.size(1)
    diag = ops.GatherRows.apply(A, plan.diag_pos, plan.diag_node, plan.Nn)
    R = ops.SegSumRows.apply(A, plan.row_seg(), plan.kind, KEEP_OFFDIAG, plan.rown)
    C = ops.SegSumRows.apply(A, plan.col_seg(), plan.kind, KEEP_OFFDIAG, plan.coln)
    eg = plan_entry_graph(plan)
    T = ops.SegSumRows.apply(A, plan.graph_seg(2), plan.kind, KEEP_DIAG, eg)
    S = ops.SegSumRows.apply(A, plan.graph_seg(2), plan.kind, KEEP_OFFDIAG, eg)
    shard = ops.shard_of(plan)
    if shard is not None:
        flat = ops.AllReduceSum.apply(torch.cat([R.reshape(-1), C.reshape(-1), S.reshape(-1)]), shard)
        R, C, S = flat[:R.numel()].view_as(R), flat[R.numel():2 * R.numel()].view_as(C), flat[2 * R.numel():].view_as(S)
    return A, diag, R, C, T, S


def basis_1_to_1(G, normalize):
    """[x, sum of the other nodes (/ (n-1))]  (linear.py:64-80); the pooled term is left unmasked on padded node
    rows exactly as in the reference (Linear masks afterwards)."""
    plan = G.plan
    x = G.values.reshape(plan.Nn, -1)
    ng = plan_node_graph(plan)
    if normalize:
        n1 = _counts(G.n_nodes, x.dtype, x.device)[ng.long()].unsqueeze(1) - 1
        n1 = torch.where(n1 == 0, torch.full_like(n1, 1e-5), n1)
        x = x / n1
        xs = G.values.reshape(plan.Nn, -1)
    else:
        xs = x
    tot = ops.SegSumRows.apply(x, plan.graph_seg(1), plan.nkind, KEEP_DIAG, ng)
    pooled = ops.GatherSeg.apply(tot, ng, plan.graph_seg(1), plan.Nn) - x
    return torch.cat([xs, pooled], -1).view(*G.values.shape[:-1], -1)


def basis_2_to_1(G, dG, normalize):
    """[A_vv, R_v/od, C_v/id, (T - A_vv)/(n-1), (S - R_v - C_v)/(Eo - od - id)]  (linear.py:99-141); the fourth term is
    not masked on padded node rows, as in the reference."""
    plan = G.plan
    c = _plan_counts(plan, G.n_nodes, G.n_edges)
    A, diag, R, C, T, S = _order2_sums(G)
    ng = plan_node_graph(plan)
    nm = (plan.nkind == 1).to(A.dtype).unsqueeze(1)
    Tn = ops.GatherSeg.apply(T, ng, plan.graph_seg(1), plan.Nn)
    Sn = ops.GatherSeg.apply(S, ng, plan.graph_seg(1), plan.Nn)
    V = [diag, _nrm(R, c['od'], normalize) * nm, _nrm(C, c['idg'], normalize) * nm,
         _nrm(Tn - diag, c['n_node'] - 1, normalize),
         _nrm(Sn - R - C, c['eo_node'] - c['od'] - c['idg'], normalize) * nm]
    return torch.cat(V, -1).view(plan.B, plan.Nmax, -1)


def basis_2_to_2(G, normalize):
    """The 15 order-2 -> order-2 basis tensors (linear.py:143-238) in closed form from the row / column /
    graph segment sums: O(E) instead of the reference's [B,E,E] masks; one expand kernel (ops.Basis22Full)."""
    plan = G.plan
    c = _plan_counts(plan, G.n_nodes, G.n_edges)
    counts = (c['od'].view(-1), c['idg'].view(-1), c['n_graph'].view(-1), c['eo_graph'].view(-1))
    V = ops.Basis22Full.apply(G.values.reshape(plan.R, -1), plan, counts, bool(normalize))
    return V.view(plan.B, plan.Emax, -1)
