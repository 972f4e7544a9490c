"""CPU (no GPU): the C-ABI library loads and exports every symbol include/hotk.h declares; host-side layout
logic of the Batch container; the product refuses to compute without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, tt
from hot_amd import _lib
from hot_amd.batch.sparse import make_batch_concatenated, Batch, batch_like
from hot_amd.models.common import KernelFeatureMap
from hot_amd.models.sparse import KernelSelfAttn, Linear


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'hotk.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hotk_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert names == _lib.EXPORTS
    assert _lib.lib().hotk_version() >= 100
    assert _lib.lib().hotk_strerror(-2) == b'workspace too small'


def test_argument_validation_without_gpu():
    """size queries and argument checks are host code: callable with no device"""
    need = ctypes.c_size_t(0)
    assert _lib.lib().hotk_plan_workspace_bytes(4, 100, 10, 300, ctypes.byref(need)) == 0 and need.value > 0
    assert _lib.lib().hotk_sort_workspace_bytes(1000, ctypes.byref(need)) == 0 and need.value >= 8000
    assert _lib.lib().hotk_plan_workspace_bytes(-1, 1, 1, 1, ctypes.byref(need)) == -1
    z = ctypes.c_void_p(0)
    assert _lib.lib().hotk_gather_rows(z, 4, z, z, 4, 10, 4, 0, z) == -1         # null pointers
    assert _lib.lib().hotk_gather_rows(z, 4, z, z, 4, 0, 4, 0, z) == 0           # empty input is fine
    assert _lib.lib().hotk_phi_fwd(z, 8, z, z, 6, z, 8, 0, 1, 8, 16, 7, z, z) == -1  # bad mode


@pytest.mark.parametrize('name', ['util_reftest.npz', 'util_random.npz', 'util_ba64.npz'])
def test_make_batch_layout_on_host(name):
    z = load_golden(name)
    G = make_batch_concatenated(tt(z['node_feature']), tt(z['edge_index']), tt(z['edge_feature']),
                                [int(x) for x in z['n_nodes']], [int(x) for x in z['n_edges']])
    assert np.array_equal(G.indices.numpy(), z['indices'])
    assert np.array_equal(G.values.numpy(), z['values'])
    assert np.array_equal(G.mask.numpy(), z['mask']) and np.array_equal(G.node_mask.numpy(), z['node_mask'])
    assert np.array_equal(G.node_ofs.numpy(), z['node_ofs']) and G.n_edges == list(z['sizes'])
    assert G.order == 2 and G.t_indices is None     # transpose info is built by the device kernel, not on the host
    H = batch_like(G, G.values, skip_masking=True)
    assert H.indices is G.indices and H.n_nodes is G.n_nodes


def test_no_cpu_fallback():
    z = load_golden('util_random.npz')
    G = make_batch_concatenated(tt(z['node_feature']), tt(z['edge_index']), tt(z['edge_feature']),
                                [int(x) for x in z['n_nodes']], [int(x) for x in z['n_edges']])
    layer = KernelSelfAttn(2, 2, 5, 8, 8, 2, 'default', 0., 0., KernelFeatureMap(4, generalized_attention=True))
    with pytest.raises(_lib.HotkError):
        layer(G)
    with pytest.raises(_lib.HotkError):
        Linear(2, 2, 5, 4, cfg='light')(G)


def test_state_dict_keys_match_reference_layout():
    z = load_golden('layer_small.npz')
    ref_keys = sorted(k[len('g22d/sd/'):] for k in z if k.startswith('g22d/sd/'))
    D, dim_v, dim_qk, H = [int(x) for x in z['dims']]
    layer = KernelSelfAttn(2, 2, D, dim_v, dim_qk, H, 'default', 0., 0., KernelFeatureMap(dim_qk // H, generalized_attention=True))
    sd = layer.state_dict()
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == z['g22d/sd/' + k].shape, k


def test_projection_matrix_construction():
    torch.manual_seed(0)
    fm = KernelFeatureMap(8, generalized_attention=True)
    P = fm.projection_matrix
    assert P.shape == (16, 8) and fm.num_features == 16
    # rows of each 8x8 block are orthogonal (scaled by chi-distributed norms)
    for blk in (P[:8], P[8:]):
        G = blk @ blk.t()
        assert torch.allclose(G - torch.diag(torch.diag(G)), torch.zeros(8, 8), atol=1e-4)


def test_dense_plan_index_arrays_brute_force():
    """DensePlan (the dense [B,N,N] layout read as the complete index set; batch/dense.py:47-54 masks) is pure index
    bookkeeping and runs on the CPU: kinds, node rows, transposed rows, diagonal positions and both CSR views against
    Python loops, on ragged sizes including an empty graph."""
    import torch
    from hot_amd.plan import DensePlan
    n_nodes, N = [3, 0, 4, 1], 4
    P = DensePlan(n_nodes, N, 'cpu')
    assert (P.R, P.Nn, P.Emax, P.Nmax, P.Ev) == (4 * 16, 16, 16, 4, 9 + 0 + 16 + 1)
    rows_of, cols_of = {}, {}
    for b, n in enumerate(n_nodes):
        for i in range(N):
            node = b * N + i
            assert int(P.nkind[node]) == (1 if i < n else 0)
            assert int(P.diag_pos[node]) == ((b * N + i) * N + i if i < n else -1)
            for j in range(N):
                r = (b * N + i) * N + j
                valid = i < n and j < n
                assert int(P.kind[r]) == (0 if not valid else 1 if i == j else 2)
                assert int(P.rown[r]) == (b * N + i if valid else -1)
                assert int(P.coln[r]) == (b * N + j if valid else -1)
                assert int(P.tpos[r]) == ((b * N + j) * N + i if valid else -1)
                if valid:
                    rows_of.setdefault(b * N + i, []).append(r)
                    cols_of.setdefault(b * N + j, []).append(r)
    for ptr, perm, ref in ((P.row_ptr, P.row_perm, rows_of), (P.col_ptr, P.col_perm, cols_of)):
        for node in range(P.Nn):
            seg = perm[int(ptr[node]):int(ptr[node + 1])].tolist()
            assert seg == ref.get(node, []), node
    assert P.g_nend.tolist() == [b * N + n for b, n in enumerate(n_nodes)]
    assert P.n_entries_dev.tolist() == [n * n for n in n_nodes]
