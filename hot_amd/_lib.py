"""ctypes binding of libhotk.so (C ABI declared in include/hotk.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is missing,
or a tensor handed to a kernel wrapper does not live on a HIP device, the call raises.
"""
import ctypes
import os
from ctypes import c_int, c_int64, c_size_t, c_uint, c_void_p, c_float

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HOTK_LIB') or os.path.join(_HERE, 'libhotk.so')   # HOTK_LIB: A/B builds of the same ABI

KEEP_PAD, KEEP_DIAG, KEEP_OFFDIAG = 1, 2, 4
KEEP_VALID = KEEP_DIAG | KEEP_OFFDIAG
SEG_CHUNK = 1024
PHI_RELU, PHI_SOFTMAX_Q, PHI_SOFTMAX_K = 0, 1, 2

_P, _I, _L, _U, _S, _F = c_void_p, c_int, c_int64, c_uint, c_size_t, c_float

# name -> argtypes; every function returns int.  Order follows include/hotk.h.
_SIGNATURES = {
    'hotk_plan_workspace_bytes': [_L, _L, _L, _L, ctypes.POINTER(_S)],
    'hotk_plan_build': [_P, _P, _P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _S, _P],
    'hotk_sort_workspace_bytes': [_L, ctypes.POINTER(_S)],
    'hotk_sort_pairs_u32': [_P, _P, _L, _I, _P, _S, _P],
    'hotk_gather_rows': [_P, _L, _P, _P, _L, _L, _L, _I, _P],
    'hotk_mask_rows': [_P, _L, _P, _U, _F, _P, _L, _L, _L, _P],
    'hotk_seg_sum_rows': [_P, _L, _P, _P, _P, _P, _U, _P, _L, _L, _L, _L, _P],
    'hotk_seg_sum_rows_pair': [_P, _L, _P, _P, _P, _P, _P, _P, _P, _U, _P, _P, _L, _L, _L, _L, _P],
    'hotk_seg_sum_rows_pair_own': [_P, _L, _P, _P, _P, _P, _P, _P, _P, _U, _P, _P, _P, _P, _L, _L, _L, _L, _P],
    'hotk_bias_mask_rows': [_P, _L, _P, _P, _P, _L, _L, _P],
    'hotk_bias_grad_workspace_bytes': [_L, _L, ctypes.POINTER(_S)],
    'hotk_bias_grad_rows': [_P, _L, _P, _P, _P, _L, _L, _P, _S, _P],
    'hotk_basis22_light_fwd': [_P, _L, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    'hotk_basis22_light_bwd': [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    'hotk_basis22_pair_fwd': [_P, _L, _P, _P, _P, _L, _L, _L, _P],
    'hotk_basis22_pair_fwd_diag': [_P, _L, _P, _P, _P, _L, _L, _L, _P, _P, _L, _L, _P],
    'hotk_basis22_pair_bwd': [_P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _L, _L, _L, _P],
    'hotk_assemble22': [_P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _L, _P],
    'hotk_basis22_full_fwd': [_P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P],
    'hotk_basis22_full_bwd_entries': [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P],
    'hotk_basis22_full_bwd_gather': [_P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P, _L, _L, _L, _L, _P],
    'hotk_thin_gemm_supported': [_L, _L],
    'hotk_thin_gemm_bias': [_P, _L, _P, _L, _L, _P, _P, _L, _P, _L, _L, _P],
    'hotk_thin_gemm_bias_bf16': [_P, _L, _P, _L, _L, _P, _P, _L, _L, _P],
    'hotk_thin_gemm_bias_bf16in': [_P, _L, _P, _L, _L, _P, _P, _L, _P, _L, _L, _P],
    'hotk_linear22_entry_fwd': [_P, _L, _P, _L, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    'hotk_linear_bwd_supported': [_L, _L],
    'hotk_linear_bwd_partials': [_L, ctypes.POINTER(_L)],
    'hotk_linear_bwd_fused': [_P, _L, _P, _L, _P, _L, _L, _P, _L, _P, _L, _P],
    'hotk_linear_bwd_fused_bf16': [_P, _L, _P, _L, _P, _L, _L, _P, _L, _P, _L, _P],
    'hotk_linear22_bwd_reduce_supported': [_L, _L, _L],
    'hotk_linear22_bwd_reduce_partials': [_L, ctypes.POINTER(_L)],
    'hotk_linear22_bwd_reduce': [_P, _L, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P, _L, _P, _P],
    'hotk_linear22_bwd_reduce_bf16': [_P, _L, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P, _L, _P, _P],
    'hotk_seg_softmax_pool_fwd': [_P, _L, _P, _L, _P, _L, _P, _P, _P, _U, _P, _L, _P, _P, _P, _L, _I, _I, _I, _P],
    'hotk_seg_softmax_pool_bwd': [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P],
    'hotk_phi_fwd': [_P, _L, _P, _P, _U, _P, _L, _L, _I, _I, _I, _I, _P, _P],
    'hotk_phi_bwd': [_P, _L, _P, _P, _U, _P, _L, _P, _L, _L, _I, _I, _I, _I, _P, _P],
    'hotk_phi_stats_workspace_bytes': [_L, _I, _I, ctypes.POINTER(_S)],
    'hotk_phi_key_stats': [_P, _L, _P, _L, _I, _I, _I, _P, _P, _S, _P],
    'hotk_phi_key_bwd_stats': [_P, _L, _P, _P, _U, _P, _L, _L, _I, _I, _I, _P, _P, _S, _P],
    'hotk_seg_outer_sum': [_P, _L, _P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _L, _P],
    'hotk_row_contract': [_P, _L, _P, _P, _P, _L, _P, _U, _P, _P, _L, _L, _I, _I, _I, _I, _P, _P],
    'hotk_apply_bwd_stats': [_P, _L, _P, _P, _L, _P, _U, _P, _L, _P, _L, _P, _L, _L, _I, _I, _I, _P, _P],
    'hotk_row_expand': [_P, _P, _L, _P, _L, _P, _P, _L, _P, _U, _P, _L, _L, _I, _I, _I, _P, _P],
}


class Attn22Args(ctypes.Structure):
    """mirror of struct hotk_attn22_args (include/hotk.h)"""
    _fields_ = ([(n, _L) for n in ('B', 'Emax', 'Nmax', 'R', 'Nn', 'D')] +
                [(n, ctypes.c_int32) for n in ('H', 'm', 'd_head', 'c')] +
                [(n, _P) for n in ('kind', 'nkind', 'rown', 'coln', 'tpos', 'diag_pos', 'col_ptr', 'col_perm',
                                   'row_ptr', 'row_perm', 'n_entries', 'n_nodes', 'col_rec', 'row_rec')] +
                [('h1', _P), ('ld_h1', _L), ('h2', _P), ('ld_h2', _L), ('v', _P), ('ld_v', _L), ('proj', _P),
                 ('phik', _P), ('ld_phik', _L),
                 ('att', _P), ('ld_att', _L), ('kvg', _P),
                 ('datt', _P), ('ld_datt', _L), ('dout', _P), ('ld_dout', _L),
                 ('dh1', _P), ('ld_dh1', _L), ('dh2', _P), ('ld_dh2', _L), ('dv', _P), ('ld_dv', _L),
                 ('ck', _P), ('dkvg', _P),
                 ('kvn', _P), ('dkvn', _P), ('nd0', _P), ('dd0', _P), ('diag_w', ctypes.c_float), ('storage', ctypes.c_int32)])
STORAGE_F32, STORAGE_BF16 = 0, 1


# phases of hotk_attn22_phase (include/hotk.h)
ATTN22_FWD_GRAPHSUM, ATTN22_FWD_NODESUM, ATTN22_FWD_APPLY, ATTN22_BWD_QUERY, ATTN22_BWD_KEYVAL = range(5)
ATTN22_FWD_APPLY_GRAPH, ATTN22_FWD_APPLY_NODE, ATTN22_BWD_KEYVAL_GRAPH, ATTN22_BWD_KEYVAL_NODE = range(5, 9)


_SIGNATURES.update({
    'hotk_attn22_workspace_bytes': [_L, _L, ctypes.POINTER(_S)],
    'hotk_attn22_forward': [ctypes.POINTER(Attn22Args), _P, _S, _P],
    'hotk_attn22_backward': [ctypes.POINTER(Attn22Args), _P, _S, _P],
    'hotk_attn22_phase': [ctypes.POINTER(Attn22Args), c_int, _P, _S, _P],
    'hotk_attn22_forward_static': [ctypes.POINTER(Attn22Args), _P],
    'hotk_attn22_static_init': [],
    'hotk_layernorm_fwd': [_P, _L, _P, _P, _P, _U, ctypes.c_float, _P, _L, _P, _L, _L, _P],
    'hotk_layernorm_bwd': [_P, _L, _P, _L, _P, _P, _P, _U, _P, _L, _P, _L, _L, _P],
    'hotk_layernorm_partials': [_L, ctypes.POINTER(_L)],
    'hotk_layernorm_supported': [_L],
    'hotk_gemm6_nn_supported': [_L, _L, _L],
    'hotk_gemm6_tn_supported': [_L, _L, _L],
    'hotk_gemm6_split_weight': [_P, _L, _L, _L, c_int, _P, _P],
    'hotk_gemm6_nn': [_P, _L, _P, _P, _P, _L, _L, _L, _L, _P],
    'hotk_gemm6_tn_partials': [_L, _L, _L, ctypes.POINTER(_L)],
    'hotk_gemm6_tn': [_P, _L, _P, _L, _P, _L, _L, _L, _P],
    'hotk_gemm6_nn_rows': [_P, _L, _P, _P, _P, _L, _L, _L, _L, _P, _L, _P],
    'hotk_gemm6_tn_rows': [_P, _L, _P, _L, _P, _L, _L, _L, _P, _L, _P],
    'hotk_gemm6_entry22': [_P, _L, _P, _P, _L, _L, _L, _L, _P, _L, _P, _L, _P, _P, _P],
    'hotk_linear22_entry_fwd_bf16': [_P, _L, _P, _L, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    'hotk_assemble22_bf16': [_P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _L, _P],
    'hotk_seg_sum_rows_bf16': [_P, _L, _P, _P, _P, _P, _U, _P, _L, _L, _L, _L, _P],
})
EXPORTS = sorted(list(_SIGNATURES) + ['hotk_strerror', 'hotk_version'])

_lib = None


class HotkError(RuntimeError):
    pass


def lib():
    """Load libhotk.so once.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HotkError(f'{LIB_PATH} is missing: hot_amd has no fallback path; build it with '
                            f'`python -m hot_amd.build` (hipcc --offload-arch=gfx950)')
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        L.hotk_strerror.argtypes = [c_int]
        L.hotk_strerror.restype = ctypes.c_char_p
        L.hotk_version.argtypes = []
        L.hotk_version.restype = c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise HotkError(f'{what} failed: {lib().hotk_strerror(rc).decode()} ({rc})')


# ------------------------------------------------------------------------------------------------
# optional live timing of C-ABI launches with HIP events on the launch stream (used by bench.py)
# ------------------------------------------------------------------------------------------------
_timing = None   # None = off; else {'only': name or None, 'events': [(name, start, end, alg_bytes)]}


def timing_begin(only=None):
    """Record a HIP event pair around every launch of `only` (or of every entry point when None)."""
    global _timing
    _timing = {'only': only, 'events': []}


def timing_end():
    """-> {entry point: (total ms, launches, algorithmic bytes)}; synchronises the device."""
    global _timing
    t, _timing = _timing, None
    out = {}
    if not t:
        return out
    torch.cuda.synchronize()
    for name, e0, e1, nbytes in t['events']:
        ms, calls, b = out.get(name, (0.0, 0, 0))
        out[name] = (ms + e0.elapsed_time(e1), calls + 1, b + nbytes)
    return out


def call(name, alg_bytes, *args):
    """Invoke one C-ABI entry point on the current torch stream and raise on a non-zero status.
    alg_bytes: the bytes this launch has to move at minimum (operands once + results once)."""
    fn = getattr(lib(), name)
    t = _timing
    if t is not None and (t['only'] is None or t['only'] == name):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, stream_ptr())
        e1.record()
        t['events'].append((name, e0, e1, int(alg_bytes)))
    else:
        rc = fn(*args, stream_ptr())
    if rc != 0:
        raise HotkError(f'{name} failed: {lib().hotk_strerror(rc).decode()} ({rc})')


def require_device(t: torch.Tensor, what='tensor'):
    if not t.is_cuda:
        raise HotkError(f'hot_amd kernels run on a HIP device only; {what} is on {t.device}. '
                        f'There is no CPU path: move the Batch / module to the GPU.')


def require_unsharded(plan, what):
    """Edge-shard mode (hot_amd.dist.shard_graph, plan.shard) is implemented for the kernel-attention sums only.  Ops whose
    sums over ENTRIES would be partial on a rank (and are not all-reduced yet) refuse to run instead of returning this
    rank's partial result."""
    shard = getattr(plan, 'shard', None)
    if shard is not None and shard.world > 1:
        raise HotkError(f'{what} is not available in edge-shard mode (world {shard.world}): its sums over entries would '
                        f'cover this rank\'s shard only; see DESIGN.md section 6')


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    """device pointer of a tensor (storage offset included) or NULL"""
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def rows2d(t: torch.Tensor):
    """View [..., C] as rows: returns (tensor_keepalive, ld, n_rows, C).  Column slices of a contiguous
    tensor are passed in place (ld = parent width); anything else is made contiguous first."""
    require_device(t)
    if t.dtype != torch.float32:
        raise HotkError(f'fp32 tensors expected, got {t.dtype}')
    C = t.size(-1)
    n = t.numel() // C if C > 0 else 0
    if t.dim() == 1:
        t = t.unsqueeze(0)
    ok = t.stride(-1) == 1 or C == 1
    if ok:
        ld = t.stride(-2) if t.dim() >= 2 else C
        # leading dims must collapse onto the row stride
        expect = ld
        for k in range(t.dim() - 2, -1, -1):
            if t.size(k) != 1 and t.stride(k) != expect:
                ok = False
                break
            expect = expect * t.size(k)
        if ld < C:
            ok = False
    if not ok:
        t = t.contiguous()
        ld = C
    return t, int(ld), int(n), int(C)
