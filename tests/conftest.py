import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """Plain numeric fixtures written by tests/golden/make_golden.py (reference outputs)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    """max-norm error relative to the reference tensor's scale."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def oracle_fp32_fp64(run, state, x0):
    """The oracle twice on the same inputs -- float32 (what the reference computes) and float64 (what it approximates):
    run(x, sd, dtype) -> output tensor, having called backward.  -> {dtype: (out, dx, {name: dparam})}.  The HIP path is
    held to max(bar, 2 x |oracle32 - oracle64|) against the float64 run: the reference's own rounding distance sets the
    scale, no flat allowance (oracle/hot_oracle.py is dtype-agnostic torch)."""
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.detach().cpu().clone().to(dt) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in state.items()}
        for k, v in sd.items():
            if v.is_floating_point() and 'projection' not in k:
                v.requires_grad_(True)
        x = x0.detach().cpu().clone().to(dt).requires_grad_(True)
        y = run(x, sd, dt)
        res[dt] = (y.detach(), x.grad, {k: v.grad for k, v in sd.items() if v.requires_grad})
    return res[torch.float32], res[torch.float64]


def err64(a, ref, floor=0.):
    """max |a - ref| over max(|ref|max, floor), in float64"""
    return float((a.detach().cpu().double() - ref).abs().max()) / max(float(ref.abs().max()), floor, 1e-300)


def hold_to_fp64(what, hip, o32, o64, bar, floor=0.):
    e, gap = err64(hip, o64, floor), err64(o32, o64, floor)
    assert e <= max(bar, 2 * gap), f'{what}: HIP vs fp64 oracle {e:.2e}, oracle fp32 vs fp64 {gap:.2e}, bar {bar:.0e}'
    return e, gap
