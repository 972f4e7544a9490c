"""Kernel selection for the library GEMMs of small batches.

Below `ops.G6_MIN_ROWS` rows the dense projections (nn.Linear / `_Weight` under models/sparse/linear.py:46-48 and their
autograd) stay rocBLAS / hipBLASLt calls: a 256-row tile of the split-bf16 GEMM would fill a quarter of the chip.  At
8 k rows those calls are 10-40 us each and the library's default choice per shape is often not its best kernel:
PyTorch's TunableOp times the candidates of each shape once, in the warm-up steps, and keeps the fastest (128-molecule
C2 step: 270 GEMMs, 10.1 -> 9.1 ms; no gain from 4 k molecules up, where the tiled GEMMs take over, nor on C1).
Same arithmetic (fp32 GEMMs of the same libraries), another tile shape.  Tuning must be switched off again before a
stream capture: `finish()`."""
import os
import tempfile

import torch


def start(max_ms_per_candidate=15, results_file=None):
    """enable TunableOp with online tuning; results_file: where the choices are kept (default: a file under the temp
    directory, so nothing is written into the working tree)"""
    import torch.cuda.tunable as tn
    if results_file is None:
        results_file = os.path.join(tempfile.gettempdir(), f'hotk_tunableop_{os.getpid()}.csv')
    tn.set_filename(results_file)
    tn.set_max_tuning_duration(int(max_ms_per_candidate))
    tn.enable(True)
    tn.tuning_enable(True)


def finish():
    """keep the choices made so far, try no new candidates (required before torch.cuda.graph capture);
    -> number of GEMM shapes with a tuned choice"""
    import torch.cuda.tunable as tn
    if not tn.is_enabled():
        return 0
    tn.tuning_enable(False)
    torch.cuda.synchronize()
    return len(tn.get_results())
