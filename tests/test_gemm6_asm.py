"""csrc/gemm6.hip issues its operand loads from inline asm and waits for them with hand-counted `s_waitcnt vmcnt(N)` (the
compiler's own wait-count pass serialises a register ring at the loop's back edge).  The compiler believes those loads
complete at once, so nothing but the source's discipline (g6_arrived / g6_drain) keeps it from touching a destination
register while the load is still in flight -- a late-landing load then overwrites whatever the register holds by then
(found the hard way: an address built in such a register before the post-loop drain faulted the GPU).  This test
compiles the file to assembly WITH THE LIBRARY'S FLAGS and scans the kernels the library launches by default for any
instruction that reads or writes a register with a load in flight (tests/asm_hazards.py).  CPU only: hipcc
cross-compiles."""
import os
import pytest

from hot_amd import build
from asm_hazards import scan_file

CSRC = os.path.join(os.path.dirname(build.__file__), 'csrc')


@pytest.mark.parametrize('src', ['gemm6.hip', 'gemm6_n128.hip'])
def test_no_instruction_touches_a_register_with_a_load_in_flight(src, tmp_path):
    out = build.assembly(os.path.join(CSRC, src), str(tmp_path / (src + '.s')))
    found = scan_file(out, 'k_gemm6')
    default = {k: v for k, v in found.items() if 'k_gemm6_nn16' not in k}      # (nn16: opt-in A/B kernels, HOTK_G6_PIPE)
    assert all(any('k_gemm6_nnILi%dE' % mode in k for k in default) for mode in (0, 1, 2))      # all rows / row list / records
    if src == 'gemm6.hip':
        assert any('k_gemm6_tnILb1' in k for k in default) and any('k_gemm6_tnILb0' in k for k in default)
    for name, report in default.items():
        assert not report, f'{name}: {report[:4]}'
