"""Build libhotk.so in-tree with hipcc for gfx950 (`python -m hot_amd.build`)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'libhotk.so')


def sources():
    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))


def build(force=False, verbose=True):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(HERE, 'csrc', '*.h')) + [os.path.join(HERE, '..', 'include', 'hotk.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in deps):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
           '-mllvm', '-amdgpu-mfma-vgpr-form',   # MFMA results in VGPRs: no v_accvgpr_read/write around the tile code
           '-o', OUT] + srcs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
