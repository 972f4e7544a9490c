"""Build libhotk.so in-tree with hipcc for gfx950 (`python -m hot_amd.build`).

One object per csrc/*.hip (compiled in parallel, only when the source or a header is newer), linked into
hot_amd/libhotk.so.  Objects live under hot_amd/csrc/_obj/ (git-ignored)."""
import glob
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'libhotk.so')
OBJ = os.path.join(HERE, 'csrc', '_obj')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value',
         '-mllvm', '-amdgpu-mfma-vgpr-form']   # MFMA results in VGPRs: no v_accvgpr_read/write around the tile code


def sources():
    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))


def _headers():
    return glob.glob(os.path.join(HERE, 'csrc', '*.h')) + [os.path.join(HERE, '..', 'include', 'hotk.h')]


def _src_time(src):
    """mtime of a source and of the csrc files it includes by name (gemm6_n128.hip is a second build of gemm6.hip)"""
    t = os.path.getmtime(src)
    for m in re.finditer(r'^#include "([^"]+\.hip)"', open(src).read(), re.M):
        inc = os.path.join(os.path.dirname(src), m.group(1))
        if os.path.exists(inc):
            t = max(t, os.path.getmtime(inc))
    return t


def build(force=False, verbose=True, extra_flags=(), out=OUT):
    """extra_flags / out: A/B builds of the same ABI (load with HOTK_LIB=<out>)"""
    srcs = sources()
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tag = '' if out == OUT else '.' + os.path.basename(out)
    os.makedirs(OBJ, exist_ok=True)
    # objects are reused only when they were compiled by the same compiler with the same flags: the stamp holds a hash
    # of (hipcc, FLAGS + extra_flags), so an A/B build with other flags under the same name never measures stale code
    cfg = hashlib.sha256('\0'.join([hipcc] + FLAGS + list(extra_flags)).encode()).hexdigest()
    stamp = os.path.join(OBJ, 'flags' + tag + '.stamp')
    if not os.path.exists(stamp) or open(stamp).read() != cfg:
        force = True
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + tag + '.o')
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(_src_time(s), hdr_time):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ['-c', s, '-o', o])
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + tag + '.o') for s in srcs]
    if not jobs and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(o) for o in objs):
        return out

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    link_extra = [f for f in extra_flags if f.startswith(('-L', '-l', '-Wl,', '-fsanitize', '-g'))]
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + link_extra + objs)
    with open(stamp, 'w') as f:
        f.write(cfg)
    return out


def assembly(src, out, extra_flags=()):
    """device assembly of one source, compiled with the library's own flags (for tests/asm_hazards.py)"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc] + FLAGS + list(extra_flags) + ['-S', '--cuda-device-only', '-o', out, src],
                          stderr=subprocess.DEVNULL)
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
