#!/bin/bash
# usage: scratch/regs.sh file.s [pattern]: kernel name, vgprs, sgprs, scratch bytes from the metadata of a -save-temps .s
python3 - "$1" "${2:-.}" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size: *\d+', t, re.S):
    b = m.group(0)
    g = lambda k: re.search(r'\.%s: *(\S+)' % k, b).group(1)
    n = g('name')
    if re.search(sys.argv[2], n):
        print(f"{n[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size')}")
PY
