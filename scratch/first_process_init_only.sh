#!/bin/bash
# does a child that only creates a context remove the first-process penalty? (fresh box)
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 -c "
import torch
x = torch.zeros(16, device='cuda'); torch.cuda.synchronize(); print('init only')
" || exit 1
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/ci1.json 2> gpurun_out/ci1.err || exit 1
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/ci2.json 2> gpurun_out/ci2.err || exit 1
for f in ci1 ci2; do python3 -c "import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
