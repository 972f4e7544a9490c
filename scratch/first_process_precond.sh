#!/bin/bash
# fresh box: the default bench (preconditioned) first, then without, then with again
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py --no-cpu-baseline > gpurun_out/pc1.json 2> gpurun_out/pc1.err || exit 1
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-precondition > gpurun_out/pc2.json 2> gpurun_out/pc2.err || exit 1
timeout -k 10 500 python3 bench.py --no-cpu-baseline > gpurun_out/pc3.json 2> gpurun_out/pc3.err || exit 1
for f in pc1 pc2 pc3; do python3 -c "import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['preconditioned'])"; done
