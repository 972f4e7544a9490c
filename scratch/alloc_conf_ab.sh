#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/ac_def$i.json 2> gpurun_out/ac_def$i.err || exit 1
  PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/ac_exp$i.json 2> gpurun_out/ac_exp$i.err || { tail -3 gpurun_out/ac_exp$i.err; exit 1; }
done
for f in ac_def1 ac_exp1 ac_def2 ac_exp2; do python3 -c "import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
