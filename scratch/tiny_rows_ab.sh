#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tr_tests.log 2>&1; rc=$?; tail -2 gpurun_out/tr_tests.log; test $rc -eq 0 || exit 1
for i in 1 2 3; do
  timeout -k 10 400 python3 bench.py --config c2 --molecules 128 > gpurun_out/tr_c2_$i.json 2> gpurun_out/tr_c2_$i.err || exit 1
  python3 -c "
import json
b=json.loads(open('gpurun_out/tr_c2_$i.json').read().strip().splitlines()[-1])
print('c2-128', round(b['ms_per_step'],3), b['kernel_census_ms_per_step'].get('hotk_attn22_backward'))"
done
