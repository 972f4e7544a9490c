#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 400 python3 bench.py --config c2 --molecules 128 --no-precondition > gpurun_out/ka_$v.json 2> gpurun_out/ka_$v.err || exit 1
  python3 -c "
import json
b=json.loads(open('gpurun_out/ka_$v.json').read().strip().splitlines()[-1])
print($v, 'c2-128', round(b['ms_per_step'],3))"
done
