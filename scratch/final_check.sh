#!/bin/bash
# the round-end sequence on a fresh box: default bench first (as the driver runs it), then smoke, then the GPU tests
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err || { tail -5 gpurun_out/fc_bench.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/fc_bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['preconditioned'])"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/fc_tests.log 2>&1; rc=$?; tail -2 gpurun_out/fc_tests.log; exit $rc
