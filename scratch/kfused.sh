#!/bin/bash
# usage: scratch/kfused.sh <tag> <entries> : per-kernel times of the fused micro-benchmark
tag=$1; n=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python scratch/bench_fused.py $n 3 > gpurun_out/$tag.log 2>&1 || { tail -5 gpurun_out/$tag.log; exit 1; }
grep entries gpurun_out/$tag.log
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | sort | tail -1)
python - "$f" <<'PY' | tee gpurun_out/$tag.stats.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '')
    n = re.sub(r'^void ', '', n).split('(')[0]
    if n.startswith('k_seg') or n.startswith('k_graph'):
        print(f"{n:28s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:10.1f} us")
PY
