#!/bin/bash
# usage: scratch/kstats.sh <tag> <cmd...> : rocprofv3 kernel stats, top rows, names shortened
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- "$@" > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    n = r['Name']
    n = n.replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"{n:60s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:10.1f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
