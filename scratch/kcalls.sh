#!/bin/bash
# usage: scratch/kcalls.sh <tag> <program> [args...] : like kstats.sh, but every kernel, sorted by number of calls
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- "$@" > gpurun_out/$tag.log 2>&1 || { tail -5 gpurun_out/$tag.log; exit 1; }
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | sort | tail -1)
python - "$f" "$*" <<'PY' | tee gpurun_out/$tag.calls.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r['Calls']))
print('# rocprofv3 --kernel-trace --stats -- ' + sys.argv[2])
print(f"{'kernel':110s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s}")
tot = 0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '')
    n = re.sub(r'^void ', '', n)[:110]
    tot += int(r['Calls'])
    print(f"{n:110s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f}")
print('total calls', tot)
PY
