#!/bin/bash
# usage: scratch/kstats.sh <tag> <program> [args...] : rocprofv3 kernel stats of one command; writes the compact
# per-kernel summary (names shortened) to gpurun_out/<tag>.stats.txt and prints it
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- "$@" > gpurun_out/$tag.log 2>&1 || { tail -5 gpurun_out/$tag.log; exit 1; }
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | sort | tail -1)
python - "$f" "$*" <<'PY' | tee gpurun_out/$tag.stats.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('# rocprofv3 --kernel-trace --stats -- ' + sys.argv[2])
print(f"{'kernel':58s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>7s}")
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '')
    n = re.sub(r'^void ', '', n)
    n = re.split(r'\(', n)[0][:58]
    print(f"{n:58s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):7.2f}")
PY
