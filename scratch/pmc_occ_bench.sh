#!/bin/bash
# usage: scratch/pmc_occ_bench.sh <tag> : resident waves per SIMD of every kernel of the bench step (two --pmc passes)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$tag/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precondition > gpurun_out/$tag/p$i.log 2>&1 || { tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
python - gpurun_out/$tag <<'PY' > gpurun_out/$tag.txt
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
        agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
rows = []
for name, cs in agg.items():
    if 'GRBM_GUI_ACTIVE' not in cs or 'SQ_WAVE_CYCLES' not in cs: continue
    gui = sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE']) / 8          # cycles of one XCD
    wc = sum(cs['SQ_WAVE_CYCLES']) / len(cs['SQ_WAVE_CYCLES']) * 4            # quad-cycles -> cycles
    valu = sum(cs['SQ_ACTIVE_INST_VALU']) / len(cs['SQ_ACTIVE_INST_VALU']) * 4
    rows.append((gui, name, wc / gui / 1024, valu / gui / 1024, sum(cs['SQ_WAVES']) / len(cs['SQ_WAVES'])))
print('%-60s %10s %12s %10s %10s' % ('kernel', 'Mcycles', 'waves/SIMD', 'VALU busy', 'waves'))
for gui, name, occ, vb, w in sorted(rows, reverse=True)[:40]:
    print('%-60s %10.2f %12.2f %10.2f %10.0f' % (name, gui / 1e6, occ, vb, w))
PY
rm -rf gpurun_out/$tag
cat gpurun_out/$tag.txt
