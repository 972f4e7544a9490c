#!/bin/bash
# usage: scratch/pmc_occ.sh <tag> <entries> : occupancy counters of the fused micro-benchmark (separate --pmc passes)
tag=$1; n=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$tag/p$i -- python scratch/bench_fused.py $n 1 > gpurun_out/$tag/p$i.log 2>&1 || { tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
python scratch/pmc_sum.py gpurun_out/$tag > gpurun_out/$tag.txt
rm -rf gpurun_out/$tag
cat gpurun_out/$tag.txt
