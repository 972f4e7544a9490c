#!/bin/bash
# usage: scratch/pmc_traffic.sh <tag> <entries>: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
tag=$1; n=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/$ctr -- python scratch/traffic_probe.py $n > gpurun_out/$tag/$ctr.log 2>&1 || { tail -5 gpurun_out/$tag/$ctr.log; exit 1; }
done
python scratch/pmc_traffic_sum.py gpurun_out/$tag $n | tee gpurun_out/$tag.traffic.txt
