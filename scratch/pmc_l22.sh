#!/bin/bash
# usage: scratch/pmc_gemm6.sh <tag> : SQ counters of the one-pass light-basis backward reductions at the C1 size (separate --pmc passes)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$tag/p$i -- python3 scratch/bench_l22_reduce.py > gpurun_out/$tag/p$i.log 2>&1 || { tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
python3 scratch/pmc_sum.py gpurun_out/$tag > gpurun_out/$tag.txt
rm -rf gpurun_out/$tag
cat gpurun_out/$tag.txt
